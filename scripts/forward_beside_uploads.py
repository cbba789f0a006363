#!/usr/bin/env python3
"""Development aid: the resident filter step while a background thread keeps uploading chunk-sized buffers from page-locked memory
(what the device chunk loader does beside the consumer) -- does the DMA alone slow the read-set kernels?"""
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_arrays  # noqa: E402
from permutect_amd.architecture.artifact_model import ArtifactModel  # noqa: E402
from permutect_amd.data.batch import Batch  # noqa: E402
from permutect_amd.parameters import P0_DIMS, p0_params  # noqa: E402

dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
model = ArtifactModel(p0_params(), device=dev, **P0_DIMS)
model.engine()
model.train(False)
res = [Batch.from_arrays(*synth_arrays(rng, 65536, "wgs"), pack=True).copy_to(dev) for _ in range(4)]
host = torch.empty(110 << 20, dtype=torch.uint8, pin_memory=True)
dst = torch.empty(110 << 20, dtype=torch.uint8, device=dev)
stop = threading.Event()
copied = [0]


def uploader():
    torch.cuda.set_device(dev)
    s = torch.cuda.Stream(dev)
    while not stop.is_set():
        with torch.cuda.stream(s):
            dst.copy_(host, non_blocking=True)
        s.synchronize()
        copied[0] += 1


def timed(k=200):
    with torch.inference_mode():
        for i in range(10):
            model.compute_batch_output(res[i % 4])
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        eng = model.engine()
        eng.timers = {"pmt_forward": [], "pmt_backward": []}
        c0, t0 = copied[0], time.perf_counter()
        a.record()
        for i in range(k):
            model.compute_batch_output(res[i % 4])
        b.record()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        kern = sum(s.elapsed_time(e) for s, e in eng.timers["pmt_forward"]) / len(eng.timers["pmt_forward"])
        eng.timers = None
    return a.elapsed_time(b) / k, kern, (copied[0] - c0) * 110 / 1024 / dt


print("alone:            step %.3f ms, forward kernel %.3f ms" % timed()[:2])
t = threading.Thread(target=uploader)
t.start()
step, kern, gbs = timed()
print("beside uploads:   step %.3f ms, forward kernel %.3f ms, uploads at %.1f GB/s" % (step, kern, gbs))
stop.set()
t.join()
print("alone again:      step %.3f ms, forward kernel %.3f ms" % timed()[:2])
