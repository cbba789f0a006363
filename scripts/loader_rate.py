#!/usr/bin/env python3
"""Development aid: how fast the device chunk loader produces batches on its own (no model), and the filter / train loops on top
of it, to tell a producer-bound loop from a consumer-bound one.   python scripts/loader_rate.py [variants] [batch]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_arrays  # noqa: E402
from permutect_amd.architecture.artifact_model import ArtifactModel  # noqa: E402
from permutect_amd.data.memory_mapped_data import MemoryMappedData  # noqa: E402
from permutect_amd.data.reads_dataset import ReadsDataset  # noqa: E402
from permutect_amd.parameters import P0_DIMS, p0_params  # noqa: E402
from permutect_amd.training.optimizer import FusedClipAdamW  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
b = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
t0 = time.perf_counter()
ints, floats, packed = synth_arrays(rng, n, "wgs")
print(f"synth {n} variants: {time.perf_counter() - t0:.1f} s", flush=True)
ds = ReadsDataset(MemoryMappedData.from_arrays(ints, floats, packed))
if os.environ.get("PMT_PIN_DATASET", "1") != "0":
    t0 = time.perf_counter()
    ds.pin_memory()
    print(f"dataset pinned: {time.perf_counter() - t0:.2f} s", flush=True)
model = ArtifactModel(p0_params(), device=dev, **P0_DIMS)
model.engine()  # (built outside inference mode: its buffers are ordinary tensors)
opt = FusedClipAdamW(model, lr=1e-3, weight_decay=0.01)


def loop(name, fn, shuffle, passes=2):
    for p in range(passes):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        k = 0
        for cb in ds.device_loader(b, dev, chunk_variants=1 << 18, rng=rng, shuffle=shuffle):
            if cb.size() == b:
                fn(cb)
                k += 1
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{name}: pass {p}: {k} batches in {1e3 * dt:.1f} ms = {1e3 * dt / k:.3f} ms/batch = {k * b / dt / 1e6:.1f} M read-sets/s", flush=True)


def filt(cb):
    with torch.inference_mode():
        model.compute_batch_output(cb)


def train(cb):
    opt.zero_grad()
    out = model.compute_batch_output(cb)
    model.compute_batch_losses(out, cb).total_loss.backward()
    opt.step()


loop("loader only (no model), in order", lambda cb: None, False)
loop("loader only (no model), shuffled", lambda cb: None, True)
loop("loader + read_index only", lambda cb: cb.read_index(), False)
model.train(False)
loop("filter through the loader", filt, False, passes=3)
model.train(True)
loop("train through the loader", train, True, passes=3)
