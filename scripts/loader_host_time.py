#!/usr/bin/env python3
"""Development aid: is the filter loop over the device chunk loader bound by the consumer's HOST thread?  Splits the wall time of one
pass into the time the consumer spends (a) waiting in the iterator, (b) in compute_batch_output (enqueueing), and the final drain."""
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

dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
ds = ReadsDataset(MemoryMappedData.from_arrays(*synth_arrays(rng, 5 << 20, "wgs"))).pin_memory()
model = ArtifactModel(p0_params(), device=dev, **P0_DIMS)
model.engine()
model.train(False)
for p in range(3):
    torch.cuda.synchronize()
    t_iter = t_call = 0.0
    k = 0
    t0 = time.perf_counter()
    with torch.inference_mode():
        it = iter(ds.device_loader(65536, dev, chunk_variants=1 << 18, shuffle=False))
        while True:
            a = time.perf_counter()
            cb = next(it, None)
            b = time.perf_counter()
            if cb is None:
                break
            model.compute_batch_output(cb)
            c = time.perf_counter()
            t_iter += b - a
            t_call += c - b
            k += 1
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"pass {p}: {k} batches, wall {1e3 * (t2 - t0) / k:.3f} ms/batch: iterator {1e3 * t_iter / k:.3f}, compute_batch_output {1e3 * t_call / k:.3f}, "
          f"drain {1e3 * (t2 - t1):.2f} ms in all", flush=True)
