#!/usr/bin/env python3
"""Development aid: where a filter pass's start-up time goes -- host time until the loader hands out its k-th batch, and the GPU time
of each of the first batches (events), against the steady state."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_arrays
from permutect_amd.architecture.artifact_model import ArtifactModel
from permutect_amd.data.memory_mapped_data import MemoryMappedData
from permutect_amd.data.reads_dataset import ReadsDataset
from permutect_amd.parameters import P0_DIMS, p0_params
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 21
dev = torch.device("cuda:0")
ds = ReadsDataset(MemoryMappedData.from_arrays(*synth_arrays(np.random.default_rng(0), n, "wgs"))).pin_memory()
model = ArtifactModel(p0_params(), device=dev, **P0_DIMS)
model.train(False)
for p in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    host, evs = [], []
    with torch.inference_mode():
        loader = ds.device_loader(65536, dev, chunk_variants=1 << 18, shuffle=False)
        t_made = time.perf_counter() - t0
        for cb in loader:
            host.append(time.perf_counter() - t0)
            s = torch.cuda.Event(enable_timing=True); s.record()
            model.compute_batch_output(cb)
            e = torch.cuda.Event(enable_timing=True); e.record()
            evs.append((s, e))
    t_loop = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    gpu = [a.elapsed_time(b) for a, b in evs]
    first = evs[0][0]
    starts = [first.elapsed_time(a) for a, _ in evs]
    print(f"pass {p}: loader made {1e3*t_made:.2f} ms; batch k handed out at (ms) {[round(1e3*h,2) for h in host[:6]]} ... last {1e3*host[-1]:.2f}; loop done {1e3*t_loop:.2f}, all done {1e3*t_all:.2f}", flush=True)
    print(f"        GPU ms of batches 0..5 {[round(g,3) for g in gpu[:6]]}, median {np.median(gpu):.3f}; GPU start of batch k after batch 0's start {[round(s,2) for s in starts[:6]]} ... last {starts[-1]:.2f}", flush=True)
