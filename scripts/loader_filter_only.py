#!/usr/bin/env python3
"""Development aid: the filter forward over a pinned synthetic dataset through the device chunk loader, nothing else (for timeline
traces: rocprofv3 --kernel-trace --memory-copy-trace -- python3 scripts/loader_filter_only.py [variants])."""
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

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 21
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
ds = ReadsDataset(MemoryMappedData.from_arrays(*synth_arrays(rng, n, "wgs"))).pin_memory()
model = ArtifactModel(p0_params(), device=dev, **P0_DIMS)
model.engine()
model.train(False)
for p in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    k = 0
    with torch.inference_mode():
        for cb in ds.device_loader(65536, dev, chunk_variants=1 << 18, shuffle=False):
            model.compute_batch_output(cb)
            k += 1
    torch.cuda.synchronize()
    print(f"pass {p}: {1e3 * (time.perf_counter() - t0) / k:.3f} ms/batch", flush=True)
