#!/usr/bin/env python3
"""Development aid: where the posterior hand-off (tools/posterior_data.make_posterior_mmap) over 5 x 2^20 candidates spends its time."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PMT_POSTERIOR_TIMING"] = "1"
from bench import synth_arrays  # noqa: E402
from permutect_amd.architecture.artifact_model import ArtifactModel  # noqa: E402
from permutect_amd.data.memory_mapped_data import MemoryMappedData  # noqa: E402
from permutect_amd.data.reads_dataset import ReadsDataset  # noqa: E402
from permutect_amd.parameters import P0_DIMS, p0_params  # noqa: E402
from permutect_amd.tools.posterior_data import make_posterior_mmap  # noqa: E402

dev = torch.device("cuda:0")
ints, floats, packed = synth_arrays(np.random.default_rng(0), 1 << 20, "wgs")
ds = ReadsDataset(MemoryMappedData.from_arrays(np.concatenate([ints] * 5), np.concatenate([floats] * 5), np.concatenate([packed] * 5))).pin_memory()
model = ArtifactModel(p0_params(), device=dev, **P0_DIMS)
model.engine()
for rep in range(4):
    torch.cuda.synchronize()
    t = time.perf_counter()
    post = make_posterior_mmap(ds, model, 65536, chunk_variants=1 << 18)
    print(f"pass {rep}: {1e3 * (time.perf_counter() - t):.1f} ms for {len(ds)} candidates", flush=True)
    del post
