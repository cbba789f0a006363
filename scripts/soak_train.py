#!/usr/bin/env python3
"""A longer end-to-end run (development aid): train_artifact_model for a few epochs on a synthetic dataset that mixes
WGS-shaped variants with deep and oversized read sets, two sources, through the device loader, fused downsampling, balancer,
layered kernels, scheduler, checkpoints and a calibration epoch; prints the epoch history."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_arrays  # noqa: E402
from permutect_amd.architecture.artifact_model import ArtifactModel  # noqa: E402
from permutect_amd.data.memory_mapped_data import MemoryMappedData  # noqa: E402
from permutect_amd.data.reads_dataset import ReadsDataset, all_but_last_fold, last_fold_only  # noqa: E402
from permutect_amd.parameters import P0_DIMS, TrainingParameters, p0_params  # noqa: E402
from permutect_amd.training.model_training import train_artifact_model  # noqa: E402

rng = np.random.default_rng(3)
n = 120000
ints, floats, packed = synth_arrays(rng, n, "wgs")
# every 400th variant deep (up to 900 reads), sources alternate; rebuild the read rows for the changed counts
deep = np.arange(0, n, 400)
ints[deep, 0] = rng.integers(50, 500, len(deep))
ints[deep, 1] = rng.integers(50, 400, len(deep))
ints[:, 4] = np.arange(n) % 2
packed = rng.integers(0, 256, (int(ints[:, 0].astype(np.int64).sum() + ints[:, 1].astype(np.int64).sum()), 12), dtype=np.uint8)
mm = MemoryMappedData.from_arrays(ints, floats, packed)
train = ReadsDataset(mm, num_folds=10, folds_to_use=all_but_last_fold(10))
valid = ReadsDataset(mm, num_folds=10, folds_to_use=last_fold_only(10))
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = ArtifactModel(p0_params(), device=dev, **P0_DIMS)
t0 = time.perf_counter()
hist = train_artifact_model(model, train, valid, TrainingParameters(batch_size=4096, num_epochs=3, num_calibration_epochs=1, learning_rate=1e-3),
                            chunk_variants=1 << 15, seed=5, log=lambda *_: None)
torch.cuda.synchronize()
print(f"{time.perf_counter() - t0:.1f} s")
for h in hist:
    print(h[:3])
assert all(np.isfinite(h[2]) for h in hist)
print("ok")
