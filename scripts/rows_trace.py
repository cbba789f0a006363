#!/usr/bin/env python3
"""Development aid (library built with -DPMT_ROWS_TRACE=1): where one wave of pmt_rows_forward spends its cycles."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from permutect_amd.architecture.artifact_model import ArtifactModel  # noqa: E402
from permutect_amd.engine import lib as L  # noqa: E402
from permutect_amd.engine.runtime import RowsMlpFunction  # noqa: E402
from permutect_amd.parameters import P0_DIMS, p0_params  # noqa: E402

dev = torch.device("cuda:0")
n = 65536
model = ArtifactModel(p0_params(), device=dev, **P0_DIMS)
eng = model.engine()
eng.pack(eng.plan.materialize_phi(model).detach().contiguous())
x = torch.from_numpy(np.random.default_rng(0).standard_normal((n, 71)).astype(np.float32)).to(dev)
names = ["weights staged", "rows loaded", "first linear", "rest of the MLP", "stored"]
with torch.inference_mode():
    for _ in range(3):
        y = RowsMlpFunction.apply(eng, L.ROWS_INFO, x, eng.trigger, None)
    torch.cuda.synchronize()
    row = y[(n // 256 // 2) * 256].cpu().numpy()
prev = 0.0
for k, nm in enumerate(names):
    print(f"{nm:18s} at {row[k]:9.0f} cycles  (+{row[k] - prev:8.0f})")
    prev = row[k]
