#!/usr/bin/env python3
"""Haplotype-CNN kernel times at B = 65 536 (development aid; PMT_CNN_DBG ablates phases of the backward kernel)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from permutect_amd.architecture.artifact_model import ArtifactModel  # noqa: E402
from permutect_amd.engine.runtime import HaplotypeCnnFunction, PhiFunction  # noqa: E402
from permutect_amd.parameters import P0_DIMS, p0_params  # noqa: E402

dev = torch.device("cuda:0")
model = ArtifactModel(p0_params(), device=dev, **P0_DIMS)
model.train(True)
eng = model.engine()
b = 65536
hap = torch.from_numpy(np.random.default_rng(0).integers(0, 5, (b, 42))).to(dev)
phi = PhiFunction.apply(eng, eng.plan.phi_program(model), eng.trigger)
eng.pack(phi.detach().contiguous())
fw, bw = [], []
for i in range(8):
    s0, s1, s2, s3 = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    s0.record()
    out = HaplotypeCnnFunction.apply(eng, hap, eng.trigger)
    s1.record()
    g = torch.ones_like(out)
    s2.record()
    out.backward(g)
    s3.record()
    torch.cuda.synchronize()
    fw.append(s0.elapsed_time(s1))
    bw.append(s2.elapsed_time(s3))
print(f"PMT_CNN_DBG={os.environ.get('PMT_CNN_DBG', '0'):>2s}: forward {np.median(fw[2:]):.3f} ms, backward {np.median(bw[2:]):.3f} ms")
