#!/usr/bin/env python3
"""Phase timeline of one wave of pmt_cnn3_backward_kernel (development aid; needs a library built with EXTRA=-DC3_TRACE=1)."""
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
b = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
hap = torch.from_numpy(np.random.default_rng(0).integers(0, 5, (b, 42))).to(dev)
phi = PhiFunction.apply(eng, eng.plan.phi_program(model), eng.trigger)
eng.pack(phi.detach().contiguous())
for it in range(3):
    out = HaplotypeCnnFunction.apply(eng, hap, eng.trigger)
    stash = out.grad_fn.stash
    s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s0.record()
    out.backward(torch.ones_like(out))
    s1.record()
    torch.cuda.synchronize()
def show(tr, names_cycle, last):
    n = int(tr[127])
    names = ["weights built"] + names_cycle * 64
    prev = t0 = tr[0]
    for i in range(1, min(n, 120)):
        label = names[i - 1] if (i < n - 1 or last is None) else last
        print(f"{i:3d} {label:18s} +{tr[i] - prev:8d}  at {tr[i] - t0:9d}")
        prev = tr[i]


# forward (training: the log lies over the stash rows of the wave's first batch)
out = HaplotypeCnnFunction.apply(eng, hap, eng.trigger)
torch.cuda.synchronize()
trf = out.grad_fn.stash[:256].view(torch.int64).cpu().numpy()
print(f"forward: {int(trf[127])} events (the first batch is not logged)")
show(trf, ["records stored", "batch begins", "haplotypes", "conv1 + pool", "conv2", "linear"], None)
del out
tr = stash[:256].view(torch.int64).cpu().numpy()
n = int(tr[127])
print(f"backward {s0.elapsed_time(s1):.3f} ms; {n} events")
names = ["weights built"] + ["inputs", "1 linear dgrad", "2 act2 + dWl", "3 dW2", "4 conv2 dgrad", "5 dW1"] * 64
t0 = tr[0]
prev = t0
for i in range(1, min(n, 120)):
    label = names[i - 1] if i < n - 1 else "gradients out"
    print(f"{i:3d} {label:18s} +{tr[i] - prev:8d}  at {tr[i] - t0:9d}")
    prev = tr[i]
