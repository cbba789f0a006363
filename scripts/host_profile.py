#!/usr/bin/env python3
"""Host-side cost of one training step at a small batch (development aid): python scripts/host_profile.py [B]"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_arrays  # noqa: E402
from permutect_amd.architecture.artifact_model import ArtifactModel  # noqa: E402
from permutect_amd.data.batch import Batch  # noqa: E402
from permutect_amd.parameters import P0_DIMS, p0_params  # noqa: E402
from permutect_amd.training.optimizer import FusedClipAdamW  # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
model = ArtifactModel(p0_params(), device=dev, **P0_DIMS)
model.train(True)
opt = FusedClipAdamW(model, lr=1e-3, weight_decay=0.01)
rng = np.random.default_rng(0)
batches = [Batch.from_arrays(*synth_arrays(rng, b, "wgs")).copy_to(dev) for _ in range(4)]


def step(i):
    batch = batches[i % 4]
    opt.zero_grad()
    out = model.compute_batch_output(batch)
    losses = model.compute_batch_losses(out, batch)
    losses.total_loss.backward()
    opt.step()


for i in range(20):
    step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(200):
    step(i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"B={b}: host {1e3 * (t1 - t0) / 200:.3f} ms/step, with sync {1e3 * (t2 - t0) / 200:.3f} ms/step", flush=True)
pr = cProfile.Profile()
pr.enable()
for i in range(200):
    step(i)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(40)

# ---- host time inside the autograd backward functions (they run on autograd's device thread, which cProfile does not see)
import permutect_amd.engine.runtime as RT  # noqa: E402

acc = {}
for name in ("LossesFunction", "PhiFunction", "ReadSetFunction", "HaplotypeCnnFunction", "RowsMlpFunction"):
    cls = getattr(RT, name)
    orig = cls.backward

    def make(orig, name):
        def timed(ctx, *grads):
            t = time.perf_counter()
            out = orig(ctx, *grads)
            acc[name] = acc.get(name, 0.0) + time.perf_counter() - t
            return out
        return staticmethod(timed)
    cls.backward = make(orig, name)
t0 = time.perf_counter()
for i in range(200):
    step(i)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"host {1e3 * (t1 - t0) / 200:.3f} ms/step; inside backward functions (us/step):", {k: round(1e6 * v / 200, 1) for k, v in acc.items()})
