#!/usr/bin/env python3
"""Development aid: the stress-depth (600 reads per set) train / filter step of bench.py alone: python scripts/stress_time.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_arrays
from permutect_amd.architecture.artifact_model import ArtifactModel
from permutect_amd.data.batch import Batch
from permutect_amd.parameters import P0_DIMS, p0_params
from permutect_amd.training.optimizer import FusedClipAdamW
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = ArtifactModel(p0_params(), device=dev, **P0_DIMS)
opt = FusedClipAdamW(model, lr=1e-3, weight_decay=0.01)
rng = np.random.default_rng(4040)
pool = []
for _ in range(2):
    b = Batch.from_arrays(*synth_arrays(rng, 1420, "stress"), pack=True)
    b.plan(allow_split=True)
    pool.append(b.copy_to(dev))
eng = model.engine()
def step(b, train):
    if not train:
        with torch.inference_mode():
            return model.compute_batch_output(b)
    opt.zero_grad()
    model.compute_batch_losses(model.compute_batch_output(b), b).total_loss.backward()
    opt.step()
for train in (True, False):
    model.train(train)
    for i in range(6): step(pool[i % 2], train)
    eng.timers = {"pmt_forward": [], "pmt_backward": []}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(30): step(pool[i % 2], train)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
    k = {n: (sum(a.elapsed_time(b) for a, b in v) / len(v) if v else None) for n, v in eng.timers.items()}
    eng.timers = None
    print(f"{'train' if train else 'filter'}: {1e3 * dt:.3f} ms/step, kernels {k}", flush=True)
eng.check_join_fault()
