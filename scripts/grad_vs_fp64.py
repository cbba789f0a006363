#!/usr/bin/env python3
"""Development aid: how far the HIP gradients, and the fp32 oracle's, are from an fp64 evaluation of the same training step
(relative L2 over all parameters; B WGS-shaped read sets, P0).   [PMT_LIB=<build>] python scripts/grad_vs_fp64.py [B]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_arrays  # noqa: E402
from oracle import artifact_oracle as O  # noqa: E402
from permutect_amd.architecture.artifact_model import ArtifactModel  # noqa: E402
from permutect_amd.data.batch import Batch  # noqa: E402
from permutect_amd.parameters import P0_DIMS, p0_params  # noqa: E402
from permutect_amd.training.optimizer import FusedClipAdamW  # noqa: E402
from tests.helpers import config_for  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
torch.set_num_threads(16)
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = ArtifactModel(p0_params(), device=dev, **P0_DIMS)
with torch.no_grad():
    for p in model.parameters():
        p.add_(0.05 * torch.randn_like(p))
ints, floats, packed = synth_arrays(np.random.default_rng(1), B, "wgs")
batch = Batch.from_arrays(ints, floats, packed).copy_to(dev)
model.train(True)
out = model.compute_batch_output(batch)
losses = model.compute_batch_losses(out, batch)
opt = FusedClipAdamW(model, lr=1e-3, weight_decay=0.01)
opt.zero_grad()
losses.total_loss.backward()
torch.cuda.synchronize()
names = [n for n, _ in model.named_parameters()]
ours = np.concatenate([p.grad.detach().cpu().numpy().ravel().astype(np.float64) for _, p in model.named_parameters()])
i64 = torch.from_numpy(ints.astype(np.int64))
ob = dict(reads_re=torch.from_numpy(O.decode_packed_reads(packed).astype(np.float32)), nref=i64[:, O.REF_COUNT], nalt=i64[:, O.ALT_COUNT],
          labels=i64[:, O.LABEL], sources=i64[:, O.SOURCE], info_be=torch.from_numpy(floats[:, O.INFO_START:].astype(np.float32)),
          haplotypes_bh=i64[:, O.HAPLOTYPES_START:])
sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
cfg = config_for("p0")
res = {}
for dt in (torch.float32, torch.float64):
    O.COMPUTE_DTYPE = dt
    sdd = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in sd.items()}
    _, _, g = O.train_step_grads(sdd, cfg, ob)
    res[dt] = np.concatenate([g[n].numpy().ravel().astype(np.float64) for n in names])
O.COMPUTE_DTYPE = torch.float32
ref = res[torch.float64]
rel = lambda a: float(np.linalg.norm(a - ref) / np.linalg.norm(ref))  # noqa: E731
print(f"B = {B}: |HIP - fp64| / |fp64| = {rel(ours):.3e};  |fp32 oracle - fp64| / |fp64| = {rel(res[torch.float32]):.3e};  "
      f"|HIP - fp32 oracle| / |fp32 oracle| = {float(np.linalg.norm(ours - res[torch.float32]) / np.linalg.norm(res[torch.float32])):.3e}")
