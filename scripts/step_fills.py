"""Development aid: which ATen launches (fills, copies, reductions) a resident training step still makes, with their Python call sites."""
import os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import synth_arrays
from permutect_amd.architecture.artifact_model import ArtifactModel
from permutect_amd.data.batch import Batch
from permutect_amd.parameters import P0_DIMS, p0_params
from permutect_amd.training.optimizer import FusedClipAdamW
from torch.utils._python_dispatch import TorchDispatchMode

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = ArtifactModel(p0_params(), device=dev, **P0_DIMS)
model.train(True)
batch = Batch.from_arrays(*synth_arrays(np.random.default_rng(0), 4096, "wgs"), pack=True).copy_to(dev)
opt = FusedClipAdamW(model, lr=1e-3, weight_decay=0.01)


def step():
    opt.zero_grad()
    out = model.compute_batch_output(batch)
    model.compute_batch_losses(out, batch).total_loss.backward()
    opt.step()


for _ in range(3):
    step()
seen = collections.Counter()


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(k in name for k in ("view", "detach", "alias", "slice", "select", "as_strided", "_unsafe_view", "expand", "t.default", "reshape", "unsqueeze", "squeeze")):
            frames = [f for f in traceback.extract_stack()[:-1] if "permutect_amd" in f.filename or "scripts/step_fills" in f.filename]
            where = " < ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in frames[-3:])
            seen[(name, where)] += 1
        return func(*args, **(kwargs or {}))


with Log():
    step()
torch.cuda.synchronize()
for (name, where), n in sorted(seen.items(), key=lambda kv: kv[0][1]):
    print(f"{n:3d}  {name:40s} {where}")
