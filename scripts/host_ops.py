"""Which lines of the package launch the small torch fills / copies of one train step (development aid): counts calls of the
usual suspects by their first permutect_amd / bench frame."""
import sys, os, traceback, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import synth_arrays
from permutect_amd.architecture.artifact_model import ArtifactModel
from permutect_amd.data.batch import Batch
from permutect_amd.parameters import P0_DIMS, p0_params
from permutect_amd.training.optimizer import FusedClipAdamW, backpropagate
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = ArtifactModel(p0_params(), device=dev, **P0_DIMS)
model.train(True)
batch = Batch.from_arrays(*synth_arrays(np.random.default_rng(0), B, "wgs"), pack=True).copy_to(dev)
opt = FusedClipAdamW(model, lr=1e-3, weight_decay=0.01)
def step():
    out = model.compute_batch_output(batch)
    backpropagate(opt, model.compute_batch_losses(out, batch).total_loss, params_to_clip=model.parameters())
for i in range(3): step()
torch.cuda.synchronize()
counts = collections.Counter()
def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "permutect_amd" in fr.filename or fr.filename.endswith("bench.py"):
            return f"{os.path.relpath(fr.filename)}:{fr.lineno} {fr.line}"
    return "?"
def wrap(obj, name):
    orig = getattr(obj, name)
    def f(*a, **k):
        counts[(name, site())] += 1
        return orig(*a, **k)
    setattr(obj, name, f)
for name in ("zeros", "zeros_like", "ones", "ones_like", "empty", "empty_like", "tensor", "hstack", "cat", "sum", "full"):
    wrap(torch, name)
for name in ("copy_", "to", "contiguous", "float", "long", "clone", "zero_", "fill_", "int"):
    wrap(torch.Tensor, name)
step()
torch.cuda.synchronize()
for (name, s), n in sorted(counts.items(), key=lambda kv: -kv[1]):
    print(f"{n:3d}  {name:12s} {s[:150]}")
