"""torch.profiler view of one eager train step (development aid: which torch ops launch fills and copies)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from torch.profiler import profile, ProfilerActivity
from bench import synth_arrays
from permutect_amd.architecture.artifact_model import ArtifactModel
from permutect_amd.data.batch import Batch
from permutect_amd.parameters import P0_DIMS, p0_params
from permutect_amd.training.optimizer import FusedClipAdamW, backpropagate
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = ArtifactModel(p0_params(), device=dev, **P0_DIMS)
model.train(True)
batch = Batch.from_arrays(*synth_arrays(np.random.default_rng(0), B, "wgs"), pack=True).copy_to(dev)
opt = FusedClipAdamW(model, lr=1e-3, weight_decay=0.01)
def step():
    out = model.compute_batch_output(batch)
    backpropagate(opt, model.compute_batch_losses(out, batch).total_loss, params_to_clip=model.parameters())
for i in range(5): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
for e in prof.events():
    if e.name.startswith("aten::") and e.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::zeros", "aten::ones_like", "aten::zeros_like", "aten::to", "aten::_to_copy", "aten::contiguous", "aten::cat", "aten::add_", "aten::add", "aten::sum", "aten::mul"):
        st = [s for s in (e.stack or []) if "permutect_amd" in s or "bench" in s or "scripts" in s]
        print(e.name, [str(x) for x in (e.input_shapes or [])][:2] if hasattr(e, "input_shapes") else "", st[:2])
