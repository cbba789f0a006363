"""kernel_times.py for another depth: python scripts/kernel_times_depth.py stress 1420 [steps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from torch.profiler import profile, ProfilerActivity
from bench import synth_arrays
from permutect_amd.architecture.artifact_model import ArtifactModel
from permutect_amd.data.batch import Batch
from permutect_amd.parameters import P0_DIMS, p0_params
from permutect_amd.training.optimizer import FusedClipAdamW, backpropagate
depth = sys.argv[1] if len(sys.argv) > 1 else "stress"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1420
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = ArtifactModel(p0_params(), device=dev, **P0_DIMS)
model.train(True)
batch = Batch.from_arrays(*synth_arrays(np.random.default_rng(0), B, depth), pack=True).copy_to(dev)
opt = FusedClipAdamW(model, lr=1e-3, weight_decay=0.01)
def step():
    out = model.compute_batch_output(batch)
    backpropagate(opt, model.compute_batch_losses(out, batch).total_loss, params_to_clip=model.parameters())
for i in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for i in range(steps): step()
    torch.cuda.synchronize()
rows = sorted(((e.key, e.device_time_total / steps, e.count / steps) for e in prof.key_averages() if e.device_time_total > 0), key=lambda r: -r[1])
total = sum(r[1] for r in rows)
for k, t, n in rows[:14]:
    print(f"{t:9.1f} us  x{n:4.1f}  {k[:120]}")
print(f"KT total {total:.0f} us; plan: groups {batch.plan().num_groups} layered {batch.plan().layered}")
