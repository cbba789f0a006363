"""Development aid: where the HOST time of a small-batch training step goes (cProfile over 300 eager steps at B = 64; the step's
~25 launches are a dependent chain of ~0.45 ms, everything above that is Python)."""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import synth_arrays
from permutect_amd.architecture.artifact_model import ArtifactModel
from permutect_amd.data.batch import Batch
from permutect_amd.parameters import P0_DIMS, p0_params
from permutect_amd.training.optimizer import FusedClipAdamW
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = ArtifactModel(p0_params(), device=dev, **P0_DIMS)
model.train(True)
batches = [Batch.from_arrays(*synth_arrays(np.random.default_rng(i), B, "wgs"), pack=True).copy_to(dev) for i in range(4)]
opt = FusedClipAdamW(model, lr=1e-3, weight_decay=0.01)


def step(b):
    opt.zero_grad()
    out = model.compute_batch_output(b)
    model.compute_batch_losses(out, b).total_loss.backward()
    opt.step()


for i in range(50):
    step(batches[i % 4])
torch.cuda.synchronize()
t = time.perf_counter()
for i in range(300):
    step(batches[i % 4])
torch.cuda.synchronize()
print(f"B={B}: {1e3 * (time.perf_counter() - t) / 300:.3f} ms per step (wall)")
pr = cProfile.Profile()
pr.enable()
for i in range(300):
    step(batches[i % 4])
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
