"""A few training steps and filter forwards at the bench workload (development aid for rocprofv3 --pmc runs: few dispatches, no CPU baseline)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import synth_arrays
from permutect_amd.architecture.artifact_model import ArtifactModel
from permutect_amd.data.batch import Batch
from permutect_amd.parameters import P0_DIMS, p0_params
from permutect_amd.training.optimizer import FusedClipAdamW, backpropagate

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = ArtifactModel(p0_params(), device=dev, **P0_DIMS)
model.train(True)
ints, floats, packed = synth_arrays(np.random.default_rng(0), B, "wgs")
batch = Batch.from_arrays(ints, floats, packed, pack=True).copy_to(dev)  # as bench.py builds its batches
opt = FusedClipAdamW(model, lr=1e-3, weight_decay=0.01)
for i in range(steps):
    out = model.compute_batch_output(batch)
    loss = model.compute_batch_losses(out, batch).total_loss
    backpropagate(opt, loss, params_to_clip=model.parameters())
model.train(False)
with torch.inference_mode():  # and the filter forward (its kernel instance has its own counters)
    for i in range(steps):
        model.compute_batch_output(batch)
torch.cuda.synchronize()
print("done", float(loss))
