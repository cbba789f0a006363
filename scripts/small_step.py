"""A few eager train steps at a small batch (development aid: kernel-trace the reference's default batch size)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
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
for i in range(20):
    out = model.compute_batch_output(batch)
    backpropagate(opt, model.compute_batch_losses(out, batch).total_loss, params_to_clip=model.parameters())
torch.cuda.synchronize()
