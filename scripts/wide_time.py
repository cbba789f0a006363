"""Development aid: step times of the wide configuration (parameters.wide_params: d_model 98) against P0 on the headline batch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import synth_arrays
from permutect_amd.architecture.artifact_model import ArtifactModel
from permutect_amd.data.batch import Batch
from permutect_amd.parameters import P0_DIMS, p0_params, wide_params
from permutect_amd.training.optimizer import FusedClipAdamW

B = int(os.environ.get("B", 65536))
dev = torch.device("cuda:0")
ints, floats, packed = synth_arrays(np.random.default_rng(0), B, "wgs")
batch = Batch.from_arrays(ints, floats, packed).copy_to(dev)
for name, params in (("P0", p0_params()), ("wide_d98", wide_params())):
    torch.manual_seed(0)
    model = ArtifactModel(params, device=dev, **P0_DIMS)
    opt = FusedClipAdamW(model, lr=1e-3, weight_decay=0.01)
    model.train(True)
    def step():
        out = model.compute_batch_output(batch)
        loss = model.compute_batch_losses(out, batch).total_loss
        opt.zero_grad(); loss.backward(); opt.step()
    for _ in range(3): step()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize(); tr = (time.perf_counter() - t) / 10
    model.train(False)
    with torch.no_grad():
        for _ in range(3): model.compute_batch_output(batch)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10): model.compute_batch_output(batch)
        torch.cuda.synchronize(); fl = (time.perf_counter() - t) / 10
    print(f"{name:10s} shape_id {model.engine().shape_id}  train {1e3 * tr:8.2f} ms  filter {1e3 * fl:8.2f} ms  (B = {B})", flush=True)
