"""Development aid: f16 forward vs bf16x3 forward of wide configurations (same instance library), max logit difference."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, warnings
warnings.simplefilter("ignore")
from bench import synth_arrays
from permutect_amd.architecture.artifact_model import ArtifactModel
from permutect_amd.data.batch import Batch
from permutect_amd.parameters import P0_DIMS, P0_CNN, ModelParameters

dev = torch.device("cuda:0")
ints, floats, packed = synth_arrays(np.random.default_rng(0), 256, "wgs")
batch = Batch.from_arrays(ints, floats, packed).copy_to(dev)
DFFN = int(os.environ.get("DFFN", 32))
CASES = {
    "p0h24": ([30, -2, -2, -2], 48, 6, [20, -2, -2, -2], [-2, -2, 10]),
    "p0h24 no skips 2 blocks": ([30], 48, 2, [20], [10]),
    "full": ([48, -2], DFFN, 2, [40, -1], [-1, 20]),
    "no blocks": ([48, -2], 32, 0, [40, -1], [-1, 20]),
    "no skips": ([48], DFFN, 2, [40], [20]),
    "no skips no blocks": ([48], 32, 0, [40], [20]),
    "read skip only": ([48, -2], 32, 0, [40], [20]),
    "reducer skip only": ([48], 32, 0, [40], [-1, 20]),
    "blocks only": ([48], DFFN, 1, [40], [20]),
}
for name, (rl, dffn, nb, il, al) in CASES.items():
    res = {}
    for shape in ("", "bf16x3"):
        os.environ["PMT_SHAPE"] = shape
        torch.manual_seed(1)
        model = ArtifactModel(ModelParameters(rl, dffn, nb, il, al, 4, [10, 10], list(P0_CNN), 0.0, 0.3), device=dev, **P0_DIMS)
        with torch.no_grad():
            for q in model.parameters():
                q.add_(0.05 * torch.randn_like(q))
        model.train(False)
        with torch.no_grad():
            out = model.compute_batch_output(batch)
        res[shape] = (out.logits_b.cpu().numpy(), out.features_be.cpu().numpy(), model.engine().shape_id)
    d = np.abs(res[""][0] - res["bf16x3"][0]).max()
    df = np.abs(res[""][1] - res["bf16x3"][1]).max()
    print(f"{name:22s} shape ids {res[''][2]} {res['bf16x3'][2]}  max logit diff {d:.3e}  max feature diff {df:.3e}", flush=True)
