"""Development aid: per-parameter gradient error of the wide configuration on split read sets (PMT_SHAPE from the environment)."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.simplefilter("ignore")
import numpy as np, torch
from oracle import artifact_oracle as O
from permutect_amd.architecture.artifact_model import ArtifactModel
from permutect_amd.data.batch import Batch
from permutect_amd.parameters import P0_DIMS, P0_CNN, ModelParameters
from tests.test_forward_gpu import _arrays

dev = torch.device("cuda")
which = sys.argv[1] if len(sys.argv) > 1 else "wide"
if which == "wide":
    params = ModelParameters([48, -2], 32, 2, [40, -1], [-1, 20], 4, [10, 10], list(P0_CNN), 0.0, 0.3)
    cfg = O.Config([48, -2], [40, -1], [-1, 20], 32, 2, 4, list(P0_CNN), 61, 71, 42)
elif which == "A":
    params = ModelParameters([40, -2], 24, 1, [20, -1], [-1, 24], 4, [10, 10], list(P0_CNN), 0.0, 0.3)
    cfg = O.Config([40, -2], [20, -1], [-1, 24], 24, 1, 4, list(P0_CNN), 61, 71, 42)
elif which == "wide64":
    params = ModelParameters([48, -2], 64, 2, [40, -1], [-1, 20], 4, [10, 10], list(P0_CNN), 0.0, 0.3)
    cfg = O.Config([48, -2], [40, -1], [-1, 20], 64, 2, 4, list(P0_CNN), 61, 71, 42)
elif which == "p0h24":
    params = ModelParameters([30, -2, -2, -2], 48, 6, [20, -2, -2, -2], [-2, -2, 10], 4, [10, 10], list(P0_CNN), 0.0, 0.3)
    cfg = O.Config([30, -2, -2, -2], [20, -2, -2, -2], [-2, -2, 10], 48, 6, 4, list(P0_CNN), 61, 71, 42)
else:
    params = ModelParameters([30, -2, -2, -2], 20, 6, [20, -2, -2, -2], [-2, -2, 10], 4, [10, 10], list(P0_CNN), 0.0, 0.3)
    cfg = O.Config([30, -2, -2, -2], [20, -2, -2, -2], [-2, -2, 10], 20, 6, 4, list(P0_CNN), 61, 71, 42)
torch.manual_seed(6)
model = ArtifactModel(params, device=dev, **P0_DIMS)
with torch.no_grad():
    for q in model.parameters():
        q.add_(0.05 * torch.randn_like(q))
sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
nref2, nalt2 = (np.array(eval(os.environ['NREF'])), np.array(eval(os.environ['NALT']))) if 'NREF' in os.environ else (np.array([5, 330, 2, 40]), np.array([3, 280, 9, 600])) if os.environ.get('DEEP', '1') == '1' else (np.array([5, 33, 2, 40, 0, 7]), np.array([3, 28, 9, 60, 4, 1]))
ints2, floats2, packed2 = _arrays(nref2, nalt2, seed=81)
batch2 = Batch.from_arrays(ints2, floats2, packed2).copy_to(dev)
model.train(True)
out2 = model.compute_batch_output(batch2)
model.compute_batch_losses(out2, batch2).total_loss.backward()
torch.cuda.synchronize()
print("shape id", model.engine().shape_id, "PMT_SHAPE", os.environ.get("PMT_SHAPE"))
try:
    model.engine().check_join_fault()
    print("fault word clear")
except Exception as exc:
    print("FAULT:", str(exc)[:200])
i64 = torch.from_numpy(ints2.astype(np.int64))
ob2 = dict(reads_re=torch.from_numpy(O.decode_packed_reads(packed2).astype(np.float32)), nref=i64[:, O.REF_COUNT], nalt=i64[:, O.ALT_COUNT],
           labels=i64[:, O.LABEL], sources=i64[:, O.SOURCE], info_be=torch.from_numpy(floats2[:, O.INFO_START:].astype(np.float32)),
           haplotypes_bh=i64[:, O.HAPLOTYPES_START:])
ref_out2, _, ref_grads2 = O.train_step_grads(sd, cfg, ob2)
O.COMPUTE_DTYPE = torch.float64
ref_out64, _, ref_grads64 = O.train_step_grads({k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}, cfg, ob2)
O.COMPUTE_DTYPE = torch.float32
names = [n for n, _ in model.named_parameters()]
g64 = np.concatenate([ref_grads64[n].numpy().ravel() for n in names])
g32 = np.concatenate([ref_grads2[n].numpy().ravel().astype(np.float64) for n in names])
gh = np.concatenate([p.grad.detach().cpu().numpy().ravel().astype(np.float64) for _, p in model.named_parameters()])
print("vs fp64: logit err HIP", float((out2.logits_b.detach().cpu().double() - ref_out64["logits_b"]).abs().max()), "fp32 oracle", float((ref_out2["logits_b"].double() - ref_out64["logits_b"]).abs().max()))
print("vs fp64: grad rel HIP", np.linalg.norm(gh - g64) / np.linalg.norm(g64), "fp32 oracle", np.linalg.norm(g32 - g64) / np.linalg.norm(g64))
print("logit err", float((out2.logits_b.detach().cpu() - ref_out2["logits_b"]).abs().max()))
tot = 0.0; tn = 0.0
for n, p in model.named_parameters():
    g, r = p.grad.detach().cpu().numpy(), ref_grads2[n].numpy()
    e = np.linalg.norm(g - r); tot += e * e; tn += np.linalg.norm(r) ** 2
    if e > 2e-4 * max(np.linalg.norm(r), 1e-3):
        print(f"{n:70s} |ref| {np.linalg.norm(r):10.4f} err {e:10.3e} rel {e / max(np.linalg.norm(r), 1e-12):.2e}")
print("total rel", (tot / tn) ** 0.5)
if os.environ.get("DUMP_GRADS"):
    np.save(os.environ["DUMP_GRADS"], gh)
    open(os.environ["DUMP_GRADS"] + ".names", "w").write("\n".join(f"{n} {p.numel()}" for n, p in model.named_parameters()))
