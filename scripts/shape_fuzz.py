"""Development aid: other model shapes through their exact instances.  `python scripts/shape_fuzz.py shapes` prints the SHAPE of every
configuration (CPU: build them with `make -C permutect_amd/csrc instance SHAPE="..."`); `python scripts/shape_fuzz.py run` (GPU)
compares forward and gradients of every instance kind with fp32 / fp64 oracle evaluations on ordinary and split read sets."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.simplefilter("ignore")
import numpy as np, torch
from permutect_amd.parameters import P0_DIMS, P0_CNN, ModelParameters

CONFIGS = {
    "A_4_3_5_2": ([40, -2], 24, 3, [20, -1], [-1, 24]),
    "B_4_1_3_1_h20": ([16], 40, 2, [8], [12]),
    "C_4_4_8_2_h32": ([64, -2], 64, 2, [40], [-2, 32]),
    "D_4_2_6_1": ([30, -2], 20, 4, [50, -1], [-2, 10]),
}


if os.environ.get("FUZZ_ONLY"):
    CONFIGS = {k: v for k, v in CONFIGS.items() if k.startswith(os.environ["FUZZ_ONLY"])}
if os.environ.get("FUZZ_BLOCKS"):
    CONFIGS = {k: (v[0], v[1], int(os.environ["FUZZ_BLOCKS"]), v[3], v[4]) for k, v in CONFIGS.items()}


def params_of(c):
    rl, dffn, nb, il, al = c
    return ModelParameters(rl, dffn, nb, il, al, 4, [10, 10], list(P0_CNN), 0.0, 0.3)


if sys.argv[1] == "shapes":
    from permutect_amd.architecture.artifact_model import ArtifactModel
    from permutect_amd.engine import instances as I
    from permutect_amd.engine.plan import EnginePlan, ParamSpace
    os.environ["PMT_JIT"] = "0"
    for name, c in CONFIGS.items():
        model = ArtifactModel(params_of(c), device=torch.device("cpu"), **P0_DIMS)
        d = EnginePlan(model, ParamSpace(model, torch.device("cpu")), torch.device("cpu")).desc
        print(name, " ".join(str(v) for v in I.exact_shape_of(d)))
    sys.exit(0)

from oracle import artifact_oracle as O
from permutect_amd.architecture.artifact_model import ArtifactModel
from permutect_amd.data.batch import Batch
from tests.test_forward_gpu import _arrays
os.environ["PMT_JIT"] = "0"
dev = torch.device("cuda")
for name, c in CONFIGS.items():
    rl, dffn, nb, il, al = c
    cfg = O.Config(rl, il, al, dffn, nb, 4, list(P0_CNN), 61, 71, 42)
    for shape in ("", "bf16x3", "tile", "any"):
        os.environ["PMT_SHAPE"] = shape
        torch.manual_seed(6)
        model = ArtifactModel(params_of(c), device=dev, **P0_DIMS)
        with torch.no_grad():
            for q in model.parameters():
                q.add_(0.05 * torch.randn_like(q))
        sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        names = [n for n, _ in model.named_parameters()]
        for deep in (0, 1):
            nref, nalt = (np.array([5, 330, 2, 40]), np.array([3, 280, 9, 600])) if deep else (np.array([5, 33, 2, 40, 0, 7, 12, 1]), np.array([3, 28, 9, 60, 4, 1, 15, 2]))
            ints, floats, packed = _arrays(nref, nalt, seed=81 + deep)
            batch = Batch.from_arrays(ints, floats, packed).copy_to(dev)
            res = {}
            for train in (True, False):
                model.train(train)
                model.zero_grad()
                model.engine().space.gtheta.zero_()  # (the flat gradient buffer the .grad views alias: FusedClipAdamW.zero_grad does this)
                with torch.set_grad_enabled(train):
                    out = model.compute_batch_output(batch)
                    if train:
                        model.compute_batch_losses(out, batch).total_loss.backward()
                torch.cuda.synchronize()
                res[train] = out.logits_b.detach().cpu().double()
                if train:
                    gh = np.concatenate([p.grad.detach().cpu().numpy().ravel().astype(np.float64) for _, p in model.named_parameters()])
            i64 = torch.from_numpy(ints.astype(np.int64))
            ob = dict(reads_re=torch.from_numpy(O.decode_packed_reads(packed).astype(np.float32)), nref=i64[:, O.REF_COUNT], nalt=i64[:, O.ALT_COUNT],
                      labels=i64[:, O.LABEL], sources=i64[:, O.SOURCE], info_be=torch.from_numpy(floats[:, O.INFO_START:].astype(np.float32)),
                      haplotypes_bh=i64[:, O.HAPLOTYPES_START:])
            O.COMPUTE_DTYPE = torch.float32
            o32, _, g32d = O.train_step_grads(sd, cfg, ob)
            O.COMPUTE_DTYPE = torch.float64
            o64, _, g64d = O.train_step_grads({k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}, cfg, ob)
            O.COMPUTE_DTYPE = torch.float32
            g64 = np.concatenate([g64d[n].numpy().ravel() for n in names])
            g32 = np.concatenate([g32d[n].numpy().ravel().astype(np.float64) for n in names])
            l64 = o64["logits_b"]
            print(f"{name:16s} {shape or 'auto':7s} id {model.engine().shape_id} deep {deep}: logit err vs fp64 train {float((res[True] - l64).abs().max()):.2e} eval {float((res[False] - l64).abs().max()):.2e} "
                  f"(fp32 oracle {float((o32['logits_b'].double() - l64).abs().max()):.2e});  grad rel {np.linalg.norm(gh - g64) / np.linalg.norm(g64):.2e} (fp32 oracle {np.linalg.norm(g32 - g64) / np.linalg.norm(g64):.2e})", flush=True)
