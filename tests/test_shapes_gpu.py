"""Other model shapes than the production one, each on the kernel instances compiled for ITS tile counts (csrc/Makefile:
INSTANCE_TABLE, built by __graft_entry__.build()): every instance kind, ordinary and split read sets, forward and every gradient
against the oracle evaluated in fp64 (the yardstick: how far fp32 arithmetic itself is from the exact result is printed beside it).

Shape A (tiles 4, 3, 5, 2) is the regression of round 4's "race": on read sets split unevenly between ref and alt over several
workgroups its 16-bit backward gave O(1)-wrong gradients in seven runs of eight, with a handful of values repeating bit for bit.
The cause: d(u) = W2^T d(y) contracts FIVE tiles of d(y) = two 32-deep k blocks and a half-filled one, and the half-filled block ran
as `v_mfma_f32_16x16x16_bf16` accumulating onto the register the 32-deep `v_mfma_f32_16x16x32_bf16` before it writes.  The hardware
interlocks an accumulation chain of one opcode only and hipcc 7.2 puts no wait state between the two, so whenever the pair issued
back to back -- which depends on what the SIMD's other wave is doing, hence on how full the group's waves are -- the 16-deep MFMA
read a stale accumulator (scripts/microbench/mfma_chain_hazard.hip shows it in isolation).  The half-filled block now takes the
32-deep MFMA on zero-padded operands (pmt_device.hpp: linear_acc_bf16), like the forward's f16 products since round 4.  The
repeated-run test below runs the three splits the verdict names twenty times each."""
import numpy as np
import pytest
import torch

from oracle import artifact_oracle as O
from permutect_amd.architecture.artifact_model import ArtifactModel
from permutect_amd.data.batch import Batch
from permutect_amd.parameters import P0_CNN, P0_DIMS, ModelParameters
from tests.test_forward_gpu import _arrays

pytestmark = pytest.mark.gpu

# name: (read layers, d_ffn, blocks, info layers, aggregation layers) -- scripts/shape_fuzz.py
CONFIGS = {
    "A_4_3_5_2": ([40, -2], 24, 3, [20, -1], [-1, 24]),
    "B_4_1_3_1_h20": ([16], 40, 2, [8], [12]),
    "C_4_4_8_2_h32": ([64, -2], 64, 2, [40], [-2, 32]),
    "D_4_2_6_1": ([30, -2], 20, 4, [50, -1], [-2, 10]),
}
KINDS = ["auto", "bf16x3", "tile", "any"]


def _params(c):
    rl, dffn, nb, il, al = c
    return ModelParameters(rl, dffn, nb, il, al, 4, [10, 10], list(P0_CNN), 0.0, 0.3)


def _config(c):
    rl, dffn, nb, il, al = c
    return O.Config(rl, il, al, dffn, nb, 4, list(P0_CNN), 61, 71, 42)


def _model(c, kind, monkeypatch):
    if kind != "auto":
        monkeypatch.setenv("PMT_SHAPE", kind)
    torch.manual_seed(6)
    model = ArtifactModel(_params(c), device=torch.device("cuda"), **P0_DIMS)
    with torch.no_grad():
        for q in model.parameters():
            q.add_(0.05 * torch.randn_like(q))  # (away from the initialisation: zero-initialised parameters hide their gradients' paths)
    return model


def _oracle(model, c, ints, floats, packed, dtype):
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    i64 = torch.from_numpy(ints.astype(np.int64))
    ob = dict(reads_re=torch.from_numpy(O.decode_packed_reads(packed).astype(np.float32)), nref=i64[:, O.REF_COUNT], nalt=i64[:, O.ALT_COUNT],
              labels=i64[:, O.LABEL], sources=i64[:, O.SOURCE], info_be=torch.from_numpy(floats[:, O.INFO_START:].astype(np.float32)),
              haplotypes_bh=i64[:, O.HAPLOTYPES_START:])
    old = O.COMPUTE_DTYPE
    O.COMPUTE_DTYPE = dtype
    try:
        out, _, grads = O.train_step_grads({k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}, _config(c), ob)
    finally:
        O.COMPUTE_DTYPE = old
    names = [n for n, _ in model.named_parameters()]
    return out["logits_b"].detach().double().numpy(), np.concatenate([grads[n].numpy().ravel().astype(np.float64) for n in names])


def _step(model, batch):
    model.train(True)
    model.zero_grad()
    model.engine().space.gtheta.zero_()  # (the flat gradient buffer the .grad views alias)
    out = model.compute_batch_output(batch)
    model.compute_batch_losses(out, batch).total_loss.backward()
    torch.cuda.synchronize()
    model.engine().check_join_fault()
    grads = np.concatenate([p.grad.detach().cpu().numpy().ravel().astype(np.float64) for _, p in model.named_parameters()])
    return out.logits_b.detach().cpu().double().numpy(), grads


def _require_instances(model, name, kind):
    """the exact instances must be the ones that run: a missing per-shape library would quietly test the generic instance four times"""
    sid = model.engine().shape_id
    if kind in ("auto", "bf16x3"):
        assert sid in (2, 6), f"{name}: no exact-width library for this shape (shape id {sid}); __graft_entry__.build() builds INSTANCE_TABLE"


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("deep", [False, True], ids=["ordinary", "split"])
@pytest.mark.parametrize("name", list(CONFIGS))
def test_shape_forward_and_gradients_against_fp64(name, deep, kind, monkeypatch):
    c = CONFIGS[name]
    model = _model(c, kind, monkeypatch)
    _require_instances(model, name, kind)
    nref, nalt = ((np.array([5, 330, 2, 40]), np.array([3, 280, 9, 600])) if deep
                  else (np.array([5, 33, 2, 40, 0, 7, 12, 1]), np.array([3, 28, 9, 60, 4, 1, 15, 2])))
    ints, floats, packed = _arrays(nref, nalt, seed=81 + int(deep))
    batch = Batch.from_arrays(ints, floats, packed).copy_to(torch.device("cuda"))
    assert batch.plan(allow_split=True).layered == deep
    logits, grads = _step(model, batch)
    model.train(False)
    with torch.no_grad():
        logits_eval = model.compute_batch_output(batch).logits_b.detach().cpu().double().numpy()
    l64, g64 = _oracle(model, c, ints, floats, packed, torch.float64)
    l32, g32 = _oracle(model, c, ints, floats, packed, torch.float32)
    rel, rel32 = np.linalg.norm(grads - g64) / np.linalg.norm(g64), np.linalg.norm(g32 - g64) / np.linalg.norm(g64)
    err, err_eval, err32 = np.abs(logits - l64).max(), np.abs(logits_eval - l64).max(), np.abs(l32 - l64).max()
    print(f"{name} {kind} split={deep}: logit err vs fp64 train {err:.2e} eval {err_eval:.2e} (fp32 oracle {err32:.2e}); grad rel {rel:.2e} (fp32 oracle {rel32:.2e})")
    # the contract: 1e-4 on the capped logits; deep sets (summed log-likelihoods in the thousands) get the fp32 oracle's own distance on top
    tol = 1e-4 + (2.0 * err32 if deep else 0.0)
    assert err <= tol and err_eval <= tol, (err, err_eval, err32)
    assert np.all(np.isfinite(grads)) and rel <= 1e-4, (rel, rel32)


SPLITS = {"10+300": ([10], [300]), "300+10": ([300], [10]), "1+600": ([1], [600]), "10+300 beside small sets": ([3, 10, 7], [2, 300, 5])}


@pytest.mark.parametrize("blocks", [1, 3])
@pytest.mark.parametrize("split", list(SPLITS))
def test_uneven_splits_of_shape_a_twenty_runs(split, blocks, monkeypatch):
    """Round 4's failing case and its relatives, on the 16-bit instances, twenty consecutive steps each: every run within 1e-4 of the
    fp64 gradient, and the runs within float-atomic noise of each other."""
    rl, dffn, _, il, al = CONFIGS["A_4_3_5_2"]
    c = (rl, dffn, blocks, il, al)
    model = _model(c, "auto", monkeypatch)
    _require_instances(model, "A", "auto")
    nref, nalt = (np.array(v) for v in SPLITS[split])
    ints, floats, packed = _arrays(nref, nalt, seed=81)
    batch = Batch.from_arrays(ints, floats, packed).copy_to(torch.device("cuda"))
    assert batch.plan(allow_split=True).layered
    _, g64 = _oracle(model, c, ints, floats, packed, torch.float64)
    rels, first = [], None
    for _ in range(20):
        _, g = _step(model, batch)
        rels.append(float(np.linalg.norm(g - g64) / np.linalg.norm(g64)))
        first = g if first is None else first
        assert np.linalg.norm(g - first) <= 1e-5 * np.linalg.norm(first), rels
    print(f"A x {blocks} blocks, {split}: gradient rel. L2 vs fp64 over 20 runs: min {min(rels):.2e} max {max(rels):.2e}")
    assert max(rels) <= 1e-4, rels


def test_production_shape_uneven_split_fifty_steps_are_stable():
    """The production shape on an unevenly split batch (the stress configuration's code path), fifty consecutive steps: every step
    within float-atomic noise of the first (the per-set sums and the small-parameter gradients are float atomics; nothing else moves)."""
    from permutect_amd.parameters import p0_params
    torch.manual_seed(7)
    model = ArtifactModel(p0_params(), device=torch.device("cuda"), **P0_DIMS)
    with torch.no_grad():
        for q in model.parameters():
            q.add_(0.05 * torch.randn_like(q))
    nref, nalt = np.array([4, 10, 300, 7, 1, 9]), np.array([6, 300, 10, 2, 600, 5])
    ints, floats, packed = _arrays(nref, nalt, seed=83)
    batch = Batch.from_arrays(ints, floats, packed).copy_to(torch.device("cuda"))
    assert batch.plan(allow_split=True).layered and model.engine().shape_id == 2
    c = ([30, -2, -2, -2], 20, 6, [20, -2, -2, -2], [-2, -2, 10])
    _, g64 = _oracle(model, c, ints, floats, packed, torch.float64)
    first, worst = None, 0.0
    for _ in range(50):
        _, g = _step(model, batch)
        first = g if first is None else first
        worst = max(worst, float(np.linalg.norm(g - first) / np.linalg.norm(first)))
    rel = float(np.linalg.norm(first - g64) / np.linalg.norm(g64))
    print(f"P0 uneven split: run-to-run {worst:.2e}, vs fp64 {rel:.2e}")
    assert worst <= 2e-6 and rel <= 1e-4, (worst, rel)
