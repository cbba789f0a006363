"""GPU parity of one training step: fused HIP forward + backward (through autograd) + fused clip/AdamW against the
reference's gradients and post-step parameters (golden fixtures).  Tolerances: global gradient vector
1e-4 relative in L2; each tensor 5e-4 of max(its own scale, 1e-3 of the global scale); parameters after the step 1e-5."""
import numpy as np
import pytest
import torch

from permutect_amd.data.batch import Batch
from permutect_amd.training.optimizer import FusedClipAdamW, backpropagate
from tests.helpers import CASES, load_case
from tests.test_forward_gpu import build

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["auto", "tile", "any"], autouse=True)
def kernel_shape(request, monkeypatch):
    """Every case runs three times: with the kernel instance the library picks for the model (for the P0 fixtures ShapeP0X,
    which has the production widths compiled in), with the tile-exact instance (PMT_SHAPE=tile: ShapeP0, widths read at run
    time) and with the generic instance (PMT_SHAPE=any); the variable is read ONCE, when the model is lowered (engine/plan.py:
    PmtModel.force_shape / force_cnn), which is why the fixture sets it before the model is built."""
    if request.param == "tile":
        monkeypatch.setenv("PMT_SHAPE", "tile")
        monkeypatch.setenv("PMT_CNN_STASH", "0")  # and the haplotype-CNN backward that recomputes its forward
    if request.param == "any":
        monkeypatch.setenv("PMT_SHAPE", "any")
        monkeypatch.setenv("PMT_CNN", "general")  # and the general (workgroup-per-chunk) haplotype-CNN kernels


def run_step(name, fmt="packed"):
    z, sd, b = load_case(name)
    model, dev = build(name, sd)
    model.train(True)
    if "source_strength" in z.files:
        model.source_predictor.set_adversarial_strength(float(z["source_strength"]))
    batch = Batch.from_arrays(b["int_array"], b["float_array"], b["packed_reads"]).copy_to(dev)
    out = model.compute_batch_output(batch)
    losses = model.compute_batch_losses(out, batch)
    return z, model, out, losses


@pytest.mark.parametrize("name", CASES)
def test_losses_match_reference(name):
    z, model, out, losses = run_step(name)
    for k in ("supervised_losses_b", "unsupervised_losses_b", "alt_count_losses_b", "source_prediction_losses_b",
              "total_losses_b"):
        ref = z["loss/" + k]
        np.testing.assert_allclose(getattr(losses, k).detach().cpu().numpy(), ref, rtol=1e-4, atol=1e-4 + 1e-5 * np.abs(ref).max(), err_msg=k)


@pytest.mark.parametrize("name", CASES)
def test_gradients_match_reference(name):
    z, model, out, losses = run_step(name)
    opt = FusedClipAdamW(model, lr=float(z["lr"]), weight_decay=float(z["weight_decay"]))
    opt.zero_grad()
    losses.total_loss.backward()
    torch.cuda.synchronize()
    names = [n for n, _ in model.named_parameters()]
    assert set(names) == {k[5:] for k in z.files if k.startswith("grad/")}
    gref = np.concatenate([z["grad/" + n].ravel() for n in names])
    gour = np.concatenate([p.grad.detach().cpu().numpy().ravel() for _, p in model.named_parameters()])
    assert np.all(np.isfinite(gour))
    gscale = np.abs(gref).max()
    bad = []
    for n, p in model.named_parameters():
        ref = z["grad/" + n]
        err = np.abs(p.grad.detach().cpu().numpy() - ref).max()
        tol = 5e-4 * max(np.abs(ref).max(), 1e-3 * gscale)
        if err > tol:
            bad.append((n, float(err), float(np.abs(ref).max())))
    assert not bad, bad[:12]
    assert np.linalg.norm(gour - gref) <= 1e-4 * np.linalg.norm(gref)


@pytest.mark.parametrize("name", ["t0_b8", "p0_b16", "p0_deep"])
def test_fused_clip_adamw_kernel_on_reference_gradients(name):
    """The optimizer kernel alone: fed the reference's raw gradients it must reproduce the reference's parameters after
    clip_grad_norm_(1.0) + AdamW.step to fp32 rounding."""
    z, sd, b = load_case(name)
    model, dev = build(name, sd)
    opt = FusedClipAdamW(model, lr=float(z["lr"]), weight_decay=float(z["weight_decay"]))
    opt.zero_grad()
    with torch.no_grad():
        for n, p in model.named_parameters():
            p.grad.copy_(torch.from_numpy(z["grad/" + n]))
    opt.step()
    torch.cuda.synchronize()
    ref_norm = float(np.sqrt(sum((z["grad/" + n].astype(np.float64) ** 2).sum() for n, _ in model.named_parameters())))
    assert abs(float(opt.grad_norm.item()) - ref_norm) <= 1e-5 * ref_norm
    for n, p in model.named_parameters():
        np.testing.assert_allclose(p.detach().cpu().numpy(), z["after/" + n], rtol=1e-5, atol=1e-6, err_msg=n)


@pytest.mark.parametrize("name", ["t0_b8", "p0_b16", "p0_deep"])
def test_full_train_step_matches_reference(name):
    """forward + losses + backward + clip + AdamW end to end.  The first Adam step is lr * g / (|g| + eps): elements whose
    gradient is within a few orders of eps = 1e-8 amplify fp32 gradient noise (the step flips from 0 to +-lr across
    |g| ~ eps).  So: elements with a clipped reference gradient above 1e-6 must agree to 5% of one step, and no element
    may be off by more than 10% of a step."""
    z, model, out, losses = run_step(name)
    lr = float(z["lr"])
    opt = FusedClipAdamW(model, lr=lr, weight_decay=float(z["weight_decay"]))
    backpropagate(opt, losses.total_loss, params_to_clip=model.parameters())
    torch.cuda.synchronize()
    ref_norm = float(np.sqrt(sum((z["grad/" + n].astype(np.float64) ** 2).sum() for n, _ in model.named_parameters())))
    assert abs(float(opt.grad_norm.item()) - ref_norm) <= 1e-4 * ref_norm
    clip = min(1.0, 1.0 / (ref_norm + 1e-6))
    worst = worst_big = 0.0
    for n, p in model.named_parameters():
        err = np.abs(p.detach().cpu().numpy() - z["after/" + n])
        big = np.abs(z["grad/" + n]) * clip > 1e-6
        worst = max(worst, float(err.max()))
        if big.any():
            worst_big = max(worst_big, float(err[big].max()))
    assert worst_big <= 0.05 * lr and worst <= 0.10 * lr, (worst_big, worst)


@pytest.mark.parametrize("name", ["t0_b8", "p0_b16", "p0_saturated"])
def test_phi_kernels_match_torch_parametrizations(name):
    """pmt_phi_forward / pmt_phi_backward (one launch each) against torch.nn.utils.parametrize + autograd for every
    parametrized tensor, including torch's orthogonal (matrix_exp with trivialization) and its adjoint."""
    from permutect_amd.engine.runtime import PhiFunction
    z, sd, b = load_case(name)
    model, dev = build(name, sd)
    model.train(True)
    eng = model.engine()
    torch.manual_seed(11)
    with torch.no_grad():  # move the rotation away from its initial point so that expm and its adjoint are exercised
        model.pre_clustering_transform.rotation_ee.parametrizations.weight.original.add_(
            0.3 * torch.randn_like(model.pre_clustering_transform.rotation_ee.parametrizations.weight.original))
    prog = eng.plan.phi_program(model)
    assert prog is not None
    phi_ref = eng.plan.materialize_phi(model)
    phi = PhiFunction.apply(eng, prog, eng.trigger)
    np.testing.assert_allclose(phi.detach().cpu().numpy(), phi_ref.detach().cpu().numpy(), rtol=2e-6, atol=2e-6)
    q = phi.detach()[eng.plan.phi_layout[len(model.ref_alt_reads_encoder.blocks)][1]:][:model.reducer.output_dimension() ** 2]
    q = q.view(model.reducer.output_dimension(), -1)
    assert torch.allclose(q @ q.T, torch.eye(q.shape[0], device=q.device), atol=1e-5)
    gphi = torch.randn_like(phi_ref)
    originals = [p for n, p in model.named_parameters() if n.endswith(".original")]
    gref = torch.autograd.grad(phi_ref, originals, gphi)
    eng.space.gtheta.zero_()
    phi.backward(gphi)
    torch.cuda.synchronize()
    for p, g in zip(originals, gref):
        np.testing.assert_allclose(p.grad.detach().cpu().numpy(), g.cpu().numpy(), rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("name", ["p0_b16", "t0_two_sources", "p0_saturated"])
def test_fused_losses_match_torch_composition(name):
    """pmt_losses_forward / _backward against the reference's torch composition (BCEWithLogits, clip, logsumexp, MSE on a
    sigmoid, squared error on a softmax) with arbitrary upstream gradients on all five loss vectors and non-unit weights."""
    z, model, out, _ = run_step(name)
    z2, sd, b = load_case(name)
    batch = Batch.from_arrays(b["int_array"], b["float_array"], b["packed_reads"]).copy_to(out.logits_b.device)
    g = torch.Generator(device="cpu").manual_seed(3)
    n = out.logits_b.shape[0]
    out.weights = (0.5 + torch.rand(n, generator=g)).to(out.logits_b.device)
    out.source_weights = (0.5 + torch.rand(n, generator=g)).to(out.logits_b.device)
    ups = [torch.randn(n, generator=g).to(out.logits_b.device) for _ in range(5)]
    results = []
    for fn in (model.compute_batch_losses, model.compute_batch_losses_torch):
        leaves = [t.detach().clone().requires_grad_(True) for t in (out.logits_b, out.logits_bk, out.features_be)]
        o = type(out)(features_be=leaves[2], ref_features_be=out.ref_features_be, logits_b=leaves[0], logits_bk=leaves[1],
                      weights=out.weights, source_weights=out.source_weights)
        model.engine().space.gtheta.zero_()
        losses = fn(o, batch)
        vecs = [losses.supervised_losses_b, losses.unsupervised_losses_b, losses.alt_count_losses_b,
                losses.source_prediction_losses_b, losses.total_losses_b]
        scalar = sum((u * v).sum() for u, v in zip(ups, vecs))
        scalar.backward()
        results.append(([v.detach().cpu().numpy() for v in vecs], [t.grad.cpu().numpy() for t in leaves]))
    (v1, g1), (v2, g2) = results
    for a, c in zip(v1, v2):
        np.testing.assert_allclose(a, c, rtol=2e-5, atol=2e-6)
    for a, c in zip(g1, g2):
        np.testing.assert_allclose(a, c, rtol=2e-4, atol=2e-6)


def _compare_gradients(model, gref_by_name):
    names = [n for n, _ in model.named_parameters()]
    gref = np.concatenate([gref_by_name[n].ravel() for n in names])
    gour = np.concatenate([p.grad.detach().cpu().numpy().ravel() for _, p in model.named_parameters()])
    assert np.all(np.isfinite(gour))
    gscale = np.abs(gref).max()
    bad = []
    for n, p in model.named_parameters():
        ref = gref_by_name[n]
        err = np.abs(p.grad.detach().cpu().numpy() - ref).max()
        if err > 5e-4 * max(np.abs(ref).max(), 1e-3 * gscale):
            bad.append((n, float(err), float(np.abs(ref).max())))
    assert not bad, bad[:12]
    assert np.linalg.norm(gour - gref) <= 1e-4 * np.linalg.norm(gref)


def test_training_on_read_sets_beyond_one_workgroup_matches_oracle():
    """BASELINE config 'mean 600 reads per variant' in training: read sets split over several workgroups, forward and
    backward as num_blocks + 1 launches each (pmt_forward_layered / pmt_backward_layered); losses and every parameter
    gradient against the oracle's autograd on the same inputs."""
    from oracle import artifact_oracle as O
    from tests.helpers import config_for
    from tests.test_forward_gpu import _arrays
    _, sd, _ = load_case("p0_b16")
    cfg = config_for("p0_b16")
    nref = np.array([5, 300, 0, 10, 700, 2, 256, 9])
    nalt = np.array([3, 350, 600, 15, 1, 7, 255, 1])
    ints, floats, packed = _arrays(nref, nalt, seed=21)
    model, dev = build("p0_b16", sd)
    model.train(True)
    batch = Batch.from_arrays(ints, floats, packed).copy_to(dev)
    out = model.compute_batch_output(batch)
    assert batch.plan(allow_split=True).layered
    losses = model.compute_batch_losses(out, batch)
    opt = FusedClipAdamW(model, lr=1e-3, weight_decay=0.01)
    opt.zero_grad()
    losses.total_loss.backward()
    torch.cuda.synchronize()
    i64 = torch.from_numpy(ints.astype(np.int64))
    ob = dict(reads_re=torch.from_numpy(O.decode_packed_reads(packed).astype(np.float32)), nref=i64[:, O.REF_COUNT],
              nalt=i64[:, O.ALT_COUNT], labels=i64[:, O.LABEL], sources=i64[:, O.SOURCE],
              info_be=torch.from_numpy(floats[:, O.INFO_START:].astype(np.float32)), haplotypes_bh=i64[:, O.HAPLOTYPES_START:])
    _, ref_losses, ref_grads = O.train_step_grads(sd, cfg, ob)
    ref_total = ref_losses["total_losses_b"].detach().numpy()
    np.testing.assert_allclose(losses.total_losses_b.detach().cpu().numpy(), ref_total, rtol=1e-4, atol=1e-4 + 1e-5 * np.abs(ref_total).max())
    _compare_gradients(model, {k: v.numpy() for k, v in ref_grads.items()})


def test_layered_backward_on_forced_splits_matches_reference():
    """The layered forward + backward on a fixture batch whose read sets are cut into many small groups (at most two
    tiles of ref rows and two of alt rows each, so every larger set spans several workgroups): the reference's gradients."""
    from permutect_amd.data.batch import GroupPlan
    z, sd, b = load_case("p0_deep")
    model, dev = build("p0_deep", sd)
    model.train(True)
    batch = Batch.from_arrays(b["int_array"], b["float_array"], b["packed_reads"]).copy_to(dev)
    nref, nalt = batch.host_counts()
    plan = GroupPlan(nref, nalt, allow_split=True)
    spans, tiles, tile_base = [], 0, []
    ro, ao = np.concatenate([[0], np.cumsum(nref)]), np.concatenate([[0], np.cumsum(nalt)])
    for v in range(len(nref)):
        r, a, first = int(ro[v]), int(ao[v]), True
        while first or r < ro[v + 1] or a < ao[v + 1]:
            first = False
            r1, a1 = min(r + 32, int(ro[v + 1])), min(a + 32, int(ao[v + 1]))
            if r < ro[v + 1]:  # ref rows first, alt rows start once the ref rows are done (as the planner does) ...
                a1 = a if r1 < ro[v + 1] else a1
            spans.append([v, v + 1, r, r1, a, a1])
            tile_base.append(tiles)
            tiles += (r1 - r + 15) // 16 + (a1 - a + 15) // 16
            r, a = r1, a1
    tile_base.append(tiles)
    plan.use_span(np.array(spans, dtype=np.int32), np.array(tile_base, dtype=np.int32), len(nref))
    assert plan.num_groups > 2 * len(nref) and plan.set_groups.max() > 2
    batch._plan = plan
    out = model.compute_batch_output(batch)
    losses = model.compute_batch_losses(out, batch)
    opt = FusedClipAdamW(model, lr=float(z["lr"]), weight_decay=float(z["weight_decay"]))
    opt.zero_grad()
    losses.total_loss.backward()
    torch.cuda.synchronize()
    ref = z["loss/total_losses_b"]
    np.testing.assert_allclose(losses.total_losses_b.detach().cpu().numpy(), ref, rtol=1e-4, atol=1e-4 + 1e-5 * np.abs(ref).max())
    _compare_gradients(model, {k[5:]: z[k] for k in z.files if k.startswith("grad/")})


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_mixed_batches_match_oracle(seed):
    """Randomly mixed batches -- singletons, zero-ref sets, ordinary sets, sets that fill a workgroup exactly and sets of
    several hundred reads (split over workgroups) side by side, random labels: losses and every parameter gradient against
    the oracle's autograd."""
    from oracle import artifact_oracle as O
    from tests.helpers import config_for
    from tests.test_forward_gpu import _arrays
    rng = np.random.default_rng(100 + seed)
    kinds = rng.integers(0, 6, 28)
    nref = np.where(kinds == 0, 0, np.where(kinds == 1, 1, np.where(kinds == 2, rng.integers(0, 11, 28), np.where(
        kinds == 3, 128, np.where(kinds == 4, rng.integers(100, 420, 28), rng.integers(0, 60, 28))))))
    nalt = np.where(kinds == 0, 1, np.where(kinds == 1, 1, np.where(kinds == 2, rng.integers(1, 16, 28), np.where(
        kinds == 3, 128, np.where(kinds == 4, rng.integers(1, 420, 28), rng.integers(1, 60, 28))))))
    _, sd, _ = load_case("p0_b16")
    cfg = config_for("p0_b16")
    ints, floats, packed = _arrays(nref, nalt, seed=200 + seed)
    model, dev = build("p0_b16", sd)
    model.train(True)
    batch = Batch.from_arrays(ints, floats, packed).copy_to(dev)
    out = model.compute_batch_output(batch)
    losses = model.compute_batch_losses(out, batch)
    opt = FusedClipAdamW(model, lr=1e-3, weight_decay=0.01)
    opt.zero_grad()
    losses.total_loss.backward()
    torch.cuda.synchronize()
    i64 = torch.from_numpy(ints.astype(np.int64))
    ob = dict(reads_re=torch.from_numpy(O.decode_packed_reads(packed).astype(np.float32)), nref=i64[:, O.REF_COUNT],
              nalt=i64[:, O.ALT_COUNT], labels=i64[:, O.LABEL], sources=i64[:, O.SOURCE],
              info_be=torch.from_numpy(floats[:, O.INFO_START:].astype(np.float32)), haplotypes_bh=i64[:, O.HAPLOTYPES_START:])
    _, ref_losses, ref_grads = O.train_step_grads(sd, cfg, ob)
    ref_total = ref_losses["total_losses_b"].detach().numpy()
    np.testing.assert_allclose(losses.total_losses_b.detach().cpu().numpy(), ref_total, rtol=1e-4, atol=1e-4 + 1e-5 * np.abs(ref_total).max())
    _compare_gradients(model, {k: v.numpy() for k, v in ref_grads.items()})


def test_zero_adversarial_strength_sends_no_gradient_upstream():
    """Epoch 1 of train_artifact_model sets the source adversary's strength to 2 / (1 + e^0) - 1 = 0 (reference
    training/model_training.py:92); GradientReversal then returns -0 * g = 0 to the features (reference
    gradient_reversal/functional.py:18-22) while the adversary's own parameters still learn.  `None` (the info MLP) passes
    the gradient through unchanged."""
    from permutect_amd.engine import lib as L
    from permutect_amd.engine.runtime import RowsMlpFunction
    z, sd, b = load_case("t0_two_sources")
    model, dev = build("t0_two_sources", sd)
    model.train(True)
    eng = model.engine()
    eng.pack(eng.plan.materialize_phi(model).detach().contiguous())
    torch.manual_seed(5)
    feats = torch.randn(24, model.reducer.output_dimension(), device=dev)
    grads = {}
    for alpha in (0.0, 0.3):
        leaf = feats.clone().requires_grad_(True)
        eng.space.gtheta.zero_()
        out = RowsMlpFunction.apply(eng, L.ROWS_SOURCE, leaf, eng.trigger, alpha)
        out.square().sum().backward()
        torch.cuda.synchronize()
        grads[alpha] = (leaf.grad.clone(), eng.space.gtheta.clone())
    assert torch.count_nonzero(grads[0.0][0]) == 0               # nothing flows upstream at strength 0
    assert torch.count_nonzero(grads[0.0][1]) > 0                # ... but the adversary's parameters get their gradient
    assert torch.allclose(grads[0.0][1], grads[0.3][1])          # (which does not depend on the strength)
    assert torch.count_nonzero(grads[0.3][0]) > 0
    # `None` passes the gradient through unchanged: reversal at 0.3 == -0.3 x the un-reversed gradient of the same head
    leaf = feats.clone().requires_grad_(True)
    out = RowsMlpFunction.apply(eng, L.ROWS_SOURCE, leaf, eng.trigger, None)
    out.square().sum().backward()
    torch.cuda.synchronize()
    assert torch.allclose(grads[0.3][0], -0.3 * leaf.grad, rtol=1e-6, atol=1e-7)


def test_check_for_nan_fails_the_run_on_a_poisoned_gradient(capsys):
    """reference misc_utils.py:159-166 (called after every epoch, model_training.py:168): names the parameter, then asserts."""
    from permutect_amd.training.model_training import check_for_nan
    z, sd, b = load_case("t0_b8")
    model, dev = build("t0_b8", sd)
    model.train(True)
    batch = Batch.from_arrays(b["int_array"], b["float_array"], b["packed_reads"]).copy_to(dev)
    opt = FusedClipAdamW(model, lr=1e-3, weight_decay=0.01)
    opt.zero_grad()
    model.compute_batch_losses(model.compute_batch_output(batch), batch).total_loss.backward()
    check_for_nan(model)  # a healthy step passes
    name, param = next((n, p) for n, p in model.named_parameters() if n.startswith("reducer."))
    param.grad.view(-1)[0] = float("inf")
    with pytest.raises(AssertionError):
        check_for_nan(model)
    assert name in capsys.readouterr().out
