"""GPU parity: the fused HIP forward (through the C ABI) against the reference's outputs (golden fixtures) and the
CPU oracle on the same inputs.  Tolerances: capped logits 1e-4 abs (the north-star contract), everything else 2e-5
relative to the tensor's scale; deep sets (|log-likelihood| ~ 1e3) get 4 ulp of their magnitude."""
import numpy as np
import pytest
import torch

from oracle import artifact_oracle as O
from permutect_amd.architecture.artifact_model import ArtifactModel
from permutect_amd.data.batch import Batch
from permutect_amd.parameters import P0_DIMS, p0_params, t0_params
from tests.helpers import CASES, config_for, load_case

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["auto", "tile", "any"], autouse=True)
def kernel_shape(request, monkeypatch):
    """Every case runs three times: with the kernel instance the library picks for the model (for the P0 fixtures ShapeP0X,
    which has the production widths compiled in), with the tile-exact instance (PMT_SHAPE=tile: ShapeP0, widths read at run
    time) and with the generic instance (PMT_SHAPE=any); the library reads the variable at every launch."""
    if request.param == "tile":
        monkeypatch.setenv("PMT_SHAPE", "tile")
        monkeypatch.setenv("PMT_CNN_STASH", "0")  # and the haplotype-CNN backward that recomputes its forward
    if request.param == "any":
        monkeypatch.setenv("PMT_SHAPE", "any")
        monkeypatch.setenv("PMT_CNN", "general")  # and the general (workgroup-per-chunk) haplotype-CNN kernels


def build(name, sd):
    dev = torch.device("cuda")
    params = t0_params() if name.startswith("t0") else p0_params()
    model = ArtifactModel(params, device=dev, **P0_DIMS)
    if name == "t0_two_sources":
        model.reset_source_predictor(2)
    model.load_state_dict(sd)
    return model, dev


def _record(name, logit_err, logit_over_tol, lk_over_tol):
    """measured errors per fixture, for DESIGN.md section 2 (gpurun_out/parity_errors.jsonl)"""
    import json
    import os
    try:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        with open(os.path.join(root, "gpurun_out", "parity_errors.jsonl"), "a") as f:
            f.write(json.dumps({"test": "forward_fixture", "case": name, "instance": os.environ.get("PMT_SHAPE", "auto"),
                                "max_logit_err": logit_err, "max_logit_err_over_tol": logit_over_tol, "max_lk_err_over_tol": lk_over_tol}) + "\n")
    except OSError:
        pass


def check_outputs(out, z, name, lk_ulps=8):
    lk = out.logits_bk.cpu().numpy()
    ref_lk = z["out/logits_bk"]
    mag = np.abs(ref_lk).max(axis=1)
    tol_b = 1e-4 + 4 * np.spacing(mag.astype(np.float32))  # fp32 resolution of the summed log-likelihoods
    logit_err = np.abs(out.logits_b.cpu().numpy() - z["out/logits_b"])
    assert np.all(logit_err <= tol_b), name
    # every summed log-likelihood on ITS OWN scale: 2e-5 + 8 ulp of that element (a cluster's 5.0 next to another's 2000.0 is
    # held to 2e-5, not to 2e-5 of the batch maximum)
    # (sums over several hundred reads carry ~sqrt(N) ulp of rounding in ANY summation order, the oracle's included: the
    #  tests with 300-700-read sets pass lk_ulps = 16)
    lk_tol = 2e-5 + lk_ulps * np.spacing(np.abs(ref_lk).astype(np.float32))
    assert np.all(np.abs(lk - ref_lk) <= lk_tol), (name, float((np.abs(lk - ref_lk) / lk_tol).max()))
    _record(name, float(logit_err.max()), float((logit_err / tol_b).max()), float((np.abs(lk - ref_lk) / lk_tol).max()))
    for k, t in (("features_be", out.features_be), ("ref_features_be", out.ref_features_be),
                 ("artifact_probs_b", out.artifact_probs_b)):
        ref = z["out/" + k]
        np.testing.assert_allclose(t.cpu().numpy(), ref, rtol=2e-5, atol=2e-5 * max(1.0, float(np.abs(ref).max())), err_msg=k)
    ref = z["out/outlier_binary_logits"]
    # a difference of two summed log-likelihoods (each good to a few ulp of ITS magnitude; the per-set sums are float atomics,
    # so their rounding varies from run to run): 8 ulp of the larger one
    np.testing.assert_allclose(out.outlier_binary_logits.cpu().numpy(), ref, rtol=2e-5, atol=1e-4 + 8 * np.spacing(mag.astype(np.float32)).max())


def test_eval_forward_of_a_dropout_model_matches_reference():
    """A reference model built with dropout_p = 0.25, run in eval mode as filter_variants does (tests/golden/p0_dropout_eval.npz):
    the nn.Dropout modules only shift the state_dict keys; the forward is the dropout-free one."""
    z, sd, b = load_case("p0_dropout_eval")
    params = p0_params()
    params.dropout_p = float(z["dropout_p"])
    model = ArtifactModel(params, device=torch.device("cuda"), **P0_DIMS)
    model.load_state_dict(sd)
    model.eval()
    batch = Batch.from_arrays(b["int_array"], b["float_array"], b["packed_reads"]).copy_to(torch.device("cuda"))
    with torch.inference_mode():
        out = model.compute_batch_output(batch)
    check_outputs(out, z, "p0_dropout_eval")
    model.train(True)
    with pytest.raises(NotImplementedError, match="dropout"):
        model.compute_batch_output(batch)


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("fmt", ["packed", "f16", "f32"])
def test_forward_matches_reference(name, fmt):
    z, sd, b = load_case(name)
    model, dev = build(name, sd)
    reads = b["packed_reads"] if fmt == "packed" else z["reads_re_f16"]
    batch = Batch.from_arrays(b["int_array"], b["float_array"], reads).copy_to(dev, torch.float32 if fmt == "f32" else torch.float16)
    with torch.no_grad():
        out = model.compute_batch_output(batch)
    torch.cuda.synchronize()
    # (the deep fixture's sets are summed with float atomics whose order varies from run to run: measured 0.7 - 1.1 x the
    #  8-ulp bound, so it gets the 16 ulp of the other tests with long sums)
    check_outputs(out, z, name, lk_ulps=16 if "deep" in name else 8)


@pytest.mark.parametrize("name", ["t0_b8", "p0_b16"])
def test_forward_matches_oracle_on_fresh_inputs(name):
    """Seeded inputs that are NOT in the fixtures: the oracle is the checker."""
    _, sd, _ = load_case(name)
    cfg = config_for(name)
    rng = np.random.default_rng(123)
    nb = 300
    nref, nalt = rng.integers(0, 11, nb), rng.integers(1, 16, nb)
    ints = np.zeros((nb, 16 + 42), dtype=np.int16)
    ints[:, 0], ints[:, 1], ints[:, 2] = nref, nalt, rng.integers(0, 3, nb)
    ints[:, 16:] = rng.integers(0, 5, (nb, 42))
    floats = np.zeros((nb, 6 + 71), dtype=np.float16)
    floats[:, 6:] = rng.standard_normal((nb, 71)).astype(np.float16)
    packed = rng.integers(0, 256, (int(nref.sum() + nalt.sum()), 12), dtype=np.uint8)
    model, dev = build(name, sd)
    batch = Batch.from_arrays(ints, floats, packed).copy_to(dev)
    assert batch.plan().num_groups > 5
    with torch.no_grad():
        out = model.compute_batch_output(batch)
        ref = O.compute_batch_output(sd, cfg, torch.from_numpy(O.decode_packed_reads(packed).astype(np.float32)),
                                     torch.from_numpy(nref), torch.from_numpy(nalt),
                                     torch.from_numpy(floats[:, 6:].astype(np.float32)), torch.from_numpy(ints[:, 16:].astype(np.int64)))
    z = {"out/" + k: v.numpy() for k, v in ref.items()}
    check_outputs(out, z, name)


def _arrays(nref, nalt, seed=7):
    rng = np.random.default_rng(seed)
    nb = len(nref)
    ints = np.zeros((nb, 16 + 42), dtype=np.int16)
    ints[:, 0], ints[:, 1], ints[:, 2] = nref, nalt, rng.integers(0, 3, nb)
    ints[:, 16:] = rng.integers(0, 5, (nb, 42))
    floats = np.zeros((nb, 6 + 71), dtype=np.float16)
    floats[:, 6:] = rng.standard_normal((nb, 71)).astype(np.float16)
    packed = rng.integers(0, 256, (int(np.sum(nref) + np.sum(nalt)), 12), dtype=np.uint8)
    return ints, floats, packed


def test_sets_at_the_group_capacity_boundary_match_oracle():
    """Read sets that fill a whole workgroup (8 + 8 tiles, 16 + 0, 1 + 15 waves' worth), neighbours that must start a new
    group, zero-ref sets and single-read sets in one batch; forward and input-side gradients against the oracle."""
    _, sd, _ = load_case("p0_b16")
    cfg = config_for("p0_b16")
    nref = np.array([128, 0, 256, 3, 16, 0, 1, 127, 10])
    nalt = np.array([128, 256, 0 + 1, 1, 224, 1, 15, 113, 15])
    nalt[2] = 1  # 256 ref + 1 alt does not fit (16 + 1 tiles): use 240 + 1
    nref[2] = 224
    ints, floats, packed = _arrays(nref, nalt)
    model, dev = build("p0_b16", sd)
    batch = Batch.from_arrays(ints, floats, packed).copy_to(dev)
    plan = batch.plan()
    assert plan.num_groups >= 6
    with torch.no_grad():
        out = model.compute_batch_output(batch)
        ref = O.compute_batch_output(sd, cfg, torch.from_numpy(O.decode_packed_reads(packed).astype(np.float32)),
                                     torch.from_numpy(nref), torch.from_numpy(nalt),
                                     torch.from_numpy(floats[:, 6:].astype(np.float32)), torch.from_numpy(ints[:, 16:].astype(np.int64)))
    z = {"out/" + k: v.numpy() for k, v in ref.items()}
    check_outputs(out, z, "p0_deep")


def test_capacity_error_and_empty_batch():
    _, sd, _ = load_case("p0_b16")
    model, dev = build("p0_b16", sd)
    from permutect_amd.engine.lib import PmtError
    ints, floats, packed = _arrays(np.array([3, 241]), np.array([2, 17]))  # 16 + 2 tiles: more than a workgroup holds
    with pytest.raises(PmtError, match="variant 1"):
        Batch.from_arrays(ints, floats, packed).plan()
    ints, floats, packed = _arrays(np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64))
    empty = Batch.from_arrays(ints, floats, packed).copy_to(dev)
    with torch.no_grad():
        out = model.compute_batch_output(empty)
    assert out.logits_b.shape == (0,) and out.features_be.shape[0] == 0


def test_read_sets_beyond_one_workgroup_run_layered_and_match_oracle():
    """BASELINE config 'mean 600 reads per variant': read sets split over several workgroups, num_blocks + 1 launches with
    the per-set sums accumulated in HBM (pmt_forward_layered), mixed with ordinary sets; against the oracle."""
    _, sd, _ = load_case("p0_b16")
    cfg = config_for("p0_b16")
    nref = np.array([5, 300, 0, 10, 700, 2, 256, 9])
    nalt = np.array([3, 350, 600, 15, 1, 7, 255, 1])
    ints, floats, packed = _arrays(nref, nalt, seed=21)
    model, dev = build("p0_b16", sd)
    batch = Batch.from_arrays(ints, floats, packed).copy_to(dev)
    with torch.no_grad():
        out = model.compute_batch_output(batch)
    plan = batch.plan()
    assert plan.layered and plan.num_groups > len(nref)
    with torch.no_grad():
        ref = O.compute_batch_output(sd, cfg, torch.from_numpy(O.decode_packed_reads(packed).astype(np.float32)),
                                     torch.from_numpy(nref), torch.from_numpy(nalt),
                                     torch.from_numpy(floats[:, 6:].astype(np.float32)), torch.from_numpy(ints[:, 16:].astype(np.int64)))
    z = {"out/" + k: v.numpy() for k, v in ref.items()}
    check_outputs(out, z, "p0_deep", lk_ulps=16)


def test_layered_forward_equals_the_single_launch_forward():
    """The layered path on a batch the ordinary kernel accepts (forced split plan) gives the same outputs."""
    z, sd, b = load_case("p0_deep")
    model, dev = build("p0_deep", sd)
    batch = Batch.from_arrays(b["int_array"], b["float_array"], b["packed_reads"]).copy_to(dev)
    with torch.no_grad():
        a = model.compute_batch_output(batch)
    forced = Batch.from_arrays(b["int_array"], b["float_array"], b["packed_reads"]).copy_to(dev)
    from permutect_amd.data.batch import GroupPlan
    plan = GroupPlan(*forced.host_counts(), allow_split=True)
    if plan.span is None:  # nothing oversized: build the explicit spans of the ordinary groups
        gs, counts = plan.group_start, forced.host_counts()
        ro = np.concatenate([[0], np.cumsum(counts[0])])
        ao = np.concatenate([[0], np.cumsum(counts[1])])
        plan.span = np.array([[gs[g], gs[g + 1], ro[gs[g]], ro[gs[g + 1]], ao[gs[g]], ao[gs[g + 1]]] for g in range(plan.num_groups)], dtype=np.int32)
    forced._plan = plan
    with torch.no_grad():
        c = model.compute_batch_output(forced)
    assert torch.allclose(a.logits_b, c.logits_b, rtol=1e-5, atol=1e-5)
    assert torch.allclose(a.features_be, c.features_be, rtol=1e-5, atol=1e-5)
    assert torch.allclose(a.logits_bk, c.logits_bk, rtol=1e-5, atol=2e-4)


def test_packed_order_gives_the_same_outputs_per_variant():
    """Batch.from_arrays(pack=True) only reorders the variants (to fill the workgroups): every variant's outputs are the same."""
    _, sd, _ = load_case("p0_b16")
    rng = np.random.default_rng(9)
    nb = 600
    nref, nalt = rng.integers(0, 11, nb), rng.integers(1, 16, nb)
    ints, floats, packed = _arrays(nref, nalt, seed=31)
    model, dev = build("p0_b16", sd)
    plain = Batch.from_arrays(ints, floats, packed).copy_to(dev)
    tight = Batch.from_arrays(ints, floats, packed, pack=True).copy_to(dev)
    assert tight.plan().num_groups < plain.plan().num_groups
    with torch.no_grad():
        a = model.compute_batch_output(plain)
        c = model.compute_batch_output(tight)
    o = torch.from_numpy(tight.order).to(dev)
    assert torch.allclose(a.logits_b[o], c.logits_b, rtol=1e-5, atol=1e-5)
    assert torch.allclose(a.features_be[o], c.features_be, rtol=1e-5, atol=1e-5)
    assert torch.allclose(a.logits_bk[o], c.logits_bk, rtol=1e-5, atol=2e-4)
