"""The training loop's per-step bookkeeping as fused launches (the loop the CLI runs): the balancer's step
(pmt_balance_step, two launches for ~60 torch ones) against this package's torch form of reference training/balancer.py:55-119 --
itself pinned by the reference fixture tests/golden/training_helpers.npz (tests/test_training_helpers_cpu.py) -- and the downsampler's
mixture-weight lookup inside pmt_downsample_counts against the per-variant weights handed in."""
import numpy as np
import pytest
import torch

from permutect_amd.data.batch import Batch, DownsampledBatch
from permutect_amd.training.balancer import Balancer
from permutect_amd.training.downsampler import Downsampler
from tests.test_forward_gpu import _arrays

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")


def _batch(rng, n, num_sources):
    nref, nalt = rng.integers(0, 14, n), rng.integers(1, 19, n)  # (beyond the last count bins on both sides)
    ints, floats, packed = _arrays(nref, nalt, seed=int(rng.integers(1 << 30)))
    ints[:, 3] = rng.integers(0, 5, n)
    ints[:, 4] = rng.integers(0, num_sources, n)
    return Batch.from_arrays(ints, floats, packed).copy_to(DEV)


@pytest.mark.parametrize("num_sources", [1, 3])
def test_fused_balancer_step_matches_the_torch_form(num_sources):
    rng = np.random.default_rng(7 + num_sources)
    fused, ref = Balancer(num_sources, DEV), Balancer(num_sources, DEV)
    recomputed = 0
    for step, n in enumerate((3000, 4000, 4500, 200, 11000, 64)):  # recomputations at steps 2 and 4 (more than 10 000 variants since the last)
        batch = _batch(rng, n, num_sources)
        parent_or_down = batch if step % 2 == 0 else DownsampledBatch.on_device(batch, seed=100 + step)  # int64 strided and int32 dense counts
        logits = torch.from_numpy(rng.normal(0, 3, n).astype(np.float32)).to(DEV)
        before = ref.count_since_last_recomputation
        w_ref, sw_ref = ref.process_batch_and_compute_weights(parent_or_down, torch.sigmoid(logits))
        recomputed += int(ref.count_since_last_recomputation == 0 and before + n > Balancer.DATA_BEFORE_RECOMPUTE)
        w, wsw = fused.weights_from_logits(parent_or_down, logits)
        torch.cuda.synchronize()
        assert fused.count_since_last_recomputation == ref.count_since_last_recomputation
        np.testing.assert_allclose(w.cpu().numpy(), w_ref.cpu().numpy(), rtol=2e-6, atol=1e-7)
        np.testing.assert_allclose(wsw.cpu().numpy(), (w_ref * sw_ref).cpu().numpy(), rtol=3e-6, atol=1e-7)
        for name in ("counts_slvra", "pseudo_counts_slvra", "weights_slvra", "unlabeled_weights_slvra", "source_weights_s"):
            a, b = getattr(fused, name).cpu().numpy(), getattr(ref, name).cpu().numpy()
            np.testing.assert_allclose(a, b, rtol=2e-5, atol=1e-6, err_msg=f"{name} after step {step}")
    assert recomputed == 2 and float(fused.weights_slvra.std()) > 0  # the tables moved


def test_downsampler_lookup_inside_the_kernel_is_the_per_variant_gather():
    rng = np.random.default_rng(3)
    ds = Downsampler(num_sources=2)
    with torch.no_grad():  # weights away from uniform, different in every cell
        for p in ds.weights_parameters():
            p.copy_(torch.from_numpy(rng.normal(0, 1.5, tuple(p.shape)).astype(np.float32)))
    ds = ds.to(DEV)
    batch = _batch(rng, 5000, 2)
    ref_w, alt_w = ds._weights_bk(batch)
    a = DownsampledBatch.on_device(batch, seed=77, ref_weights_b4=ref_w, alt_weights_b4=alt_w, force_random=13)
    b = DownsampledBatch.on_device(batch, seed=77, weight_tables=ds.weight_tables(), num_sources=2, force_random=13)
    torch.cuda.synchronize()
    for name in ("ref_fracs", "alt_fracs", "ref_counts", "alt_counts", "read_indices"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    assert float(a.ref_fracs.std()) > 0.1 and int(a.alt_counts.min()) >= 1
    # the cached tables follow the weights
    t0 = ds.weight_tables()[0]
    with torch.no_grad():
        ds.weights_parameters()[0].add_(1.0 * torch.randn_like(ds.weights_parameters()[0]))
    assert ds.weight_tables()[0] is not t0 and not torch.equal(ds.weight_tables()[0], t0)


def _host_plan(ref, alt):
    from permutect_amd.data.batch import GroupPlan
    p = GroupPlan(ref, alt)
    return p.group_start, p.group_tile_base


@pytest.mark.parametrize("n", [1, 255, 256, 257, 5000, 65536, 300000])
def test_device_planner_is_the_host_planner_chunk_by_chunk(n):
    """pmt_plan_groups_device against pmt_plan_groups: the same next-fit packing inside every chunk of consecutive variants, a group
    closed at every chunk boundary, group and tile bases continued across chunks -- and never more groups than a plan of LARGER counts
    in the same order plus the number of chunks (the capacity a DownsampledBatch allocates)."""
    import ctypes as C  # noqa: F401
    from permutect_amd.engine import lib as L
    lib = L.load()
    rng = np.random.default_rng(n)
    parent_ref, parent_alt = rng.integers(0, 11, n).astype(np.int32), rng.integers(1, 16, n).astype(np.int32)
    ref = rng.binomial(parent_ref, rng.beta(1, 1, n)).astype(np.int32)  # what a downsampling keeps
    alt = np.maximum(rng.binomial(parent_alt, rng.beta(1, 1, n)), 1).astype(np.int32)
    chunks, chunk = lib.pmt_plan_device_chunks(n), 256  # (one wave per chunk of 256 consecutive variants)
    assert chunks == -(-n // chunk)
    scratch = torch.empty(2 * chunks, dtype=torch.int32, device=DEV)
    pgs, _ = _host_plan(parent_ref, parent_alt)
    capacity = (len(pgs) - 1) + chunks
    ro = torch.from_numpy(np.concatenate([[0], np.cumsum(ref)]).astype(np.int32)).to(DEV)
    ao = torch.from_numpy(np.concatenate([[0], np.cumsum(alt)]).astype(np.int32)).to(DEV)
    gs = torch.full((capacity + 1,), -1, dtype=torch.int32, device=DEV)
    gt = torch.full((capacity + 1,), -1, dtype=torch.int32, device=DEV)
    ng = torch.zeros(1, dtype=torch.int32, device=DEV)
    fault = torch.zeros(1, dtype=torch.int32, device=DEV)
    L.check(lib.pmt_plan_groups_device(ro.data_ptr(), ao.data_ptr(), n, gs.data_ptr(), gt.data_ptr(), capacity, ng.data_ptr(), fault.data_ptr(),
                                       scratch.data_ptr(), torch.cuda.current_stream().cuda_stream), "pmt_plan_groups_device")
    torch.cuda.synchronize()
    want_gs, want_gt = [0], [0]
    for lo in range(0, n, chunk):
        hs, ht = _host_plan(ref[lo:lo + chunk], alt[lo:lo + chunk])
        want_gs += [lo + int(v) for v in hs[1:]]
        want_gt += [want_gt[-1] + int(v) for v in ht[1:]] if False else [int(v) for v in (ht[1:].astype(np.int64) + want_gt[-1])]
    g = int(ng.item())
    assert int(fault.item()) == 0 and g == len(want_gs) - 1 <= capacity
    assert gs[: g + 1].cpu().tolist() == want_gs and gt[: g + 1].cpu().tolist() == [int(v) for v in want_gt]
    # a capacity one short raises the fault word (and the kernels' group count stays inside what was allocated)
    if g > 1:
        L.check(lib.pmt_plan_groups_device(ro.data_ptr(), ao.data_ptr(), n, gs.data_ptr(), gt.data_ptr(), g - 1, ng.data_ptr(), fault.data_ptr(),
                                           scratch.data_ptr(), torch.cuda.current_stream().cuda_stream), "pmt_plan_groups_device")
        assert int(fault.item()) == 4 and int(ng.item()) == g - 1


def test_a_downsampled_training_step_on_its_own_plan_equals_the_step_on_its_parents(monkeypatch):
    """The same DownsampledBatch through forward, losses and backward on the device-made plan of ITS counts and (PMT_DEVICE_PLAN=0) on its
    parent's plan as rounds 1 - 4 ran it: the same logits and gradients up to the order of the float atomics -- from about half the
    workgroups."""
    from permutect_amd.architecture.artifact_model import ArtifactModel
    from permutect_amd.parameters import P0_DIMS, p0_params
    rng = np.random.default_rng(21)
    torch.manual_seed(5)
    model = ArtifactModel(p0_params(), device=DEV, **P0_DIMS)
    with torch.no_grad():
        for q in model.parameters():
            q.add_(0.05 * torch.randn_like(q))
    batch = _batch(rng, 6000, 1)
    results = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("PMT_DEVICE_PLAN", mode)
        db = DownsampledBatch.on_device(batch, seed=5, force_random=3)
        model.train(True)
        model.zero_grad()
        model.engine().space.gtheta.zero_()
        out = model.compute_batch_output(db)
        model.compute_batch_losses(out, db).total_loss.backward()
        torch.cuda.synchronize()
        model.engine().check_join_fault()
        plan = db.plan(allow_split=True)
        groups = int(plan.num_groups_dev.item()) if mode == "1" else plan.num_groups
        results[mode] = (out.logits_b.detach().cpu().numpy(), out.features_be.detach().cpu().numpy(),
                         model.engine().space.gtheta.detach().cpu().numpy().copy(), groups)
    (l1, f1, g1, n1), (l0, f0, g0, n0) = results["1"], results["0"]
    assert n1 < 0.75 * n0, (n1, n0)  # about half the reads kept -> about half the workgroups
    # (a set's log-likelihood sums are float atomics, added in the order its reads' lanes arrive: another grouping is another order, and a
    #  capped logit is a difference of sums of tens to hundreds -- both results lie within the 1e-4 contract of the oracle)
    np.testing.assert_allclose(l1, l0, rtol=0, atol=1e-4)
    assert np.median(np.abs(l1 - l0)) < 2e-6
    np.testing.assert_allclose(f1, f0, rtol=1e-5, atol=1e-5)
    assert np.linalg.norm(g1 - g0) <= 2e-5 * np.linalg.norm(g0)


@pytest.mark.parametrize("num_sources", [1, 3])  # (one source: the histogram joined in LDS; three: global atomics)
def test_fused_evaluation_tallies_match_the_torch_form(num_sources):
    """pmt_record_evaluation against the torch composition it replaces (EvaluationCounts.record_batch_torch; the reference's tallies
    themselves are pinned through it by tests/test_metrics_gpu.py): histogram over (source, label, type, count bins, logit bin) of the
    LABELED variants' weights, and per label the weight called artifact / not and the weighted logit sum -- both epoch types."""
    from permutect_amd.training.loss_recorder import EvaluationCounts
    rng = np.random.default_rng(17 + num_sources)
    fused, ref = EvaluationCounts(DEV, num_sources), EvaluationCounts(DEV, num_sources)
    for step, n in enumerate((5000, 37, 9000)):
        batch = _batch(rng, n, num_sources)
        b = batch if step != 1 else DownsampledBatch.on_device(batch, seed=9)
        logits = torch.from_numpy(rng.normal(0, 6, n).astype(np.float32)).to(DEV)  # (beyond +-10 on both sides: the clamped end bins)
        logits[:4] = torch.tensor([-10.0, 10.0, 0.0, 9.999], device=DEV)
        weights = torch.from_numpy(rng.uniform(0.1, 3.0, n).astype(np.float32)).to(DEV)
        fused.record_batch(step % 2, b, logits, weights)
        ref.record_batch_torch(step % 2, b, logits, weights)
    torch.cuda.synchronize()
    assert fused.batches == 3
    np.testing.assert_allclose(fused.hist.cpu().numpy(), ref.hist.cpu().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(fused.stats.cpu().numpy(), ref.stats.cpu().numpy(), rtol=2e-5, atol=1e-3)
    assert float(ref.hist.sum()) > 0 and abs(fused.accuracy(0) - ref.accuracy(0)) < 1e-6
