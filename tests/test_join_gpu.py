"""Read sets beyond one workgroup, JOINED inside one launch (BASELINE configs[4], mean 600 reads per variant): the groups of a
split read set exchange their per-set sums through HBM -- device-scope float atomics, an arrival counter per (variant, block),
groups handed out by ticket (pmt_device.hpp: PmtJoin) -- while the activations stay in registers.  One launch each way instead
of num_blocks + 1 with every activation parked in between (PMT_LAYERED_JOIN=0 keeps that path; both are run here).

Checked: the joined path against the oracle (forward, losses, every gradient); against the layered launches on the same batch
(the same arithmetic up to the order in which the groups' parts are added); more groups than can be resident at once (the
forward's workgroups then start in several rounds and wait for groups of a later round); forced spans that cut every set into
many small groups (a set's sums then have five and more contributors); and that no wait ever gave up (the fault word)."""
import numpy as np
import pytest
import torch

from oracle import artifact_oracle as O
from permutect_amd.data.batch import Batch, GroupPlan
from permutect_amd.training.optimizer import FusedClipAdamW
from tests.helpers import config_for, load_case, oracle_forward
from tests.test_forward_gpu import _arrays, build, check_outputs

pytestmark = pytest.mark.gpu


def _stress_counts(rng, n_big, n_small):
    nref = np.concatenate([rng.poisson(300, n_big), rng.integers(0, 11, n_small)])
    nalt = np.concatenate([np.maximum(rng.poisson(300, n_big), 1), rng.integers(1, 16, n_small)])
    order = rng.permutation(len(nref))
    return nref[order], nalt[order]


def _train_step(model, batch):
    model.train(True)
    out = model.compute_batch_output(batch)
    losses = model.compute_batch_losses(out, batch)
    opt = FusedClipAdamW(model, lr=1e-3, weight_decay=0.01)
    opt.zero_grad()
    losses.total_loss.backward()
    torch.cuda.synchronize()
    model.engine().check_join_fault()
    grads = {n: p.grad.detach().cpu().numpy().copy() for n, p in model.named_parameters()}
    return out, losses, grads


def test_joined_training_step_matches_oracle_and_the_layered_launches():
    _, sd, _ = load_case("p0_b16")
    cfg = config_for("p0_b16")
    nref, nalt = _stress_counts(np.random.default_rng(5), 24, 12)
    ints, floats, packed = _arrays(nref, nalt, seed=41)
    joined, dev = build("p0_b16", sd)
    layered, _ = build("p0_b16", sd)
    layered.engine().join_layered = False
    batch = Batch.from_arrays(ints, floats, packed).copy_to(dev)
    plan = batch.plan(allow_split=True)
    assert plan.layered and plan.set_groups.max() >= 3 and plan.num_groups > 40
    out_j, losses_j, g_j = _train_step(joined, batch)
    out_l, losses_l, g_l = _train_step(layered, Batch.from_arrays(ints, floats, packed).copy_to(dev))
    # the two paths: same arithmetic, the groups' parts of a sum added in a different order
    assert torch.allclose(out_j.logits_b, out_l.logits_b, rtol=1e-5, atol=2e-5)
    assert torch.allclose(out_j.features_be, out_l.features_be, rtol=1e-5, atol=1e-5)
    flat_j, flat_l = np.concatenate([g.ravel() for g in g_j.values()]), np.concatenate([g_l[n].ravel() for n in g_j])
    assert np.linalg.norm(flat_j - flat_l) <= 1e-5 * np.linalg.norm(flat_l)
    # the oracle
    i64 = torch.from_numpy(ints.astype(np.int64))
    ob = dict(reads_re=torch.from_numpy(O.decode_packed_reads(packed).astype(np.float32)), nref=i64[:, O.REF_COUNT],
              nalt=i64[:, O.ALT_COUNT], labels=i64[:, O.LABEL], sources=i64[:, O.SOURCE],
              info_be=torch.from_numpy(floats[:, O.INFO_START:].astype(np.float32)), haplotypes_bh=i64[:, O.HAPLOTYPES_START:])
    ref_out, ref_losses, ref_grads = O.train_step_grads(sd, cfg, ob)
    check_outputs(out_j, {"out/" + k: v.detach().numpy() for k, v in ref_out.items()}, "p0_deep", lk_ulps=32)
    ref_total = ref_losses["total_losses_b"].detach().numpy()
    np.testing.assert_allclose(losses_j.total_losses_b.detach().cpu().numpy(), ref_total, rtol=1e-4, atol=1e-4 + 1e-5 * np.abs(ref_total).max())
    gref = np.concatenate([ref_grads[n].numpy().ravel() for n in g_j])
    assert np.linalg.norm(flat_j - gref) <= 1e-4 * np.linalg.norm(gref)
    gscale = np.abs(gref).max()
    for n, g in g_j.items():
        ref = ref_grads[n].numpy()
        assert np.abs(g - ref).max() <= 5e-4 * max(np.abs(ref).max(), 1e-3 * gscale), n


def test_more_groups_than_resident_workgroups():
    """~1 400 groups: the forward's grid exceeds what the card holds at once (512 workgroups), so its workgroups start in rounds and
    the last groups of one round wait for the first of the next; the backward's 256 persistent workgroups draw ~5 tickets each."""
    _, sd, _ = load_case("p0_b16")
    cfg = config_for("p0_b16")
    nref, nalt = _stress_counts(np.random.default_rng(6), 560, 40)
    ints, floats, packed = _arrays(nref, nalt, seed=42)
    joined, dev = build("p0_b16", sd)
    layered, _ = build("p0_b16", sd)
    layered.engine().join_layered = False
    batch = Batch.from_arrays(ints, floats, packed, pack=True).copy_to(dev)
    assert batch.plan(allow_split=True).num_groups > 1200
    out_j, _, g_j = _train_step(joined, batch)
    out_l, _, g_l = _train_step(layered, batch)
    assert torch.allclose(out_j.logits_b, out_l.logits_b, rtol=1e-5, atol=2e-5)
    assert torch.allclose(out_j.logits_bk, out_l.logits_bk, rtol=2e-6, atol=2e-3)
    flat_j, flat_l = np.concatenate([g.ravel() for g in g_j.values()]), np.concatenate([g_l[n].ravel() for n in g_j])
    assert np.all(np.isfinite(flat_j)) and np.linalg.norm(flat_j - flat_l) <= 1e-5 * np.linalg.norm(flat_l)
    # forward against the oracle (in the batch's packed order)
    o = batch.order
    nr, na = ints[:, 0].astype(np.int64), ints[:, 1].astype(np.int64)
    ro, ao = np.concatenate([[0], np.cumsum(nr)]), int(nr.sum()) + np.concatenate([[0], np.cumsum(na)])
    rows = np.concatenate([np.concatenate([np.arange(ro[v], ro[v + 1]) for v in o]), np.concatenate([np.arange(ao[v], ao[v + 1]) for v in o])])
    ref = oracle_forward(sd, cfg, ints[o], floats[o], packed[rows])
    joined.eval()
    with torch.inference_mode():
        out = joined.compute_batch_output(batch)
    joined.engine().check_join_fault()
    check_outputs(out, {"out/" + k: v.numpy() for k, v in ref.items()}, "p0_deep", lk_ulps=32)


def test_joined_path_on_forced_small_groups_matches_reference():
    """the deep fixture cut into groups of at most two ref and two alt tiles: every set's sums have many contributors, and
    neighbouring groups wait for each other at every block"""
    z, sd, b = load_case("p0_deep")
    model, dev = build("p0_deep", sd)
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
            if r < ro[v + 1]:
                a1 = a if r1 < ro[v + 1] else a1
            spans.append([v, v + 1, r, r1, a, a1])
            tile_base.append(tiles)
            tiles += (r1 - r + 15) // 16 + (a1 - a + 15) // 16
            r, a = r1, a1
    tile_base.append(tiles)
    plan.use_span(np.array(spans, dtype=np.int32), np.array(tile_base, dtype=np.int32), len(nref))
    assert plan.set_groups.max() >= 3
    batch._plan = plan
    out, losses, grads = _train_step(model, batch)
    check_outputs(out, z, "p0_deep", lk_ulps=32)
    gref = np.concatenate([z["grad/" + n].ravel() for n in grads])
    gour = np.concatenate([g.ravel() for g in grads.values()])
    assert np.linalg.norm(gour - gref) <= 1e-4 * np.linalg.norm(gref)


def test_a_join_that_gives_up_is_reported_through_the_persistent_fault_word():
    """a wait that times out carries on with partial sums -- wrong numbers -- so it must not pass silently.  One variant
    is told to expect ONE MORE group than covers it: its groups' bounded wait gives up (no hang), the kernels raise the engine's
    persistent fault word (PmtBatch.join_fault), and the host raises at its next check -- also when other, healthy launches ran in
    between (the old per-launch words kept only the last eight and were read by tests only).  The check clears the word."""
    from permutect_amd.engine.lib import PmtError
    _, sd, _ = load_case("p0_b16")
    nref, nalt = _stress_counts(np.random.default_rng(8), 6, 6)
    ints, floats, packed = _arrays(nref, nalt, seed=43)
    model, dev = build("p0_b16", sd)
    eng = model.engine()
    good = Batch.from_arrays(ints, floats, packed).copy_to(dev)
    bad = Batch.from_arrays(ints, floats, packed).copy_to(dev)
    plan = bad.plan(allow_split=True)
    assert plan.layered and plan.set_groups is not None
    v = int(np.argmax(plan.set_groups))
    plan.set_groups = plan.set_groups.copy()
    plan.set_groups[v] += 1  # a group that will never arrive
    model.eval()
    with torch.inference_mode():
        model.compute_batch_output(bad)
        for _ in range(10):  # healthy launches afterwards must not hide it
            model.compute_batch_output(good)
    with pytest.raises(PmtError, match="timed out"):
        eng.check_join_fault()
    eng.check_join_fault()  # cleared: the next check passes
    with torch.inference_mode():
        out = model.compute_batch_output(good)
    eng.check_join_fault()
    assert torch.isfinite(out.logits_b).all()
