"""Training WITH dropout (reference architecture/mlp.py:57-58: an nn.Dropout behind every Linear of read_embedding, info_embedding,
reducer, source_predictor and their skip blocks; parameters.py:142 --dropout_p).  The kernels generate the masks from
(seed, linear, batch row, feature) (pmt_dropout.hpp) -- in the forward, again in the backward's recomputation, and the same in every
group of a split read set; the host exports the very same function (pmt_dropout_mask).

Checked here: a train-mode step of a dropout_p = 0.25 model (two sources, so that the source adversary runs) against the oracle GIVEN
the exported masks -- outputs, losses and every gradient; the same with read sets split over workgroups (joined and layered
launches); that eval mode ignores the masks (the reference fixture covers the numbers); that one torch.manual_seed replays a step bit
for bit and another seed gives other masks; that the masks really bite (the train-mode output differs from eval mode)."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import artifact_oracle as O
from permutect_amd.architecture.artifact_model import ArtifactModel
from permutect_amd.data.batch import Batch
from permutect_amd.engine import lib as L
from permutect_amd.parameters import P0_DIMS, p0_params, t0_params, wide_params
from permutect_amd.training.optimizer import FusedClipAdamW
from tests.helpers import config_for
from tests.test_forward_gpu import _arrays

pytestmark = pytest.mark.gpu
P = 0.25


def dropout_model(family, num_sources=2, seed=3):
    torch.manual_seed(seed)
    params = t0_params() if family == "t0" else wide_params() if family == "wide" else p0_params()
    params.dropout_p = P
    model = ArtifactModel(params, device=torch.device("cuda"), **P0_DIMS)
    if num_sources > 1:
        model.reset_source_predictor(num_sources)
    with torch.no_grad():  # (the skip blocks' alphas start at 0.1; make every branch carry weight)
        for n, p in model.named_parameters():
            if n.endswith(".alpha"):
                p.fill_(0.5)
    return model


def linear_ids(model):
    """state_dict prefix of a Linear ("reducer._model.0") -> its index in the descriptor (PmtModel.lin)"""
    eng = model.engine()
    d = eng.plan.desc
    by_offset = {d.lin[i].w_src: i for i in range(d.n_linear) if d.lin[i].w_src >= 0}
    return {n[:-len(".weight")]: by_offset[eng.space.offset_of(p)] for n, p in model.named_parameters()
            if n.endswith(".weight") and p.dim() == 2 and eng.space.offset_of(p) in by_offset}


def mask_provider(model, seed):
    ids, lib = linear_ids(model), L.load()

    def provide(key, y, row0=0):
        out = np.empty(tuple(y.shape), dtype=np.float32)
        L.check(lib.pmt_dropout_mask(seed, C.c_float(P), ids[key], row0, y.shape[0], y.shape[1], out.ctypes.data), "pmt_dropout_mask")
        return torch.from_numpy(out)
    return provide


def oracle_batch(ints, floats, packed):
    i64 = torch.from_numpy(ints.astype(np.int64))
    return dict(reads_re=torch.from_numpy(O.decode_packed_reads(packed).astype(np.float32)), nref=i64[:, O.REF_COUNT],
                nalt=i64[:, O.ALT_COUNT], labels=i64[:, O.LABEL], sources=i64[:, O.SOURCE],
                info_be=torch.from_numpy(floats[:, O.INFO_START:].astype(np.float32)), haplotypes_bh=i64[:, O.HAPLOTYPES_START:])


def train_step(model, batch):
    model.train(True)
    out = model.compute_batch_output(batch)
    seed = model.engine().dropout_seed
    losses = model.compute_batch_losses(out, batch)
    opt = FusedClipAdamW(model, lr=1e-3, weight_decay=0.01)
    opt.zero_grad()
    losses.total_loss.backward()
    torch.cuda.synchronize()
    model.engine().check_join_fault()
    grads = {n: p.grad.detach().cpu().numpy().copy() for n, p in model.named_parameters()}
    return out, losses, grads, seed


def check_against_oracle(model, family, ints, floats, packed, out, losses, grads, seed):
    cfg = config_for(family + "_dropout")
    cfg.num_sources = model.num_sources
    cfg.dropout = mask_provider(model, seed)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    ref_out, ref_losses, ref_grads = O.train_step_grads(sd, cfg, oracle_batch(ints, floats, packed))
    for k in ("logits_b", "features_be", "ref_features_be"):
        ref = ref_out[k].detach().numpy()
        np.testing.assert_allclose(getattr(out, k).detach().cpu().numpy(), ref, rtol=2e-5, atol=1e-4 * max(1.0, float(np.abs(ref).max())), err_msg=k)
    for k in ("alt_count_losses_b", "source_prediction_losses_b", "total_losses_b"):
        ref = ref_losses[k].detach().numpy()
        ours = getattr(losses, {"alt_count_losses_b": "alt_count_losses_b", "source_prediction_losses_b": "source_prediction_losses_b",
                                "total_losses_b": "total_losses_b"}[k]).detach().cpu().numpy()
        np.testing.assert_allclose(ours, ref, rtol=1e-4, atol=1e-4 + 1e-5 * np.abs(ref).max(), err_msg=k)
    names = list(grads)
    gref = np.concatenate([ref_grads[n].numpy().ravel() for n in names])
    gour = np.concatenate([grads[n].ravel() for n in names])
    assert np.all(np.isfinite(gour))
    assert np.linalg.norm(gour - gref) <= 1e-4 * np.linalg.norm(gref)
    gscale = np.abs(gref).max()
    bad = [(n, float(np.abs(grads[n] - ref_grads[n].numpy()).max())) for n in names
           if np.abs(grads[n] - ref_grads[n].numpy()).max() > 5e-4 * max(np.abs(ref_grads[n].numpy()).max(), 1e-3 * gscale)]
    assert not bad, bad[:10]
    # the MLPs that carry dropout have gradients worth comparing on their own scale (they are a part of the global vector)
    for prefix in ("read_embedding", "info_embedding", "reducer", "source_predictor"):
        sel = [n for n in names if n.startswith(prefix)]
        a, b = np.concatenate([grads[n].ravel() for n in sel]), np.concatenate([ref_grads[n].numpy().ravel() for n in sel])
        assert np.linalg.norm(b) > 0 and np.linalg.norm(a - b) <= 2e-4 * np.linalg.norm(b), prefix


def small_batch(seed, n=48):
    rng = np.random.default_rng(seed)
    nref, nalt = rng.integers(0, 12, n), rng.integers(1, 16, n)
    ints, floats, packed = _arrays(nref, nalt, seed=seed + 100)
    ints[:, O.SOURCE] = rng.integers(0, 2, n)
    return ints, floats, packed


@pytest.mark.parametrize("family", ["p0", "t0", "wide"])  # (wide: layers beyond 64, the dropout instance of its own build of the library)
def test_train_step_with_dropout_matches_oracle_given_the_masks(family):
    model = dropout_model(family)
    d = model.engine().plan.desc
    assert abs(d.dropout_p - P) < 1e-7 and d.read_mlp.dropout == 1 and d.reducer.dropout == 1
    assert d.row_mlp[L.ROWS_INFO].dropout == 1 and d.row_mlp[L.ROWS_SOURCE].dropout == 1 and d.row_mlp[L.ROWS_ALT_COUNT].dropout == 0
    ints, floats, packed = small_batch(21)
    batch = Batch.from_arrays(ints, floats, packed).copy_to(torch.device("cuda"))
    out, losses, grads, seed = train_step(model, batch)
    assert seed != 0
    check_against_oracle(model, family, ints, floats, packed, out, losses, grads, seed)
    # the masks bite: eval mode gives other numbers, and is the mask-free oracle
    model.eval()
    with torch.inference_mode():
        ev = model.compute_batch_output(batch)
    assert model.engine().dropout_seed == 0
    assert float((ev.logits_b - out.logits_b.detach()).abs().max()) > 1e-3
    cfg = config_for(family + "_dropout")
    cfg.num_sources, cfg.dropout = model.num_sources, (lambda key, y, row0=0: None)
    ob = oracle_batch(ints, floats, packed)
    with torch.no_grad():
        ref = O.compute_batch_output({k: v.detach().cpu() for k, v in model.state_dict().items()}, cfg, ob["reads_re"], ob["nref"], ob["nalt"],
                                     ob["info_be"], ob["haplotypes_bh"])
    np.testing.assert_allclose(ev.logits_b.cpu().numpy(), ref["logits_b"].numpy(), rtol=2e-5, atol=1e-4)


def test_split_read_sets_see_the_same_masks_in_every_group():
    """read sets beyond one workgroup: every group regenerates the masks of ITS rows (joined launches and layered launches)"""
    rng = np.random.default_rng(8)
    nref = np.concatenate([rng.poisson(300, 6), rng.integers(0, 11, 10)])
    nalt = np.concatenate([np.maximum(rng.poisson(300, 6), 1), rng.integers(1, 16, 10)])
    ints, floats, packed = _arrays(nref, nalt, seed=77)
    ints[:, O.SOURCE] = rng.integers(0, 2, len(nref))
    for join in (True, False):
        model = dropout_model("p0")
        model.engine().join_layered = join
        batch = Batch.from_arrays(ints, floats, packed).copy_to(torch.device("cuda"))
        assert batch.plan(allow_split=True).layered
        out, losses, grads, seed = train_step(model, batch)
        check_against_oracle(model, "p0", ints, floats, packed, out, losses, grads, seed)


def test_seeded_replay_and_fresh_masks():
    ints, floats, packed = small_batch(22)
    dev = torch.device("cuda")
    runs = []
    for s in (5, 5, 6):
        model = dropout_model("p0")  # (same weights every time: its own manual_seed)
        torch.manual_seed(s)
        out, _, grads, seed = train_step(model, Batch.from_arrays(ints, floats, packed).copy_to(dev))
        # a second step of the same model draws a NEW seed
        out2 = model.compute_batch_output(Batch.from_arrays(ints, floats, packed).copy_to(dev))
        assert model.engine().dropout_seed not in (0, seed)
        assert float((out2.logits_b - out.logits_b).detach().abs().max()) > 1e-3
        runs.append((seed, out.logits_b.detach().cpu().numpy(), np.concatenate([g.ravel() for g in grads.values()])))
    assert runs[0][0] == runs[1][0] != runs[2][0]
    np.testing.assert_array_equal(runs[0][1], runs[1][1])
    # (gradients: float atomics, so equal up to the order of the additions)
    assert np.linalg.norm(runs[0][2] - runs[1][2]) <= 1e-5 * np.linalg.norm(runs[0][2])
    assert np.abs(runs[0][1] - runs[2][1]).max() > 1e-3


def test_keep_rate_seen_through_the_kernels():
    """zero weights and unit biases in the info MLP: its output IS the last Linear's mask (0 or 1 / (1 - p)), read back through the
    row kernel (the wide first Linear included) over many variants"""
    model = dropout_model("t0", num_sources=1)
    with torch.no_grad():
        for name, p in model.info_embedding.named_parameters():
            p.zero_()
            if name.endswith("bias"):
                p.fill_(1.0)
    n = 4096
    rng = np.random.default_rng(1)
    ints, floats, packed = _arrays(rng.integers(0, 3, n), rng.integers(1, 3, n), seed=9)
    batch = Batch.from_arrays(ints, floats, packed).copy_to(torch.device("cuda"))
    model.train(True)
    with torch.no_grad():
        model.compute_batch_output(batch)  # draws the step's seed, packs the weights
        seed = model.engine().dropout_seed
        ve = model.variant_embedding(batch)
    e_info = model.info_embedding.output_dimension()
    y = ve[:, :e_info].cpu().numpy()
    vals = np.unique(y.round(5))
    assert len(vals) == 2 and vals[0] == 0.0 and abs(vals[1] - 1 / (1 - P)) < 1e-4, vals
    keep = (y != 0)
    assert abs(keep.mean() - (1 - P)) < 5 * np.sqrt(P * (1 - P) / keep.size)
    ids = linear_ids(model)
    last = max((k for k in ids if k.startswith("info_embedding")), key=lambda k: int(k.rsplit(".", 1)[1]))
    expect = np.empty_like(y)
    L.check(L.load().pmt_dropout_mask(seed, C.c_float(P), ids[last], 0, n, y.shape[1], expect.ctypes.data), "mask")
    np.testing.assert_allclose(y, expect, rtol=1e-6)


def test_training_loop_with_dropout_runs_end_to_end(monkeypatch):
    """train_artifact_model (reference model_training.py:49-201) with --dropout_p 0.25: train epochs draw masks (DownsampledBatch
    parents included), validation / evaluation epochs run in eval mode; the losses stay finite and the model learns something"""
    import os
    from permutect_amd.data.memory_mapped_data import MemoryMappedData
    from permutect_amd.data.reads_dataset import ReadsDataset, all_but_last_fold, last_fold_only
    from permutect_amd.parameters import TrainingParameters
    from permutect_amd.training.model_training import train_artifact_model
    from tests.helpers import GOLDEN
    dev = torch.device("cuda:0")
    mm = MemoryMappedData.load_from_tarfile(os.path.join(GOLDEN, "tiny_dataset.tar"))
    train = ReadsDataset(mm, num_folds=5, folds_to_use=all_but_last_fold(5))
    valid = ReadsDataset(mm, num_folds=5, folds_to_use=last_fold_only(5))
    torch.manual_seed(0)
    params = p0_params()
    params.dropout_p = P
    model = ArtifactModel(params, device=dev, **P0_DIMS)
    seeds = []
    from permutect_amd.engine.runtime import ReadSetEngine
    draw = ReadSetEngine.draw_dropout_seed  # (patched on the class: the loop rebuilds the engine when it resets the source predictor)

    def recording(self, training):
        seeds.append(draw(self, training))
        return seeds[-1]
    monkeypatch.setattr(ReadSetEngine, "draw_dropout_seed", recording)
    hist = train_artifact_model(model, train, valid, TrainingParameters(batch_size=16, num_epochs=2, num_calibration_epochs=1, learning_rate=1e-3),
                                chunk_variants=24, seed=1, log=lambda s: None)
    assert [h[:2] for h in hist] == [(1, "TRAIN"), (1, "VALID"), (2, "TRAIN"), (2, "VALID"), (3, "TRAIN"), (3, "VALID")]
    assert all(np.isfinite(h[2]) and h[2] > 0 for h in hist)
    drawn = [s for s in seeds if s != 0]
    assert len(drawn) >= 4 and len(set(drawn)) == len(drawn) and seeds.count(0) >= 2  # new masks every train step; none in eval mode
    with torch.no_grad():
        model.eval()
        out = model.compute_batch_output(valid.host_batch(np.arange(len(valid))).copy_to(dev))
    assert torch.isfinite(out.logits_b).all()
