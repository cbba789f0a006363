"""Batches composed on the device from an HBM-resident chunk (pmt_build_read_index + the kernels' gather index) against
the same variants collated on the host: the gather index must list exactly the host batch's rows, and the outputs agree
to the order-dependence of the in-kernel float atomics."""
import os

import numpy as np
import pytest
import torch

from permutect_amd.architecture.artifact_model import ArtifactModel
from permutect_amd.data.batch import DownsampledBatch
from permutect_amd.data.memory_mapped_data import MemoryMappedData
from permutect_amd.data.reads_dataset import ChunkBatch, DeviceChunk, ReadsDataset
from permutect_amd.parameters import P0_DIMS, p0_params

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _dataset():
    return ReadsDataset(MemoryMappedData.load_from_tarfile(os.path.join(GOLDEN, "tiny_dataset.tar")))


def test_device_composed_batch_equals_host_batch():
    dev = torch.device("cuda:0")
    ds = _dataset()
    torch.manual_seed(0)
    model = ArtifactModel(p0_params(), device=dev, **P0_DIMS)
    chunk = DeviceChunk(ds, 3, 37, dev)
    ids = np.array([30, 2, 2, 17, 0, 33, 9, 21], dtype=np.int64)
    cb = ChunkBatch(chunk, ids)
    hb = ds.host_batch(3 + ids).copy_to(dev)
    # the gather index lists the chunk rows in batch order
    np.testing.assert_array_equal(chunk.reads[cb.read_index()].cpu().numpy(), hb.packed_reads.cpu().numpy())
    with torch.no_grad():
        a = model.compute_batch_output(cb)
        b = model.compute_batch_output(hb)
    assert torch.allclose(a.logits_b, b.logits_b, rtol=1e-5, atol=1e-5) and torch.allclose(a.features_be, b.features_be, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("mode", ["consumer", "prefetch"])
def test_loader_batches_equal_host_batches_in_every_composition_mode(mode, monkeypatch):
    """Every batch the device chunk loader hands out -- composed in ONE launch from host-side count scans (pmt_compose_batch_planned),
    on the consumer's stream (default) or by the prefetch thread -- holds exactly the rows, tables, offsets and gather index of the
    same variants collated on the host; and the unplanned composition (pmt_compose_batch: scans on the device) gives the same."""
    monkeypatch.setenv("PMT_LOADER_COMPOSE", mode)
    dev = torch.device("cuda:0")
    ds = _dataset()
    seen = 0
    for cb in ds.device_loader(batch_size=16, device=dev, chunk_variants=24, rng=np.random.default_rng(9), shuffle=True):
        hb = ds.host_batch(cb.dataset_index).copy_to(dev)
        np.testing.assert_array_equal(cb._chunk.reads[cb.read_index()].cpu().numpy(), hb.packed_reads.cpu().numpy())
        np.testing.assert_array_equal(cb.int_tensor.cpu().numpy(), hb.int_tensor.cpu().numpy())
        np.testing.assert_allclose(cb.float_tensor.cpu().numpy(), hb.float_tensor.cpu().numpy(), rtol=0, atol=0)
        nref, nalt = hb.host_counts()
        ref_off, alt_off = cb._offsets
        np.testing.assert_array_equal(ref_off.cpu().numpy(), np.concatenate([[0], np.cumsum(nref)]))
        np.testing.assert_array_equal(alt_off.cpu().numpy(), np.concatenate([[0], np.cumsum(nalt)]))
        # the same batch composed with the scans made on the device
        plain = ChunkBatch.compose_on_device(cb._chunk, cb.chunk_ids, int(nref.sum() + nalt.sum()))
        for a, b in zip(plain, (cb.int_tensor, cb.float_tensor, cb._row_start, ref_off, alt_off, cb.read_index())):
            assert torch.equal(a, b)
        seen += cb.size()
    assert seen == len(ds)


def test_device_loader_and_downsampling_compose():
    dev = torch.device("cuda:0")
    ds = _dataset()
    torch.manual_seed(0)
    model = ArtifactModel(p0_params(), device=dev, **P0_DIMS)
    loader = ds.device_loader(batch_size=16, device=dev, chunk_variants=20, rng=np.random.default_rng(5))
    total = 0
    for cb in loader:
        total += cb.size()
        ones = torch.ones(cb.size(), device=dev)
        db = DownsampledBatch(cb, ones, ones, fix_alt_gather=True)  # keeps every read: must equal the parent
        with torch.no_grad():
            a = model.compute_batch_output(cb)
            b = model.compute_batch_output(db)
        assert torch.allclose(a.logits_b, b.logits_b, rtol=1e-5, atol=1e-5)
        assert torch.isfinite(a.logits_b).all()
    assert total == len(ds) and loader.bytes_uploaded > 0


def test_posterior_handoff_matches_per_datum_semantics():
    """tools/posterior_data.make_posterior_mmap against the reference's per-variant construction (filter_variants.py:
    302-320) emulated with Datum: counts zeroed, logit through float16 into CACHED_ARTIFACT_LOGIT, info := embedding."""
    from permutect_amd.data.datum import Data, Datum
    from permutect_amd.tools.posterior_data import make_posterior_mmap
    dev = torch.device("cuda:0")
    ds = _dataset()
    torch.manual_seed(0)
    model = ArtifactModel(p0_params(), device=dev, **P0_DIMS)
    # one batch of all 41 variants: the loader's group packer reorders the variants inside it (also without shuffling),
    # the rows must come back in dataset order all the same
    reordered = [not np.all(np.diff(cb.dataset_index) > 0) for cb in ds.device_loader(64, dev, shuffle=False)]
    assert any(reordered)
    post = make_posterior_mmap(ds, model, batch_size=64)
    post16 = make_posterior_mmap(ds, model, batch_size=16, chunk_variants=20)  # (batches the packer leaves in order)
    np.testing.assert_array_equal(post.int_mmap[:len(ds)], post16.int_mmap[:len(ds)])
    np.testing.assert_allclose(post.float_mmap[:len(ds)], post16.float_mmap[:len(ds)], rtol=1e-5, atol=2e-3)
    assert len(post) == len(ds) and post.reads_mmap is None and post.num_reads == 0
    with torch.no_grad():
        out = model.compute_batch_output(ds.host_batch(np.arange(len(ds))).copy_to(dev))
    logits, feats = out.logits_b.cpu().numpy(), out.features_be.cpu().numpy()
    for i in range(len(ds)):
        d = Datum(np.array(ds._ints[i]), np.array(ds._floats[i]), np.zeros((0, 12), dtype=np.uint8), compressed=True)
        d.set(Data.REF_COUNT, 0)
        d.set(Data.ALT_COUNT, 0)
        d.set(Data.CACHED_ARTIFACT_LOGIT, logits[i])
        want_float = np.hstack((d.float_array[:6], feats[i]))
        np.testing.assert_array_equal(post.int_mmap[i], d.int_array)
        np.testing.assert_allclose(post.float_mmap[i], want_float, rtol=1e-5, atol=2e-3)  # fp16 logit, atomics-order embedding
        assert abs(float(post.float_mmap[i][5]) - float(np.float16(logits[i]))) <= 2e-2
    # the posterior dataset loads like any other (no reads)
    ReadsDataset(post)


def test_loss_recorder_kernel_matches_index_add():
    """pmt_record_losses (one launch) against the reference's eight index_add_ into [S, L, V, R, A] histograms."""
    from permutect_amd.data.datum import Data
    from permutect_amd.training.downsampler import flattened_slvra_index
    from permutect_amd.training.loss_recorder import ALT_COUNT, PRIMARY, SOURCE, LossRecorder, evaluate
    dev = torch.device("cuda:0")
    ds = _dataset()
    torch.manual_seed(0)
    model = ArtifactModel(p0_params(), device=dev, **P0_DIMS)
    rec = LossRecorder(dev, num_sources=2)
    want = torch.zeros(3, 2, rec.num_bins, device=dev, dtype=torch.float64)
    for batch in ds.device_loader(16, dev, chunk_variants=20, shuffle=False):
        with torch.no_grad():
            out = model.compute_batch_output(batch)
            out.weights = 0.5 + torch.rand(batch.size(), device=dev)
            out.source_weights = 0.5 + torch.rand(batch.size(), device=dev)
            losses = model.compute_batch_losses(out, batch)
        rec.record(out, losses, batch)
        idx = flattened_slvra_index(batch)
        il = batch.get_is_labeled_mask().double()
        w, sw = out.weights.double(), out.source_weights.double()
        want[PRIMARY, 0].index_add_(0, idx, losses.supervised_losses_b.double() * il * w + losses.unsupervised_losses_b.double() * (1 - il) * w)
        want[PRIMARY, 1].index_add_(0, idx, w)
        want[ALT_COUNT, 0].index_add_(0, idx, losses.alt_count_losses_b.double() * w)
        want[ALT_COUNT, 1].index_add_(0, idx, w)
        want[SOURCE, 0].index_add_(0, idx, losses.source_prediction_losses_b.double() * sw)
        want[SOURCE, 1].index_add_(0, idx, sw)
    np.testing.assert_allclose(rec.hist.cpu().numpy(), want.float().cpu().numpy(), rtol=1e-5, atol=1e-5)
    assert rec.mean_loss() > 0 and rec.averages().shape == (2, 3, 5, 4, 5)
    ev = evaluate(model, ds.device_loader(16, dev, chunk_variants=20, shuffle=False), num_sources=2)
    assert abs(float(ev.counts().sum()) - len(ds)) < 1e-3


def test_train_artifact_model_loop_end_to_end():
    """The reference's training loop shape (model_training.py:49-201) on the engine: device-composed batches, two fused
    downsamplings per parent batch, balancer, recorder, scheduler, checkpoint, a calibration epoch that moves only the
    calibration parameters."""
    from permutect_amd.data.reads_dataset import all_but_last_fold, last_fold_only
    from permutect_amd.parameters import TrainingParameters
    from permutect_amd.training.model_training import train_artifact_model
    dev = torch.device("cuda:0")
    mm = MemoryMappedData.load_from_tarfile(os.path.join(GOLDEN, "tiny_dataset.tar"))
    train = ReadsDataset(mm, num_folds=5, folds_to_use=all_but_last_fold(5))
    valid = ReadsDataset(mm, num_folds=5, folds_to_use=last_fold_only(5))
    assert len(train) + len(valid) == len(mm)
    torch.manual_seed(0)
    model = ArtifactModel(p0_params(), device=dev, **P0_DIMS)
    logs = []
    evals = []
    hist = train_artifact_model(model, train, valid, TrainingParameters(batch_size=16, num_epochs=2, num_calibration_epochs=1,
                                                                        learning_rate=1e-3), chunk_variants=24, seed=1, log=logs.append,
                                evaluations=evals)
    assert [h[:2] for h in hist] == [(1, "TRAIN"), (1, "VALID"), (2, "TRAIN"), (2, "VALID"), (3, "TRAIN"), (3, "VALID")]
    assert all(np.isfinite(h[2]) and h[2] > 0 for h in hist)
    # one line per epoch half plus the evaluation pass after every validation epoch (3 downsamplings of train + valid)
    # (33 training variants: an epoch's mean loss can double from noise alone, and the loop then rolls back to its best checkpoint
    #  and says so -- the reference's behaviour, training/checkpoint.py:27-32; such lines are allowed, not counted)
    regular = [ln for ln in logs if "restored the best checkpoint" not in ln]
    assert len(regular) == 9 and len(logs) - len(regular) <= 2, logs
    assert len(evals) == 3 and all(0.0 <= a <= 1.0 and 0.0 <= b <= 1.0 for _, a, b in evals)
    # the model still produces finite outputs after training
    with torch.no_grad():
        out = model.compute_batch_output(valid.host_batch(np.arange(len(valid))).copy_to(dev))
    assert torch.isfinite(out.logits_b).all()


@pytest.mark.parametrize("n", [0, 1, 4095, 4096, 4097, 70001, 300000, 1 << 20])
@pytest.mark.parametrize("wide", [True, False])
def test_count_scans_of_any_length(n, wide):
    """pmt_scan_counts (exclusive scans of the ref / alt count columns, bit-exact): lengths around the 4096-element chunk,
    beyond one segment and beyond 64 x 4096; int64 counts in a strided table and contiguous int32 counts."""
    import ctypes as C
    from permutect_amd.engine import lib as L
    lib = L.load()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(n + (1 if wide else 0))
    ref, alt = rng.integers(0, 11, n), rng.integers(1, 16, n)
    if wide:
        table = torch.zeros(max(n, 1), 7, dtype=torch.int64)
        table[:n, 0], table[:n, 1] = torch.from_numpy(ref), torch.from_numpy(alt)
        table = table.to(dev)
        rc, ac, elem, stride = table[:, 0], table[:, 1], 8, table.stride(0)
    else:
        rc, ac = torch.from_numpy(ref.astype(np.int32)).to(dev), torch.from_numpy(alt.astype(np.int32)).to(dev)
        if n == 0:
            rc, ac = torch.zeros(1, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
        elem, stride = 4, 1
    ro = torch.full((n + 1,), -1, dtype=torch.int32, device=dev)
    ao = torch.full((n + 1,), -1, dtype=torch.int32, device=dev)
    L.check(lib.pmt_scan_counts(rc.data_ptr(), ac.data_ptr(), elem, stride, n, ro.data_ptr(), ao.data_ptr(),
                                torch.cuda.current_stream().cuda_stream), "pmt_scan_counts")
    assert np.array_equal(ro.cpu().numpy(), np.concatenate([[0], np.cumsum(ref)]).astype(np.int32))
    assert np.array_equal(ao.cpu().numpy(), np.concatenate([[0], np.cumsum(alt)]).astype(np.int32))


def test_training_loop_learns_a_separable_dataset():
    """System check of the whole training path (loader, fused downsampling, balancer, forward, losses, backward, clip +
    AdamW, scheduler): artifacts differ from true variants in one read feature of their alt reads; after a few epochs the
    validation loss has dropped and the held-out fold is classified."""
    from bench import synth_arrays
    from permutect_amd.data.reads_dataset import all_but_last_fold, last_fold_only
    from permutect_amd.parameters import TrainingParameters
    from permutect_amd.training.model_training import train_artifact_model
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5)
    n = 6000
    ints, floats, packed = synth_arrays(rng, n, "wgs")
    ints[:, 2] = np.arange(n) % 2  # ARTIFACT / VARIANT, everything labeled
    nref, nalt = ints[:, 0].astype(np.int64), ints[:, 1].astype(np.int64)
    # on-disk order: per datum its ref rows then its alt rows; mark the alt rows of artifacts in the first float column
    starts = np.concatenate([[0], np.cumsum(nref + nalt)])
    for v in np.nonzero(ints[:, 2] == 0)[0]:
        a0 = starts[v] + nref[v]
        packed[a0:a0 + nalt[v], 7] = rng.integers(200, 256, nalt[v])   # decodes to 2.25 .. 4
    for v in np.nonzero(ints[:, 2] == 1)[0]:
        a0 = starts[v] + nref[v]
        packed[a0:a0 + nalt[v], 7] = rng.integers(128, 184, nalt[v])   # decodes to 0 .. 1.75
    mm = MemoryMappedData.from_arrays(ints, floats, packed)
    train = ReadsDataset(mm, num_folds=5, folds_to_use=all_but_last_fold(5))
    valid = ReadsDataset(mm, num_folds=5, folds_to_use=last_fold_only(5))
    torch.manual_seed(1)
    model = ArtifactModel(p0_params(), device=dev, **P0_DIMS)
    hist = train_artifact_model(model, train, valid, TrainingParameters(batch_size=64, num_epochs=6, learning_rate=2e-3, fit_downsampler=False),
                                chunk_variants=2048, seed=2, log=lambda *_: None, fix_alt_gather=True)
    valid_losses = [h[2] for h in hist if h[1] == "VALID"]
    assert max(valid_losses[-2:]) < 0.2, valid_losses  # chance level is log 2 = 0.69 plus the adversary terms
    model.train(False)
    batch = valid.host_batch(np.arange(len(valid))).copy_to(dev)
    with torch.no_grad():
        out = model.compute_batch_output(batch)
    labels = batch.int_tensor[:, 2]
    predicted_artifact = out.logits_b > 0
    accuracy = float((predicted_artifact == (labels == 0)).float().mean())
    assert accuracy > 0.95, accuracy
    # The reference's own alt gather hands the training step ref rows in place of the kept alt reads, so a
    # signal that lives only in the alt reads is invisible to it: same data, same loop, the loss stays at chance.
    torch.manual_seed(1)
    quirk = ArtifactModel(p0_params(), device=dev, **P0_DIMS)
    hist_q = train_artifact_model(quirk, train, valid, TrainingParameters(batch_size=64, num_epochs=3, learning_rate=2e-3, fit_downsampler=False),
                                  chunk_variants=2048, seed=2, log=lambda *_: None)
    assert min(h[2] for h in hist_q if h[1] == "VALID") > 0.4, hist_q


def test_evaluation_pass_runs_three_downsamplings_over_both_loaders():
    """collect_evaluation_data (reference training/model_training.py:204-228): every parent batch of the training and of the
    validation loader is downsampled three times and run forward; the weighted label counts add up to 3 x the data."""
    from permutect_amd.data.reads_dataset import all_but_last_fold, last_fold_only
    from permutect_amd.training.balancer import Balancer
    from permutect_amd.training.downsampler import Downsampler
    from permutect_amd.training.loss_recorder import collect_evaluation_data
    dev = torch.device("cuda:0")
    mm = MemoryMappedData.load_from_tarfile(os.path.join(GOLDEN, "tiny_dataset.tar"))
    train = ReadsDataset(mm, num_folds=5, folds_to_use=all_but_last_fold(5))
    valid = ReadsDataset(mm, num_folds=5, folds_to_use=last_fold_only(5))
    torch.manual_seed(0)
    model = ArtifactModel(p0_params(), device=dev, **P0_DIMS)
    ev = collect_evaluation_data(model, None, Downsampler(2).to(dev), train.device_loader(16, dev, shuffle=False),
                                 valid.device_loader(16, dev, shuffle=False), seed=4, fix_alt_gather=True)
    counts = ev.counts.cpu().numpy()
    assert abs(counts[0].sum() - 3 * len(train)) < 1e-3 and abs(counts[1].sum() - 3 * len(valid)) < 1e-3  # unit weights
    assert ev.batches == 3 * (len(train.device_loader(16, dev)) + len(valid.device_loader(16, dev)))
    labels = np.asarray(train._ints[: len(train), 2])
    np.testing.assert_allclose(counts[0].sum(axis=1), 3 * np.bincount(labels, minlength=3), atol=1e-3)
    assert 0.0 <= ev.accuracy(0) <= 1.0


def test_device_posterior_rows_match_the_reference_fixture_directly():
    """the DEVICE hand-off pinned by the reference's own rows, not by the repo's Datum emulation.  The fixture
    (tests/golden/posterior_rows.npz, written by make_golden.py from the reference's loop in generate_posterior_data,
    tools/filter_variants.py:302-320) holds a batch, the logits / embeddings the reference model produced and the rows its loop made
    of them; the same logits / embeddings go into pmt_posterior_rows (one launch) and must give the same float rows BIT FOR BIT, in
    order and scattered through a permutation."""
    import os
    from permutect_amd.tools.posterior_data import device_posterior_rows
    from tests.helpers import GOLDEN
    z = np.load(os.path.join(GOLDEN, "posterior_rows.npz"))
    dev = torch.device("cuda:0")
    floats = torch.from_numpy(z["batch_float"]).to(dev)
    logits, feats = torch.from_numpy(z["logits_b"]).to(dev), torch.from_numpy(z["features_be"]).to(dev)
    want = z["out_float"]
    n, width = want.shape
    block = torch.full((n, width), float("nan"), dtype=torch.float32, device=dev)
    device_posterior_rows(floats, logits, feats, None, block)
    np.testing.assert_array_equal(block.cpu().numpy(), want)
    perm = torch.from_numpy(np.random.default_rng(3).permutation(n)).to(dev)
    scattered = torch.full((n + 5, width), float("nan"), dtype=torch.float32, device=dev)
    device_posterior_rows(floats, logits, feats, perm, scattered)
    got = scattered.cpu().numpy()
    np.testing.assert_array_equal(got[perm.cpu().numpy()], want)
    assert np.isnan(got[n:]).all()


def test_train_artifact_model_tool_runs_the_references_own_tool_test():
    """BASELINE configs[0] on a synthetic tar in the reference's format: `tools/train_artifact_model.main_without_parsing` with the
    Namespace of the reference's tool test (test/tools/test_train_permutect_model.py:14-64: T0 hyperparameters, two epochs, batch 64)
    -- same flag names through the same `parse_*_params` -- trains, saves a `.pt` in the reference's dictionary format, and the
    saved model loads back and filters.  (The reference asserts only that much: events load, the model re-loads.)"""
    import argparse
    import os
    import tempfile
    from permutect_amd import constants
    from permutect_amd.architecture.artifact_model import load_model
    from permutect_amd.parameters import T0_CNN
    from permutect_amd.tools import train_artifact_model as tool
    from tests.helpers import GOLDEN
    args = argparse.Namespace()
    for name, value in ((constants.READ_LAYERS_NAME, [10, 10, 10]), (constants.SELF_ATTENTION_HIDDEN_DIMENSION_NAME, 20),
                        (constants.NUM_SELF_ATTENTION_LAYERS_NAME, 2), (constants.INFO_LAYERS_NAME, [10, 10]),
                        (constants.AGGREGATION_LAYERS_NAME, [20, 20, 20]), (constants.NUM_ARTIFACT_CLUSTERS_NAME, 4),
                        (constants.CALIBRATION_LAYERS_NAME, [10, 10, 10]), (constants.REF_SEQ_LAYER_STRINGS_NAME, list(T0_CNN)),
                        (constants.DROPOUT_P_NAME, 0.0), (constants.BATCH_NORMALIZE_NAME, False),
                        (constants.TRAIN_TAR_NAME, os.path.join(GOLDEN, "tiny_dataset.tar")), (constants.PRETRAINED_ARTIFACT_MODEL_NAME, None),
                        (constants.REWEIGHTING_RANGE_NAME, 0.3), (constants.BATCH_SIZE_NAME, 64), (constants.INFERENCE_BATCH_SIZE_NAME, 64),
                        (constants.NUM_WORKERS_NAME, 2), (constants.NUM_EPOCHS_NAME, 2), (constants.NUM_CALIBRATION_EPOCHS_NAME, 0),
                        (constants.LEARNING_RATE_NAME, 0.001), (constants.WEIGHT_DECAY_NAME, 0.01)):
        setattr(args, name, value)
    with tempfile.TemporaryDirectory() as d:
        setattr(args, constants.OUTPUT_NAME, os.path.join(d, "model.pt"))
        setattr(args, constants.TENSORBOARD_DIR_NAME, os.path.join(d, "tb"))
        history = tool.main_without_parsing(args, log=lambda *_: None)
        assert [h[:2] for h in history] == [(1, "TRAIN"), (1, "VALID"), (2, "TRAIN"), (2, "VALID")] and all(np.isfinite(h[2]) for h in history)
        model, _, _ = load_model(getattr(args, constants.OUTPUT_NAME), device=torch.device("cuda:0"))
    ds = _dataset()
    model.eval()
    with torch.inference_mode():
        for cb in ds.device_loader(64, torch.device("cuda:0"), shuffle=False):
            assert torch.isfinite(model.compute_batch_output(cb).logits_b).all()
    # and the command line itself parses the reference's flags
    ns = tool.parse_arguments(["--train_tar", "x.tar", "--output", "m.pt", "--read_layers", "30", "-2", "--self_attention_hidden_dimension", "20",
                               "--num_self_attention_layers", "6", "--info_layers", "20", "-2", "--aggregation_layers", "-2", "10",
                               "--calibration_layers", "10", "10", "--ref_seq_layer_strings", "flatten", "linear/out_features=10", "--num_epochs", "3"])
    assert ns.read_layers == [30, -2] and ns.batch_size == 64 and ns.inference_batch_size == 8192 and ns.weight_decay == 0.0 and ns.num_epochs == 3
