"""The training loop's bookkeeping against vectors produced by the REFERENCE (tests/golden/make_golden.py --helpers-only):
the (source, label, variant type, ref bin, alt bin) index of every variant (reference data/batch.py:204-230,
data/count_binning.py), the Balancer's weights over a sequence of batches that crosses its recompute threshold
(training/balancer.py:57-119) and the Downsampler's per-variant mixture weights (training/downsampler.py:105-113)."""
import os

import numpy as np
import torch

from permutect_amd.data.batch import Batch
from permutect_amd.training.balancer import Balancer
from permutect_amd.training.downsampler import Downsampler, flattened_slvra_index

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "training_helpers.npz")


def _batches(z):
    ints, per = z["int_array"], int(z["batch"])
    for k in range(len(ints) // per):
        part = ints[k * per:(k + 1) * per]
        floats = np.zeros((per, 6 + 71), dtype=np.float16)
        reads = np.zeros((int(part[:, 0].sum() + part[:, 1].sum()), 12), dtype=np.uint8)
        yield k, Batch.from_arrays(part, floats, reads)


def test_bin_index_matches_reference():
    z = np.load(GOLDEN)
    per = int(z["batch"])
    for k, batch in _batches(z):
        assert np.array_equal(flattened_slvra_index(batch).numpy(), z["flattened_idx"][k * per:(k + 1) * per])


def test_balancer_sequence_matches_reference():
    z = np.load(GOLDEN)
    per = int(z["batch"])
    bal = Balancer(num_sources=2, device=torch.device("cpu"))
    for k, batch in _batches(z):
        w, sw = bal.process_batch_and_compute_weights(batch, torch.from_numpy(z["probs"][k * per:(k + 1) * per]))
        np.testing.assert_allclose(w.numpy(), z["weights"][k * per:(k + 1) * per], rtol=1e-6, atol=1e-7, err_msg=f"batch {k}")
        np.testing.assert_allclose(sw.numpy(), z["source_weights"][k * per:(k + 1) * per], rtol=1e-6, atol=1e-7)
    assert not np.allclose(z["final_weights_slvra"], 1.0)  # the sequence did cross the recompute threshold
    np.testing.assert_allclose(bal.weights_slvra.numpy(), z["final_weights_slvra"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(bal.unlabeled_weights_slvra.numpy(), z["final_unlabeled_weights_slvra"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(bal.source_weights_s.numpy(), z["final_source_weights_s"], rtol=1e-6, atol=1e-7)
    np.testing.assert_array_equal(bal.counts_slvra.numpy(), z["final_counts_slvra"])


def test_downsampler_mixture_weights_match_reference():
    z = np.load(GOLDEN)
    per = int(z["batch"])
    down = Downsampler(num_sources=2)
    with torch.no_grad():
        down.parametrizations.log_ref_weights_slvrak.original.copy_(torch.from_numpy(z["log_ref_weights_original"]))
        down.parametrizations.log_alt_weights_slvrah.original.copy_(torch.from_numpy(z["log_alt_weights_original"]))
    for k, batch in _batches(z):
        ref_w, alt_w = down._weights_bk(batch)
        np.testing.assert_allclose(ref_w.numpy(), z["ref_weights_bk"][k * per:(k + 1) * per], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(alt_w.numpy(), z["alt_weights_bk"][k * per:(k + 1) * per], rtol=1e-6, atol=1e-7)


def test_downsampler_constants_and_state_dict_keys_match_reference():
    """The binned beta-binomial transition matrices (reference training/downsampler.py:36-89) and the module's state_dict
    keys, so that a reference Downsampler's weights load."""
    z = np.load(GOLDEN)
    down = Downsampler(num_sources=2)
    np.testing.assert_allclose(down.binned_ref_trans_kry.numpy(), z["binned_ref_trans_kry"], rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(down.binned_alt_trans_haz.numpy(), z["binned_alt_trans_haz"], rtol=1e-6, atol=1e-8)
    fit = np.load(os.path.join(os.path.dirname(GOLDEN), "downsampler_fit.npz"))
    assert sorted(down.state_dict().keys()) == list(fit["state_dict_keys"])


def test_downsampling_balance_fit_matches_reference():
    """`optimize_downsampling_balance` (reference :141-158, 10 000 deterministic AdamW steps) on the fixture's dataset
    counts: the fitted mixture weights and the expected downsampled counts they give."""
    fit = np.load(os.path.join(os.path.dirname(GOLDEN), "downsampler_fit.npz"))
    counts = torch.from_numpy(fit["counts_slvra"])
    down = Downsampler(num_sources=2)
    np.testing.assert_allclose(down.calculate_expected_downsampled_counts(counts).numpy(), fit["expected_before"], rtol=1e-5, atol=1e-3)
    down.optimize_downsampling_balance(counts)
    assert np.all(np.isfinite(fit["log_ref_weights_original"]))  # (the fixture's counts leave no cell empty: see make_golden.py)
    ref_w = torch.log_softmax(torch.from_numpy(fit["log_ref_weights_original"]), dim=-1).exp().numpy()
    alt_w = torch.log_softmax(torch.from_numpy(fit["log_alt_weights_original"]), dim=-1).exp().numpy()
    np.testing.assert_allclose(down.log_ref_weights_slvrak.exp().numpy(), ref_w, rtol=0, atol=2e-4)
    np.testing.assert_allclose(down.log_alt_weights_slvrah.exp().numpy(), alt_w, rtol=0, atol=2e-4)
    after = down.calculate_expected_downsampled_counts(counts).numpy()
    np.testing.assert_allclose(after, fit["expected_after"], rtol=2e-3, atol=1e-2 * fit["expected_after"].max() * 1e-2)
    # the fit did what it is for: the downsampled counts are spread more evenly over the count bins than before
    def unevenness(e):
        p = e / e.sum(axis=(-2, -1), keepdims=True)
        return float((p ** 2).sum())
    assert unevenness(after) < 0.8 * unevenness(fit["expected_before"])


def test_downsampling_balance_fit_survives_cells_without_data():
    """A (source, label, variant type) cell without data makes the reference's loss NaN; here it is skipped."""
    counts = torch.zeros(1, 3, 5, 4, 5)
    counts[0, 0, 0] = 5.0
    counts[0, 1, 2, 1, 3] = 9.0
    down = Downsampler(num_sources=1)
    down.optimize_downsampling_balance(counts, steps=200)
    assert torch.isfinite(down.log_ref_weights_slvrak).all() and torch.isfinite(down.log_alt_weights_slvrah).all()
    assert not torch.allclose(down.log_ref_weights_slvrak[0, 0, 0], down.log_ref_weights_slvrak[0, 2, 4])  # fitted vs untouched


def test_dataset_totals_by_bin():
    from permutect_amd.data.memory_mapped_data import MemoryMappedData
    from permutect_amd.data.reads_dataset import ReadsDataset
    ds = ReadsDataset(MemoryMappedData.load_from_tarfile(os.path.join(os.path.dirname(GOLDEN), "tiny_dataset.tar")))
    tot = ds.totals_slvra
    assert tuple(tot.shape) == (ds.num_sources(), 3, 5, 4, 5) and float(tot.sum()) == len(ds)
    batch = ds.host_batch(np.arange(len(ds)))
    want = torch.bincount(flattened_slvra_index(batch), minlength=tot.numel()).float().view(tot.shape)
    assert torch.equal(tot, want)
    assert torch.equal(tot.sum(dim=(0, 2, 3, 4)), ds.totals_by_label())


def test_evaluation_tally_matches_reference_given_the_reference_logits():
    """EvaluationCounts.record_batch (the reference's EvaluationMetrics.record_batch -> AccuracyMetrics over source, label, variant
    type, ref bin, alt bin, LOGIT bin) fed the logits the reference itself computed (evaluation_metrics.npz: first_pass_logits;
    every read kept, unit balancer weights, so each of the three passes tallies the same triples): the tensors must be the
    reference's, bin for bin.  The GPU test (tests/test_metrics_gpu.py) runs the model as well."""
    from permutect_amd.data.memory_mapped_data import MemoryMappedData
    from permutect_amd.data.reads_dataset import ReadsDataset
    from permutect_amd.training.loss_recorder import EvaluationCounts, NUM_LOGIT_BINS, logit_bin_indices
    here = os.path.dirname(GOLDEN)
    z = np.load(os.path.join(here, "evaluation_metrics.npz"))
    assert np.all(z["balancer_weights_slvra"] == 1.0)  # (41 variants never reach the balancer's recompute threshold)
    dataset = ReadsDataset(MemoryMappedData.load_from_tarfile(os.path.join(here, "tiny_dataset.tar")))
    ev = EvaluationCounts(torch.device("cpu"), num_sources=2)
    logits = torch.from_numpy(z["first_pass_logits"])
    at = 0
    for e, (ids_key, sizes_key) in enumerate((("train_ids", "train_batches"), ("valid_ids", "valid_batches"))):
        ids, k = z[ids_key], 0
        for n in z[sizes_key]:
            batch = dataset.host_batch(ids[k:k + int(n)])
            for _ in range(3):
                ev.record_batch(e, batch, logits[at:at + int(n)], torch.ones(int(n)))
            k += int(n)
            at += int(n)
    np.testing.assert_allclose(ev.hist[0].numpy(), z["accuracy_train"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(ev.hist[1].numpy(), z["accuracy_valid"], rtol=0, atol=1e-6)
    assert ev.hist.shape[-1] == NUM_LOGIT_BINS == 21
    assert logit_bin_indices(torch.tensor([-25.0, -10.0, -0.5, 0.0, 0.999, 9.99, 10.0, 30.0])).tolist() == [0, 0, 9, 10, 10, 19, 20, 20]
    # the call counts of the log line: labeled + unlabeled, three passes
    assert abs(float(ev.counts[0].sum()) - 3 * len(z["train_ids"])) < 1e-4 and abs(float(ev.counts[1].sum()) - 3 * len(z["valid_ids"])) < 1e-4
