"""Generate golden vectors by running the REFERENCE implementation (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Imports /root/reference/permutect with three in-process stubs for I/O-only third-party modules that are not
installed here (cyvcf2, intervaltree, torch.utils.tensorboard); none of them carries arithmetic.
Writes tests/golden/*.npz: inputs in the reference's on-disk dtypes (uint8 packed reads, int16 int array, float16
float array), the full state_dict, and the reference's outputs / losses / gradients / post-AdamW parameters.
Only data is written -- no reference source text.  The fixtures travel to the GPU box; the reference does not.
"""
import os
import random
import sys
import types

REFERENCE = os.environ.get("PERMUTECT_REFERENCE", "/root/reference")
sys.path.insert(0, REFERENCE)
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

cy = types.ModuleType("cyvcf2"); cy.VCF = cy.Variant = cy.Writer = object; sys.modules["cyvcf2"] = cy
it = types.ModuleType("intervaltree"); it.IntervalTree = dict; sys.modules["intervaltree"] = it
tb = types.ModuleType("torch.utils.tensorboard")


class SummaryWriter:
    def __init__(self, *a, **k):
        pass

    def __getattr__(self, n):
        return lambda *a, **k: None


tb.SummaryWriter = SummaryWriter
sys.modules["torch.utils.tensorboard"] = tb

import numpy as np  # noqa: E402
import torch  # noqa: E402
from permutect.architecture.artifact_model import ArtifactModel  # noqa: E402
from permutect.data.batch import Batch, DownsampledBatch  # noqa: E402
from permutect.data.datum import Data, Datum  # noqa: E402
from permutect.misc_utils import backpropagate  # noqa: E402
from permutect.parameters import ModelParameters  # noqa: E402

from permutect_amd.parameters import P0_CNN, P0_CNN_BATCHNORM, P0_CNN_LEGACY, T0_CNN, T0_CNN_OPTIONS  # noqa: E402  (plain lists of layer strings)

CPU = torch.device("cpu")


def make_model(kind, seed, perturb=0.05, num_sources=1, cnn=None):
    torch.manual_seed(seed)
    if kind == "T0":
        p = ModelParameters([10, 10, 10], 20, 2, [10, 10], [20, 20, 20], 4, [10, 10, 10], list(cnn or T0_CNN), 0.0, 0.3, False)
    elif kind == "SKIP34":  # skip blocks of three and four layers (the reference takes any depth: architecture/mlp.py:15-22)
        p = ModelParameters([30, -3, -2], 20, 2, [20, -3], [-4, 10], 4, [10, 10], list(cnn or P0_CNN), 0.0, 0.3, False)
    elif kind in ("WIDE", "WIDE64"):  # layers beyond 64: read width 48, info width 40 -> d_model 98, a 98-wide reducer, d_ffn 32 (64), feature_dim 20
        p = ModelParameters([48, -2], 64 if kind == "WIDE64" else 32, 2, [40, -1], [-1, 20], 4, [10, 10], list(cnn or P0_CNN), 0.0, 0.3, False)
    else:
        p = ModelParameters([30, -2, -2, -2], 20, 6, [20, -2, -2, -2], [-2, -2, 10], 4, [10, 10], list(cnn or P0_CNN), 0.0, 0.3, False)
    m = ArtifactModel(p, 61, 71, 42, device=CPU)
    if num_sources > 1:
        m.reset_source_predictor(num_sources)
    with torch.no_grad():
        for q in m.parameters():
            q.add_(perturb * torch.randn_like(q))
    return m


def make_data(rng, counts, labels=None, sources=None, info_scale=1.0, byte_lo=0, byte_hi=256):
    data = []
    for i, (nr, na) in enumerate(counts):
        ints = np.zeros(16 + 42, dtype=np.int16)
        ints[16:] = rng.integers(0, 5, size=42)
        floats = np.zeros(6 + 71, dtype=np.float16)
        floats[6:] = (info_scale * rng.standard_normal(71)).astype(np.float16)
        reads = rng.integers(byte_lo, byte_hi, size=(nr + na, 12), dtype=np.uint8)
        d = Datum(ints, floats, reads, compressed=True)
        d.set(Data.REF_COUNT, nr)
        d.set(Data.ALT_COUNT, na)
        d.set(Data.LABEL, (i % 3) if labels is None else labels[i])
        d.set(Data.VARIANT_TYPE, i % 5)
        d.set(Data.SOURCE, 0 if sources is None else sources[i])
        data.append(d)
    return data


def run_case(name, model, data, train=True, lr=1e-3, wd=0.01, source_strength=None):
    batch = Batch(data)
    ref_counts = batch.get(Data.REF_COUNT)
    total_ref = int(ref_counts.sum())
    packed = np.vstack([d.get_ref_reads_re() for d in data] + [d.get_alt_reads_re() for d in data])
    assert packed.shape[0] == batch.get_reads_re().shape[0]
    out = {"packed_reads": packed, "int_array": batch.int_tensor.numpy().astype(np.int16),
           "float_array": batch.float_tensor.numpy().astype(np.float16),
           "reads_re_f16": batch.get_reads_re().numpy(), "total_ref": np.int64(total_ref)}
    for k, v in model.state_dict().items():
        out["sd/" + k] = v.detach().numpy().copy()
    if source_strength is not None:
        model.source_predictor.set_adversarial_strength(source_strength)
        out["source_strength"] = np.float64(source_strength)
    model.train(True)
    output = model.compute_batch_output(batch, None)
    losses = model.compute_batch_losses(output, batch)
    ref_bre, alt_bre, hap = model.calculate_features(batch)
    for k in ("features_be", "ref_features_be", "logits_b", "logits_bk", "artifact_probs_b", "outlier_binary_logits"):
        out["out/" + k] = getattr(output, k).detach().numpy()
    out["out/final_ref_re"] = ref_bre.flattened_tensor_nf.detach().numpy()
    out["out/final_alt_re"] = alt_bre.flattened_tensor_nf.detach().numpy()
    out["out/ref_seq_embeddings_be"] = hap.detach().numpy()
    for k in ("supervised_losses_b", "unsupervised_losses_b", "alt_count_losses_b", "source_prediction_losses_b",
              "total_losses_b", "total_loss"):
        out["loss/" + k] = getattr(losses, k).detach().numpy()
    if train:
        opt = torch.optim.AdamW(model.parameters(), lr=lr, weight_decay=wd)
        # backpropagate() = zero_grad, backward, clip_grad_norm_(1.0), step.  Capture raw grads first by hooks.
        raw = {}
        handles = [q.register_hook(lambda g, n=n: raw.__setitem__(n, g.detach().clone()))
                   for n, q in model.named_parameters()]
        backpropagate(opt, losses.total_loss, params_to_clip=model.parameters())
        for h in handles:
            h.remove()
        for n, q in model.named_parameters():
            g = raw.get(n)
            out["grad/" + n] = (torch.zeros_like(q) if g is None else g).numpy()
            out["after/" + n] = q.detach().numpy().copy()
        out["lr"], out["weight_decay"] = np.float64(lr), np.float64(wd)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "B", len(data), "R", packed.shape[0], "total_loss", float(losses.total_loss),
          "logits", output.logits_b.detach().numpy()[:4])


def main():
    rng = np.random.default_rng(0)
    random.seed(0)

    # -- case 1: reference test configuration, small batch with n_ref = 0 and n_alt = 1 members
    counts = [(3, 2), (0, 1), (10, 15), (1, 1), (0, 4), (7, 3), (2, 9), (5, 5)]
    run_case("t0_b8", make_model("T0", 1), make_data(rng, counts))

    # -- case 2: production-shaped configuration
    counts = [(int(rng.integers(0, 11)), int(rng.integers(1, 16))) for _ in range(16)]
    run_case("p0_b16", make_model("P0", 2), make_data(rng, counts))

    # -- case 3: P0, every variant has zero ref reads (artifact_model.py:253 TODO), forward + train
    counts = [(0, int(rng.integers(1, 16))) for _ in range(6)]
    try:
        run_case("p0_zero_ref", make_model("P0", 3), make_data(rng, counts))
    except Exception as exc:  # record that the reference itself cannot do it
        print("p0_zero_ref: reference raised", type(exc).__name__, exc)

    # -- case 4: large-magnitude embeddings: logits saturate the +-20 tanh cap, EMG argument on both sides of z = 5
    m = make_model("P0", 4, perturb=0.05)
    with torch.no_grad():
        m.pre_clustering_transform.translation_e.add_(3.0)
        m.feature_clustering.artifact_emg.mu_k.copy_(torch.tensor([6.0, -4.0, 1.0, 12.0]))
    counts = [(int(rng.integers(0, 11)), int(rng.integers(1, 16))) for _ in range(12)]
    run_case("p0_saturated", m, make_data(rng, counts, info_scale=3.0))

    # -- case 5: deeper sets than the reference pipeline produces (count-agnostic math), forward + train
    counts = [(40, 70), (0, 33), (64, 1), (17, 48)]
    run_case("p0_deep", make_model("P0", 5), make_data(rng, counts))

    # -- case 6: two sources -> source predictor with hidden skip blocks and the adversarial source loss
    counts = [(int(rng.integers(0, 11)), int(rng.integers(1, 16))) for _ in range(10)]
    srcs = [i % 2 for i in range(10)]
    run_case("t0_two_sources", make_model("T0", 6, num_sources=2), make_data(rng, counts, sources=srcs),
             source_strength=0.3)

    # -- quirk A: uint8 decode of every byte value in a float column, and of every bit pattern byte
    reads = np.zeros((256, 12), dtype=np.uint8)
    reads[:, 0] = np.arange(256)
    reads[:, 6] = np.arange(256)[::-1]
    reads[:, 7] = np.arange(256)
    reads[:, 11] = np.arange(256)[::-1]
    ints = np.zeros(16 + 42, dtype=np.int16)
    floats = np.zeros(6 + 71, dtype=np.float16)
    d = Datum(ints, floats, reads, compressed=True)
    d.set(Data.REF_COUNT, 128)
    d.set(Data.ALT_COUNT, 128)
    b = Batch([d])
    np.savez_compressed(os.path.join(HERE, "quirk_decode.npz"), packed_reads=reads, reads_re_f16=b.get_reads_re().numpy())

    # -- quirk B: DownsampledBatch gathers alt rows un-offset
    counts = [(10, 3), (8, 5), (6, 2)]
    data = make_data(rng, counts)
    parent = Batch(data)
    db = DownsampledBatch(parent, torch.ones(3), torch.ones(3))
    torch.manual_seed(11)
    random.seed(11)
    db2 = DownsampledBatch(parent, torch.tensor([0.5, 0.3, 0.9]), torch.tensor([0.5, 0.5, 0.5]))
    np.savez_compressed(
        os.path.join(HERE, "quirk_downsample.npz"),
        ref_counts=np.array([c[0] for c in counts]), alt_counts=np.array([c[1] for c in counts]),
        read_indices_all_kept=db.read_indices.numpy(), new_ref_counts_all_kept=db.ref_counts.numpy(),
        new_alt_counts_all_kept=db.alt_counts.numpy(),
        read_indices_half=db2.read_indices.numpy(), new_ref_counts_half=db2.ref_counts.numpy(),
        new_alt_counts_half=db2.alt_counts.numpy(),
        gathered_reads_half=db2.get_reads_re().numpy(), parent_reads=parent.get_reads_re().numpy(),
    )
    print("quirks written")


def make_dataset_fixture():
    """tiny_dataset.tar: written by the reference's own MemoryMappedData (from_generator with small capacity estimates, so
    the arrays carry unused capacity rows) + what the reference reads back from it and collates from it."""
    from permutect.data.memory_mapped_data import MemoryMappedData
    rng = np.random.default_rng(1234)
    counts = [(int(rng.integers(0, 11)), int(rng.integers(1, 16))) for _ in range(41)]
    counts[5] = (0, 1)
    data = make_data(rng, counts, sources=[i % 2 for i in range(41)])
    mm = MemoryMappedData.from_generator(iter(data), estimated_num_data=16, estimated_num_reads=64)
    tar_path = os.path.join(HERE, "tiny_dataset.tar")
    mm.save_to_tarfile(tar_path)
    back = MemoryMappedData.load_from_tarfile(tar_path)
    ids = [7, 0, 40, 5, 5, 19]
    datums = list(back.generate())
    b = Batch([datums[i] for i in ids])
    folds = [n for n, _ in enumerate(datums) if n % 3 in {0, 2}]
    restricted = back.restrict_to_folds(3, [0, 2])
    np.savez_compressed(
        os.path.join(HERE, "tiny_dataset_expected.npz"),
        num_data=back.num_data, num_reads=back.num_reads, read_end_indices=back.read_end_indices,
        int_array=np.asarray(back.int_mmap[: back.num_data]), float_array=np.asarray(back.float_mmap[: back.num_data]),
        reads=np.asarray(back.reads_mmap[: back.num_reads]),
        batch_ids=np.array(ids), batch_int=b.int_tensor.numpy(), batch_float=b.float_tensor.numpy(),
        batch_reads_f16=b.get_reads_re().numpy(),
        fold_ids=np.array(folds), fold_num_data=restricted.num_data, fold_num_reads=restricted.num_reads,
        fold_int=np.asarray(restricted.int_mmap[: restricted.num_data]),
        fold_reads=np.asarray(restricted.reads_mmap[: restricted.num_reads]),
    )
    print("dataset fixture written:", os.path.getsize(tar_path), "bytes")


def make_training_helpers_fixture():
    """training_helpers.npz: the reference's bin indexing (BatchIndices.flattened_idx), its Balancer run over a sequence of
    batches that crosses the recompute threshold, and its Downsampler's constants and per-variant mixture weights."""
    from permutect.training.balancer import Balancer
    from permutect.training.downsampler import Downsampler
    rng = np.random.default_rng(77)
    torch.manual_seed(77)
    nb, per = 12, 1000
    n = nb * per
    ref = rng.integers(0, 40, n)
    alt = rng.integers(1, 30, n)
    labels = rng.integers(0, 3, n)
    vtypes = rng.integers(0, 5, n)
    sources = rng.integers(0, 2, n)
    probs = rng.random(n).astype(np.float32)
    int_array = np.zeros((n, 16 + 42), dtype=np.int16)
    flat, weights, source_weights = [], [], []
    bal = Balancer(num_sources=2, device=CPU)
    down = Downsampler(num_sources=2)
    with torch.no_grad():
        down.parametrizations.log_ref_weights_slvrak.original.add_(torch.randn_like(down.parametrizations.log_ref_weights_slvrak.original))
        down.parametrizations.log_alt_weights_slvrah.original.add_(torch.randn_like(down.parametrizations.log_alt_weights_slvrah.original))
    ref_w, alt_w = [], []
    for k in range(nb):
        data = []
        for i in range(k * per, (k + 1) * per):
            ints = np.zeros(16 + 42, dtype=np.int16)
            floats = np.zeros(6 + 71, dtype=np.float16)
            d = Datum(ints, floats, np.zeros((1, 12), dtype=np.uint8), compressed=True)  # (reads are irrelevant here)
            d.set(Data.REF_COUNT, int(ref[i])); d.set(Data.ALT_COUNT, int(alt[i])); d.set(Data.LABEL, int(labels[i]))
            d.set(Data.VARIANT_TYPE, int(vtypes[i])); d.set(Data.SOURCE, int(sources[i]))
            int_array[i] = d.get_int_array()
            data.append(d)
        batch = Batch(data)
        flat.append(batch.batch_indices().flattened_idx.numpy())
        w, sw = bal.process_batch_and_compute_weights(batch, torch.from_numpy(probs[k * per:(k + 1) * per]))
        weights.append(w.numpy().copy()); source_weights.append(sw.numpy().copy())
        ref_w.append(torch.exp(down.log_ref_weights_slvrak).view(-1, 4)[batch.batch_indices().flattened_idx].detach().numpy())
        alt_w.append(torch.exp(down.log_alt_weights_slvrah).view(-1, 4)[batch.batch_indices().flattened_idx].detach().numpy())
    np.savez_compressed(
        os.path.join(HERE, "training_helpers.npz"),
        int_array=int_array, probs=probs, batch=np.int64(per), flattened_idx=np.concatenate(flat),
        weights=np.concatenate(weights), source_weights=np.concatenate(source_weights),
        final_weights_slvra=bal.weights_slvra.detach().numpy(), final_unlabeled_weights_slvra=bal.unlabeled_weights_slvra.detach().numpy(),
        final_source_weights_s=bal.source_weights_s.detach().numpy(), final_counts_slvra=bal.counts_slvra.detach().numpy(),
        binned_ref_trans_kry=down.binned_ref_trans_kry.detach().numpy(), binned_alt_trans_haz=down.binned_alt_trans_haz.detach().numpy(),
        log_ref_weights_original=down.parametrizations.log_ref_weights_slvrak.original.detach().numpy(),
        log_alt_weights_original=down.parametrizations.log_alt_weights_slvrah.original.detach().numpy(),
        ref_weights_bk=np.concatenate(ref_w), alt_weights_bk=np.concatenate(alt_w),
    )
    print("training helpers fixture written")


def make_downsampler_fit_fixture():
    """downsampler_fit.npz: the reference's `Downsampler.optimize_downsampling_balance` (training/downsampler.py:141-158:
    10 000 AdamW steps, deterministic) on a seeded table of dataset counts, and the expected downsampled counts before and
    after the fit."""
    import time
    from permutect.training.downsampler import Downsampler
    rng = np.random.default_rng(5)
    counts = torch.from_numpy(rng.integers(0, 2000, size=(2, 3, 5, 4, 5)).astype(np.float32))
    counts[:, :, 3, 0, :] = 0   # count bins without data are fine ...
    counts[1, 2, 4] = 0
    counts[1, 2, 4, 1, 2] = 7   # ... but a whole (source, label, variant type) cell without data makes the reference's loss
    # 0 / 0 = NaN and with it every weight (its training then dies in torch.multinomial): every cell here has some data
    torch.manual_seed(0)
    down = Downsampler(num_sources=2)
    before = down.calculate_expected_downsampled_counts(counts).detach().numpy()
    t0 = time.time()
    down.optimize_downsampling_balance(counts)
    print("reference fit took", time.time() - t0, "s")
    after = down.calculate_expected_downsampled_counts(counts).detach().numpy()
    np.savez_compressed(
        os.path.join(HERE, "downsampler_fit.npz"), counts_slvra=counts.numpy(), expected_before=before, expected_after=after,
        log_ref_weights_original=down.parametrizations.log_ref_weights_slvrak.original.detach().numpy(),
        log_alt_weights_original=down.parametrizations.log_alt_weights_slvrah.original.detach().numpy(),
        state_dict_keys=np.array(sorted(down.state_dict().keys())),
    )
    print("downsampler fit fixture written")


def make_posterior_rows_fixture():
    """posterior_rows.npz: what the reference's per-variant loop in `generate_posterior_data` (tools/filter_variants.py:
    302-320) makes of a batch -- counts zeroed, the logit stored through the float16 scalar array, the info columns replaced
    by the embedding -- for the variants of tiny_dataset.tar with seeded logits / embeddings standing in for the model's."""
    from permutect.data.datum import COMPRESSED_READS_ARRAY_DTYPE
    from permutect.data.memory_mapped_data import MemoryMappedData
    back = MemoryMappedData.load_from_tarfile(os.path.join(HERE, "tiny_dataset.tar"))
    datums = list(back.generate())
    batch = Batch(datums)
    rng = np.random.default_rng(9)
    n, e = len(datums), 10
    logits = torch.from_numpy((20 * np.tanh(rng.standard_normal(n) * 2)).astype(np.float32))
    logits[0], logits[1], logits[2] = 20.0, -20.0, 1e-4
    features = torch.from_numpy(rng.standard_normal((n, e)).astype(np.float32))
    ints, floats = [], []
    for int_array, float_array, logit, embedding in zip(batch.get_int_array_be(), batch.get_float_array_be(), logits.tolist(), features):
        d = Datum(int_array=int_array, float_array=float_array, reads_re=np.zeros((0, 0), dtype=COMPRESSED_READS_ARRAY_DTYPE), compressed=True)
        d.set(Data.REF_COUNT, 0)
        d.set(Data.ALT_COUNT, 0)
        d.set(Data.CACHED_ARTIFACT_LOGIT, logit)
        d.set_info_1d(embedding)
        ints.append(np.asarray(d.get_int_array()).copy())
        floats.append(np.asarray(d.get_float_array()).copy())
    print("posterior rows: int dtype", ints[0].dtype, "float dtype", floats[0].dtype, floats[0].shape)
    np.savez_compressed(os.path.join(HERE, "posterior_rows.npz"), batch_int=batch.int_tensor.numpy(), batch_float=batch.float_tensor.numpy(),
                        logits_b=logits.numpy(), features_be=features.numpy(), out_int=np.stack(ints), out_float=np.stack(floats),
                        out_float_dtype=str(floats[0].dtype))
    print("posterior rows fixture written")


def make_cnn_fixtures():
    """p0_cnn_legacy.npz / t0_cnn_options.npz: haplotype-CNN stacks beyond the two of the other fixtures -- the four-convolution
    stack of the shipped v0.4.0 checkpoint (kernel sizes 3, 3, 5, 5: im2col columns up to 160 wide) and the reference docstring's
    stack (dilation, selu; a 192-wide im2col column) extended by a strided and a padded convolution -- forward, losses, every
    gradient and the post-step parameters, like the other training fixtures.  Own random streams."""
    rng = np.random.default_rng(31)
    counts = [(int(rng.integers(0, 11)), int(rng.integers(1, 16))) for _ in range(20)]
    run_case("p0_cnn_legacy", make_model("P0", 31, cnn=P0_CNN_LEGACY), make_data(rng, counts))
    counts = [(int(rng.integers(0, 11)), int(rng.integers(1, 16))) for _ in range(20)]
    run_case("t0_cnn_options", make_model("T0", 32, cnn=T0_CNN_OPTIONS), make_data(rng, counts))


def make_metrics_fixtures():
    """loss_metrics.npz: the reference's LossRecorder.record (training/loss_recorder.py:14-22 -> metrics/loss_metrics.py:50-54) over
    three seeded batches with non-unit weights and two sources: the three (totals, counts) histograms it leaves.
    evaluation_metrics.npz: the reference's collect_evaluation_data (training/model_training.py:204-228) on tiny_dataset.tar with a
    seeded P0 model, a fresh reference Balancer and a downsampler stub whose fractions are all 1 (every read kept -- with the
    reference's un-offset alt gather): the AccuracyMetrics tensors [source, label, variant type, ref bin, alt bin, logit bin] of
    the TRAIN and VALID loaders, and the balancer's state afterwards."""
    from types import SimpleNamespace
    from permutect.data.memory_mapped_data import MemoryMappedData
    from permutect.training.balancer import Balancer
    from permutect.training.loss_recorder import LossRecorder
    from permutect.training.model_training import collect_evaluation_data
    from permutect.utils.enums import Epoch
    rng = np.random.default_rng(41)
    rec = LossRecorder(device=CPU, num_sources=2)
    nb, per = 3, 200
    int_array = np.zeros((nb * per, 16 + 42), dtype=np.int16)
    vec = {k: [] for k in ("weights", "source_weights", "supervised", "unsupervised", "alt_count", "source")}
    for k in range(nb):
        data = []
        for i in range(per):
            d = Datum(np.zeros(16 + 42, dtype=np.int16), np.zeros(6 + 71, dtype=np.float16), np.zeros((1, 12), dtype=np.uint8), compressed=True)
            d.set(Data.REF_COUNT, int(rng.integers(0, 40))); d.set(Data.ALT_COUNT, int(rng.integers(1, 30)))
            d.set(Data.LABEL, int(rng.integers(0, 3))); d.set(Data.VARIANT_TYPE, int(rng.integers(0, 5))); d.set(Data.SOURCE, int(rng.integers(0, 2)))
            int_array[k * per + i] = d.get_int_array()
            data.append(d)
        batch = Batch(data)
        v = {name: torch.from_numpy((scale * rng.random(per)).astype(np.float32))
             for name, scale in (("weights", 2.0), ("source_weights", 3.0), ("supervised", 5.0), ("unsupervised", 4.0), ("alt_count", 1.0), ("source", 2.0))}
        rec.record(SimpleNamespace(weights=v["weights"], source_weights=v["source_weights"]),
                   SimpleNamespace(supervised_losses_b=v["supervised"], unsupervised_losses_b=v["unsupervised"],
                                   alt_count_losses_b=v["alt_count"], source_prediction_losses_b=v["source"]), batch)
        for name in vec:
            vec[name].append(v[name].numpy())
    out = {"int_array": int_array, "batch": np.int64(per)}
    out.update({name: np.concatenate(vs) for name, vs in vec.items()})
    for name, m in (("primary", rec.primary_metrics), ("count", rec.count_metrics), ("source_m", rec.source_metrics)):
        out[name + "_totals"] = np.asarray(m.totals_slvra.detach().numpy())
        out[name + "_counts"] = np.asarray(m.counts_slvra.detach().numpy())
    out["primary_marginal_by_label"] = np.asarray(rec.primary_metrics.put_on_cpu().get_marginal(__import__("permutect.data.batch", fromlist=["BatchProperty"]).BatchProperty.LABEL).numpy())
    np.savez_compressed(os.path.join(HERE, "loss_metrics.npz"), **out)
    print("loss metrics fixture written; primary totals sum", float(out["primary_totals"].sum()))

    back = MemoryMappedData.load_from_tarfile(os.path.join(HERE, "tiny_dataset.tar"))
    datums = list(back.generate())
    model = make_model("P0", 51)
    model.eval()
    splits = {"train": [list(range(0, 14)), list(range(14, 30))], "valid": [list(range(30, 41))]}

    class AllReadsKept:
        def calculate_downsampling_fractions(self, batch):
            return torch.ones(batch.size()), torch.ones(batch.size())

    balancer = Balancer(num_sources=2, device=CPU)
    random.seed(3)
    em, _ = collect_evaluation_data(model, 2, balancer, AllReadsKept(), [Batch([datums[i] for i in ids]) for ids in splits["train"]],
                                    [Batch([datums[i] for i in ids]) for ids in splits["valid"]], report_worst=False)
    ev = {"sd/" + k: v.detach().numpy().copy() for k, v in model.state_dict().items()}
    ev["train_batches"] = np.array([len(x) for x in splits["train"]]); ev["valid_batches"] = np.array([len(x) for x in splits["valid"]])
    ev["train_ids"] = np.concatenate(splits["train"]); ev["valid_ids"] = np.concatenate(splits["valid"])
    ev["accuracy_train"] = em.accuracy_metrics_by_epoch_type[Epoch.TRAIN].detach().numpy()
    ev["accuracy_valid"] = em.accuracy_metrics_by_epoch_type[Epoch.VALID].detach().numpy()
    ev["balancer_counts_slvra"] = balancer.counts_slvra.detach().numpy()
    ev["balancer_weights_slvra"] = balancer.weights_slvra.detach().numpy()
    # the logits behind the tallies (first pass over every batch, balancer untouched), so that a test can tell a bin flip from an error
    with torch.inference_mode():
        ev["first_pass_logits"] = np.concatenate([
            model.compute_batch_output(DownsampledBatch(Batch([datums[i] for i in ids]), torch.ones(len(ids)), torch.ones(len(ids))), None).logits_b.numpy()
            for ids in splits["train"] + splits["valid"]])
    np.savez_compressed(os.path.join(HERE, "evaluation_metrics.npz"), **ev)
    print("evaluation metrics fixture written; train total", float(ev["accuracy_train"].sum()), "valid total", float(ev["accuracy_valid"].sum()),
          "shape", ev["accuracy_train"].shape)


def make_dropout_eval_fixture():
    """p0_dropout_eval.npz: a P0-shaped model built with dropout_p = 0.25 (nn.Dropout modules inside every MLP: the
    state_dict keys shift, reference architecture/mlp.py:57-58) in EVAL mode, as filter_variants runs it: state_dict,
    inputs and the forward outputs.  Leaves the other fixtures and their random streams untouched."""
    torch.manual_seed(11)
    p = ModelParameters([30, -2, -2, -2], 20, 6, [20, -2, -2, -2], [-2, -2, 10], 4, [10, 10], list(P0_CNN), 0.25, 0.3, False)
    m = ArtifactModel(p, 61, 71, 42, device=CPU)
    with torch.no_grad():
        for q in m.parameters():
            q.add_(0.05 * torch.randn_like(q))
    m.eval()
    rng = np.random.default_rng(11)
    counts = [(int(rng.integers(0, 12)), int(rng.integers(1, 9))) for _ in range(24)]
    data = make_data(rng, counts)
    batch = Batch(data)
    packed = np.vstack([d.get_ref_reads_re() for d in data] + [d.get_alt_reads_re() for d in data])
    out = {"packed_reads": packed, "int_array": batch.int_tensor.numpy().astype(np.int16),
           "float_array": batch.float_tensor.numpy().astype(np.float16), "dropout_p": np.float64(0.25)}
    for k, v in m.state_dict().items():
        out["sd/" + k] = v.detach().numpy().copy()
    with torch.inference_mode():
        output = m.compute_batch_output(batch, None)
    for k in ("features_be", "ref_features_be", "logits_b", "logits_bk", "artifact_probs_b", "outlier_binary_logits"):
        out["out/" + k] = getattr(output, k).detach().numpy()
    np.savez_compressed(os.path.join(HERE, "p0_dropout_eval.npz"), **out)
    print("dropout eval fixture written;", len(m.state_dict()), "state_dict entries; logits", out["out/logits_b"][:4])


def make_batchnorm_eval_fixture():
    """p0_batchnorm_eval.npz: a P0-shaped model built with batch_normalize = True (an nn.BatchNorm1d in front of every Linear of the
    MLPs, reference architecture/mlp.py:52-53) in EVAL mode, as filter_variants runs a checkpoint trained that way: running
    statistics away from their initial (0, 1), state_dict, inputs and the forward outputs."""
    torch.manual_seed(13)
    p = ModelParameters([30, -2, -2, -2], 20, 6, [20, -2, -2, -2], [-2, -2, 10], 4, [10, 10], list(P0_CNN), 0.0, 0.3, True)
    m = ArtifactModel(p, 61, 71, 42, device=CPU)
    with torch.no_grad():
        for q in m.parameters():
            q.add_(0.05 * torch.randn_like(q))
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.running_mean.normal_(0.0, 0.3)
                mod.running_var.uniform_(0.5, 1.5)
    m.eval()
    rng = np.random.default_rng(13)
    counts = [(int(rng.integers(0, 12)), int(rng.integers(1, 9))) for _ in range(24)]
    data = make_data(rng, counts)
    batch = Batch(data)
    packed = np.vstack([d.get_ref_reads_re() for d in data] + [d.get_alt_reads_re() for d in data])
    out = {"packed_reads": packed, "int_array": batch.int_tensor.numpy().astype(np.int16),
           "float_array": batch.float_tensor.numpy().astype(np.float16)}
    for k, v in m.state_dict().items():
        out["sd/" + k] = v.detach().numpy().copy()
    with torch.inference_mode():
        output = m.compute_batch_output(batch, None)
    for k in ("features_be", "ref_features_be", "logits_b", "logits_bk", "artifact_probs_b", "outlier_binary_logits"):
        out["out/" + k] = getattr(output, k).detach().numpy()
    np.savez_compressed(os.path.join(HERE, "p0_batchnorm_eval.npz"), **out)
    n_bn = sum(isinstance(mod, torch.nn.BatchNorm1d) for mod in m.modules())
    print("batchnorm eval fixture written;", len(m.state_dict()), "state_dict entries,", n_bn, "BatchNorm1d modules; logits", out["out/logits_b"][:4])


def make_cnn_batchnorm_eval_fixture():
    """p0_cnn_batchnorm_eval.npz: the production stack with the reference's `batch_norm` token (dna_sequence_convolution.py:82-83)
    behind the first convolution, in front of the second one and between flatten and the linear, in EVAL mode with running
    statistics away from (0, 1): state_dict, inputs, the forward outputs and the haplotype embedding itself."""
    torch.manual_seed(17)
    p = ModelParameters([30, -2, -2, -2], 20, 6, [20, -2, -2, -2], [-2, -2, 10], 4, [10, 10], list(P0_CNN_BATCHNORM), 0.0, 0.3, False)
    m = ArtifactModel(p, 61, 71, 42, device=CPU)
    with torch.no_grad():
        for q in m.parameters():
            q.add_(0.05 * torch.randn_like(q))
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.running_mean.normal_(0.0, 0.3)
                mod.running_var.uniform_(0.5, 1.5)
    m.eval()
    rng = np.random.default_rng(17)
    counts = [(int(rng.integers(0, 12)), int(rng.integers(1, 9))) for _ in range(24)]
    data = make_data(rng, counts)
    batch = Batch(data)
    packed = np.vstack([d.get_ref_reads_re() for d in data] + [d.get_alt_reads_re() for d in data])
    out = {"packed_reads": packed, "int_array": batch.int_tensor.numpy().astype(np.int16),
           "float_array": batch.float_tensor.numpy().astype(np.float16)}
    for k, v in m.state_dict().items():
        out["sd/" + k] = v.detach().numpy().copy()
    with torch.inference_mode():
        output = m.compute_batch_output(batch, None)
        out["out/ref_seq_embeddings_be"] = m.haplotypes_cnn(batch.get_one_hot_haplotypes_bcs().float()).numpy()
    for k in ("features_be", "ref_features_be", "logits_b", "logits_bk", "artifact_probs_b", "outlier_binary_logits"):
        out["out/" + k] = getattr(output, k).detach().numpy()
    np.savez_compressed(os.path.join(HERE, "p0_cnn_batchnorm_eval.npz"), **out)
    n_bn = sum(isinstance(mod, torch.nn.BatchNorm1d) for mod in m.haplotypes_cnn.modules())
    print("cnn batchnorm eval fixture written;", n_bn, "BatchNorm1d modules in the CNN; logits", out["out/logits_b"][:4])


def make_wide_fixture():
    """wide_d98: a model with layers wider than 64 (its own random streams: the other fixtures regenerate bit-exact)"""
    rng = np.random.default_rng(98)
    random.seed(98)
    counts = [(int(rng.integers(0, 11)), int(rng.integers(1, 16))) for _ in range(10)] + [(40, 37), (0, 21)]
    run_case("wide_d98", make_model("WIDE", 98), make_data(rng, counts))
    if "--wide-only" in sys.argv and "--no-wide64" in sys.argv:
        return
    counts = [(int(rng.integers(0, 11)), int(rng.integers(1, 16))) for _ in range(10)] + [(33, 41), (0, 18)]
    run_case("wide64_d98", make_model("WIDE64", 99), make_data(rng, counts))


def make_skip_depth_fixture():
    """p0_skip34: skip blocks of three and four layers in the read MLP, the info MLP and the reducer (its own random streams)"""
    rng = np.random.default_rng(34)
    random.seed(34)
    counts = [(int(rng.integers(0, 11)), int(rng.integers(1, 16))) for _ in range(12)] + [(25, 30), (0, 9)]
    run_case("p0_skip34", make_model("SKIP34", 34), make_data(rng, counts))


if __name__ == "__main__":
    if "--skip-depth-only" in sys.argv:
        make_skip_depth_fixture()
    elif "--wide-only" in sys.argv:
        make_wide_fixture()
    elif "--cnn-batchnorm-only" in sys.argv:
        make_cnn_batchnorm_eval_fixture()
    elif "--dropout-only" in sys.argv:
        make_dropout_eval_fixture()
    elif "--batchnorm-only" in sys.argv:
        make_batchnorm_eval_fixture()
    elif "--cnn-only" in sys.argv:
        make_cnn_fixtures()
    elif "--metrics-only" in sys.argv:
        make_metrics_fixtures()
    elif "--downsampler-only" in sys.argv:
        make_downsampler_fit_fixture()
    elif "--posterior-only" in sys.argv:
        make_posterior_rows_fixture()
    elif "--dataset-only" in sys.argv:  # leaves the model fixtures (and their random streams) untouched
        make_dataset_fixture()
    elif "--helpers-only" in sys.argv:
        make_training_helpers_fixture()
    else:
        main()
        make_dataset_fixture()
        make_training_helpers_fixture()
        make_downsampler_fit_fixture()
        make_posterior_rows_fixture()
        make_dropout_eval_fixture()
        make_batchnorm_eval_fixture()
        make_cnn_fixtures()
        make_metrics_fixtures()
        make_cnn_batchnorm_eval_fixture()
        make_wide_fixture()
