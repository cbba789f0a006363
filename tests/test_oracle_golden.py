"""CPU: pin the oracle (oracle/artifact_oracle.py) against vectors produced by the reference itself."""
import numpy as np
import pytest
import torch

from oracle import artifact_oracle as O
from tests.helpers import CASES, CNN_CASES, config_for, load_case, GOLDEN

FWD_TOL = dict(rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("name", CASES + CNN_CASES)
def test_forward_and_losses_match_reference(name):
    z, sd, b = load_case(name)
    cfg = config_for(name)
    np.testing.assert_array_equal(O.decode_packed_reads(b["packed_reads"]), z["reads_re_f16"])
    with torch.no_grad():
        out = O.compute_batch_output(sd, cfg, b["reads_re"], b["nref"], b["nalt"], b["info_be"], b["haplotypes_bh"])
        kw = {}
        if "source_strength" in z.files:
            kw["source_adv_strength"] = float(z["source_strength"])
        losses = O.compute_batch_losses(sd, cfg, out, b["labels"], b["nalt"], b["sources"], **kw)
    for k in ("features_be", "ref_features_be", "logits_b", "logits_bk", "artifact_probs_b",
              "outlier_binary_logits", "final_ref_re", "final_alt_re", "ref_seq_embeddings_be"):
        ref = z["out/" + k]
        scale = max(1.0, float(np.abs(ref).max())) if ref.size else 1.0
        np.testing.assert_allclose(out[k].numpy(), ref, rtol=2e-5, atol=2e-5 * scale, err_msg=k)
    for k in ("supervised_losses_b", "unsupervised_losses_b", "alt_count_losses_b", "source_prediction_losses_b",
              "total_losses_b", "total_loss"):
        np.testing.assert_allclose(losses[k].numpy(), z["loss/" + k], rtol=1e-4, atol=1e-4, err_msg=k)
    # the north-star contract: per-variant artifact logits within 1e-4 (fp32)
    assert np.abs(out["logits_b"].numpy() - z["out/logits_b"]).max() < 1e-4


@pytest.mark.parametrize("name", CASES + CNN_CASES)
def test_gradients_and_adamw_step_match_reference(name):
    z, sd, b = load_case(name)
    cfg = config_for(name)
    kw = {}
    if "source_strength" in z.files:
        kw["source_adv_strength"] = float(z["source_strength"])
    _, _, grads = O.train_step_grads(sd, cfg, b, **kw)
    names = [k[5:] for k in z.files if k.startswith("grad/")]
    assert set(names) == set(grads.keys())
    gref = np.concatenate([z["grad/" + n].ravel() for n in names])
    gour = np.concatenate([grads[n].numpy().ravel() for n in names])
    assert np.linalg.norm(gour - gref) <= 1e-4 * np.linalg.norm(gref)
    for n in names:  # per-tensor check, relative to the tensor's own scale (floor: the global grad scale)
        ref = z["grad/" + n]
        tol = 2e-4 * max(np.abs(ref).max(), 1e-3 * np.abs(gref).max())
        assert np.abs(grads[n].numpy() - ref).max() <= tol, n
    params = [sd[n].clone() for n in names]
    m = [torch.zeros_like(p) for p in params]
    v = [torch.zeros_like(p) for p in params]
    O.clip_and_adamw(params, [grads[n] for n in names], m, v, step=1, lr=float(z["lr"]),
                     weight_decay=float(z["weight_decay"]))
    for n, p in zip(names, params):
        np.testing.assert_allclose(p.numpy(), z["after/" + n], rtol=1e-5, atol=2e-6, err_msg=n)


def test_eval_forward_with_cnn_batch_norm_tokens_matches_reference():
    """the reference's `batch_norm` token of the haplotype CNN in eval mode (tests/golden/p0_cnn_batchnorm_eval.npz: three of them,
    running statistics away from (0, 1)): the oracle's forward and the haplotype embedding itself"""
    z, sd, b = load_case("p0_cnn_batchnorm_eval")
    cfg = config_for("p0_cnn_batchnorm_eval")
    with torch.no_grad():
        out = O.compute_batch_output(sd, cfg, b["reads_re"], b["nref"], b["nalt"], b["info_be"], b["haplotypes_bh"])
    for k in ("features_be", "ref_features_be", "logits_b", "logits_bk", "outlier_binary_logits", "ref_seq_embeddings_be"):
        ref = z["out/" + k]
        np.testing.assert_allclose(out[k].numpy(), ref, rtol=2e-5, atol=2e-5 * max(1.0, float(np.abs(ref).max())), err_msg=k)
    assert np.abs(out["logits_b"].numpy() - z["out/logits_b"]).max() < 1e-4


def test_decode_wrap_quirk():
    z = np.load(f"{GOLDEN}/quirk_decode.npz")
    dec = O.decode_packed_reads(z["packed_reads"])
    np.testing.assert_array_equal(dec, z["reads_re_f16"])
    assert dec.shape == (256, 61)
    assert dec[96, 56] == 7.0  # byte 96 in a float column decodes to +7.0, not -1.0


def test_downsampled_batch_gather_quirk():
    z = np.load(f"{GOLDEN}/quirk_downsample.npz")
    total_ref, total_alt = int(z["ref_counts"].sum()), int(z["alt_counts"].sum())
    idx = O.downsampled_read_indices(torch.ones(total_ref), torch.ones(total_alt)).numpy()
    np.testing.assert_array_equal(idx, z["read_indices_all_kept"])
    np.testing.assert_array_equal(idx, np.concatenate([np.arange(total_ref), np.arange(total_alt)]))
    np.testing.assert_array_equal(z["parent_reads"][z["read_indices_half"]], z["gathered_reads_half"])
    assert z["read_indices_half"][int(z["new_ref_counts_half"].sum()):].max() < total_alt  # un-offset alt rows
