"""GPU parity of the haplotype CNN beyond the two stacks of the other fixtures (reference
architecture/dna_sequence_convolution.py:49-111): the four-convolution stack of the shipped v0.4.0 checkpoint (kernel sizes
3, 3, 5, 5) and the reference docstring's stack (dilation = 2, selu) extended by a strided and a padded convolution
(tests/golden/p0_cnn_legacy.npz, t0_cnn_options.npz: written by the reference).  Every stack runs through every CNN kernel
family -- PMT_CNN = general (pmt_cnn.hip) | wave (pmt_cnn2.hip) | batched (pmt_cnn3.hip) -- where the family accepts it and
must fail loudly (PMT_E_UNSUPPORTED) where it does not; forward AND every gradient.  `calculate_features` is called directly:
the haplotype embedding (`ref_seq_embeddings_be`) and the info embedding are compared on their own, not only through the logits."""
import numpy as np
import pytest
import torch

from oracle import artifact_oracle as O
from permutect_amd.architecture.artifact_model import ArtifactModel
from permutect_amd.data.batch import Batch
from permutect_amd.engine.lib import PmtError
from permutect_amd.parameters import P0_DIMS, p0_params, t0_params
from permutect_amd.training.optimizer import FusedClipAdamW
from tests.helpers import CASES, CNN_CASES, CNN_STACKS, config_for, load_case, params_for
from tests.test_forward_gpu import check_outputs

pytestmark = pytest.mark.gpu

# which family runs which stack.  wave: at most two convolutions of at most 32 output channels (or ONE of at most 64 with <= 32 im2col columns), im2col columns <= 96 wide;
# batched: exactly conv k3 -> pool 2 -> act -> conv k3 -> act -> flatten -> linear on 21 positions (P0_CNN); general: everything.
ACCEPTS = {
    "p0_cnn_legacy": {"auto", "general"},
    "t0_cnn_options": {"auto", "general"},
    "p0_b16": {"auto", "general", "wave", "batched", "batched_fp32"},
    "t0_b8": {"auto", "general", "wave"},  # (one convolution 10 -> 64: the wave family's four-out-tile instance since round 5)
}


def build_cnn(name, sd, family, monkeypatch):
    if family == "batched_fp32":  # the batched family's forward on the fp32 matrix pipe (default: bf16 pieces, pmt_cnn3.hip)
        monkeypatch.setenv("PMT_CNN", "batched")
        monkeypatch.setenv("PMT_CNN_DBG", "256")
    elif family != "auto":
        monkeypatch.setenv("PMT_CNN", family)
    params = params_for(name)
    if name in CNN_STACKS:
        params.ref_seq_layer_strings = list(CNN_STACKS[name])
    model = ArtifactModel(params, device=torch.device("cuda"), **P0_DIMS)
    model.load_state_dict(sd)
    return model


@pytest.mark.parametrize("family", ["auto", "general", "wave", "batched", "batched_fp32"])
@pytest.mark.parametrize("name", CNN_CASES + ["p0_b16", "t0_b8"])
def test_every_family_matches_the_reference_or_refuses(name, family, monkeypatch):
    z, sd, b = load_case(name)
    model = build_cnn(name, sd, family, monkeypatch)
    dev = torch.device("cuda")
    batch = Batch.from_arrays(b["int_array"], b["float_array"], b["packed_reads"]).copy_to(dev)
    model.train(True)
    if family not in ACCEPTS[name]:
        with pytest.raises(PmtError, match="not supported"):
            model.compute_batch_output(batch)
        return
    # ---- calculate_features, as the reference's callers use it (artifact_model.py:239-265) -------------------------------
    with torch.no_grad():
        ref_sets, alt_sets, hap = model.calculate_features(batch)
        ve = model.variant_embedding(batch)
    ref_hap = z["out/ref_seq_embeddings_be"]
    np.testing.assert_allclose(hap.cpu().numpy(), ref_hap, rtol=2e-5, atol=2e-5 * max(1.0, float(np.abs(ref_hap).max())), err_msg="ref_seq_embeddings_be")
    cfg = config_for(name)
    with torch.no_grad():
        info_ref = O.mlp(sd, "info_embedding", [cfg.num_info_features] + cfg.info_layers, b["info_be"].to(torch.float32)).numpy()
    e_info = info_ref.shape[1]
    np.testing.assert_allclose(ve[:, :e_info].cpu().numpy(), info_ref, rtol=2e-5, atol=2e-5 * max(1.0, float(np.abs(info_ref).max())), err_msg="info embedding")
    np.testing.assert_allclose(ve[:, e_info:].cpu().numpy(), ref_hap, rtol=2e-5, atol=2e-5 * max(1.0, float(np.abs(ref_hap).max())))
    for view, key in ((alt_sets, "features_be"), (ref_sets, "ref_features_be")):
        ref = z["out/" + key]
        np.testing.assert_allclose(view.means_over_sets().cpu().numpy(), ref, rtol=2e-5, atol=2e-5 * max(1.0, float(np.abs(ref).max())), err_msg=key)
    # ---- outputs, losses, every gradient -----------------------------------------------------------------------------------
    out = model.compute_batch_output(batch)
    check_outputs(out, z, name)
    losses = model.compute_batch_losses(out, batch)
    ref_total = z["loss/total_losses_b"]
    np.testing.assert_allclose(losses.total_losses_b.detach().cpu().numpy(), ref_total, rtol=1e-4, atol=1e-4 + 1e-5 * np.abs(ref_total).max())
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
        if err > 5e-4 * max(np.abs(ref).max(), 1e-3 * gscale):
            bad.append((n, float(err), float(np.abs(ref).max())))
    assert not bad, bad[:12]
    assert np.linalg.norm(gour - gref) <= 1e-4 * np.linalg.norm(gref)
    # the haplotype CNN's own gradients, on their own scale (they are a small part of the global vector)
    cnn_ref = np.concatenate([z["grad/" + n].ravel() for n in names if n.startswith("haplotypes_cnn")])
    cnn_our = np.concatenate([p.grad.detach().cpu().numpy().ravel() for n, p in model.named_parameters() if n.startswith("haplotypes_cnn")])
    assert np.linalg.norm(cnn_our - cnn_ref) <= 2e-4 * np.linalg.norm(cnn_ref)


@pytest.mark.parametrize("name", CASES)
def test_embeddings_of_every_fixture_match_the_reference_directly(name):
    """`ref_seq_embeddings_be` of the six main fixtures (all carry it) and the info embedding, through calculate_features /
    variant_embedding: the per-variant branch compared on its own."""
    z, sd, b = load_case(name)
    from tests.test_forward_gpu import build
    model, dev = build(name, sd)
    batch = Batch.from_arrays(b["int_array"], b["float_array"], b["packed_reads"]).copy_to(dev)
    with torch.no_grad():
        _, _, hap = model.calculate_features(batch)
        ve = model.variant_embedding(batch)
    ref_hap = z["out/ref_seq_embeddings_be"]
    np.testing.assert_allclose(hap.cpu().numpy(), ref_hap, rtol=2e-5, atol=2e-5 * max(1.0, float(np.abs(ref_hap).max())))
    cfg = config_for(name)
    with torch.no_grad():
        info_ref = O.mlp(sd, "info_embedding", [cfg.num_info_features] + cfg.info_layers, b["info_be"].to(torch.float32)).numpy()
    np.testing.assert_allclose(ve[:, :info_ref.shape[1]].cpu().numpy(), info_ref, rtol=2e-5, atol=2e-5 * max(1.0, float(np.abs(info_ref).max())))


@pytest.mark.parametrize("family", ["auto", "general", "wave", "batched"])
def test_eval_forward_with_cnn_batch_norm_tokens_matches_reference(family, monkeypatch):
    """The reference's `batch_norm` token (dna_sequence_convolution.py:82-83) in EVAL mode, as filter_variants runs a checkpoint
    trained with it (tests/golden/p0_cnn_batchnorm_eval.npz: behind the first convolution, in front of the second, between flatten
    and the linear; running statistics away from (0, 1)).  The engine folds the three BatchNorms into the neighbouring layers'
    weights (engine/plan.py: folded_cnn_theta), so every kernel family runs the stack unchanged; state_dict keys as the
    reference's; new running statistics are new weights; training is refused loudly."""
    from permutect_amd.architecture.artifact_model import ArtifactModel
    from permutect_amd.parameters import P0_CNN_BATCHNORM, P0_DIMS, p0_params
    from tests.test_forward_gpu import check_outputs
    if family != "auto":
        monkeypatch.setenv("PMT_CNN", family)
    z, sd, b = load_case("p0_cnn_batchnorm_eval")
    params = p0_params()
    params.ref_seq_layer_strings = list(P0_CNN_BATCHNORM)
    dev = torch.device("cuda")
    model = ArtifactModel(params, device=dev, **P0_DIMS)
    assert set(model.state_dict().keys()) == set(sd.keys())
    model.load_state_dict(sd)
    model.eval()
    batch = Batch.from_arrays(b["int_array"], b["float_array"], b["packed_reads"]).copy_to(dev)
    with torch.inference_mode():
        out = model.compute_batch_output(batch)
        _, _, hap = model.calculate_features(batch)
    ref_hap = z["out/ref_seq_embeddings_be"]
    np.testing.assert_allclose(hap.cpu().numpy(), ref_hap, rtol=2e-5, atol=2e-5 * max(1.0, float(np.abs(ref_hap).max())))
    check_outputs(out, z, "p0_cnn_batchnorm_eval")
    sd2 = {k: (v * 1.5 if (k.startswith("haplotypes_cnn") and k.endswith("running_var")) else v) for k, v in sd.items()}
    model.load_state_dict(sd2)
    with torch.inference_mode():
        moved = model.compute_batch_output(batch)
    assert float((moved.logits_b - out.logits_b).abs().max()) > 1e-4
    model.train(True)
    with pytest.raises(NotImplementedError, match="batch_norm"):
        model.compute_batch_output(batch)
