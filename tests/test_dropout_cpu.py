"""Dropout (reference architecture/mlp.py:57-58, artifact_model.py:145-206), the parts that need no GPU:

* the oracle's treatment of a model built with dropout_p > 0 is pinned twice -- in eval mode against the reference's own outputs
  (tests/golden/p0_dropout_eval.npz: a reference model with dropout_p = 0.25), and in train mode against torch's nn.Dropout
  inside the very Sequential the reference builds (the masks torch drew are read back with forward hooks and handed to the
  oracle, which must then reproduce the module's output and gradients);
* the mask function the kernels use (pmt_dropout.hpp, exported as pmt_dropout_mask -- a host computation): keep rate,
  scale, independence between linears, rows and seeds."""
import ctypes as C

import numpy as np
import torch
from torch import nn

from oracle import artifact_oracle as O
from permutect_amd.architecture import modules as M
from permutect_amd.engine import lib as L
from tests.helpers import config_for, load_case


def host_mask(seed, p, lin, row0, rows, width):
    out = np.empty((rows, width), dtype=np.float32)
    L.check(L.load().pmt_dropout_mask(seed, p, lin, row0, rows, width, out.ctypes.data), "pmt_dropout_mask")
    return out


def test_oracle_eval_mode_of_a_dropout_model_matches_the_reference_outputs():
    z, sd, b = load_case("p0_dropout_eval")
    cfg = config_for("p0_dropout_eval")
    cfg.dropout = lambda key, y, row0=0: None  # eval mode: the Dropout modules only shift the Sequential indices
    with torch.no_grad():
        out = O.compute_batch_output(sd, cfg, b["reads_re"], b["nref"], b["nalt"], b["info_be"], b["haplotypes_bh"])
    for k in ("logits_b", "features_be", "ref_features_be"):
        ref = z["out/" + k]
        np.testing.assert_allclose(out[k].numpy(), ref, rtol=1e-5, atol=1e-5 * max(1.0, float(np.abs(ref).max())), err_msg=k)


def test_oracle_train_mode_reproduces_torch_dropout_given_its_masks():
    """the reference's MLP layout (Linear, Dropout, SELU; skip blocks with their own inner Sequential) run by torch in train mode;
    the oracle, given the masks torch drew, must give the same output and the same parameter gradients"""
    torch.manual_seed(11)
    sizes = [12, 16, -2, -1, 16, 5]
    mlp = M.MLP(sizes, dropout_p=0.3)
    mlp.train(True)
    masks = {}
    hooks = []
    for name, mod in mlp.named_modules():
        if isinstance(mod, nn.Dropout):
            # "<seq prefix>.<i>" is the Dropout, "<seq prefix>.<i-1>" the Linear in front of it
            head, idx = name.rsplit(".", 1)
            key = f"mlp.{head}.{int(idx) - 1}"
            hooks.append(mod.register_forward_hook(
                lambda m, inp, out, key=key: masks.__setitem__(key, torch.where(out != 0, 1 / 0.7, 0.0).to(out.dtype).detach())))
    x = torch.randn(40, 12)
    y = mlp(x)
    y.square().sum().backward()
    for h in hooks:
        h.remove()
    assert len(masks) == 6 and all(0.6 < float((m != 0).float().mean()) < 0.8 for m in masks.values())
    sd = {"mlp." + k: v.detach().clone().requires_grad_(True) for k, v in mlp.state_dict().items()}
    y2 = O.mlp(sd, "mlp", sizes, x, dropout=lambda key, t, row0=0: masks[key])
    np.testing.assert_allclose(y2.detach().numpy(), y.detach().numpy(), rtol=1e-6, atol=1e-6)
    y2.square().sum().backward()
    for k, p in mlp.named_parameters():
        np.testing.assert_allclose(sd["mlp." + k].grad.numpy(), p.grad.numpy(), rtol=1e-5, atol=1e-6, err_msg=k)
    # and eval mode is the mask-free network with the same (shifted) keys
    mlp.eval()
    with torch.no_grad():
        np.testing.assert_allclose(O.mlp(sd, "mlp", sizes, x, dropout=lambda key, t, row0=0: None).numpy(), mlp(x).numpy(), rtol=1e-6, atol=1e-6)


def test_mask_function_statistics():
    p, rows, width = 0.25, 20000, 60
    m = host_mask(0x1234_5678_9ABC_DEF1, p, 7, 0, rows, width)
    assert set(np.unique(m)) == {0.0, np.float32(1 / (1 - p))}
    keep = m != 0
    n = keep.size
    sigma = np.sqrt(p * (1 - p) / n)
    assert abs(keep.mean() - (1 - p)) < 5 * sigma
    # every feature column and every block of rows on its own
    assert np.abs(keep.mean(axis=0) - (1 - p)).max() < 5 * np.sqrt(p * (1 - p) / rows)
    assert np.abs(keep.reshape(200, -1).mean(axis=1) - (1 - p)).max() < 5 * np.sqrt(p * (1 - p) / (n / 200))
    assert abs(m.mean() - 1.0) < 5 * sigma / (1 - p)  # unbiased: E[mask] = 1
    # another linear, another seed, shifted rows: independent draws (correlation ~ N(0, 1/n))
    for other in (host_mask(0x1234_5678_9ABC_DEF1, p, 8, 0, rows, width), host_mask(0x1234_5678_9ABC_DEF2, p, 7, 0, rows, width),
                  host_mask(0x1234_5678_9ABC_DEF1, p, 7, 1, rows, width)):
        c = np.corrcoef(keep.ravel().astype(np.float64), (other != 0).ravel().astype(np.float64))[0, 1]
        assert abs(c) < 5 / np.sqrt(n)
    # neighbouring features / rows of one mask are uncorrelated too
    assert abs(np.corrcoef(keep[:, :-1].ravel(), keep[:, 1:].ravel())[0, 1]) < 5 / np.sqrt(n)
    assert abs(np.corrcoef(keep[:-1].ravel(), keep[1:].ravel())[0, 1]) < 5 / np.sqrt(n)
    # a window of the mask is the mask of those rows (row0 is the batch row of the first one)
    np.testing.assert_array_equal(host_mask(0x1234_5678_9ABC_DEF1, p, 7, 100, 50, width), m[100:150])
    # the extremes, and the seed that means "no dropout"
    assert np.all(host_mask(5, 0.0, 0, 0, 64, 16) == 1.0)
    assert np.all(host_mask(0, 0.5, 0, 0, 64, 16) == 1.0)
    for q in (0.05, 0.5, 0.9):
        k = host_mask(99, q, 3, 0, 4000, 64) != 0
        assert abs(k.mean() - (1 - q)) < 5 * np.sqrt(q * (1 - q) / k.size)
    assert L.load().pmt_dropout_mask(1, C.c_float(1.0), 0, 0, 1, 1, np.empty(1, np.float32).ctypes.data) != 0  # p = 1: invalid
