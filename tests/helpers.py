"""Shared test helpers: load golden fixtures, build oracle inputs."""
import os

import numpy as np
import torch

from oracle import artifact_oracle as O
from permutect_amd.parameters import P0_CNN, P0_CNN_BATCHNORM, P0_CNN_LEGACY, T0_CNN, T0_CNN_OPTIONS

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["t0_b8", "p0_b16", "p0_zero_ref", "p0_saturated", "p0_deep", "t0_two_sources", "wide_d98", "wide64_d98", "p0_skip34"]
# p0_skip34: skip blocks of three and four layers (generic instances); wide_d98: layers beyond 64 (the wide builds of the library); wide64_d98: the same with d_ffn = 64 (two tiles per half of the gated blocks' hidden layer)
CNN_CASES = ["p0_cnn_legacy", "t0_cnn_options"]  # haplotype-CNN stacks beyond the two of CASES (tests/golden/make_golden.py: make_cnn_fixtures)
CNN_STACKS = {"p0_cnn_legacy": P0_CNN_LEGACY, "t0_cnn_options": T0_CNN_OPTIONS, "p0_cnn_batchnorm_eval": P0_CNN_BATCHNORM}


def params_for(name: str):
    """the model hyperparameters of a fixture (tests/golden/make_golden.py: make_model)"""
    from permutect_amd.parameters import ModelParameters, P0_CNN as _CNN, p0_params, t0_params, wide64_params, wide_params
    if name == "p0_skip34":
        return ModelParameters([30, -3, -2], 20, 2, [20, -3], [-4, 10], 4, [10, 10], list(_CNN), 0.0, 0.3)
    return t0_params() if name.startswith("t0") else wide64_params() if name.startswith("wide64") else wide_params() if name.startswith("wide") else p0_params()


def config_for(name: str) -> O.Config:
    if name == "p0_skip34":
        cfg = O.Config([30, -3, -2], [20, -3], [-4, 10], 20, 2, 4, list(P0_CNN), 61, 71, 42)
    elif name.startswith("t0"):
        cfg = O.Config([10, 10, 10], [10, 10], [20, 20, 20], 20, 2, 4, list(T0_CNN), 61, 71, 42)
    elif name.startswith("wide"):
        cfg = O.Config([48, -2], [40, -1], [-1, 20], 64 if name.startswith("wide64") else 32, 2, 4, list(P0_CNN), 61, 71, 42)
    else:
        cfg = O.Config([30, -2, -2, -2], [20, -2, -2, -2], [-2, -2, 10], 20, 6, 4, list(P0_CNN), 61, 71, 42)
    if name == "t0_two_sources":
        cfg.num_sources = 2
    if name in CNN_STACKS:
        cfg.cnn_layers = list(CNN_STACKS[name])
    return cfg


def load_case(name: str):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")}
    ints = torch.from_numpy(z["int_array"].astype(np.int64))
    floats = torch.from_numpy(z["float_array"].astype(np.float32))
    batch = dict(
        packed_reads=z["packed_reads"],
        reads_re=torch.from_numpy(O.decode_packed_reads(z["packed_reads"]).astype(np.float32)),
        nref=ints[:, O.REF_COUNT], nalt=ints[:, O.ALT_COUNT], labels=ints[:, O.LABEL], sources=ints[:, O.SOURCE],
        info_be=floats[:, O.INFO_START:], haplotypes_bh=ints[:, O.HAPLOTYPES_START:],
        int_array=z["int_array"], float_array=z["float_array"],
    )
    return z, sd, batch


def oracle_forward(sd, cfg, ints, floats, packed, dtype=torch.float32):
    """The oracle's forward on batch arrays (int16 / float16 / packed uint8 rows, ref rows first) in `dtype`.  fp64 is the
    yardstick that says how far fp32 arithmetic itself is from the exact result -- never a parity target."""
    i64 = torch.from_numpy(np.asarray(ints).astype(np.int64))
    old = O.COMPUTE_DTYPE
    O.COMPUTE_DTYPE = dtype
    try:
        with torch.inference_mode():
            sdd = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}
            return O.compute_batch_output(sdd, cfg, torch.from_numpy(O.decode_packed_reads(packed).astype(np.float32)), i64[:, O.REF_COUNT],
                                          i64[:, O.ALT_COUNT], torch.from_numpy(np.asarray(floats)[:, O.INFO_START:].astype(np.float32)),
                                          i64[:, O.HAPLOTYPES_START:])
    finally:
        O.COMPUTE_DTYPE = old


def variant_rows(ints, packed, variants):
    """(ints rows, float row selector, packed rows) of a subset of a batch's variants: their ref rows, then their alt rows"""
    nref, nalt = np.asarray(ints)[:, 0].astype(np.int64), np.asarray(ints)[:, 1].astype(np.int64)
    ro, ao = np.concatenate([[0], np.cumsum(nref)]), int(nref.sum()) + np.concatenate([[0], np.cumsum(nalt)])
    variants = np.asarray(variants, dtype=np.int64)
    rows = [packed[ro[v]:ro[v + 1]] for v in variants] + [packed[ao[v]:ao[v + 1]] for v in variants]
    return np.concatenate(rows) if rows else packed[:0]
