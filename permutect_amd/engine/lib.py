"""ctypes binding of the C ABI in include/permutect_amd.h.

The shared library is built in-tree (permutect_amd/libpermutect_amd.so, see __graft_entry__.build or
permutect_amd/csrc/Makefile).  There is no fallback: if the library is missing, `load()` raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("PMT_LIB", os.path.join(_HERE, "libpermutect_amd.so"))  # (PMT_LIB: development builds for A/B runs)

# ---- limits (must match the header) -------------------------------------------------------------------------------
ABI_VERSION = 10
MAX_WIDTH, MAX_HALF_FFN, MAX_CLUSTERS = 64, 16, 16
MAX_HALF_FFN_WIDE = 32  # d_ffn / 2 of the wide32 build (two tiles per half of a gated block's hidden layer)
MAX_WIDTH_WIDE = 128  # the wide build of the library (csrc/Makefile: `make wide`; engine/instances.py loads it for wider layers)
MAX_ROW_INPUT = 128
MAX_CNN_TAPS = 192
ROWS_INFO, ROWS_ALT_COUNT, ROWS_SOURCE = 0, 1, 2
MAX_OPS, MAX_SKIP_LAYERS, MAX_BLOCKS, MAX_LINEAR = 8, 4, 16, 96
GROUP_WAVES = int(os.environ.get("PMT_GROUP_WAVES", 8))  # PMT_GROUP_WAVES of the built library (include/permutect_amd.h; the variable is for development builds
# loaded through PMT_LIB: 4 was measured)
GROUP_TILES, GROUP_MAX_SETS, TILE = 2 * GROUP_WAVES, 8 * GROUP_WAVES, 16
READS_PACKED_U8, READS_F16, READS_F32 = 0, 1, 2
OP_LINEAR, OP_SKIP = 0, 1
SLOT_FLOATS = 4 * 256

STEP_ON_DEVICE, STEP_SLOT = -1, 1000
E_INVALID, E_UNSUPPORTED, E_CAPACITY, E_LAUNCH, E_WORKSPACE = -1, -2, -3, -4, -5
_ERR = {-1: "invalid argument/descriptor", -2: "configuration not supported by the gfx950 kernels",
        -3: "a read set exceeds the register-resident group capacity", -4: "HIP launch failure",
        -5: "workspace too small"}

i32, i64, vp = C.c_int32, C.c_int64, C.c_void_p


class PmtLinear(C.Structure):
    _fields_ = [("in_dim", i32), ("out_dim", i32), ("w_frag", i32), ("wt_frag", i32), ("b_pvec", i32),
                ("w_src", i32), ("b_src", i32), ("out_split", i32), ("wb_frag", i32), ("wtb_frag", i32), ("wh_frag", i32), ("emit_tab", i32)]


class PmtOp(C.Structure):
    _fields_ = [("kind", i32), ("n_layers", i32), ("selu_after", i32), ("alpha_src", i32),
                ("lin", i32 * MAX_SKIP_LAYERS)]


class PmtMlp(C.Structure):
    _fields_ = [("n_ops", i32), ("in_dim", i32), ("out_dim", i32), ("dropout", i32), ("ops", PmtOp * MAX_OPS)]


class PmtBlock(C.Structure):
    _fields_ = [("norm_w_pvec", i32), ("norm_b_pvec", i32), ("norm_w_src", i32), ("norm_b_src", i32),
                ("proj1", i32 * 2), ("proj2", i32 * 2),
                ("sgu_norm_w_pvec", i32), ("sgu_norm_b_pvec", i32), ("sgu_norm_w_src", i32), ("sgu_norm_b_src", i32),
                ("alpha_src", i32 * 2), ("beta_src", i32 * 2), ("gamma_src", i32),
                ("ref_reg_pvec", i32), ("ref_reg_src", i32), ("reg_weight_phi", i32)]


class PmtHead(C.Structure):
    _fields_ = [("stdev_e_phi", i32), ("dirs_ke_phi", i32), ("art_stdev_k_phi", i32), ("log_w_k_phi", i32),
                ("mu_k_src", i32), ("sigma_k_phi", i32), ("lambda_k_phi", i32), ("reserved", i32)]


MAX_CNN_LAYERS = 12
CNN_CONV, CNN_POOL, CNN_LEAKY_RELU, CNN_SELU, CNN_FLATTEN, CNN_LINEAR = range(6)


class PmtCnnLayer(C.Structure):
    _fields_ = [("kind", i32), ("in_ch", i32), ("in_len", i32), ("out_ch", i32), ("out_len", i32),
                ("kernel", i32), ("stride", i32), ("padding", i32), ("dilation", i32), ("w_src", i32), ("b_src", i32),
                ("in_off", i32), ("out_off", i32), ("lin", i32), ("reserved", i32 * 2)]


class PmtCnn(C.Structure):
    _fields_ = [("n_layers", i32), ("seq_len", i32), ("out_dim", i32), ("max_act", i32), ("sum_act", i32),
                ("reserved", i32 * 3), ("layers", PmtCnnLayer * MAX_CNN_LAYERS)]


class PmtModel(C.Structure):
    _fields_ = [("abi_version", i32), ("num_read_features", i32), ("read_embed_dim", i32),
                ("variant_embed_dim", i32), ("d_model", i32), ("d_ffn", i32), ("num_blocks", i32),
                ("feature_dim", i32), ("num_clusters", i32), ("n_linear", i32),
                ("theta_size", i32), ("phi_size", i32), ("packed_size", i32),
                ("translation_src", i32), ("translation_pvec", i32), ("rotation_lin", i32),
                ("read_mlp", PmtMlp), ("reducer", PmtMlp), ("row_mlp", PmtMlp * 3), ("blocks", PmtBlock * MAX_BLOCKS), ("head", PmtHead), ("cnn", PmtCnn),
                ("lin", PmtLinear * MAX_LINEAR),
                ("force_shape", i32), ("force_cnn", i32), ("cnn_debug", i32), ("emit_base", i32), ("emit_len", i32),
                ("dropout_p", C.c_float)]


class PmtBatch(C.Structure):
    _fields_ = [("num_variants", i32), ("num_groups", i32), ("read_format", i32), ("read_row_bytes", i32),
                ("reads", vp), ("read_index", vp), ("ref_offsets", vp), ("alt_offsets", vp), ("variant_embed", vp),
                ("group_start", vp), ("group_tile_base", vp), ("total_tiles", i64), ("debug_flags", vp), ("group_span", vp),
                ("num_groups_dev", vp), ("set_groups", vp), ("dropout_seed", C.c_uint64), ("join_fault", vp)]


class PmtOutputs(C.Structure):
    _fields_ = [("logits_b", vp), ("logits_bk", vp), ("features_be", vp), ("ref_features_be", vp)]


class PmtOutputGrads(C.Structure):
    _fields_ = [("d_logits_b", vp), ("d_logits_bk", vp), ("d_features_be", vp), ("d_ref_features_be", vp)]


class PmtAdamW(C.Structure):
    _fields_ = [("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("weight_decay", C.c_float), ("max_grad_norm", C.c_float), ("step", i32), ("reserved", i32)]


PHI_EXP, PHI_BOUNDED, PHI_UNIT_ROWS, PHI_LOG_SOFTMAX, PHI_ORTHOGONAL = range(5)
MAX_PHI_SEGS, MAX_ORTHO_DIM = 48, 32


class PmtPhiSeg(C.Structure):
    _fields_ = [("kind", i32), ("theta_off", i32), ("phi_off", i32), ("rows", i32), ("cols", i32), ("reserved", i32),
                ("p0", C.c_float), ("p1", C.c_float), ("base_rs", i32), ("base_cs", i32), ("base", vp)]


class PmtPhiProgram(C.Structure):
    _fields_ = [("n_segs", i32), ("reserved", i32), ("seg", PmtPhiSeg * MAX_PHI_SEGS)]


class PmtIntColumn(C.Structure):
    _fields_ = [("ptr", vp), ("stride", i64), ("elem_bytes", i32), ("reserved", i32)]


class PmtBinning(C.Structure):
    _fields_ = [("num_sources", i32), ("num_variant_types", i32), ("num_ref_bins", i32), ("num_alt_bins", i32),
                ("count_bin_skip", i32), ("max_ref_count", i32), ("max_alt_count", i32), ("reserved", i32)]


def int_column(t) -> PmtIntColumn:
    """a 1-D int32 / int64 tensor (any stride: a column of Batch.int_tensor, a DownsampledBatch's counts) as the kernels read it"""
    c = PmtIntColumn()
    if t is None:
        return c
    assert t.dim() == 1 and t.element_size() in (4, 8) and not t.is_floating_point(), (t.shape, t.dtype)
    c.ptr, c.stride, c.elem_bytes = t.data_ptr(), (t.stride(0) if t.numel() > 1 else 1), t.element_size()
    return c


class PmtDownsample(C.Structure):
    _fields_ = [("num_variants", i32), ("reference_alt_gather", i32), ("seed", C.c_uint64), ("force_random", i64),
                ("ref_offsets", vp), ("alt_offsets", vp), ("ref_weights_b4", vp), ("alt_weights_b4", vp),
                ("ref_fracs_in", vp), ("alt_fracs_in", vp), ("ref_weight_table", vp), ("alt_weight_table", vp),
                ("labels", PmtIntColumn), ("variant_types", PmtIntColumn), ("sources", PmtIntColumn), ("bins", PmtBinning)]


class PmtBalanceArgs(C.Structure):
    _fields_ = [("num_variants", i32), ("recompute", i32), ("attenuation", C.c_float), ("reserved", i32), ("bins", PmtBinning),
                ("labels", PmtIntColumn), ("variant_types", PmtIntColumn), ("sources", PmtIntColumn), ("ref_counts", PmtIntColumn),
                ("alt_counts", PmtIntColumn), ("logits_b", vp), ("counts", vp), ("pseudo_counts", vp),
                ("weights_in", vp), ("unlabeled_weights_in", vp), ("source_weights_in", vp),
                ("weights_out", vp), ("unlabeled_weights_out", vp), ("source_weights_out", vp),
                ("weights_b", vp), ("source_weights_b", vp)]


class PmtEvalArgs(C.Structure):
    _fields_ = [("num_variants", i32), ("epoch_index", i32), ("num_logit_bins", i32), ("min_logit", i32), ("max_logit", i32),
                ("logit_bin_skip", i32), ("bins", PmtBinning), ("labels", PmtIntColumn), ("variant_types", PmtIntColumn),
                ("sources", PmtIntColumn), ("ref_counts", PmtIntColumn), ("alt_counts", PmtIntColumn), ("logits_b", vp), ("weights_b", vp),
                ("nhist", i64)]


class PmtRecordArgs(C.Structure):
    _fields_ = [("num_variants", i32), ("num_bins", i32), ("num_variant_types", i32), ("num_ref_bins", i32),
                ("num_alt_bins", i32), ("count_bin_skip", i32), ("max_ref_count", i32), ("max_alt_count", i32),
                ("labels", PmtIntColumn), ("variant_types", PmtIntColumn), ("sources", PmtIntColumn), ("ref_counts", PmtIntColumn),
                ("alt_counts", PmtIntColumn), ("weights", vp), ("source_weights", vp),
                ("supervised_b", vp), ("unsupervised_b", vp), ("alt_count_b", vp), ("source_b", vp)]


class PmtLossArgs(C.Structure):
    _fields_ = [("num_variants", i32), ("num_clusters", i32), ("num_sources", i32), ("reserved", i32),
                ("max_outlier_logit", C.c_float), ("max_alt_count", C.c_float),
                ("logits_b", vp), ("logits_bk", vp), ("alt_count_raw", vp), ("source_logits", vp),
                ("labels", vp), ("label_stride", i64), ("alt_counts", vp), ("alt_count_stride", i64),
                ("sources", vp), ("source_stride", i64), ("weights", vp), ("source_weights", vp)]


class PmtLossOutputs(C.Structure):
    _fields_ = [("supervised_b", vp), ("unsupervised_b", vp), ("alt_count_b", vp), ("source_b", vp), ("total_b", vp)]


class PmtLossInputGrads(C.Structure):
    _fields_ = [("d_logits_b", vp), ("d_logits_bk", vp), ("d_alt_count_raw", vp), ("d_source_logits", vp)]


EXPORTS = ["pmt_abi_version", "pmt_build_id", "pmt_shape_info", "pmt_shape_id", "pmt_limits", "pmt_struct_bytes", "pmt_model_check", "pmt_plan_groups", "pmt_plan_groups_device", "pmt_plan_device_chunks", "pmt_stash_bytes", "pmt_pack_params",
           "pmt_scan_counts", "pmt_forward", "pmt_backward", "pmt_clip_adamw",
           "pmt_dropout_mask", "pmt_rows_stash_bytes", "pmt_rows_forward", "pmt_rows_backward", "pmt_rows_workspace_floats", "pmt_cnn_forward", "pmt_cnn_backward", "pmt_cnn_stash_floats", "pmt_cnn_workspace_floats",
           "pmt_phi_forward", "pmt_phi_backward", "pmt_build_read_index", "pmt_losses_forward", "pmt_losses_backward",
           "pmt_downsample_counts", "pmt_downsample_index", "pmt_record_losses", "pmt_record_evaluation", "pmt_balance_step", "pmt_posterior_rows",
           "pmt_plan_groups_split", "pmt_layered_scratch_floats", "pmt_forward_layered",
           "pmt_layered_backward_scratch_floats", "pmt_backward_layered", "pmt_host_copy", "pmt_pack_order", "pmt_pack_order_batches", "pmt_prepare_chunk", "pmt_host_copy_rows", "pmt_compose_batch", "pmt_compose_batch_planned"]

_lib = None
_libs = {}  # path -> CDLL: the default library and the per-shape instance libraries (engine/instances.py)


class PmtError(RuntimeError):
    pass


def check(rc: int, what: str) -> int:
    if rc < 0:
        raise PmtError(f"{what} failed: {_ERR.get(rc, rc)} (code {rc})")
    return rc


def load(path: str = None) -> C.CDLL:
    """Load libpermutect_amd.so (once), or the build of it at `path` (a per-shape instance library, engine/instances.py).  Raises if
    it has not been built -- there is no CPU fallback."""
    global _lib
    if path is None and _lib is not None:
        return _lib
    path = LIB_PATH if path is None else path
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise PmtError(f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       f"or `make -C permutect_amd/csrc`.  permutect_amd has no CPU fallback.")
    lib = C.CDLL(path)
    P = C.POINTER
    lib.pmt_abi_version.restype = i32
    lib.pmt_model_check.argtypes = [P(PmtModel)]
    lib.pmt_plan_groups.argtypes = [vp, vp, i32, vp, vp, P(i32)]
    lib.pmt_plan_groups_device.argtypes = [vp, vp, i32, vp, vp, i32, vp, vp, vp, vp]
    lib.pmt_plan_device_chunks.argtypes = [i32]
    lib.pmt_stash_bytes.argtypes = [P(PmtModel), i64, i32]
    lib.pmt_stash_bytes.restype = C.c_size_t
    lib.pmt_pack_params.argtypes = [P(PmtModel), vp, vp, vp, vp, vp]
    lib.pmt_scan_counts.argtypes = [vp, vp, i32, i64, i32, vp, vp, vp]
    lib.pmt_forward.argtypes = [P(PmtModel), vp, vp, vp, vp, P(PmtBatch), P(PmtOutputs), vp, vp]
    lib.pmt_backward.argtypes = [P(PmtModel), vp, vp, vp, vp, P(PmtBatch), P(PmtOutputs), P(PmtOutputGrads), vp, vp, vp, vp, vp, i32, vp]
    lib.pmt_clip_adamw.argtypes = [vp, vp, vp, vp, i64, P(PmtAdamW), vp, vp, vp]
    lib.pmt_cnn_forward.argtypes = [P(PmtModel), vp, vp, vp, vp, i64, i32, vp, i64, vp, vp]
    lib.pmt_cnn_backward.argtypes = [P(PmtModel), vp, vp, vp, vp, i64, i32, vp, i64, vp, vp, vp, C.c_size_t, vp]
    lib.pmt_cnn_workspace_floats.argtypes = [P(PmtModel)]
    lib.pmt_cnn_workspace_floats.restype = C.c_size_t
    lib.pmt_cnn_stash_floats.argtypes = [P(PmtModel)]
    lib.pmt_cnn_stash_floats.restype = C.c_size_t
    lib.pmt_rows_stash_bytes.argtypes = [P(PmtModel), i32, i32]
    lib.pmt_rows_stash_bytes.restype = C.c_size_t
    lib.pmt_rows_forward.argtypes = [P(PmtModel), vp, i32, vp, vp, vp, i64, i32, vp, i64, vp, C.c_uint64, vp]
    lib.pmt_rows_backward.argtypes = [P(PmtModel), vp, i32, vp, vp, vp, i64, i32, vp, i64, vp, vp, vp, i64, C.c_float, vp, C.c_size_t, C.c_uint64, vp]
    lib.pmt_dropout_mask.argtypes = [C.c_uint64, C.c_float, i32, i64, i64, i32, vp]
    lib.pmt_rows_workspace_floats.argtypes = [P(PmtModel), i32]
    lib.pmt_rows_workspace_floats.restype = C.c_size_t
    lib.pmt_build_read_index.argtypes = [vp, vp, vp, i32, vp, vp]
    lib.pmt_record_losses.argtypes = [P(PmtRecordArgs), vp, vp]
    lib.pmt_plan_groups_split.argtypes = [vp, vp, i32, vp, vp, i32, P(i32)]
    lib.pmt_layered_scratch_floats.argtypes = [P(PmtModel), i64, i32]
    lib.pmt_layered_scratch_floats.restype = C.c_size_t
    lib.pmt_forward_layered.argtypes = [P(PmtModel), vp, vp, vp, vp, P(PmtBatch), P(PmtOutputs), vp, vp, vp]
    lib.pmt_host_copy.argtypes = [vp, vp, C.c_size_t, i32]
    lib.pmt_host_copy_rows.argtypes = [vp, vp, i64, i64, i64, i64, i32]
    lib.pmt_compose_batch.argtypes = [vp, i32, vp, i32, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp]
    lib.pmt_compose_batch_planned.argtypes = [vp, i32, vp, i32, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp]
    lib.pmt_pack_order.argtypes = [vp, vp, i32, i32, vp]
    lib.pmt_pack_order_batches.argtypes = [vp, vp, i32, i32, i32, i32, vp]
    lib.pmt_prepare_chunk.argtypes = [vp, i64, i32, i32, i32, i32, C.c_uint64, i32, i32, i32, vp, vp, vp, vp, i64, vp, vp]
    lib.pmt_layered_backward_scratch_floats.argtypes = [P(PmtModel), i64, i32]
    lib.pmt_layered_backward_scratch_floats.restype = C.c_size_t
    lib.pmt_backward_layered.argtypes = [P(PmtModel), vp, vp, vp, vp, P(PmtBatch), P(PmtOutputs), P(PmtOutputGrads), vp, vp, vp,
                                         vp, vp, vp, i32, vp]
    lib.pmt_downsample_counts.argtypes = [P(PmtDownsample), vp, vp, vp, vp, vp]
    lib.pmt_downsample_index.argtypes = [P(PmtDownsample), vp, vp, vp, vp, vp, vp]
    lib.pmt_balance_step.argtypes = [P(PmtBalanceArgs), vp]
    lib.pmt_record_evaluation.argtypes = [P(PmtEvalArgs), vp, vp]
    lib.pmt_posterior_rows.argtypes = [vp, i64, i32, i32, vp, vp, i32, vp, i32, vp, i64, vp]
    lib.pmt_losses_forward.argtypes = [P(PmtLossArgs), P(PmtLossOutputs), vp]
    lib.pmt_losses_backward.argtypes = [P(PmtLossArgs), P(PmtLossOutputs), P(PmtLossInputGrads), vp]
    lib.pmt_phi_forward.argtypes = [P(PmtPhiProgram), vp, vp, vp]
    lib.pmt_phi_backward.argtypes = [P(PmtPhiProgram), vp, vp, vp, vp, vp]
    for name in EXPORTS:
        fn = getattr(lib, name)
        if name not in ("pmt_abi_version", "pmt_stash_bytes", "pmt_rows_stash_bytes", "pmt_layered_scratch_floats",
                        "pmt_layered_backward_scratch_floats", "pmt_cnn_stash_floats", "pmt_cnn_workspace_floats",
                        "pmt_rows_workspace_floats"):
            fn.restype = i32
    lib.pmt_struct_bytes.argtypes = [i32]
    if lib.pmt_abi_version() != ABI_VERSION:
        raise PmtError("libpermutect_amd.so ABI version mismatch; rebuild it")
    for which, st in enumerate([PmtModel, PmtBatch, PmtOutputs, PmtOutputGrads, PmtAdamW, PmtLinear, PmtOp, PmtMlp,
                                PmtBlock, PmtHead, PmtPhiProgram, PmtLossArgs, PmtDownsample, PmtRecordArgs, PmtBalanceArgs, PmtEvalArgs]):
        if lib.pmt_struct_bytes(which) != C.sizeof(st):
            raise PmtError(f"ctypes layout of {st.__name__} ({C.sizeof(st)} B) does not match the library "
                           f"({lib.pmt_struct_bytes(which)} B)")
    lib.pmt_shape_info.argtypes = [P(i32)]
    lib.pmt_shape_id.argtypes = [P(PmtModel)]
    lib.pmt_build_id.argtypes = [C.c_char_p, i32]
    # the planner's constants below are this module's; the kernels' are the library's: they must be the same numbers
    if limits_of(lib)["group_waves"] != GROUP_WAVES:
        raise PmtError(f"{path} is built for {limits_of(lib)['group_waves']} waves per workgroup, the host side plans for {GROUP_WAVES} "
                       "(PMT_GROUP_WAVES in the environment must match a development build loaded through PMT_LIB)")
    _libs[path] = lib
    if path == LIB_PATH:
        _lib = lib
    return lib


def raw_stream(device=None) -> int:
    """the current HIP stream's handle of `device` (default: the current device) in ONE C call: torch.cuda.current_stream() builds a
    Stream object through four Python layers, ~10 us each time, which is a tenth of a host-bound small-batch training step"""
    import torch
    idx = torch.cuda.current_device() if device is None or getattr(device, "index", None) is None else device.index
    return torch._C._cuda_getCurrentRawStream(idx)


def limits_of(lib: C.CDLL) -> dict:
    """the compile-time limits of a build of the library (pmt_limits)"""
    v = (i32 * 4)()
    check(lib.pmt_limits(v), "pmt_limits")
    return {"max_width": int(v[0]), "max_half_ffn": int(v[1]), "slot_floats": int(v[2]), "group_waves": int(v[3])}


def build_id(lib: C.CDLL) -> str:
    """the hash of the sources a build of the library was compiled from (pmt_build_id)"""
    buf = C.create_string_buffer(32)
    check(lib.pmt_build_id(buf, 32), "pmt_build_id")
    return buf.value.decode()


def shape_of(lib: C.CDLL):
    """(NTF, NTR, NTD, NTE, F, R, D, H, E): the shape a build's exact-width instances are compiled for (pmt_shape_info)"""
    v = (i32 * 9)()
    check(lib.pmt_shape_info(v), "pmt_shape_info")
    return tuple(int(x) for x in v)
