"""Lowering of an ArtifactModel module tree to the device-side representation.

  * `theta`  : ONE flat fp32 buffer holding every learnable leaf tensor in natural layout.  The nn.Parameters of the
               model are re-bound as views into it, so the fused optimizer, the RCCL gradient all-reduce and the HIP
               kernels all address the same memory, and state_dict()/load_state_dict() keep working unchanged.
  * `gtheta` : same layout, gradients.  `param.grad` of every leaf is a view into it.
  * `phi`    : small buffer of materialised parametrizations (exp / bounded sigmoid / unit vectors / log_softmax /
               orthogonal matrix), produced by torch each step so that autograd carries d(phi) back to the
               `.original` leaves (reference architecture/parameterizations.py).
  * `packed` : weights in MFMA A-fragment order + per-feature vectors in tile-position order (csrc/pmt_device.hpp),
               rebuilt on device by pmt_pack_params after every parameter update.
  * `PmtModel` descriptor with all offsets (include/permutect_amd.h).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional

import torch
from torch import nn

from permutect_amd.architecture import modules as M
from permutect_amd.engine import lib as L


def _split0(h: int) -> int:
    """positions of the FIRST half in the virtual output of a gated block's first projection (PmtLinear.out_split = h = d_ffn / 2): one
    16-feature tile, two for h in 17 .. 32 (pmt_device.hpp: PMT_SPLIT0 of the build that runs such a model, PMT_MAX_HALF_FFN = 32)"""
    return 16 * ((h + 15) // 16)


def _ceil16(n: int) -> int:
    return (n + 15) // 16 * 16


class ParamSpace:
    """Flat parameter / gradient buffers with the model's Parameters re-bound as views."""

    # Gradients become final in two waves during loss.backward(): the read-set kernel (pmt_backward) and the adversaries'
    # row kernels finish every leaf they own first; the per-variant branches that autograd runs AFTER it (info MLP,
    # haplotype CNN) and the parametrization adjoint (pmt_phi_backward, the last node of the graph) finish the rest.  The
    # flat buffer is laid out in that order, [early | late], so that the data-parallel all-reduce of the early bucket can
    # run on a side stream underneath the late kernels (training/distributed.py: BucketedGradAllReduce).
    LATE_PREFIXES = ("info_embedding.", "haplotypes_cnn.")

    @classmethod
    def is_late(cls, name: str) -> bool:
        return name.startswith(cls.LATE_PREFIXES) or name.endswith(".original")

    def __init__(self, module: nn.Module, device: torch.device):
        named = list(module.named_parameters())
        named = [(n, p) for n, p in named if not self.is_late(n)] + [(n, p) for n, p in named if self.is_late(n)]
        params = [p for _, p in named]
        self.offsets: Dict[int, int] = {}
        off = 0
        self.late_start = None
        for n, p in named:
            assert p.dtype == torch.float32, "the engine computes in fp32 (reference data/datum.py:37-38)"
            if self.late_start is None and self.is_late(n):
                self.late_start = off
            self.offsets[id(p)] = off
            off += (p.numel() + 3) // 4 * 4  # keep every tensor 16-byte aligned
        self.size = max(off, 4)
        if self.late_start is None:
            self.late_start = self.size
        self.theta = torch.zeros(self.size, dtype=torch.float32, device=device)
        self.gtheta = torch.zeros(self.size, dtype=torch.float32, device=device)
        self.params = params
        with torch.no_grad():
            for p in params:
                o, n = self.offsets[id(p)], p.numel()
                self.theta[o:o + n].copy_(p.detach().reshape(-1))
                p.data = self.theta[o:o + n].view(p.shape)
        self.bind_grads()

    def bind_grads(self, quick: bool = False):
        """Every parameter's .grad is a view into the flat gradient buffer (the kernels accumulate there; torch's
        AccumulateGrad adds in place and keeps the tensor object).  Called once per step: the common case -- nothing was
        re-bound -- is ~130 identity checks."""
        views = getattr(self, "_grad_views", None)
        if views is None:
            views = self._grad_views = [self.gtheta[self.offsets[id(p)]:self.offsets[id(p)] + p.numel()].view(p.shape) for p in self.params]
        if quick and self.params and self.params[0].grad is views[0] and self.params[-1].grad is views[-1]:
            return  # (the backward nodes of ONE step: zero_grad has just checked every parameter; whoever drops gradients drops all of them)
        for p, v in zip(self.params, views):
            if p.grad is not v:
                p.grad = v

    def offset_of(self, p: nn.Parameter) -> int:
        return self.offsets[id(p)]


class EnginePlan:
    """Descriptor + buffers for one ArtifactModel instance on one device."""

    def __init__(self, model, space: ParamSpace, device: torch.device):
        self.device = device
        self.space = space
        d = L.PmtModel()
        d.abi_version = L.ABI_VERSION
        self.desc = d
        self._packed_off = 0
        self._n_lin = 0
        self._phi_off = 0
        self._phi_prog = None
        self.phi_layout: List[tuple] = []  # (name, offset, numel)
        self.bn_folds: List[tuple] = []    # (BatchNorm1d, Linear) pairs folded into phi, in phi order (materialize_phi)
        self.cnn_bn_folds: List[tuple] = []  # haplotype CNN: (BatchNorm1d, Conv1d | Linear, "out" | "in", features per BN channel), stack order

        read_mlp: M.MLP = model.read_embedding
        enc: M.GatedRefAltMLP = model.ref_alt_reads_encoder
        reducer: M.MLP = model.reducer
        fc: M.FeatureClustering = model.feature_clustering
        d.num_read_features = read_mlp.input_dimension()
        d.read_embed_dim = read_mlp.output_dimension()
        d.d_model = enc.dimension
        d.variant_embed_dim = d.d_model - d.read_embed_dim
        d.d_ffn = enc.d_ffn
        d.num_blocks = len(enc.blocks)
        d.feature_dim = reducer.output_dimension()
        d.num_clusters = fc.num_artifact_clusters
        h, e, k = d.d_ffn // 2, d.feature_dim, d.num_clusters
        if d.num_blocks > L.MAX_BLOCKS:
            raise L.PmtError(f"{d.num_blocks} gated blocks exceed the kernel limit {L.MAX_BLOCKS}")

        self._lower_mlp(d.read_mlp, read_mlp)
        for i, blk in enumerate(enc.blocks):
            b = d.blocks[i]
            s = blk.sgu
            b.norm_w_src, b.norm_b_src = space.offset_of(blk.norm.weight), space.offset_of(blk.norm.bias)
            b.sgu_norm_w_src, b.sgu_norm_b_src = space.offset_of(s.norm.weight), space.offset_of(s.norm.bias)
            b.ref_reg_src = space.offset_of(s.ref_regularizer)
            # One packed region per projection pair: [W_ref | W_alt | vectors used with them | pad to 256].
            # The first region also carries the block's LayerNorm / gating vectors.
            vecs = {}
            b.proj1[0], b.proj1[1] = self._add_linear_pair(
                blk.proj1_ref, blk.proj1_alt, out_split=h,
                extra_vectors=[("norm_w", _ceil16(d.d_model)), ("norm_b", _ceil16(d.d_model)), ("sgu_w", _ceil16(h)),
                               ("sgu_b", _ceil16(h)), ("rho", _ceil16(h))], out=vecs)
            b.norm_w_pvec, b.norm_b_pvec = vecs["norm_w"], vecs["norm_b"]
            b.sgu_norm_w_pvec, b.sgu_norm_b_pvec, b.ref_reg_pvec = vecs["sgu_w"], vecs["sgu_b"], vecs["rho"]
            b.proj2[0], b.proj2[1] = self._add_linear_pair(blk.proj2_ref, blk.proj2_alt)
            b.alpha_src[0], b.alpha_src[1] = space.offset_of(s.alpha_ref), space.offset_of(s.alpha_alt)
            b.beta_src[0], b.beta_src[1] = space.offset_of(s.beta_ref), space.offset_of(s.beta_alt)
            b.gamma_src = space.offset_of(s.gamma)
            b.reg_weight_phi = self._alloc_phi(f"reg_weight.{i}", 1)
        self._lower_mlp(d.reducer, reducer)
        n_readset_lin = self._n_lin  # the linears of the read-set kernels so far (+ the rotation below)
        # per-variant row MLPs (pmt_rows_*)
        self._lower_mlp(d.row_mlp[L.ROWS_INFO], model.info_embedding, max_in=L.MAX_ROW_INPUT)
        self._lower_mlp(d.row_mlp[L.ROWS_ALT_COUNT], model.alt_count_predictor.wrapped_module)
        if model.num_sources > 1:
            self._lower_mlp(d.row_mlp[L.ROWS_SOURCE], model.source_predictor.wrapped_module)

        tr = model.pre_clustering_transform
        d.translation_src = space.offset_of(tr.translation_e)
        q_phi = self._alloc_phi("rotation", e * e)
        vecs = {}
        d.rotation_lin = self._add_raw_linear(e, e, w_src=-(q_phi + 2), b_src=-1, has_bias=False,
                                              extra_vectors=[("translation", _ceil16(e))], out=vecs)
        d.translation_pvec = vecs["translation"]

        hd = d.head
        hd.stdev_e_phi = self._alloc_phi("stdev_e", e)
        hd.dirs_ke_phi = self._alloc_phi("dirs_ke", k * e)
        hd.art_stdev_k_phi = self._alloc_phi("art_stdev_k", k)
        hd.log_w_k_phi = self._alloc_phi("log_w_k", k)
        hd.sigma_k_phi = self._alloc_phi("sigma_k", k)
        hd.lambda_k_phi = self._alloc_phi("lambda_k", k)
        hd.mu_k_src = space.offset_of(fc.artifact_emg.mu_k)

        self._lower_cnn(d.cnn, model.haplotypes_cnn, model.haplotypes_length() // 2)
        d.n_linear = self._n_lin
        # the weights once more as three bf16 pieces per value (PmtLinear.wb_frag / wtb_frag): 1 KiB per (32-wide k block,
        # 16-row tile, piece), one more fragment of slack behind each range
        for i in range(self._n_lin):
            lin = d.lin[i]
            out_v = _split0(lin.out_split) + lin.out_split if lin.out_split else lin.out_dim
            nmt, nkt = (out_v + 15) // 16, (lin.in_dim + 15) // 16
            lin.wb_frag = self._alloc_packed(nmt * ((nkt + 1) // 2) * 3 * 256 + 256)
            lin.wtb_frag = self._alloc_packed(nkt * ((nmt + 1) // 2) * 3 * 256 + 256)
            # and as two f16 pieces (PmtLinear.wh_frag): the forward's operands (three MFMAs per product instead of six)
            lin.wh_frag = self._alloc_packed(nmt * ((nkt + 1) // 2) * 2 * 256 + 256)
        # the emit tables (int32 destinations of the dW blocks and the bias rows) lie back to back: a row of the backward's
        # private partial sums mirrors this region entry for entry (pmt_backward: grad_partials)
        # (only the read-set backward emits through tables; the row kernels compute their few destinations)
        d.emit_base = self._packed_off
        for i in range(d.n_linear):
            lin = d.lin[i]
            lin.emit_tab = -1
            if i >= n_readset_lin and i != d.rotation_lin:
                continue
            out_v = _split0(lin.out_split) + lin.out_split if lin.out_split > 0 else lin.out_dim
            nmt, nkt = (out_v + 15) // 16, (lin.in_dim + 15) // 16
            lin.emit_tab = self._alloc_packed(nmt * nkt * 256 + nmt * 16)
        d.emit_len = self._packed_off - d.emit_base
        d.theta_size, d.phi_size, d.packed_size = space.size, max(self._phi_off, 4), self._packed_off + 512  # slack: the kernels prefetch two fragments ahead
        # Which kernel instances run is part of the descriptor (the library itself reads no environment).  The parity tests
        # select the non-default instances through these variables, read HERE, once, when the model is lowered.
        d.force_shape = {"tile": 1, "any": 2, "bf16": 3, "bf16x3": 5}.get(os.environ.get("PMT_SHAPE", ""), 0)
        d.force_cnn = {"general": 1, "wave": 2, "batched": 3}.get(os.environ.get("PMT_CNN", ""), 0)
        d.cnn_debug = int(os.environ.get("PMT_CNN_DBG", "0"))
        d.dropout_p = max((float(getattr(m, "dropout_p", 0.0)) for m in model.modules() if isinstance(m, M.MLP)), default=0.0)

        from permutect_amd.engine.instances import library_for
        lib = self.lib = library_for(d)  # the build whose kernels run this model (the default one, a per-shape one, the wide one)
        L.check(lib.pmt_model_check(C.byref(d)), "pmt_model_check")
        self.packed = torch.zeros(d.packed_size, dtype=torch.float32, device=device)
        # private partial sums of the backward's weight-gradient blocks: one row per compute unit (pmt_backward: grad_partials;
        # the kernel leaves them zero).  PMT_GRAD_PARTIALS=0 falls back to global atomics (the parity tests run both).
        rows = torch.cuda.get_device_properties(device).multi_processor_count if torch.device(device).type == "cuda" else 0
        rows = int(os.environ.get("PMT_GRAD_PARTIALS", rows))
        self.partial_rows = rows if d.emit_len > 0 else 0
        self.grad_partials = torch.zeros(max(self.partial_rows * d.emit_len, 4), dtype=torch.float32, device=device)
        self.gphi_size = d.phi_size
        raw = bytes(d)
        self.desc_dev = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)
        self.debug_flags = torch.zeros(64 + 8 * 512, dtype=torch.int32, device=device)  # [0] unused, [1] debug switches, [2] traced workgroup + 1, [8:56] 24 x u64 cycle counters, [64:] event log of the traced workgroup (8 waves x 256 x (id, clock))
        self.stash_slots = (d.read_mlp.n_ops - 1) + (d.num_blocks + 1) + (d.reducer.n_ops - 1) + d.num_blocks  # = pmt_stash_slots

    # ---- allocation helpers --------------------------------------------------------------------------------------
    def _alloc_packed(self, n: int) -> int:
        off = self._packed_off
        self._packed_off += (n + 3) // 4 * 4
        return off

    def _alloc_phi(self, name: str, n: int) -> int:
        off = self._phi_off
        self.phi_layout.append((name, off, n))
        self._phi_off += n
        return off

    def _close_region(self, start: int) -> int:
        """Pad the packed cursor to a multiple of 256 floats past `start` (LDS-DMA moves 1 KiB pieces); returns the
        region length."""
        n = (self._packed_off - start + 255) // 256 * 256
        self._packed_off = start + n
        return n

    def _add_raw_linear(self, in_dim: int, out_dim: int, w_src: int, b_src: int, has_bias: bool, out_split: int = 0,
                        alloc: bool = True, extra_vectors=(), out=None, max_in: int = L.MAX_WIDTH_WIDE) -> int:
        if self._n_lin >= L.MAX_LINEAR:
            raise L.PmtError("too many linear layers for the kernel descriptor")
        if in_dim > max_in or out_dim > L.MAX_WIDTH_WIDE:  # (beyond 64: the wide build of the library, engine/instances.py)
            raise L.PmtError(f"layer width {in_dim}->{out_dim} exceeds the register-resident limit {L.MAX_WIDTH_WIDE}")
        lin = self.desc.lin[self._n_lin]
        out_v = _split0(out_split) + out_split if out_split else out_dim
        nmt, nkt = (out_v + 15) // 16, (in_dim + 15) // 16
        lin.in_dim, lin.out_dim, lin.out_split = in_dim, out_dim, out_split
        if alloc:
            # packed region: [forward fragments | bias | extra vectors | pad]; the transposed fragments follow
            self._packed_off = (self._packed_off + 255) // 256 * 256
            lin.w_frag = self._alloc_packed(nmt * nkt * 256)
            lin.b_pvec = self._alloc_packed(nmt * 16) if has_bias else -1
            for name, n in extra_vectors:
                out[name] = self._alloc_packed(n)
            self._close_region(lin.w_frag)
            lin.wt_frag = self._alloc_packed(nmt * nkt * 256)
        lin.w_src, lin.b_src = w_src, b_src
        self._n_lin += 1
        return self._n_lin - 1

    def _add_linear_pair(self, ref: nn.Linear, alt: nn.Linear, out_split: int = 0, extra_vectors=(), out=None):
        ids = []
        for layer in (ref, alt):
            ids.append(self._add_raw_linear(layer.in_features, layer.out_features, self.space.offset_of(layer.weight),
                                            self.space.offset_of(layer.bias), True, out_split, alloc=False))
        lr, la = self.desc.lin[ids[0]], self.desc.lin[ids[1]]
        out_v = _split0(out_split) + out_split if out_split else lr.out_dim
        nfl = ((out_v + 15) // 16) * ((lr.in_dim + 15) // 16) * 256
        nb = ((out_v + 15) // 16) * 16
        self._packed_off = (self._packed_off + 255) // 256 * 256
        lr.w_frag = self._alloc_packed(nfl); la.w_frag = self._alloc_packed(nfl)
        lr.b_pvec = self._alloc_packed(nb); la.b_pvec = self._alloc_packed(nb)
        for name, n in extra_vectors:
            out[name] = self._alloc_packed(n)
        self._close_region(lr.w_frag)
        lr.wt_frag = self._alloc_packed(nfl); la.wt_frag = self._alloc_packed(nfl)
        return ids[0], ids[1]

    def _add_linear(self, layer: nn.Linear, out_split: int = 0, max_in: int = L.MAX_WIDTH_WIDE, bn: Optional[nn.BatchNorm1d] = None) -> int:
        if bn is not None:
            # eval-mode BatchNorm1d in front of the Linear (reference mlp.py:52-53) is an affine map per input feature,
            # x -> s x + t with s = w / sqrt(running_var + eps), t = b - running_mean s: the Linear the kernels run is
            # W' = W diag(s), b' = b + W t, materialised into phi by materialize_phi (bn_folds) like any parametrization
            n = self._n_lin
            w_phi = self._alloc_phi(f"bnfold.w.{n}", layer.out_features * layer.in_features)
            b_phi = self._alloc_phi(f"bnfold.b.{n}", layer.out_features)
            self.bn_folds.append((bn, layer))
            return self._add_raw_linear(layer.in_features, layer.out_features, -(w_phi + 2), -(b_phi + 2), True, out_split, max_in=max_in)
        b_src = self.space.offset_of(layer.bias) if layer.bias is not None else -1
        return self._add_raw_linear(layer.in_features, layer.out_features, self.space.offset_of(layer.weight), b_src,
                                    layer.bias is not None, out_split, max_in=max_in)

    def _lower_mlp(self, dst: L.PmtMlp, mlp: M.MLP, max_in: int = L.MAX_WIDTH_WIDE):
        # nn.Dropout (reference mlp.py:57-58: one behind every Linear when dropout_p > 0) becomes a flag of the MLP: the kernels
        # mask every Linear's output of a flagged MLP when the batch brings a seed (train mode), and ignore it otherwise (eval)
        dst.dropout = int(any(isinstance(c, nn.Dropout) for c in mlp._model.modules()))
        def strip(seq):
            """children without the Dropouts, and the BatchNorm1d (if any) in front of every Linear"""
            kept, bn_of, pending = [], {}, None
            for m in seq:
                if isinstance(m, nn.Dropout):
                    continue
                if isinstance(m, nn.BatchNorm1d):
                    pending = m
                    continue
                if pending is not None:
                    if not isinstance(m, nn.Linear):
                        raise L.PmtError("a BatchNorm1d that is not directly in front of a Linear")
                    bn_of[m] = pending
                    pending = None
                kept.append(m)
            return kept, bn_of
        children, bn_of = strip(mlp._model.children())
        ops = []
        i = 0
        while i < len(children):
            c = children[i]
            if isinstance(c, nn.Linear):
                selu_after = i + 1 < len(children) and isinstance(children[i + 1], nn.SELU)
                ops.append(("lin", c, selu_after))
                i += 2 if selu_after else 1
            elif isinstance(c, M.DenseSkipBlock):
                inner, inner_bn = strip(c.mlp._model.children())  # (SELU, Linear) * n
                bn_of.update(inner_bn)
                lins = [m for m in inner if isinstance(m, nn.Linear)]
                assert len(inner) == 2 * len(lins) and all(isinstance(m, nn.SELU) for m in inner[0::2])
                ops.append(("skip", c, lins))
                i += 1
            else:
                raise L.PmtError(f"unsupported layer in MLP: {type(c).__name__}")
        if len(ops) > L.MAX_OPS:
            raise L.PmtError(f"MLP with {len(ops)} top-level ops exceeds the kernel limit {L.MAX_OPS}")
        dst.n_ops, dst.in_dim, dst.out_dim = len(ops), mlp.input_dimension(), mlp.output_dimension()
        for j, op in enumerate(ops):
            o = dst.ops[j]
            if op[0] == "lin":
                o.kind, o.n_layers, o.selu_after, o.alpha_src = L.OP_LINEAR, 1, int(op[2]), -1
                o.lin[0] = self._add_linear(op[1], max_in=max_in if j == 0 else L.MAX_WIDTH_WIDE, bn=bn_of.get(op[1]))
            else:
                blk, lins = op[1], op[2]
                if len(lins) > L.MAX_SKIP_LAYERS:  # (three and four layers: the generic instances' interpreter, pmt_shape_id 0)
                    raise L.PmtError(f"skip blocks deeper than {L.MAX_SKIP_LAYERS} layers are not supported by the gfx950 kernels")
                o.kind, o.n_layers, o.selu_after = L.OP_SKIP, len(lins), 0
                o.alpha_src = self.space.offset_of(blk.alpha)
                for t, lin in enumerate(lins):
                    o.lin[t] = self._add_linear(lin, bn=bn_of.get(lin))

    def _lower_cnn(self, dst: L.PmtCnn, cnn: M.DNASequenceConvolution, seq_len: int):
        layers = self._strip_cnn_batchnorm(list(cnn._model.children()))
        if len(layers) > L.MAX_CNN_LAYERS:
            raise L.PmtError(f"haplotype CNN with {len(layers)} layers exceeds the kernel limit {L.MAX_CNN_LAYERS}")
        ch, length = M.INITIAL_NUM_CHANNELS, seq_len
        off = ch * length  # the one-hot input occupies [0, 10*S)
        in_off, max_act = 0, ch * length
        for i, layer in enumerate(layers):
            c = dst.layers[i]
            c.in_ch, c.in_len, c.in_off = ch, length, in_off
            c.kernel, c.stride, c.padding, c.dilation = 1, 1, 0, 1
            c.w_src = c.b_src = c.lin = -1
            if isinstance(layer, nn.Conv1d):
                if layer.groups != 1 or layer.padding_mode != "zeros" or isinstance(layer.padding, str):
                    raise L.PmtError("unsupported Conv1d options in the haplotype CNN")
                c.kind = L.CNN_CONV
                c.kernel, c.stride, c.padding, c.dilation = layer.kernel_size[0], layer.stride[0], layer.padding[0], layer.dilation[0]
                c.w_src, c.b_src = self.space.offset_of(layer.weight), self.space.offset_of(layer.bias)
                # the weight [out][in][k] is contiguous = an [out][in*k] matrix: packed like any linear (implicit GEMM)
                c.lin = self._add_raw_linear(ch * c.kernel, layer.out_channels, c.w_src, c.b_src, True, max_in=L.MAX_CNN_TAPS)
                ch, length = layer.out_channels, M._conv_len(length, kernel_size=c.kernel, stride=c.stride, padding=c.padding, dilation=c.dilation)
            elif isinstance(layer, nn.MaxPool1d):
                if layer.padding != 0 or layer.dilation != 1 or layer.ceil_mode:
                    raise L.PmtError("unsupported MaxPool1d options in the haplotype CNN")
                c.kind = L.CNN_POOL
                c.kernel = layer.kernel_size
                c.stride = layer.stride if layer.stride is not None else layer.kernel_size
                length = M._pool_len(length, kernel_size=c.kernel, stride=c.stride)
            elif isinstance(layer, (nn.LeakyReLU, nn.SELU)):
                # activations run in place (their derivative is taken from the output): no region of their own
                c.kind = L.CNN_LEAKY_RELU if isinstance(layer, nn.LeakyReLU) else L.CNN_SELU
                c.out_ch, c.out_len, c.out_off = ch, length, in_off
                continue
            elif isinstance(layer, nn.Flatten):
                c.kind = L.CNN_FLATTEN
                ch, length = ch * length, 1
                c.out_ch, c.out_len, c.out_off = ch, length, in_off
                continue
            elif isinstance(layer, nn.Linear):
                c.kind = L.CNN_LINEAR
                c.w_src, c.b_src = self.space.offset_of(layer.weight), self.space.offset_of(layer.bias)
                ch, length = layer.out_features, 1
            else:
                raise L.PmtError(f"unsupported layer in the haplotype CNN: {type(layer).__name__}")
            c.out_ch, c.out_len, c.out_off = ch, length, off
            in_off = off
            off += ch * length
            max_act = max(max_act, ch * length)
        dst.n_layers, dst.seq_len, dst.out_dim = len(layers), seq_len, ch * length
        dst.max_act, dst.sum_act = max_act, off

    def _strip_cnn_batchnorm(self, layers):
        """The reference's `batch_norm` token (dna_sequence_convolution.py:82-83) in EVAL mode is x -> s x + t per channel, s = w /
        sqrt(running_var + eps), t = b - running_mean s.  The kernels have no such layer; every BatchNorm1d is folded into a
        neighbour (recorded in cnn_bn_folds, applied by folded_cnn_theta):
          "out": directly behind a Conv1d / Linear -> that layer's output rows:   W' = diag(s) W,  b' = s b + t;
          "in":  otherwise in front of (a Flatten and) an un-padded Conv1d / a Linear -> its input channels:  W' = W diag(s),
                 b' = b + W t  (a zero-padded convolution would see t only inside the sequence: refused).
        Returns the layer list without the BatchNorms.  Training with batch statistics is refused by ArtifactModel."""
        kept = []
        for i, layer in enumerate(layers):
            if not isinstance(layer, nn.BatchNorm1d):
                kept.append(layer)
                continue
            prev = layers[i - 1] if i > 0 else None
            if isinstance(prev, (nn.Conv1d, nn.Linear)):
                self.cnn_bn_folds.append((layer, prev, "out", 1))
                continue
            j = i + 1
            while j < len(layers) and isinstance(layers[j], nn.Flatten):
                j += 1
            nxt = layers[j] if j < len(layers) else None
            if isinstance(nxt, nn.Conv1d) and nxt.padding[0] == 0 and nxt.in_channels == layer.num_features:
                self.cnn_bn_folds.append((layer, nxt, "in", 1))
            elif isinstance(nxt, nn.Linear) and nxt.in_features % layer.num_features == 0:
                self.cnn_bn_folds.append((layer, nxt, "in", nxt.in_features // layer.num_features))
            else:
                raise L.PmtError("a batch_norm token of the haplotype CNN that is neither directly behind a convolution / linear nor in "
                                 "front of an un-padded convolution / a linear cannot be folded (permutect_amd runs it in eval mode only)")
        return kept

    def folded_cnn_theta(self, theta: torch.Tensor) -> torch.Tensor:
        """A copy of the flat parameter buffer in which the haplotype CNN's convolutions / linear carry their folded BatchNorms
        (eval mode; cnn_bn_folds in stack order, so an input-side fold of a layer precedes its output-side fold)."""
        out = theta.detach().clone()
        space = self.space
        with torch.no_grad():
            for bn, layer, side, per in self.cnn_bn_folds:
                s = bn.weight / torch.sqrt(bn.running_var + bn.eps)
                t = bn.bias - bn.running_mean * s
                wo, bo = space.offset_of(layer.weight), space.offset_of(layer.bias)
                w = out[wo:wo + layer.weight.numel()].view_as(layer.weight)
                b = out[bo:bo + layer.bias.numel()]
                if side == "out":
                    b.mul_(s).add_(t)
                    w.mul_(s.view(-1, *([1] * (w.dim() - 1))))
                elif w.dim() == 3:  # Conv1d [out][in][k]
                    b.add_((w * t.view(1, -1, 1)).sum(dim=(1, 2)))
                    w.mul_(s.view(1, -1, 1))
                else:               # Linear [out][in], `per` consecutive input features per BatchNorm channel (flatten: c * len + p)
                    se, te = s.repeat_interleave(per), t.repeat_interleave(per)
                    b.add_(w @ te)
                    w.mul_(se.view(1, -1))
        return out

    # ---- phi ------------------------------------------------------------------------------------------------------
    def phi_program(self, model):
        """The PmtPhiProgram evaluating every parametrization on the device (pmt_phi_forward / _backward), or None when
        the model has a parametrization the kernels do not cover (orthogonal matrices wider than MAX_ORTHO_DIM): the torch
        evaluation `materialize_phi` is then used instead.  Built once; the base-matrix pointer is refreshed per call."""
        tr, fc, space = model.pre_clustering_transform, model.feature_clustering, self.space
        e = self.desc.feature_dim
        if e > L.MAX_ORTHO_DIM or self.bn_folds:  # (folded BatchNorm weights are materialised by torch: eval mode only)
            return None
        if self._phi_prog is None:
            offs = {name: off for name, off, _ in self.phi_layout}
            prog = L.PmtPhiProgram()
            segs = []

            def original(mod, name):
                return space.offset_of(getattr(mod.parametrizations, name).original)

            def bounded(mod, name):
                p = getattr(mod.parametrizations, name)[0]
                return float(p.min_val), float(p.size)

            for i, blk in enumerate(model.ref_alt_reads_encoder.blocks):
                segs.append((L.PHI_EXP, original(blk.sgu, "reg_weight"), offs[f"reg_weight.{i}"], 1, 1, 0.0, 0.0))
            segs.append((L.PHI_ORTHOGONAL, original(tr.rotation_ee, "weight"), offs["rotation"], e, e, 0.0, 0.0))
            k = self.desc.num_clusters
            segs.append((L.PHI_BOUNDED, original(fc, "nonartifact_stdev_e"), offs["stdev_e"], 1, e) + bounded(fc, "nonartifact_stdev_e"))
            segs.append((L.PHI_UNIT_ROWS, original(fc, "artifact_directions_ke"), offs["dirs_ke"], k, e, 0.0, 0.0))
            segs.append((L.PHI_BOUNDED, original(fc, "artifact_stdev_k"), offs["art_stdev_k"], 1, k) + bounded(fc, "artifact_stdev_k"))
            segs.append((L.PHI_LOG_SOFTMAX, original(fc, "log_cluster_weights_k"), offs["log_w_k"], 1, k, 0.0, 0.0))
            segs.append((L.PHI_BOUNDED, original(fc.artifact_emg, "sigma_k"), offs["sigma_k"], 1, k) + bounded(fc.artifact_emg, "sigma_k"))
            segs.append((L.PHI_BOUNDED, original(fc.artifact_emg, "lambda_k"), offs["lambda_k"], 1, k) + bounded(fc.artifact_emg, "lambda_k"))
            assert len(segs) <= L.MAX_PHI_SEGS
            prog.n_segs = len(segs)
            for sg, (kind, toff, poff, rows, cols, p0, p1) in zip(prog.seg, segs):
                sg.kind, sg.theta_off, sg.phi_off, sg.rows, sg.cols, sg.p0, sg.p1 = kind, toff, poff, rows, cols, p0, p1
            self._phi_prog = prog
            self._phi_rot_index = len(model.ref_alt_reads_encoder.blocks)
        base = getattr(tr.rotation_ee.parametrizations.weight[0], "base", None)
        sg = self._phi_prog.seg[self._phi_rot_index]
        if base is None:
            sg.base, sg.base_rs, sg.base_cs = None, 0, 0
        else:  # torch keeps the base as a transposed view: pass its strides
            sg.base, sg.base_rs, sg.base_cs = base.data_ptr(), base.stride(0), base.stride(1)
        return self._phi_prog

    def materialize_phi(self, model) -> torch.Tensor:
        """Evaluate every parametrization with torch (autograd-tracked), laid out as phi_layout says.  The product path
        uses the device program above; this is the general path for configurations outside it and the test reference."""
        fc = model.feature_clustering
        vals = {f"reg_weight.{i}": blk.sgu.reg_weight for i, blk in enumerate(model.ref_alt_reads_encoder.blocks)}
        vals.update({"rotation": model.pre_clustering_transform.rotation_ee.weight, "stdev_e": fc.nonartifact_stdev_e,
                     "dirs_ke": fc.artifact_directions_ke, "art_stdev_k": fc.artifact_stdev_k, "log_w_k": fc.log_cluster_weights_k,
                     "sigma_k": fc.artifact_emg.sigma_k, "lambda_k": fc.artifact_emg.lambda_k})
        names = [n for n, _, _ in self.phi_layout if n.startswith("bnfold.w.")]
        for name, (bn, lin) in zip(names, self.bn_folds):  # W' = W diag(s), b' = b + W t  (eval-mode BatchNorm1d folded into its Linear)
            s = bn.weight / torch.sqrt(bn.running_var + bn.eps)
            t = bn.bias - bn.running_mean * s
            vals[name] = lin.weight * s[None, :]
            vals["bnfold.b." + name[len("bnfold.w."):]] = lin.bias + lin.weight @ t
        parts, at = [], 0
        for name, off, n in self.phi_layout:
            assert off == at and vals[name].numel() == n, (name, off, at)
            parts.append(vals[name].reshape(-1))
            at += n
        phi = torch.cat(parts)
        if phi.numel() < self.desc.phi_size:
            phi = torch.cat([phi, phi.new_zeros(self.desc.phi_size - phi.numel())])
        return phi
