"""Runtime glue between torch tensors (device memory + streams only) and the C ABI.

`ReadSetEngine` owns the flat parameter space and the device descriptor of one ArtifactModel;
`ReadSetFunction` is the single torch.autograd.Function through which the read-set encoder + clustering head run:
forward = pmt_forward (one fused HIP launch), backward = pmt_backward (one fused HIP launch).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

import torch
from torch import Tensor

from permutect_amd.engine import lib as L
from permutect_amd.engine.plan import EnginePlan, ParamSpace


def _ptr(t: Optional[Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _stream() -> int:
    """the current HIP stream's handle.  torch.cuda.current_stream() builds a Stream object and resolves the device through four Python
    layers every time -- 10 us a call, eight calls per training step, a tenth of the host time of a B = 64 step (which is host-bound:
    scripts/small_step_profile.py); the raw getter is one C call."""
    return L.raw_stream()


class ReadSetEngine:
    def __init__(self, model, device: torch.device):
        if device.type != "cuda":
            raise L.PmtError("permutect_amd computes on an MI355X (ROCm device 'cuda') only; there is no CPU fallback. "
                             "Model construction, state_dict and save/load work on any device.")
        self.device = device
        self.space = ParamSpace(model, device)
        self.plan = EnginePlan(model, self.space, device)
        # the build of the library whose exact-width instances fit THIS model's tile counts (engine/instances.py): the default
        # library for the production shape, a per-shape build otherwise (found under permutect_amd/instances/, or built once)
        self.lib = self.plan.lib  # (engine/instances.py: library_for, chosen when the model was lowered)
        self.shape_id = int(self.lib.pmt_shape_id(C.byref(self.plan.desc)))  # which read-set instance runs (pmt_host.hip: pmt_shape_id)
        self._trigger = torch.zeros(1, dtype=torch.float32, device=device, requires_grad=True)  # see RowsMlpFunction
        self._no_trigger = torch.zeros(1, dtype=torch.float32, device=device)
        self._cnn_ws = None
        self._rows_ws = None
        # `packed` / phi are functions of the parameters: params_key() changes whenever theta may have changed -- torch's version
        # counter sees every in-place torch op on theta or on a parameter view (load_state_dict, checkpoint restore, a torch
        # optimizer on the calibration parameters), `params_changed()` is called by whatever writes theta through a raw pointer
        # (the fused optimizer kernel, a captured-graph replay, a collective on the flat buffer)
        self.join_layered = os.environ.get("PMT_LAYERED_JOIN", "1") != "0"
        self.dropout_seed = 0  # this step's dropout masks (draw_dropout_seed; 0 = none)
        # ONE persistent fault word for every joined launch of this engine (PmtBatch.join_fault): a launch whose bounded wait for
        # another workgroup gave up stores 1 there and its numbers are wrong.  Read by check_join_fault() wherever the callers
        # synchronise anyway: end of a training / evaluation epoch, end of a filtering pass, bench.py, the tests.
        self.join_fault = torch.zeros(1, dtype=torch.int32, device=device)
        self._param_epoch = 0
        self.packed_for = None  # (params_key, phi) the packed weights were built from, under no_grad only
        self.timers = None  # bench.py sets {'pmt_forward': [], 'pmt_backward': []} to collect (start, end) HIP events
        self.timer_stride, self._timer_calls = 1, {}
        self.grad_hook = None  # data parallel: BucketedGradAllReduce, told when the early gradient bucket is final

    def _event_start(self, name: str):
        if self.timers is None:
            return None
        # timer_stride k: around every k-th launch of a kernel only (an event pair costs the stream ~10 us: 0.6 % of a training step
        # for the two read-set kernels)
        n = self._timer_calls.get(name, 0)
        self._timer_calls[name] = n + 1
        if n % max(1, int(self.timer_stride)) != 0:
            return None
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()  # on torch's current stream, which is the stream the kernel is launched on (_stream())
        return ev

    def _event_stop(self, name: str, start):
        if start is not None:
            end = torch.cuda.Event(enable_timing=True)
            end.record()
            self.timers[name].append((start, end))

    def check_join_fault(self):
        """Synchronises and raises if ANY joined launch since the last check gave up waiting for another workgroup (its results,
        and everything computed from them since, are wrong).  The word is cleared so that a caller may catch and carry on."""
        word = int(self.join_fault.item())
        if word != 0:
            self.join_fault.zero_()
            what = []
            if word & 1:
                what.append("a joined layered launch timed out waiting for the other workgroups of a split read set "
                            "(PMT_LAYERED_JOIN=0 runs the layered launches instead)")
            if word & 2:
                what.append("an activation left the range of the forward's f16 operand pieces (|x| > 65504 where the residual stream enters the "
                            "reducer or its output the rotation): those operands saturate (PMT_SHAPE=bf16x3 runs the bf16-piece forward, which has no such limit)")
            if word & 4:
                what.append("the device-side group planner of a downsampled batch ran out of its group capacity (pmt_plan_groups_device)")
            raise L.PmtError("; ".join(what) + ": the logits / gradients computed since the last check are wrong")

    # ---- parameters -------------------------------------------------------------------------------------------------
    def params_changed(self):
        self._param_epoch += 1

    def params_key(self):
        # (a Parameter re-bound with `p.data = view` keeps its OWN version counter: in-place writes through a parameter --
        #  load_state_dict, a torch optimizer -- show up there, writes to the flat buffer on theta's)
        try:
            stats = sum(bn.running_mean._version + bn.running_var._version  # (folded into phi / the CNN's weights: buffers, not leaves)
                        for bn in [f[0] for f in self.plan.bn_folds] + [f[0] for f in self.plan.cnn_bn_folds])
            return (self._param_epoch, self.space.theta._version, sum(p._version for p in self.space.params), stats)
        except RuntimeError:  # inference tensors (a model built under torch.inference_mode) carry no version counter:
            return None        # nothing can be proved unchanged, so nothing is reused

    def pack(self, phi: Tensor):
        d = self.plan
        L.check(self.lib.pmt_pack_params(C.byref(d.desc), d.desc_dev.data_ptr(), self.space.theta.data_ptr(),
                                         phi.data_ptr(), d.packed.data_ptr(), _stream()), "pmt_pack_params")

    def cnn_params(self) -> Tuple[Tensor, Tensor]:
        """(theta, packed) as the haplotype-CNN kernels read them.  A stack with `batch_norm` tokens (eval mode only) reads a COPY of
        the flat buffer in which the BatchNorms are folded into their neighbouring convolutions / linear, and fragments packed from
        that copy (plan.folded_cnn_theta); rebuilt when the parameters or the running statistics change."""
        if not self.plan.cnn_bn_folds:
            return self.space.theta, self.plan.packed
        key = self.params_key()
        if key is None or getattr(self, "_cnn_fold_key", None) != key:
            theta_f = self.plan.folded_cnn_theta(self.space.theta)
            packed_f = torch.zeros_like(self.plan.packed)
            phi0 = torch.zeros(max(self.plan.desc.phi_size, 4), dtype=torch.float32, device=self.device)  # (the CNN's linears read theta only)
            L.check(self.lib.pmt_pack_params(C.byref(self.plan.desc), self.plan.desc_dev.data_ptr(), theta_f.data_ptr(), phi0.data_ptr(),
                                             packed_f.data_ptr(), _stream()), "pmt_pack_params")
            self._cnn_fold, self._cnn_fold_key = (theta_f, packed_f), key
        return self._cnn_fold

    def cnn_workspace(self) -> Optional[Tensor]:
        """private rows for the haplotype CNN's weight-gradient sums (pmt_cnn_backward: workspace); PMT_CNN_WORKSPACE=0: atomics"""
        if self._cnn_ws is None:
            n = self.lib.pmt_cnn_workspace_floats(C.byref(self.plan.desc)) if os.environ.get("PMT_CNN_WORKSPACE", "1") != "0" else 0
            self._cnn_ws = torch.empty(n, dtype=torch.float32, device=self.device) if n > 0 else False
        return self._cnn_ws if self._cnn_ws is not False else None

    def rows_workspace(self) -> Optional[Tensor]:
        """zeroed gradient replicas shared by the three row MLPs' backward (pmt_rows_backward: workspace; every call leaves them
        zero); PMT_ROWS_WORKSPACE=0: every workgroup adds to the gradient buffer itself"""
        if self._rows_ws is None:
            n = 0
            if os.environ.get("PMT_ROWS_WORKSPACE", "1") != "0":
                n = max(self.lib.pmt_rows_workspace_floats(C.byref(self.plan.desc), w) if self.plan.desc.row_mlp[w].n_ops > 0 else 0
                        for w in range(3))
            self._rows_ws = torch.zeros(n, dtype=torch.float32, device=self.device) if n > 0 else False
        return self._rows_ws if self._rows_ws is not False else None

    # ---- batch views --------------------------------------------------------------------------------------------------
    def offsets(self, batch) -> Tuple[Tensor, Tensor]:
        if getattr(batch, "_offsets", None) is None:
            ref_c, alt_c, elem, stride = batch.device_counts()
            b = batch.size()
            ref_off = torch.empty(b + 1, dtype=torch.int32, device=self.device)
            alt_off = torch.empty(b + 1, dtype=torch.int32, device=self.device)
            L.check(self.lib.pmt_scan_counts(ref_c.data_ptr(), alt_c.data_ptr(), elem, stride, b, ref_off.data_ptr(),
                                             alt_off.data_ptr(), _stream()), "pmt_scan_counts")
            batch._offsets = (ref_off, alt_off)
        return batch._offsets

    @property
    def trigger(self) -> Tensor:
        """The 1-element leaf that makes autograd call the backward of the engine's Functions (their parameters live in the flat
        buffer, not among their inputs).  A Function's `ctx.needs_input_grad` reports its inputs' requires_grad flags WHATEVER
        the grad mode, so under no_grad / inference_mode a tensor without grad is handed out instead: the forward kernels then
        keep nothing for a backward that cannot follow (the haplotype CNN's records alone are 145 MB per 65 536 variants)."""
        return self._trigger if torch.is_grad_enabled() else self._no_trigger

    def draw_dropout_seed(self, training: bool) -> int:
        """The seed of this step's dropout masks (0 = none: eval mode or dropout_p = 0).  Drawn from torch's CPU generator, so
        torch.manual_seed replays a training run mask for mask; every MLP of the step uses it (the masks differ by linear
        and row, pmt_dropout.hpp), the forward keeps it for its backward."""
        self.dropout_seed = int(torch.randint(1, 2 ** 62, (1,)).item()) if (training and self.plan.desc.dropout_p > 0) else 0
        return self.dropout_seed

    def batch_view(self, batch, variant_embed: Tensor, allow_split: bool = True, dropout_seed: int = 0):
        plan = batch.plan(allow_split=allow_split, fault=self.join_fault)
        gs, gt, span = plan.on(self.device)
        ref_off, alt_off = self.offsets(batch)
        reads, fmt, row_bytes, index = batch.read_rows()
        if reads.device != self.device:
            raise L.PmtError("batch tensors must be on the model's device (call batch.copy_to(device, dtype))")
        bv = L.PmtBatch()
        bv.num_variants, bv.num_groups = batch.size(), plan.num_groups
        bv.read_format, bv.read_row_bytes = fmt, row_bytes
        bv.reads, bv.read_index = reads.data_ptr(), _ptr(index)
        bv.ref_offsets, bv.alt_offsets = ref_off.data_ptr(), alt_off.data_ptr()
        bv.variant_embed = variant_embed.data_ptr()
        bv.group_start, bv.group_tile_base = gs.data_ptr(), gt.data_ptr()
        bv.total_tiles = plan.total_tiles
        bv.debug_flags = self.plan.debug_flags.data_ptr()
        bv.group_span = _ptr(span)
        bv.num_groups_dev = _ptr(getattr(plan, "num_groups_dev", None))  # (tests/graph_capture.py: a plan sized for a capacity)
        # split read sets: with the number of groups per variant the layered entry points run ONE joined launch each way
        # (PMT_LAYERED_JOIN=0: num_blocks + 1 launches with the activations parked in between; the parity tests run both)
        sets = plan.set_groups_on(self.device) if (getattr(plan, "set_groups", None) is not None and self.join_layered) else None
        bv.set_groups = _ptr(sets)
        bv.dropout_seed = dropout_seed
        bv.join_fault = self.join_fault.data_ptr()
        keep = (gs, gt, span, ref_off, alt_off, reads, index, variant_embed, sets)
        return bv, keep, plan

    # ---- passes -----------------------------------------------------------------------------------------------------
    def forward(self, batch, phi: Tensor, variant_embed: Tensor, train: bool, dropout_seed: int = 0):
        d = self.plan.desc
        b, k, e = batch.size(), d.num_clusters, d.feature_dim
        variant_embed = variant_embed.contiguous().float()
        assert variant_embed.shape == (b, d.variant_embed_dim), (variant_embed.shape, d.variant_embed_dim)
        phi = phi.contiguous()
        # read sets of any size: beyond one workgroup they are split over several groups (layered execution)
        bv, keep, plan = self.batch_view(batch, variant_embed, dropout_seed=dropout_seed)
        dev = self.device
        logits_b = torch.empty(b, dtype=torch.float32, device=dev)
        logits_bk = torch.empty(b, k + 2, dtype=torch.float32, device=dev)
        feats = torch.empty(b, e, dtype=torch.float32, device=dev)
        ref_feats = torch.empty(b, e, dtype=torch.float32, device=dev)
        out = L.PmtOutputs(logits_b.data_ptr(), logits_bk.data_ptr(), feats.data_ptr(), ref_feats.data_ptr())
        stash = None
        if train:
            nbytes = self.lib.pmt_stash_bytes(C.byref(d), plan.total_tiles, b)
            stash = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
        ev = self._event_start("pmt_forward")
        if plan.layered:
            scratch = torch.empty(self.lib.pmt_layered_scratch_floats(C.byref(d), plan.total_tiles, b), dtype=torch.float32, device=dev)
            L.check(self.lib.pmt_forward_layered(C.byref(d), self.plan.desc_dev.data_ptr(), self.space.theta.data_ptr(),
                                                 phi.data_ptr(), self.plan.packed.data_ptr(), C.byref(bv), C.byref(out),
                                                 _ptr(stash), scratch.data_ptr(), _stream()), "pmt_forward_layered")
        else:
            L.check(self.lib.pmt_forward(C.byref(d), self.plan.desc_dev.data_ptr(), self.space.theta.data_ptr(),
                                         phi.data_ptr(), self.plan.packed.data_ptr(), C.byref(bv), C.byref(out),
                                         _ptr(stash), _stream()), "pmt_forward")
        self._event_stop("pmt_forward", ev)
        return (logits_b, logits_bk, feats, ref_feats), stash, variant_embed, phi

    def backward(self, batch, phi: Tensor, variant_embed: Tensor, stash: Tensor, outs, grads, dropout_seed: int = 0):
        d = self.plan.desc
        bv, keep, plan = self.batch_view(batch, variant_embed, dropout_seed=dropout_seed)
        g = [None if t is None else t.contiguous().float() for t in grads]
        dout = L.PmtOutputGrads(_ptr(g[0]), _ptr(g[1]), _ptr(g[2]), _ptr(g[3]))
        out = L.PmtOutputs(*[t.data_ptr() for t in outs])
        gphi = torch.zeros(d.phi_size, dtype=torch.float32, device=self.device)
        # (the one-launch backward writes every row of d(variant embedding); the layered one adds into it from several groups)
        gvar = torch.zeros_like(variant_embed) if plan.layered else torch.empty_like(variant_embed)
        ev = self._event_start("pmt_backward")
        if plan.layered:
            n = self.lib.pmt_layered_backward_scratch_floats(C.byref(d), plan.total_tiles, batch.size())
            scratch = torch.empty(n, dtype=torch.float32, device=self.device)
            L.check(self.lib.pmt_backward_layered(C.byref(d), self.plan.desc_dev.data_ptr(), self.space.theta.data_ptr(),
                                                  phi.data_ptr(), self.plan.packed.data_ptr(), C.byref(bv), C.byref(out),
                                                  C.byref(dout), stash.data_ptr(), scratch.data_ptr(),
                                                  self.space.gtheta.data_ptr(), gphi.data_ptr(), gvar.data_ptr(),
                                                  self.plan.grad_partials.data_ptr(), self.plan.partial_rows, _stream()),
                    "pmt_backward_layered")
        else:
            L.check(self.lib.pmt_backward(C.byref(d), self.plan.desc_dev.data_ptr(), self.space.theta.data_ptr(),
                                          phi.data_ptr(), self.plan.packed.data_ptr(), C.byref(bv), C.byref(out),
                                          C.byref(dout), stash.data_ptr(), self.space.gtheta.data_ptr(), gphi.data_ptr(),
                                          gvar.data_ptr(), self.plan.grad_partials.data_ptr(), self.plan.partial_rows, _stream()),
                    "pmt_backward")
        self._event_stop("pmt_backward", ev)
        if self.grad_hook is not None:  # every leaf in [0, late_start) has its final gradient at this point of the stream
            self.grad_hook.early(self.space.gtheta, self.space.late_start)
        return gphi, gvar


class LossesFunction(torch.autograd.Function):
    """(logits_b, logits_bk, alt-count prediction, source prediction) -> the five per-variant loss vectors of
    reference artifact_model.py:267-325, one launch forward and one backward (pmt_losses_*)."""

    @staticmethod
    def _args(engine, logits_b, logits_bk, alt_raw, source_logits, labels, alt_counts, sources, weights, source_weights):
        from permutect_amd import constants
        a = L.PmtLossArgs()
        a.num_variants, a.num_clusters = logits_b.shape[0], logits_bk.shape[1] - 2
        a.num_sources = 1 if source_logits is None else source_logits.shape[1]
        a.max_outlier_logit, a.max_alt_count = constants.MAX_OUTLIER_LOGIT, float(constants.MAX_ALT_COUNT)
        a.logits_b, a.logits_bk, a.alt_count_raw = logits_b.data_ptr(), logits_bk.data_ptr(), alt_raw.data_ptr()
        a.source_logits = _ptr(source_logits)
        a.labels, a.label_stride = labels.data_ptr(), labels.stride(0)
        a.alt_counts, a.alt_count_stride = alt_counts.data_ptr(), alt_counts.stride(0)
        a.sources, a.source_stride = _ptr(sources), (0 if sources is None else sources.stride(0))
        a.weights, a.source_weights = weights.data_ptr(), source_weights.data_ptr()
        return a

    @staticmethod
    def forward(ctx, engine, logits_b, logits_bk, alt_raw, source_logits, labels, alt_counts, sources, weights, source_weights):
        alt_shape = alt_raw.shape
        logits_b, logits_bk, alt_raw = logits_b.contiguous(), logits_bk.contiguous(), alt_raw.contiguous().view(-1)
        source_logits = None if source_logits is None else source_logits.contiguous()
        weights, source_weights = weights.contiguous().float(), source_weights.contiguous().float()
        assert labels.dtype == torch.int64 and alt_counts.dtype == torch.int64 and (sources is None or sources.dtype == torch.int64)
        b = logits_b.shape[0]
        outs = [torch.empty(b, dtype=torch.float32, device=logits_b.device) for _ in range(5)]
        a = LossesFunction._args(engine, logits_b, logits_bk, alt_raw, source_logits, labels, alt_counts, sources, weights, source_weights)
        o = L.PmtLossOutputs(*[t.data_ptr() for t in outs])
        L.check(engine.lib.pmt_losses_forward(C.byref(a), C.byref(o), _stream()), "pmt_losses_forward")
        ctx.engine = engine
        ctx.alt_shape = alt_shape
        ctx.set_materialize_grads(False)  # loss vectors nobody differentiates arrive as None (a null pointer), not as zero fills
        ctx.save_for_backward(logits_b, logits_bk, alt_raw, labels, alt_counts, weights, source_weights,
                              *(() if source_logits is None else (source_logits, sources)))
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        saved = ctx.saved_tensors
        logits_b, logits_bk, alt_raw, labels, alt_counts, weights, source_weights = saved[:7]
        source_logits, sources = (saved[7], saved[8]) if len(saved) > 7 else (None, None)
        grads = [None if g is None else g.contiguous().float() for g in grads]
        a = LossesFunction._args(ctx.engine, logits_b, logits_bk, alt_raw, source_logits, labels, alt_counts, sources, weights, source_weights)
        g = L.PmtLossOutputs(*[_ptr(t) for t in grads])
        d_b, d_bk, d_alt = torch.empty_like(logits_b), torch.empty_like(logits_bk), torch.empty_like(alt_raw)
        d_src = None if source_logits is None else torch.empty_like(source_logits)
        d = L.PmtLossInputGrads(d_b.data_ptr(), d_bk.data_ptr(), d_alt.data_ptr(), _ptr(d_src))
        L.check(ctx.engine.lib.pmt_losses_backward(C.byref(a), C.byref(g), C.byref(d), _stream()), "pmt_losses_backward")
        return None, d_b, d_bk, d_alt.view(ctx.alt_shape), d_src, None, None, None, None, None


class PhiFunction(torch.autograd.Function):
    """theta -> phi (the materialised parametrizations) in one launch; the backward adds J^T d(phi) into the flat gradient
    buffer that the `.original` leaves' `.grad` alias (one launch).  `trigger` is the engine's dummy leaf."""

    @staticmethod
    def forward(ctx, engine: "ReadSetEngine", prog, trigger: Tensor):
        phi = torch.zeros(engine.plan.desc.phi_size, dtype=torch.float32, device=engine.device)
        L.check(engine.lib.pmt_phi_forward(C.byref(prog), engine.space.theta.data_ptr(), phi.data_ptr(), _stream()), "pmt_phi_forward")
        ctx.engine, ctx.prog = engine, prog
        ctx.save_for_backward(phi)
        return phi

    @staticmethod
    def backward(ctx, gphi: Tensor):
        (phi,) = ctx.saved_tensors
        eng = ctx.engine
        gphi = gphi.contiguous()
        L.check(eng.lib.pmt_phi_backward(C.byref(ctx.prog), eng.space.theta.data_ptr(), phi.data_ptr(), gphi.data_ptr(),
                                         eng.space.gtheta.data_ptr(), _stream()), "pmt_phi_backward")
        return None, None, None


class ReadSetFunction(torch.autograd.Function):
    """(phi, variant_embed) -> (logits_b, logits_bk, features_be, ref_features_be).

    Gradients w.r.t. the direct (un-parametrized) leaves are accumulated by the backward kernel straight into the flat
    gradient buffer that `param.grad` views alias; gradients w.r.t. phi and the per-variant embedding are returned to
    autograd."""

    @staticmethod
    def forward(ctx, engine: ReadSetEngine, batch, phi: Tensor, variant_embed: Tensor):
        train = bool(ctx.needs_input_grad[2] or ctx.needs_input_grad[3])  # (phi / variant_embed carry no grad under no_grad)
        ctx.dropout_seed = engine.dropout_seed  # (set by the model at the start of the step; 0 in eval mode)
        outs, stash, ve, ph = engine.forward(batch, phi.detach(), variant_embed.detach(), train, ctx.dropout_seed)
        ctx.engine, ctx.batch, ctx.train = engine, batch, train
        ctx.set_materialize_grads(False)  # outputs the loss does not use (ref_features_be) arrive as None = a null pointer
        if train:
            ctx.save_for_backward(ph, ve, stash, *outs)
        return outs

    @staticmethod
    def backward(ctx, d_logits_b, d_logits_bk, d_feats, d_ref_feats):
        if not ctx.train:
            return None, None, None, None
        phi, ve, stash, *outs = ctx.saved_tensors
        ctx.engine.space.bind_grads(quick=True)
        gphi, gvar = ctx.engine.backward(ctx.batch, phi, ve, stash, outs, (d_logits_b, d_logits_bk, d_feats, d_ref_feats),
                                         ctx.dropout_seed)
        return None, None, gphi, gvar


class HaplotypeCnnFunction(torch.autograd.Function):
    """haplotypes int64 [B, H] -> haplotype embedding [B, E_h] (pmt_cnn_forward / pmt_cnn_backward).  Replaces
    Batch.get_one_hot_haplotypes_bcs + DNASequenceConvolution.forward (reference artifact_model.py:245).  The CNN's
    parameters get their gradients by atomics into the flat buffer; `trigger` makes autograd call backward."""

    @staticmethod
    def forward(ctx, engine: ReadSetEngine, haplotypes: Tensor, trigger: Tensor):
        d = engine.plan.desc
        hap = haplotypes if haplotypes.dtype == torch.int64 else haplotypes.long()
        assert hap.stride(-1) == 1 and hap.shape[1] == 2 * d.cnn.seq_len
        n = hap.shape[0]
        out = torch.empty(n, d.cnn.out_dim, dtype=torch.float32, device=engine.device)
        train = bool(ctx.needs_input_grad[2])  # (callers pass engine.mode_trigger(): no grad under no_grad / inference_mode)
        # (PMT_CNN_STASH=0: let the backward recompute the layer outputs instead; the parity tests cover both)
        per = engine.lib.pmt_cnn_stash_floats(C.byref(d)) if train and os.environ.get("PMT_CNN_STASH", "1") != "0" else 0
        stash = torch.empty(n * per, dtype=torch.float32, device=engine.device) if per > 0 and n > 0 else None
        if train and engine.plan.cnn_bn_folds:
            raise NotImplementedError("a haplotype CNN with batch_norm tokens runs under no_grad / inference_mode only (its BatchNorms are folded "
                                      "into the neighbouring layers' weights); a backward through it is not built")
        cnn_theta, cnn_packed = engine.cnn_params()
        L.check(engine.lib.pmt_cnn_forward(C.byref(d), engine.plan.desc_dev.data_ptr(), cnn_theta.data_ptr(),
                                           cnn_packed.data_ptr(), hap.data_ptr(), hap.stride(0), n, out.data_ptr(), out.stride(0),
                                           _ptr(stash), _stream()),
                "pmt_cnn_forward")
        ctx.engine, ctx.train, ctx.stash = engine, train, stash
        ctx.hap = hap  # integer tensor: kept on ctx (save_for_backward is for differentiable tensors' bookkeeping)
        return out

    @staticmethod
    def backward(ctx, d_out):
        if not ctx.train:
            return None, None, None
        eng, d, hap = ctx.engine, ctx.engine.plan.desc, ctx.hap
        eng.space.bind_grads(quick=True)
        if d_out.dtype != torch.float32 or d_out.stride(-1) != 1:
            d_out = d_out.float().contiguous()
        ws = eng.cnn_workspace()
        L.check(eng.lib.pmt_cnn_backward(C.byref(d), eng.plan.desc_dev.data_ptr(), eng.space.theta.data_ptr(),
                                         eng.plan.packed.data_ptr(), hap.data_ptr(), hap.stride(0), hap.shape[0], d_out.data_ptr(), d_out.stride(0),
                                         _ptr(ctx.stash), eng.space.gtheta.data_ptr(), _ptr(ws), 0 if ws is None else ws.numel(), _stream()),
                "pmt_cnn_backward")
        ctx.stash = None
        return None, None, None  # (the trigger only makes autograd call this node; it needs no gradient of its own)


class VariantEmbedFunction(torch.autograd.Function):
    """(info vectors [B, I], haplotypes int64 [B, H]) -> the per-variant embedding [B, E_info + E_hap] (reference
    artifact_model.py:244-246: hstack of info_embedding(info) and haplotypes_cnn(one-hot)).  The info MLP (pmt_rows_forward) and the
    haplotype CNN (pmt_cnn_forward) write their column blocks of ONE buffer (both entry points take an output stride), and their
    backward kernels read the column blocks of its gradient in place: no concatenation launch either way."""

    @staticmethod
    def forward(ctx, engine: ReadSetEngine, info: Tensor, haplotypes: Tensor, trigger: Tensor):
        lib, d = engine.lib, engine.plan.desc
        mlp = d.row_mlp[L.ROWS_INFO]
        n = info.shape[0]
        x = info.detach()
        if x.dtype != torch.float32 or x.stride(-1) != 1:
            x = x.float().contiguous()
        hap = haplotypes if haplotypes.dtype == torch.int64 else haplotypes.long()
        assert x.shape[1] == mlp.in_dim and hap.stride(-1) == 1 and hap.shape[1] == 2 * d.cnn.seq_len
        e_info, e_hap = mlp.out_dim, d.cnn.out_dim
        ve = torch.empty(n, e_info + e_hap, dtype=torch.float32, device=engine.device)
        train = bool(ctx.needs_input_grad[3])
        rows_stash = cnn_stash = None
        if train:
            rows_stash = torch.empty(lib.pmt_rows_stash_bytes(C.byref(d), L.ROWS_INFO, n) // 4, dtype=torch.float32, device=engine.device)
            per = lib.pmt_cnn_stash_floats(C.byref(d)) if os.environ.get("PMT_CNN_STASH", "1") != "0" else 0
            cnn_stash = torch.empty(n * per, dtype=torch.float32, device=engine.device) if per > 0 and n > 0 else None
        L.check(lib.pmt_rows_forward(C.byref(d), engine.plan.desc_dev.data_ptr(), L.ROWS_INFO, engine.space.theta.data_ptr(),
                                     engine.plan.packed.data_ptr(), x.data_ptr(), x.stride(0), n, ve.data_ptr(), ve.stride(0),
                                     _ptr(rows_stash), engine.dropout_seed, _stream()), "pmt_rows_forward")
        if train and engine.plan.cnn_bn_folds:
            raise NotImplementedError("a haplotype CNN with batch_norm tokens runs under no_grad / inference_mode only (its BatchNorms are folded "
                                      "into the neighbouring layers' weights); a backward through it is not built")
        cnn_theta, cnn_packed = engine.cnn_params()
        L.check(lib.pmt_cnn_forward(C.byref(d), engine.plan.desc_dev.data_ptr(), cnn_theta.data_ptr(),
                                    cnn_packed.data_ptr(), hap.data_ptr(), hap.stride(0), n, ve.data_ptr() + 4 * e_info, ve.stride(0),
                                    _ptr(cnn_stash), _stream()), "pmt_cnn_forward")
        ctx.engine, ctx.train, ctx.dropout_seed, ctx.e_info = engine, train, engine.dropout_seed, e_info
        ctx.hap, ctx.cnn_stash = hap, cnn_stash
        if train:
            ctx.save_for_backward(x, rows_stash)
        return ve

    @staticmethod
    def backward(ctx, d_ve):
        if not ctx.train:
            return None, None, None, None
        x, rows_stash = ctx.saved_tensors
        eng, d, hap = ctx.engine, ctx.engine.plan.desc, ctx.hap
        eng.space.bind_grads(quick=True)
        if d_ve.dtype != torch.float32 or d_ve.stride(-1) != 1:
            d_ve = d_ve.float().contiguous()
        n = x.shape[0]
        ws = eng.rows_workspace()
        L.check(eng.lib.pmt_rows_backward(C.byref(d), eng.plan.desc_dev.data_ptr(), L.ROWS_INFO, eng.space.theta.data_ptr(),
                                          eng.plan.packed.data_ptr(), x.data_ptr(), x.stride(0), n, d_ve.data_ptr(), d_ve.stride(0),
                                          rows_stash.data_ptr(), eng.space.gtheta.data_ptr(), None, 0, 1.0, _ptr(ws),
                                          0 if ws is None else ws.numel(), ctx.dropout_seed, _stream()), "pmt_rows_backward")
        cws = eng.cnn_workspace()
        L.check(eng.lib.pmt_cnn_backward(C.byref(d), eng.plan.desc_dev.data_ptr(), eng.space.theta.data_ptr(),
                                         eng.plan.packed.data_ptr(), hap.data_ptr(), hap.stride(0), n, d_ve.data_ptr() + 4 * ctx.e_info,
                                         d_ve.stride(0), _ptr(ctx.cnn_stash), eng.space.gtheta.data_ptr(), _ptr(cws),
                                         0 if cws is None else cws.numel(), _stream()), "pmt_cnn_backward")
        ctx.cnn_stash = None
        return None, None, None, None


class RowsMlpFunction(torch.autograd.Function):
    """One of the per-variant row MLPs (pmt_rows_forward / pmt_rows_backward): info embedding, alt-count adversary,
    source adversary.  `x` is [N, in_dim] fp32 (any row stride); the result is [N, out_dim].  `trigger` is the engine's
    1-element leaf that makes autograd call backward even when x itself needs no gradient (the MLP's own parameters
    live in the flat buffer and receive their gradients by atomics inside the kernel).  `reverse_alpha`: None = d(x) passes
    unchanged; a number (0.0 included: the source adversary's strength in epoch 1, reference model_training.py:92) applies
    the reference's gradient reversal d(x) <- -alpha * d(x) (gradient_reversal/functional.py:18-22)."""

    @staticmethod
    def forward(ctx, engine: ReadSetEngine, which: int, x: Tensor, trigger: Tensor, reverse_alpha: Optional[float]):
        lib, d = engine.lib, engine.plan.desc
        mlp = d.row_mlp[which]
        n = x.shape[0]
        x = x.detach()
        if x.dtype != torch.float32 or x.stride(-1) != 1:
            x = x.float().contiguous()
        assert x.shape[1] == mlp.in_dim
        out = torch.empty(n, mlp.out_dim, dtype=torch.float32, device=engine.device)
        train = bool(ctx.needs_input_grad[2] or ctx.needs_input_grad[3])
        stash = None
        if train:
            stash = torch.empty(lib.pmt_rows_stash_bytes(C.byref(d), which, n) // 4, dtype=torch.float32, device=engine.device)
        L.check(lib.pmt_rows_forward(C.byref(d), engine.plan.desc_dev.data_ptr(), which, engine.space.theta.data_ptr(),
                                     engine.plan.packed.data_ptr(), x.data_ptr(), x.stride(0), n, out.data_ptr(),
                                     out.stride(0), _ptr(stash), engine.dropout_seed, _stream()), "pmt_rows_forward")
        ctx.dropout_seed = engine.dropout_seed
        ctx.engine, ctx.which, ctx.train, ctx.alpha = engine, which, train, reverse_alpha
        ctx.x_needs_grad = bool(ctx.needs_input_grad[2])
        if train:
            ctx.save_for_backward(x, stash)
        return out

    @staticmethod
    def backward(ctx, d_out):
        if not ctx.train:
            return None, None, None, None, None
        x, stash = ctx.saved_tensors
        eng, d = ctx.engine, ctx.engine.plan.desc
        eng.space.bind_grads(quick=True)
        if d_out.dtype != torch.float32 or d_out.stride(-1) != 1:
            d_out = d_out.float().contiguous()
        n = x.shape[0]
        d_in = torch.empty_like(x) if ctx.x_needs_grad else None
        scale = 1.0 if ctx.alpha is None else -float(ctx.alpha)
        ws = eng.rows_workspace()
        L.check(eng.lib.pmt_rows_backward(C.byref(d), eng.plan.desc_dev.data_ptr(), ctx.which, eng.space.theta.data_ptr(),
                                          eng.plan.packed.data_ptr(), x.data_ptr(), x.stride(0), n, d_out.data_ptr(),
                                          d_out.stride(0), stash.data_ptr(), eng.space.gtheta.data_ptr(), _ptr(d_in),
                                          d_in.stride(0) if d_in is not None else 0, scale, _ptr(ws), 0 if ws is None else ws.numel(),
                                          ctx.dropout_seed, _stream()), "pmt_rows_backward")
        return None, None, d_in, None, None
