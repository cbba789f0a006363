"""Loss bookkeeping and the evaluation pass (reference training/loss_recorder.py, metrics/loss_metrics.py:31-60,
training/model_training.py:204-271).

`LossRecorder.record` adds a step's four loss vectors, weighted like the reference, into the three (totals, counts)
histograms over (source, label, variant type, ref-count bin, alt-count bin) with ONE launch (pmt_record_losses) instead of
8 index_add_; nothing is copied to the host until `averages()` / `mean_loss()` are asked for at the end of the epoch
(the reference's `put_on_cpu`).  `evaluate` is the no-grad pass of an evaluation epoch: forward kernels + losses +
recording per batch, no per-step `.item()`."""
from __future__ import annotations

import ctypes as C

import torch

from permutect_amd.data.datum import Data
from permutect_amd.engine import lib as L
from permutect_amd.enums import Label, Variation
from permutect_amd.training.downsampler import (COUNT_BIN_SKIP, MAX_ALT_COUNT, MAX_REF_COUNT, NUM_ALT_COUNT_BINS,
                                                NUM_REF_COUNT_BINS)

PRIMARY, ALT_COUNT, SOURCE = 0, 1, 2


class LossRecorder:
    def __init__(self, device, num_sources: int):
        self.device, self.num_sources = torch.device(device), num_sources
        self.shape = (num_sources, len(Label), len(Variation), NUM_REF_COUNT_BINS, NUM_ALT_COUNT_BINS)
        self.num_bins = 1
        for d in self.shape:
            self.num_bins *= d
        # [metric][totals | counts][bins]
        self.hist = torch.zeros(3, 2, self.num_bins, dtype=torch.float32, device=self.device)

    def record(self, output, losses, batch):
        a = L.PmtRecordArgs()
        a.num_variants, a.num_bins = batch.size(), self.num_bins
        a.num_variant_types, a.num_ref_bins, a.num_alt_bins = len(Variation), NUM_REF_COUNT_BINS, NUM_ALT_COUNT_BINS
        a.count_bin_skip, a.max_ref_count, a.max_alt_count = COUNT_BIN_SKIP, MAX_REF_COUNT, MAX_ALT_COUNT
        keep = []

        def col(field):  # int64 columns of the batch's integer tensor, a DownsampledBatch's int32 counts: read as they are
            t = batch.get(field)
            t = t if t.dtype in (torch.int64, torch.int32) else t.long()
            keep.append(t)
            return L.int_column(t)

        a.labels, a.variant_types, a.sources = col(Data.LABEL), col(Data.VARIANT_TYPE), col(Data.SOURCE)
        a.ref_counts, a.alt_counts = col(Data.REF_COUNT), col(Data.ALT_COUNT)
        vecs = [output.weights, output.source_weights, losses.supervised_losses_b, losses.unsupervised_losses_b,
                losses.alt_count_losses_b, losses.source_prediction_losses_b]
        vecs = [v.detach().contiguous().float() for v in vecs]
        keep.extend(vecs)
        (a.weights, a.source_weights, a.supervised_b, a.unsupervised_b, a.alt_count_b, a.source_b) = [v.data_ptr() for v in vecs]
        L.check(L.load().pmt_record_losses(C.byref(a), self.hist.data_ptr(), L.raw_stream()),
                "pmt_record_losses")

    # ---- read-out (one device -> host copy, at the end of the epoch) -----------------------------------------------------
    def totals(self, metric: int = PRIMARY) -> torch.Tensor:
        return self.hist[metric, 0].view(self.shape)

    def counts(self, metric: int = PRIMARY) -> torch.Tensor:
        return self.hist[metric, 1].view(self.shape)

    def averages(self, metric: int = PRIMARY) -> torch.Tensor:
        """reference loss_metrics.py:56-57"""
        return self.totals(metric) / (0.001 + self.counts(metric))

    def marginal_by_label(self, metric: int = PRIMARY):
        """(totals_l, counts_l): the histograms summed over every axis but the label's (reference
        metrics/loss_metrics.py:59-60 `get_marginal(BatchProperty.LABEL)` is their quotient)."""
        return self.totals(metric).sum(dim=(0, 2, 3, 4)), self.counts(metric).sum(dim=(0, 2, 3, 4))

    def mean_loss(self, metric: int = PRIMARY) -> float:
        """What the training loop feeds its LR scheduler and its checkpoint decisions (reference model_training.py:170,198):
        the UNWEIGHTED mean over labels of each label's average loss, `torch.mean(get_marginal(LABEL))` -- not the pooled
        total / count, which follows the most frequent label.  Deviation, on purpose: a label without any data makes the
        reference's quotient 0 / 0 = NaN (its scheduler and checkpoint then stop working for the whole run); here the
        mean runs over the labels that have data."""
        totals_l, counts_l = self.marginal_by_label(metric)
        present = counts_l > 0
        per_label = torch.where(present, totals_l / counts_l.clamp_min(1e-30), torch.zeros_like(totals_l))
        return float((per_label.sum() / present.sum().clamp_min(1)).item())

    def all_reduce(self, dist):
        """Sum the histograms over data-parallel ranks before the epoch-level decisions."""
        dist.all_reduce(self.hist, op=dist.ReduceOp.SUM)


@torch.inference_mode()
def evaluate(model, loader, num_sources: int = 1, balancer=None) -> LossRecorder:
    """One evaluation pass over `loader` (reference model_training.py:204-271's validation epoch without the plots)."""
    was_training = model.training
    model.train(False)
    rec = LossRecorder(model._device, num_sources)
    for batch in loader:
        out = model.compute_batch_output(batch, balancer)
        rec.record(out, model.compute_batch_losses(out, batch), batch)
    model.train(was_training)
    return rec


MIN_LOGIT, MAX_LOGIT, LOGIT_BIN_SKIP = -10, 10, 1  # reference data/count_binning.py:14-26 (logit 0 is a bin boundary)
NUM_LOGIT_BINS = (MAX_LOGIT - MIN_LOGIT) // LOGIT_BIN_SKIP + 1


def logit_bin_indices(logits: torch.Tensor) -> torch.Tensor:
    """reference data/count_binning.py:40-45"""
    return torch.div(torch.clamp(logits, min=MIN_LOGIT, max=MAX_LOGIT) - MIN_LOGIT, LOGIT_BIN_SKIP, rounding_mode="floor").long()


class EvaluationCounts:
    """What an evaluation pass leaves behind, on the device, read once at the end.

    `hist[epoch type]` is the reference's `EvaluationMetrics.accuracy_metrics_by_epoch_type[...]` (metrics/evaluation_metrics.py:
    49-66 -> AccuracyMetrics, metrics/loss_metrics.py:226-243): the weights of the LABELED variants tallied over (source, label,
    variant type, ref-count bin, alt-count bin, logit bin); its plots are out of scope, the tensor is what they are drawn from
    (pinned by tests/golden/evaluation_metrics.npz).  `stats[epoch type][label]` = (weight called not artifact, weight called
    artifact (logit > 0), weighted sum of logits) over ALL variants, for the log line.  Both live in ONE flat buffer so that data
    parallel ranks join them with one collective (`all_reduce`)."""

    def __init__(self, device, num_sources: int = 1):
        self.num_sources = num_sources
        self.shape = (num_sources, len(Label), len(Variation), NUM_REF_COUNT_BINS, NUM_ALT_COUNT_BINS, NUM_LOGIT_BINS)
        self._nhist = 1
        for d in self.shape:
            self._nhist *= d
        self.flat = torch.zeros(2 * self._nhist + 2 * len(Label) * 3, dtype=torch.float32, device=device)
        self.batches = 0

    @property
    def hist(self) -> torch.Tensor:       # [epoch type][S][L][V][R][A][G]
        return self.flat[: 2 * self._nhist].view(2, *self.shape)

    @property
    def stats(self) -> torch.Tensor:      # [epoch type][label][not artifact | artifact | logit sum]
        return self.flat[2 * self._nhist:].view(2, len(Label), 3)

    @property
    def counts(self) -> torch.Tensor:     # [epoch type][label][called artifact]
        return self.stats[:, :, :2]

    @property
    def logit_sums(self) -> torch.Tensor:
        return self.stats[:, :, 2]

    def record_batch(self, epoch_index: int, batch, logits: torch.Tensor, weights: torch.Tensor):
        if self.flat.is_cuda:  # one launch (pmt_record_evaluation) for the ~30 tensor ops below
            a = L.PmtEvalArgs()
            a.num_variants, a.epoch_index = batch.size(), epoch_index
            a.num_logit_bins, a.min_logit, a.max_logit, a.logit_bin_skip = NUM_LOGIT_BINS, MIN_LOGIT, MAX_LOGIT, LOGIT_BIN_SKIP
            g = a.bins
            g.num_sources, g.num_variant_types, g.num_ref_bins, g.num_alt_bins = self.num_sources, len(Variation), NUM_REF_COUNT_BINS, NUM_ALT_COUNT_BINS
            g.count_bin_skip, g.max_ref_count, g.max_alt_count = COUNT_BIN_SKIP, MAX_REF_COUNT, MAX_ALT_COUNT
            cols = [batch.get(f) for f in (Data.LABEL, Data.VARIANT_TYPE, Data.SOURCE, Data.REF_COUNT, Data.ALT_COUNT)]
            a.labels, a.variant_types, a.sources, a.ref_counts, a.alt_counts = [L.int_column(t) for t in cols]
            lg, wt = logits.detach().contiguous().float(), weights.detach().contiguous().float()
            a.logits_b, a.weights_b, a.nhist = lg.data_ptr(), wt.data_ptr(), self._nhist
            L.check(L.load().pmt_record_evaluation(C.byref(a), self.flat.data_ptr(), L.raw_stream(self.flat.device)), "pmt_record_evaluation")
            self._keep = (cols, lg, wt)
            self.batches += 1
            return
        self.record_batch_torch(epoch_index, batch, logits, weights)

    def record_batch_torch(self, epoch_index: int, batch, logits: torch.Tensor, weights: torch.Tensor):
        """the same tallies composed from torch ops (the CPU path, and the test reference of the fused launch)"""
        from permutect_amd.training.downsampler import flattened_slvra_index
        labels = batch.get(Data.LABEL).long()
        weights, logits = weights.float(), logits.float()
        called = (logits > 0).long()
        st = self.flat[2 * self._nhist:].view(2, -1)[epoch_index]
        st.index_add_(0, labels * 3 + called, weights)
        st.index_add_(0, labels * 3 + 2, weights * logits)
        # reference evaluation_metrics.py:58-66: unlabeled data are not tallied (weight x is_labeled)
        idx = flattened_slvra_index(batch) * NUM_LOGIT_BINS + logit_bin_indices(logits)
        self.flat[: 2 * self._nhist].view(2, -1)[epoch_index].index_add_(0, idx, weights * (labels != int(Label.UNLABELED)).float())
        self.batches += 1

    def accuracy(self, epoch_index: int) -> float:
        """weighted fraction of labeled variants on the right side of logit 0"""
        c = self.counts[epoch_index]
        right = c[int(Label.ARTIFACT), 1] + c[int(Label.VARIANT), 0]
        total = c[int(Label.ARTIFACT)].sum() + c[int(Label.VARIANT)].sum()
        return float((right / total.clamp_min(1e-12)).item())

    def all_reduce(self, dist):
        """every tally of the pass, summed over the data-parallel ranks in one collective"""
        self.flat = self.flat.clone()  # (filled under inference_mode: not updatable in place outside it)
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)


@torch.inference_mode()
def collect_evaluation_data(model, balancer, downsampler, train_loader, valid_loader, seed: int = 0,
                            fix_alt_gather: bool = False, passes: int = 3) -> EvaluationCounts:
    """The evaluation pass as the reference runs it after every validation epoch (training/model_training.py:204-228):
    over the training AND the validation loader, every parent batch downsampled `passes` = 3 times (fresh fractions each
    time), forward with the balancer's weights, results recorded.  Here each downsampling is two launches and each forward
    the fused filter kernel; nothing returns to the host inside the loop."""
    was_training = model.training
    model.train(False)
    ev = EvaluationCounts(model._device, num_sources=getattr(balancer, "num_sources", None) or getattr(downsampler, "num_sources", 1))
    step = seed * 7_368_787
    for epoch_index, loader in enumerate((train_loader, valid_loader)):
        if loader is None:
            continue
        for parent in loader:
            for _ in range(passes):
                step += 1
                batch = downsampler.downsample(parent, seed=step, fix_alt_gather=fix_alt_gather)
                out = model.compute_batch_output(batch, balancer)
                ev.record_batch(epoch_index, batch, out.logits_b, out.weights)
    model.train(was_training)
    return ev
