"""Label / source balancing weights (reference permutect/training/balancer.py:27-119): running counts over
(source, label, variant type, ref-count bin, alt-count bin), pseudo-counts for unlabeled data from the model's artifact
probability, and weights recomputed every DATA_BEFORE_RECOMPUTE variants.  O(B) index arithmetic on small tensors, on
the device with torch ops and no host sync (replica-local under data parallelism)."""
from __future__ import annotations

import math

import torch
from torch import Tensor

from permutect_amd.data.batch import Batch
from permutect_amd.data.datum import Data
from permutect_amd.enums import Label, Variation
from permutect_amd.training.downsampler import NUM_ALT_COUNT_BINS, NUM_REF_COUNT_BINS, flattened_slvra_index


class Balancer:
    ATTENUATION_PER_DATUM = 0.99999
    DATA_BEFORE_RECOMPUTE = 10000

    def __init__(self, num_sources: int, device):
        self.device, self.num_sources = torch.device(device), num_sources
        shape = (num_sources, len(Label), len(Variation), NUM_REF_COUNT_BINS, NUM_ALT_COUNT_BINS)
        self.counts_slvra = torch.zeros(shape, device=self.device)
        self.pseudo_counts_slvra = torch.zeros(shape, device=self.device)
        self.weights_slvra = torch.ones(shape, device=self.device)
        self.unlabeled_weights_slvra = torch.ones(shape, device=self.device)
        self.source_weights_s = torch.ones(num_sources, device=self.device)
        self.count_since_last_recomputation = 0
        self._label_stride = len(Variation) * NUM_REF_COUNT_BINS * NUM_ALT_COUNT_BINS

    def weights_from_logits(self, batch: Batch, logits_b: Tensor):
        """`process_batch_and_compute_weights(batch, sigmoid(logits_b))` and the product BatchOutput.source_weights = weights *
        source weight (reference artifact_model.py:285-288) in TWO launches (pmt_balance_step) instead of ~60: counts and pseudo-counts
        added, the tables re-derived when due, the batch's weights looked up.  Returns (weights_b, weights_b * source_weights_b).
        The tables are double-buffered: a recomputation reads one set and writes the other."""
        import ctypes as C

        from permutect_amd.engine import lib as L
        from permutect_amd.training.downsampler import COUNT_BIN_SKIP, MAX_ALT_COUNT, MAX_REF_COUNT
        b = batch.size()
        logits_b = logits_b.detach().contiguous().float()
        self.count_since_last_recomputation += b
        recompute = self.count_since_last_recomputation > Balancer.DATA_BEFORE_RECOMPUTE
        a = L.PmtBalanceArgs()
        a.num_variants, a.recompute = b, int(recompute)
        a.attenuation = math.pow(Balancer.ATTENUATION_PER_DATUM, self.count_since_last_recomputation) if recompute else 1.0
        g = a.bins
        g.num_sources, g.num_variant_types, g.num_ref_bins, g.num_alt_bins = self.num_sources, len(Variation), NUM_REF_COUNT_BINS, NUM_ALT_COUNT_BINS
        g.count_bin_skip, g.max_ref_count, g.max_alt_count = COUNT_BIN_SKIP, MAX_REF_COUNT, MAX_ALT_COUNT
        cols = [batch.get(f) for f in (Data.LABEL, Data.VARIANT_TYPE, Data.SOURCE, Data.REF_COUNT, Data.ALT_COUNT)]
        a.labels, a.variant_types, a.sources, a.ref_counts, a.alt_counts = [L.int_column(t) for t in cols]
        a.logits_b, a.counts, a.pseudo_counts = logits_b.data_ptr(), self.counts_slvra.data_ptr(), self.pseudo_counts_slvra.data_ptr()
        ins = (self.weights_slvra, self.unlabeled_weights_slvra, self.source_weights_s)
        if recompute:
            if getattr(self, "_spare", None) is None:
                self._spare = tuple(torch.empty_like(t) for t in ins)
            outs = self._spare
        else:
            outs = ins
        a.weights_in, a.unlabeled_weights_in, a.source_weights_in = [t.data_ptr() for t in ins]
        a.weights_out, a.unlabeled_weights_out, a.source_weights_out = [t.data_ptr() for t in outs]
        weights_b = torch.empty(b, dtype=torch.float32, device=self.device)
        source_weights_b = torch.empty(b, dtype=torch.float32, device=self.device)
        a.weights_b, a.source_weights_b = weights_b.data_ptr(), source_weights_b.data_ptr()
        L.check(L.load().pmt_balance_step(C.byref(a), L.raw_stream(self.device)), "pmt_balance_step")
        if recompute:
            self._spare = ins
            self.weights_slvra, self.unlabeled_weights_slvra, self.source_weights_s = outs
            self.count_since_last_recomputation = 0
        self._keep = (cols, logits_b)  # (alive until the launches have read them: the next step replaces the references)
        return weights_b, source_weights_b

    def process_batch_and_compute_weights(self, batch: Batch, artifact_probs_b: Tensor):
        idx = flattened_slvra_index(batch)
        labels = batch.get(Data.LABEL).long()
        art_idx = idx + self._label_stride * (int(Label.ARTIFACT) - labels)
        non_idx = idx + self._label_stride * (int(Label.VARIANT) - labels)
        p = artifact_probs_b.to(self.device).float()
        unlabeled = (1 - batch.get_is_labeled_mask()).float()
        self.counts_slvra.view(-1).index_add_(0, idx, torch.ones_like(p))
        self.pseudo_counts_slvra.view(-1).index_add_(0, art_idx, unlabeled * p)
        self.pseudo_counts_slvra.view(-1).index_add_(0, non_idx, unlabeled * (1 - p))
        self.count_since_last_recomputation += batch.size()
        if self.count_since_last_recomputation > Balancer.DATA_BEFORE_RECOMPUTE:
            att = math.pow(Balancer.ATTENUATION_PER_DATUM, self.count_since_last_recomputation)
            for counts, old in ((self.counts_slvra, self.weights_slvra), (self.pseudo_counts_slvra, self.unlabeled_weights_slvra)):
                ratio = (counts[:, Label.ARTIFACT] + 0.01) / (counts[:, Label.VARIANT] + 0.01)
                new = torch.zeros_like(counts)
                new[:, Label.ARTIFACT] = torch.clip((1 + 1 / ratio) / 2, min=0.01, max=100)
                new[:, Label.VARIANT] = torch.clip((1 + ratio) / 2, min=0.01, max=100)
                old.copy_(att * old + (1 - att) * new)
            counts_s = self.counts_slvra.sum(dim=(1, 2, 3, 4))
            new_source = (counts_s.sum(dim=0, keepdim=True) / counts_s) / self.num_sources
            self.source_weights_s.copy_(att * self.source_weights_s + (1 - att) * new_source)
            self.count_since_last_recomputation = 0
        labeled_w = self.weights_slvra.view(-1)[idx]
        unl_w = p * self.unlabeled_weights_slvra.view(-1)[art_idx] + (1 - p) * self.unlabeled_weights_slvra.view(-1)[non_idx]
        weights_b = unlabeled * unl_w + (1 - unlabeled) * labeled_w
        return weights_b, self.source_weights_s[batch.get(Data.SOURCE).long()]
