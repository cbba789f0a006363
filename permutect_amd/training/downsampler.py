"""Downsampling fractions per variant (reference permutect/training/downsampler.py:26-123, data/count_binning.py,
data/batch.py:204-230): a mixture of four fixed Beta shapes whose weights are indexed by (source, label, variant type,
ref-count bin, alt-count bin).

`calculate_downsampling_fractions` keeps the reference's signature and sampling scheme in torch; `downsample` is the
product path: it looks up the per-variant mixture weights (one gather) and hands them to the fused device kernels
(`DownsampledBatch.on_device` -> pmt_downsample_counts / _index), which draw component, fraction and per-read keep
decisions without returning to the host.  `optimize_downsampling_balance` is the reference's one-off fit of the mixture
weights before training (:125-158: 10 000 AdamW steps that spread the expected downsampled counts evenly over the count
bins; deterministic, pinned by tests/golden/downsampler_fit.npz).  Module layout and state_dict keys are the reference's
(`parametrizations.log_ref_weights_slvrak.original`, ..., `beta_basis`, `binned_ref_trans_kry`, `binned_alt_trans_haz`)."""
from __future__ import annotations

import torch
from torch import Tensor, nn
from torch.nn.utils import parametrize

from permutect_amd.data.batch import Batch, DownsampledBatch
from permutect_amd.data.datum import Data
from permutect_amd.enums import Label, Variation

# reference data/count_binning.py:9-26
MAX_REF_COUNT, MIN_ALT_COUNT, MAX_ALT_COUNT, COUNT_BIN_SKIP = 10, 1, 15, 3
NUM_REF_COUNT_BINS = (MAX_REF_COUNT // COUNT_BIN_SKIP) + 1
NUM_ALT_COUNT_BINS = ((MAX_ALT_COUNT - MIN_ALT_COUNT) // COUNT_BIN_SKIP) + 1
BETA_BASIS_SHAPES = ((1.0, 1.0), (1.0, 5.0), (5.0, 1.0), (5.0, 5.0))  # reference downsampler.py:27 (the kernels use the same)


def ref_count_bin_indices(counts: Tensor) -> Tensor:
    return torch.div(torch.clip(counts, max=MAX_REF_COUNT), COUNT_BIN_SKIP, rounding_mode="floor")


def alt_count_bin_indices(counts: Tensor) -> Tensor:
    return torch.div(torch.clip(counts, max=MAX_ALT_COUNT) - 1, COUNT_BIN_SKIP, rounding_mode="floor")


def flattened_slvra_index(batch: Batch) -> Tensor:
    """Row-major index into a [S, L, V, R, A] tensor (reference data/batch.py:228-230)."""
    s, lab, v = batch.get(Data.SOURCE).long(), batch.get(Data.LABEL).long(), batch.get(Data.VARIANT_TYPE).long()
    r = ref_count_bin_indices(batch.get(Data.REF_COUNT).long())
    a = alt_count_bin_indices(batch.get(Data.ALT_COUNT).long())
    return (((s * len(Label) + lab) * len(Variation) + v) * NUM_REF_COUNT_BINS + r) * NUM_ALT_COUNT_BINS + a


def beta_binomial_log_lk(n: Tensor, k: Tensor, alpha: Tensor, beta: Tensor) -> Tensor:
    """log P(k | n, alpha, beta) of the beta-binomial (reference utils/stats_utils.py:28-40)."""
    comb = torch.lgamma(n + 1) - torch.lgamma(n - k + 1) - torch.lgamma(k + 1)
    return (comb + torch.lgamma(k + alpha) + torch.lgamma(n - k + beta) + torch.lgamma(alpha + beta)
            - torch.lgamma(n + alpha + beta) - torch.lgamma(alpha) - torch.lgamma(beta))


class _LogWeights(nn.Module):  # the reference's LogWeights parametrization (architecture/parameterizations.py)
    def forward(self, x: Tensor) -> Tensor:
        return torch.log_softmax(x, dim=-1)


class Downsampler(nn.Module):
    def __init__(self, num_sources: int):
        super().__init__()
        self.num_sources = num_sources
        basis = torch.tensor(BETA_BASIS_SHAPES)
        self.beta_basis = nn.Parameter(basis.clone(), requires_grad=False)
        # Binned transition matrices count bin -> downsampled count bin per basis Beta (reference :36-89): a beta-binomial
        # over raw counts, SUMMED over the downsampled counts of a bin and AVERAGED over the original counts of a bin.
        a_k11, b_k11 = basis[:, 0].view(-1, 1, 1), basis[:, 1].view(-1, 1, 1)
        raw_r = torch.arange(MAX_REF_COUNT + 1, dtype=torch.int32)
        raw_a = torch.arange(MAX_ALT_COUNT + 1, dtype=torch.int32)
        ref, downref = raw_r.view(1, -1, 1), raw_r.view(1, 1, -1)
        alt, downalt = raw_a.view(1, -1, 1), raw_a.view(1, 1, -1)
        ref_trans = torch.where(ref >= downref, torch.exp(beta_binomial_log_lk(ref, downref, a_k11, b_k11)), 0)
        alt_trans = torch.where(alt >= downalt, torch.exp(beta_binomial_log_lk(alt, downalt, a_k11, b_k11)), 0)
        binned_ref = torch.zeros(len(basis), NUM_REF_COUNT_BINS, NUM_REF_COUNT_BINS)
        binned_alt = torch.zeros(len(basis), NUM_ALT_COUNT_BINS, NUM_ALT_COUNT_BINS)
        for r in range(MAX_REF_COUNT + 1):
            for d in range(MAX_REF_COUNT + 1):
                binned_ref[:, r // COUNT_BIN_SKIP, d // COUNT_BIN_SKIP] += ref_trans[:, r, d]
        alt_bin = lambda c: 0 if c < MIN_ALT_COUNT else (c - MIN_ALT_COUNT) // COUNT_BIN_SKIP  # noqa: E731 (no alt read kept -> one is forced: bin 0)
        for a in range(MAX_ALT_COUNT + 1):
            for d in range(MAX_ALT_COUNT + 1):
                binned_alt[:, alt_bin(a), alt_bin(d)] += alt_trans[:, a, d]
        self.binned_ref_trans_kry = nn.Parameter(binned_ref / COUNT_BIN_SKIP, requires_grad=False)
        self.binned_alt_trans_haz = nn.Parameter(binned_alt / COUNT_BIN_SKIP, requires_grad=False)
        shape = (num_sources, len(Label), len(Variation), NUM_REF_COUNT_BINS, NUM_ALT_COUNT_BINS, len(BETA_BASIS_SHAPES))
        self.log_ref_weights_slvrak = nn.Parameter(torch.zeros(shape), requires_grad=False)
        parametrize.register_parametrization(self, "log_ref_weights_slvrak", _LogWeights())
        self.log_alt_weights_slvrah = nn.Parameter(torch.zeros(shape), requires_grad=False)
        parametrize.register_parametrization(self, "log_alt_weights_slvrah", _LogWeights())

    def weights_parameters(self):
        return [self.parametrizations.log_ref_weights_slvrak.original, self.parametrizations.log_alt_weights_slvrah.original]

    def _weights_bk(self, batch: Batch):
        idx = flattened_slvra_index(batch)
        k = len(BETA_BASIS_SHAPES)
        ref = torch.exp(self.log_ref_weights_slvrak).view(-1, k).index_select(0, idx)
        alt = torch.exp(self.log_alt_weights_slvrah).view(-1, k).index_select(0, idx)
        return ref, alt

    def calculate_expected_downsampled_counts(self, counts_slvra: Tensor) -> Tensor:
        """sum_{r a k h} counts_slvra w_ref_slvrak w_alt_slvrah T_ref_kry T_alt_haz  (reference :125-139)"""
        return torch.einsum("slvra, slvrak, slvrah, kry, haz->slvyz", counts_slvra, torch.exp(self.log_ref_weights_slvrak),
                            torch.exp(self.log_alt_weights_slvrah), self.binned_ref_trans_kry, self.binned_alt_trans_haz)

    def optimize_downsampling_balance(self, counts_slvra: Tensor, steps: int = 10000):
        """Reference :141-158: AdamW (torch defaults) on the two `.original` weight tensors, minimising the sum over
        (source, label, variant type) of the squared normalised expected downsampled counts -- i.e. spreading them evenly
        over the ref / alt count bins.  Deterministic; runs where the module lives (a few seconds on the CPU)."""
        params = self.weights_parameters()
        for p in params:
            p.requires_grad_(True)
        optimizer = torch.optim.AdamW([p for p in self.parameters() if p.requires_grad])
        counts_slvra = counts_slvra.to(device=params[0].device, dtype=params[0].dtype)
        # 10 000 steps over tensors of a few thousand elements: on a many-core host torch's default intra-op pool (one thread per
        # hardware thread, 256 on the MI355X hosts, of which a job may own 16) turns every small op into a thread rendezvous --
        # 12 ms per step there against 1.5 ms with a handful of threads (two minutes of every training run's start-up)
        threads = torch.get_num_threads()
        if params[0].device.type == "cpu":
            torch.set_num_threads(min(threads, 4))
        try:
            for _ in range(steps):
                expected = self.calculate_expected_downsampled_counts(counts_slvra)
                total = torch.sum(expected, dim=(-2, -1), keepdim=True)
                # a (source, label, variant type) cell without any data: the reference divides 0 / 0 there and every weight turns
                # NaN; here such a cell contributes nothing (its weights only see the weight decay).  Cells with data are
                # independent terms of the loss, so wherever the reference's result is finite this is the same fit.
                normalized = expected / torch.where(total > 0, total, torch.ones_like(total))
                loss = torch.sum(torch.sum(torch.square(normalized), dim=(-2, -1)))
                optimizer.zero_grad(set_to_none=True)
                loss.backward()
                optimizer.step()
        finally:
            torch.set_num_threads(threads)
        for p in params:
            p.requires_grad_(False)

    def calculate_downsampling_fractions(self, batch: Batch):
        """Reference :105-123, in torch (multinomial over the mixture, then a Beta draw)."""
        ref_w, alt_w = self._weights_bk(batch)
        refk = torch.multinomial(ref_w, num_samples=1).flatten()
        altk = torch.multinomial(alt_w, num_samples=1).flatten()
        rs, as_ = self.beta_basis[refk], self.beta_basis[altk]
        beta = torch.distributions.Beta
        return beta(rs[:, 0], rs[:, 1]).sample().view(-1), beta(as_[:, 0], as_[:, 1]).sample().view(-1)

    def weight_tables(self):
        """exp of the two log-weight tables as flat [S L V R A][4] device arrays, made once per state of the weights (they stand still
        while a model trains: `optimize_downsampling_balance` runs before, reference model_training.py:58-60); each access of the
        parametrized attributes is a log_softmax launch, and the per-variant lookup was ~25 more per training step"""
        o_r, o_a = self.weights_parameters()
        key = (o_r._version, o_a._version, o_r.device, o_r.data_ptr(), o_a.data_ptr())
        if getattr(self, "_tables_key", None) != key:
            k = len(BETA_BASIS_SHAPES)
            with torch.no_grad():
                self._tables = (torch.exp(self.log_ref_weights_slvrak).reshape(-1, k).contiguous().float(),
                                torch.exp(self.log_alt_weights_slvrah).reshape(-1, k).contiguous().float())
            self._tables_key = key
        return self._tables

    def downsample(self, batch: Batch, seed: int, fix_alt_gather: bool = False) -> DownsampledBatch:
        """The training step's `DownsampledBatch(batch, *calculate_downsampling_fractions(batch))` in two device launches: the
        mixture weights of a variant's (source, label, variant type, count bins) cell are looked up inside the first."""
        if not batch.int_tensor.is_cuda:
            ref_w, alt_w = self._weights_bk(batch)
            return DownsampledBatch.on_device(batch, seed=seed, ref_weights_b4=ref_w, alt_weights_b4=alt_w, fix_alt_gather=fix_alt_gather)
        return DownsampledBatch.on_device(batch, seed=seed, weight_tables=self.weight_tables(), num_sources=self.num_sources, fix_alt_gather=fix_alt_gather)
