"""Downsampling fractions per variant (reference permutect/training/downsampler.py:26-123, data/count_binning.py,
data/batch.py:204-230): a mixture of four fixed Beta shapes whose weights are indexed by (source, label, variant type,
ref-count bin, alt-count bin).

`calculate_downsampling_fractions` keeps the reference's signature and sampling scheme in torch; `downsample` is the
product path: it looks up the per-variant mixture weights (one gather) and hands them to the fused device kernels
(`DownsampledBatch.on_device` -> pmt_downsample_counts / _index), which draw component, fraction and per-read keep
decisions without returning to the host.  The balance optimisation of the weights
(reference :125-158 `optimize_downsampling_balance`) is a separate, offline fit and is not part of the hot path; weights
can be loaded with `load_state_dict`."""
from __future__ import annotations

import torch
from torch import Tensor, nn

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


class Downsampler(nn.Module):
    def __init__(self, num_sources: int):
        super().__init__()
        self.num_sources = num_sources
        shape = (num_sources, len(Label), len(Variation), NUM_REF_COUNT_BINS, NUM_ALT_COUNT_BINS, len(BETA_BASIS_SHAPES))
        # stored as log weights; log_softmax over the last axis on use (the reference's LogWeights parametrization)
        self.log_ref_weights_slvrak = nn.Parameter(torch.zeros(shape), requires_grad=False)
        self.log_alt_weights_slvrah = nn.Parameter(torch.zeros(shape), requires_grad=False)
        self.register_buffer("beta_basis", torch.tensor(BETA_BASIS_SHAPES))

    def _weights_bk(self, batch: Batch):
        idx = flattened_slvra_index(batch)
        k = len(BETA_BASIS_SHAPES)
        ref = torch.softmax(self.log_ref_weights_slvrak, dim=-1).view(-1, k).index_select(0, idx)
        alt = torch.softmax(self.log_alt_weights_slvrah, dim=-1).view(-1, k).index_select(0, idx)
        return ref, alt

    def calculate_downsampling_fractions(self, batch: Batch):
        """Reference :105-123, in torch (multinomial over the mixture, then a Beta draw)."""
        ref_w, alt_w = self._weights_bk(batch)
        refk = torch.multinomial(ref_w, num_samples=1).flatten()
        altk = torch.multinomial(alt_w, num_samples=1).flatten()
        rs, as_ = self.beta_basis[refk], self.beta_basis[altk]
        beta = torch.distributions.Beta
        return beta(rs[:, 0], rs[:, 1]).sample().view(-1), beta(as_[:, 0], as_[:, 1]).sample().view(-1)

    def downsample(self, batch: Batch, seed: int, fix_alt_gather: bool = False) -> DownsampledBatch:
        """The training step's `DownsampledBatch(batch, *calculate_downsampling_fractions(batch))` in two device launches."""
        ref_w, alt_w = self._weights_bk(batch)
        return DownsampledBatch.on_device(batch, seed=seed, ref_weights_b4=ref_w, alt_weights_b4=alt_w, fix_alt_gather=fix_alt_gather)
