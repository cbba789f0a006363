// Dropout in training (reference architecture/mlp.py:57-58: an nn.Dropout behind every nn.Linear of the MLPs built with
// dropout_p > 0 -- read_embedding, info_embedding, reducer, source_predictor and their skip blocks; artifact_model.py:145-206).
//
// The mask is a FUNCTION, not a tensor: element (row, feature) of the output of linear `lin` is kept iff
//     mix32(row_key(seed, lin, row) + feature * GOLDEN) >= threshold,     threshold = p * 2^32,
// and kept elements are scaled by 1 / (1 - p) as nn.Dropout does.  `row` is the read's row in the batch (read-set kernels) or the
// variant (row kernels), so the forward, the backward's recomputation, a split read set's groups and the host (pmt_dropout_mask,
// for the parity tests) all see the same mask without storing or exchanging it.  One step = one seed (the host draws it from
// torch's generator), so a model replays bit-identically under torch.manual_seed.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define PMT_HD __host__ __device__ inline
#else
#define PMT_HD inline
#endif

PMT_HD unsigned pmt_mix32(unsigned x) {  // murmur3's finalizer: every input bit reaches every output bit
    x ^= x >> 16; x *= 0x85EBCA6Bu;
    x ^= x >> 13; x *= 0xC2B2AE35u;
    x ^= x >> 16;
    return x;
}
PMT_HD unsigned pmt_drop_row_key(unsigned s0, unsigned s1, int lin, int row) {
    return pmt_mix32((s0 ^ ((unsigned)row * 0x9E3779B1u)) + (unsigned)lin * 0x7FEB352Du) ^ s1;
}
PMT_HD bool pmt_drop_keep(unsigned row_key, int feat, unsigned thresh) {
    return pmt_mix32(row_key + (unsigned)feat * 0x9E3779B9u) >= thresh;
}
PMT_HD unsigned pmt_drop_threshold(float p) {
    const double t = (double)p * 4294967296.0;
    return t >= 4294967295.0 ? 0xFFFFFFFFu : (t <= 0.0 ? 0u : (unsigned)t);
}
