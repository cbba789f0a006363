// The balancer's training-step bookkeeping on the device (reference permutect/training/balancer.py:55-119,
// `Balancer.process_batch_and_compute_weights`): running counts per (source, label, variant type, ref-count bin, alt-count bin),
// pseudo-counts of the unlabeled data from the model's artifact probability, weight tables re-derived from the counts, the batch's
// weights looked up.  The reference (and this package's torch form of it, training/balancer.py) composes ~60 small tensor ops per
// step -- half a millisecond of launch chain in a 3.8 ms step, between the forward and the losses where nothing else can run.
// Here: two launches (include/permutect_amd.h: PmtBalanceArgs).
#include <hip/hip_runtime.h>
#include <math.h>

#include "permutect_amd.h"

#define BAL_THREADS 256
#define BAL_MAX_BINS 2048  // S * 3 * V * R * A = 300 S floats per table: up to 6 sources
#define BAL_LABEL_ARTIFACT 0
#define BAL_LABEL_VARIANT 1
#define BAL_LABEL_UNLABELED 2

__device__ __forceinline__ long long col_at(const PmtIntColumn& c, int i) {
    if (c.ptr == nullptr) return 0;
    return c.elem_bytes == 8 ? reinterpret_cast<const long long*>(c.ptr)[(size_t)i * c.stride]
                             : (long long)reinterpret_cast<const int*>(c.ptr)[(size_t)i * c.stride];
}
// reference data/count_binning.py:9-26, data/batch.py:228-230
__device__ __forceinline__ int cell_of(const PmtBinning& g, long long src, long long label, long long vt, long long nr, long long na) {
    const int rbin = (int)min(nr, (long long)g.max_ref_count) / g.count_bin_skip;
    const int abin = ((int)min(na, (long long)g.max_alt_count) - 1) / g.count_bin_skip;
    return (int)((((src * 3 + label) * g.num_variant_types + vt) * g.num_ref_bins + rbin) * g.num_alt_bins + abin);
}
__device__ __forceinline__ int label_stride_of(const PmtBinning& g) { return g.num_variant_types * g.num_ref_bins * g.num_alt_bins; }
__device__ __forceinline__ float sigmoid_of(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ __launch_bounds__(BAL_THREADS) void pmt_balance_accumulate_kernel(PmtBalanceArgs a) {
    __shared__ float sh[2][BAL_MAX_BINS];
    const int nb = a.bins.num_sources * 3 * label_stride_of(a.bins);
    for (int i = threadIdx.x; i < 2 * BAL_MAX_BINS; i += BAL_THREADS) (&sh[0][0])[i] = 0.f;
    __syncthreads();
    const int b = blockIdx.x * BAL_THREADS + threadIdx.x;
    if (b < a.num_variants) {
        const long long label = col_at(a.labels, b);
        const int idx = cell_of(a.bins, col_at(a.sources, b), label, col_at(a.variant_types, b), col_at(a.ref_counts, b), col_at(a.alt_counts, b));
        if (idx >= 0 && idx < nb) {
            atomicAdd(&sh[0][idx], 1.0f);
            if (label == BAL_LABEL_UNLABELED) {
                const float p = sigmoid_of(a.logits_b[b]);
                const int ls = label_stride_of(a.bins);
                atomicAdd(&sh[1][idx + ls * (BAL_LABEL_ARTIFACT - BAL_LABEL_UNLABELED)], p);
                atomicAdd(&sh[1][idx + ls * (BAL_LABEL_VARIANT - BAL_LABEL_UNLABELED)], 1.0f - p);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nb; i += BAL_THREADS) {
        if (sh[0][i] != 0.f) atomicAdd(&a.counts[i], sh[0][i]);
        if (sh[1][i] != 0.f) atomicAdd(&a.pseudo_counts[i], sh[1][i]);
    }
}

// one table entry after the step (reference balancer.py:76-93): att * old + (1 - att) * new, new = 0 outside the two labeled rows
__device__ __forceinline__ float table_entry(const float* __restrict__ counts, const float* __restrict__ old, int i, int ls, float att, int recompute) {
    const float o = old[i];
    if (!recompute) return o;
    const int label = (i / ls) % 3;
    float fresh = 0.f;
    if (label != BAL_LABEL_UNLABELED) {
        const int base = i - label * ls;
        const float ratio = (counts[base + BAL_LABEL_ARTIFACT * ls] + 0.01f) / (counts[base + BAL_LABEL_VARIANT * ls] + 0.01f);
        const float v = label == BAL_LABEL_ARTIFACT ? (1.0f + 1.0f / ratio) * 0.5f : (1.0f + ratio) * 0.5f;
        fresh = fminf(fmaxf(v, 0.01f), 100.0f);
    }
    return att * o + (1.0f - att) * fresh;
}

__global__ __launch_bounds__(BAL_THREADS) void pmt_balance_weights_kernel(PmtBalanceArgs a) {
    __shared__ float w[BAL_MAX_BINS], uw[BAL_MAX_BINS];
    __shared__ float sw[16], per_source[16];
    const int ls = label_stride_of(a.bins), S = a.bins.num_sources, nb = S * 3 * ls;
    for (int i = threadIdx.x; i < nb; i += BAL_THREADS) {
        w[i] = table_entry(a.counts, a.weights_in, i, ls, a.attenuation, a.recompute);
        uw[i] = table_entry(a.pseudo_counts, a.unlabeled_weights_in, i, ls, a.attenuation, a.recompute);
    }
    if (threadIdx.x < 16) per_source[threadIdx.x] = 0.f;
    __syncthreads();
    if (a.recompute) {  // counts per source (reference :94-96)
        for (int s = 0; s < S; ++s) {
            float part = 0.f;
            for (int i = threadIdx.x; i < 3 * ls; i += BAL_THREADS) part += a.counts[s * 3 * ls + i];
            for (int d = 32; d > 0; d >>= 1) part += __shfl_xor(part, d);
            if ((threadIdx.x & 63) == 0) atomicAdd(&per_source[s], part);
        }
    }
    __syncthreads();
    if (threadIdx.x < S) {
        float v = a.source_weights_in[threadIdx.x];
        if (a.recompute) {
            float total = 0.f;
            for (int s = 0; s < S; ++s) total += per_source[s];
            v = a.attenuation * v + (1.0f - a.attenuation) * ((total / per_source[threadIdx.x]) / (float)S);
        }
        sw[threadIdx.x] = v;
    }
    __syncthreads();
    if (blockIdx.x == 0 && a.weights_out != a.weights_in) {
        for (int i = threadIdx.x; i < nb; i += BAL_THREADS) {
            a.weights_out[i] = w[i];
            a.unlabeled_weights_out[i] = uw[i];
        }
        if (threadIdx.x < S) a.source_weights_out[threadIdx.x] = sw[threadIdx.x];
    }
    const int b = blockIdx.x * BAL_THREADS + threadIdx.x;
    if (b >= a.num_variants) return;
    const long long label = col_at(a.labels, b), src = col_at(a.sources, b);
    const int idx = cell_of(a.bins, src, label, col_at(a.variant_types, b), col_at(a.ref_counts, b), col_at(a.alt_counts, b));
    float wb = 1.0f;
    if (idx >= 0 && idx < nb) {
        if (label == BAL_LABEL_UNLABELED) {
            const float p = sigmoid_of(a.logits_b[b]);
            wb = p * uw[idx + ls * (BAL_LABEL_ARTIFACT - BAL_LABEL_UNLABELED)] + (1.0f - p) * uw[idx + ls * (BAL_LABEL_VARIANT - BAL_LABEL_UNLABELED)];
        } else {
            wb = w[idx];
        }
    }
    a.weights_b[b] = wb;
    a.source_weights_b[b] = wb * ((src >= 0 && src < S) ? sw[src] : 1.0f);
}

extern "C" int pmt_balance_step(const PmtBalanceArgs* args, void* stream) {
    if (!args || args->num_variants < 0) return PMT_E_INVALID;
    const PmtBinning& g = args->bins;
    if (g.num_sources < 1 || g.num_sources > 16 || g.num_variant_types < 1 || g.num_ref_bins < 1 || g.num_alt_bins < 1 || g.count_bin_skip < 1) return PMT_E_INVALID;
    if (g.num_sources * 3 * g.num_variant_types * g.num_ref_bins * g.num_alt_bins > BAL_MAX_BINS) return PMT_E_UNSUPPORTED;
    if (args->num_variants == 0) return PMT_OK;
    if (!args->labels.ptr || !args->variant_types.ptr || !args->ref_counts.ptr || !args->alt_counts.ptr || !args->logits_b || !args->counts ||
        !args->pseudo_counts || !args->weights_in || !args->unlabeled_weights_in || !args->source_weights_in || !args->weights_out ||
        !args->unlabeled_weights_out || !args->source_weights_out || !args->weights_b || !args->source_weights_b)
        return PMT_E_INVALID;
    for (const PmtIntColumn* c : {&args->labels, &args->variant_types, &args->sources, &args->ref_counts, &args->alt_counts})
        if (c->ptr != nullptr && c->elem_bytes != 4 && c->elem_bytes != 8) return PMT_E_INVALID;
    if (args->recompute && (args->weights_out == args->weights_in || args->unlabeled_weights_out == args->unlabeled_weights_in ||
                            args->source_weights_out == args->source_weights_in))
        return PMT_E_INVALID;  // workgroup 0 would overwrite what the others still read
    const hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid((args->num_variants + BAL_THREADS - 1) / BAL_THREADS);
    hipLaunchKernelGGL(pmt_balance_accumulate_kernel, grid, dim3(BAL_THREADS), 0, s, *args);
    hipLaunchKernelGGL(pmt_balance_weights_kernel, grid, dim3(BAL_THREADS), 0, s, *args);
    return hipGetLastError() == hipSuccess ? PMT_OK : PMT_E_LAUNCH;
}

// ---- the evaluation pass's tallies (PmtEvalArgs) -------------------------------------------------------------------------------
#define EVAL_LDS_BINS 8192  // floats of LDS for a workgroup's share of the histogram (one source: 6 300 bins); beyond it: global atomics
__global__ __launch_bounds__(BAL_THREADS) void pmt_record_evaluation_kernel(PmtEvalArgs a, float* __restrict__ flat) {
    __shared__ float hist[EVAL_LDS_BINS];
    __shared__ float stats[9];
    const bool in_lds = a.nhist <= EVAL_LDS_BINS;
    if (in_lds)
        for (int i = threadIdx.x; i < (int)a.nhist; i += BAL_THREADS) hist[i] = 0.f;
    if (threadIdx.x < 9) stats[threadIdx.x] = 0.f;
    __syncthreads();
    float* ghist = flat + (size_t)a.epoch_index * (size_t)a.nhist;
    float* gstats = flat + 2 * (size_t)a.nhist + (size_t)a.epoch_index * 9;
    const int b = blockIdx.x * BAL_THREADS + threadIdx.x;
    if (b < a.num_variants) {
        const long long label = col_at(a.labels, b);
        const float w = a.weights_b[b], logit = a.logits_b[b];
        if (label >= 0 && label < 3) {
            atomicAdd(&stats[label * 3 + (logit > 0.f ? 1 : 0)], w);
            atomicAdd(&stats[label * 3 + 2], w * logit);
        }
        if (label == BAL_LABEL_ARTIFACT || label == BAL_LABEL_VARIANT) {  // (unlabeled data are not tallied: reference evaluation_metrics.py:58-66)
            const int cell = cell_of(a.bins, col_at(a.sources, b), label, col_at(a.variant_types, b), col_at(a.ref_counts, b), col_at(a.alt_counts, b));
            const float clamped = fminf(fmaxf(logit, (float)a.min_logit), (float)a.max_logit);
            const int lbin = (int)floorf((clamped - (float)a.min_logit) / (float)a.logit_bin_skip);
            const long long idx = (long long)cell * a.num_logit_bins + lbin;
            if (cell >= 0 && idx < a.nhist && lbin >= 0 && lbin < a.num_logit_bins) {
                if (in_lds) atomicAdd(&hist[idx], w); else atomicAdd(&ghist[idx], w);
            }
        }
    }
    __syncthreads();
    if (in_lds)
        for (int i = threadIdx.x; i < (int)a.nhist; i += BAL_THREADS)
            if (hist[i] != 0.f) atomicAdd(&ghist[i], hist[i]);
    if (threadIdx.x < 9 && stats[threadIdx.x] != 0.f) atomicAdd(&gstats[threadIdx.x], stats[threadIdx.x]);
}

extern "C" int pmt_record_evaluation(const PmtEvalArgs* args, float* flat, void* stream) {
    if (!args || !flat || args->num_variants < 0 || args->epoch_index < 0 || args->epoch_index > 1 || args->num_logit_bins < 1 ||
        args->logit_bin_skip < 1 || args->nhist < 1)
        return PMT_E_INVALID;
    const PmtBinning& g = args->bins;
    if (g.num_sources < 1 || g.num_variant_types < 1 || g.num_ref_bins < 1 || g.num_alt_bins < 1 || g.count_bin_skip < 1) return PMT_E_INVALID;
    if ((int64_t)g.num_sources * 3 * g.num_variant_types * g.num_ref_bins * g.num_alt_bins * args->num_logit_bins != args->nhist) return PMT_E_INVALID;
    if (args->num_variants == 0) return PMT_OK;
    if (!args->labels.ptr || !args->variant_types.ptr || !args->ref_counts.ptr || !args->alt_counts.ptr || !args->logits_b || !args->weights_b) return PMT_E_INVALID;
    for (const PmtIntColumn* c : {&args->labels, &args->variant_types, &args->sources, &args->ref_counts, &args->alt_counts})
        if (c->ptr != nullptr && c->elem_bytes != 4 && c->elem_bytes != 8) return PMT_E_INVALID;
    hipLaunchKernelGGL(pmt_record_evaluation_kernel, dim3((args->num_variants + BAL_THREADS - 1) / BAL_THREADS), dim3(BAL_THREADS), 0,
                       reinterpret_cast<hipStream_t>(stream), *args, flat);
    return hipGetLastError() == hipSuccess ? PMT_OK : PMT_E_LAUNCH;
}
