// Per-variant losses of the artifact model, forward and backward, one launch each (reference
// architecture/artifact_model.py:267-325: supervised BCE on the capped logit, outlier BCE on the clipped outlier-vs-rest
// logit, alt-count adversary MSE on sigmoid(prediction), source adversary squared error on softmax(prediction), and the
// weighted total).  Through torch these are ~85 elementwise launches per training step.
#include <hip/hip_runtime.h>
#include <math.h>

#include "permutect_amd.h"

#define LOSS_THREADS 256

__device__ __forceinline__ float softplus_f(float x) { return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))); }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

struct OutlierLogit {
    float o;        // lk[1] - logsumexp(lk[0], lk[2..])
    float mx, se;   // of the non-outlier set
};
__device__ __forceinline__ OutlierLogit outlier_logit(const float* __restrict__ lk, int K) {
    OutlierLogit r;
    r.mx = lk[0];
    for (int k = 0; k < K; ++k) r.mx = fmaxf(r.mx, lk[2 + k]);
    r.se = expf(lk[0] - r.mx);
    for (int k = 0; k < K; ++k) r.se += expf(lk[2 + k] - r.mx);
    r.o = lk[1] - (r.mx + logf(r.se));
    return r;
}

__global__ __launch_bounds__(LOSS_THREADS) void pmt_losses_forward_kernel(PmtLossArgs a, PmtLossOutputs out) {
    const int b = blockIdx.x * LOSS_THREADS + threadIdx.x;
    if (b >= a.num_variants) return;
    const int K = a.num_clusters, S = a.num_sources;
    const long long label = a.labels[(size_t)b * a.label_stride];
    const float is_labeled = label != 2 ? 1.f : 0.f;                     // Label.UNLABELED = 2
    const float target = label == 0 ? 1.f : (label == 2 ? 0.5f : 0.f);  // ARTIFACT 1.0, UNLABELED 0.5, VARIANT 0.0
    const float x = a.logits_b[b];
    const float sup = is_labeled * (fmaxf(x, 0.f) - x * target + log1pf(expf(-fabsf(x))));
    const OutlierLogit ol = outlier_logit(a.logits_bk + (size_t)b * (K + 2), K);
    const float unsup = (1.f - is_labeled) * softplus_f(fminf(ol.o, a.max_outlier_logit));
    const float pred = sigmoid_f(a.alt_count_raw[b]);
    const float tgt = (float)a.alt_counts[(size_t)b * a.alt_count_stride] / a.max_alt_count;
    const float altl = (pred - tgt) * (pred - tgt);
    float src = 0.f;
    if (S > 1 && a.source_logits != nullptr) {
        const float* sl = a.source_logits + (size_t)b * S;
        const long long sid = a.sources[(size_t)b * a.source_stride];
        float mx = sl[0];
        for (int s = 1; s < S; ++s) mx = fmaxf(mx, sl[s]);
        float se = 0.f;
        for (int s = 0; s < S; ++s) se += expf(sl[s] - mx);
        for (int s = 0; s < S; ++s) {
            const float d = expf(sl[s] - mx) / se - (s == sid ? 1.f : 0.f);
            src += d * d;
        }
    }
    out.supervised_b[b] = sup;
    out.unsupervised_b[b] = unsup;
    out.alt_count_b[b] = altl;
    out.source_b[b] = src;
    out.total_b[b] = a.weights[b] * (sup + unsup + altl) + a.source_weights[b] * src;
}

__global__ __launch_bounds__(LOSS_THREADS) void pmt_losses_backward_kernel(PmtLossArgs a, PmtLossOutputs g, PmtLossInputGrads d) {
    const int b = blockIdx.x * LOSS_THREADS + threadIdx.x;
    if (b >= a.num_variants) return;
    const int K = a.num_clusters, S = a.num_sources;
    const float gt = g.total_b ? g.total_b[b] : 0.f;
    const float g_sup = (g.supervised_b ? g.supervised_b[b] : 0.f) + gt * a.weights[b];
    const float g_unsup = (g.unsupervised_b ? g.unsupervised_b[b] : 0.f) + gt * a.weights[b];
    const float g_alt = (g.alt_count_b ? g.alt_count_b[b] : 0.f) + gt * a.weights[b];
    const float g_src = (g.source_b ? g.source_b[b] : 0.f) + gt * a.source_weights[b];
    const long long label = a.labels[(size_t)b * a.label_stride];
    const float is_labeled = label != 2 ? 1.f : 0.f;
    const float target = label == 0 ? 1.f : (label == 2 ? 0.5f : 0.f);
    d.d_logits_b[b] = g_sup * is_labeled * (sigmoid_f(a.logits_b[b]) - target);
    const float* lk = a.logits_bk + (size_t)b * (K + 2);
    float* dlk = d.d_logits_bk + (size_t)b * (K + 2);
    const OutlierLogit ol = outlier_logit(lk, K);
    const float go = ol.o <= a.max_outlier_logit ? g_unsup * (1.f - is_labeled) * sigmoid_f(ol.o) : 0.f;  // clip(max=...)
    dlk[1] = go;
    dlk[0] = -go * expf(lk[0] - ol.mx) / ol.se;
    for (int k = 0; k < K; ++k) dlk[2 + k] = -go * expf(lk[2 + k] - ol.mx) / ol.se;
    const float pred = sigmoid_f(a.alt_count_raw[b]);
    const float tgt = (float)a.alt_counts[(size_t)b * a.alt_count_stride] / a.max_alt_count;
    d.d_alt_count_raw[b] = g_alt * 2.f * (pred - tgt) * pred * (1.f - pred);
    if (S > 1 && a.source_logits != nullptr && d.d_source_logits != nullptr) {
        const float* sl = a.source_logits + (size_t)b * S;
        float* dsl = d.d_source_logits + (size_t)b * S;
        const long long sid = a.sources[(size_t)b * a.source_stride];
        float mx = sl[0];
        for (int s = 1; s < S; ++s) mx = fmaxf(mx, sl[s]);
        float se = 0.f;
        for (int s = 0; s < S; ++s) se += expf(sl[s] - mx);
        float dot = 0.f;  // sum_i 2 (p_i - t_i) p_i
        for (int s = 0; s < S; ++s) {
            const float p = expf(sl[s] - mx) / se;
            dot += 2.f * (p - (s == sid ? 1.f : 0.f)) * p;
        }
        for (int s = 0; s < S; ++s) {
            const float p = expf(sl[s] - mx) / se;
            dsl[s] = g_src * (2.f * (p - (s == sid ? 1.f : 0.f)) * p - p * dot);
        }
    }
}

static int loss_args_check(const PmtLossArgs* a) {
    if (!a || a->num_variants < 0 || a->num_clusters < 0 || a->num_clusters > PMT_MAX_CLUSTERS || a->num_sources < 1) return PMT_E_INVALID;
    if (a->num_variants == 0) return PMT_OK;
    if (!a->logits_b || !a->logits_bk || !a->alt_count_raw || !a->labels || !a->alt_counts || !a->weights || !a->source_weights)
        return PMT_E_INVALID;
    if (a->num_sources > 1 && a->source_logits && !a->sources) return PMT_E_INVALID;
    return PMT_OK;
}

extern "C" int pmt_losses_forward(const PmtLossArgs* args, const PmtLossOutputs* out, void* stream) {
    const int rc = loss_args_check(args);
    if (rc) return rc;
    if (!out || !out->supervised_b || !out->unsupervised_b || !out->alt_count_b || !out->source_b || !out->total_b) return PMT_E_INVALID;
    if (args->num_variants == 0) return PMT_OK;
    hipLaunchKernelGGL(pmt_losses_forward_kernel, dim3((args->num_variants + LOSS_THREADS - 1) / LOSS_THREADS), dim3(LOSS_THREADS), 0,
                       reinterpret_cast<hipStream_t>(stream), *args, *out);
    return hipGetLastError() == hipSuccess ? PMT_OK : PMT_E_LAUNCH;
}

extern "C" int pmt_losses_backward(const PmtLossArgs* args, const PmtLossOutputs* grad_out, const PmtLossInputGrads* grad_in,
                                   void* stream) {
    const int rc = loss_args_check(args);
    if (rc) return rc;
    if (!grad_out || !grad_in || !grad_in->d_logits_b || !grad_in->d_logits_bk || !grad_in->d_alt_count_raw) return PMT_E_INVALID;
    if (args->num_variants == 0) return PMT_OK;
    hipLaunchKernelGGL(pmt_losses_backward_kernel, dim3((args->num_variants + LOSS_THREADS - 1) / LOSS_THREADS), dim3(LOSS_THREADS), 0,
                       reinterpret_cast<hipStream_t>(stream), *args, *grad_out, *grad_in);
    return hipGetLastError() == hipSuccess ? PMT_OK : PMT_E_LAUNCH;
}

// ---------------------------------------------------------------------------------------------------------------------
// Loss bookkeeping of a training / evaluation step (reference training/loss_recorder.py:15-24, metrics/loss_metrics.py:
// 50-54): three LossMetrics (primary, alt-count, source), each a (totals, counts) pair of [S][L][V][R][A] histograms
// indexed by source, label, variant type, ref-count bin and alt-count bin (reference data/count_binning.py).  The
// reference issues 8 index_add_ launches per step; here one launch bins every variant once, accumulates the six
// histograms of the workgroup in LDS and adds them to global memory once per workgroup.
// ---------------------------------------------------------------------------------------------------------------------
#define REC_THREADS 256
#define REC_MAX_BINS 2048  // S * 3 * 5 * 4 * 5 = 300 S floats per histogram: up to 6 sources

__device__ __forceinline__ long long rec_col_at(const PmtIntColumn& c, int i) {
    if (c.ptr == nullptr) return 0;
    return c.elem_bytes == 8 ? reinterpret_cast<const long long*>(c.ptr)[(size_t)i * c.stride]
                             : (long long)reinterpret_cast<const int*>(c.ptr)[(size_t)i * c.stride];
}

__global__ __launch_bounds__(REC_THREADS) void pmt_record_losses_kernel(PmtRecordArgs a, float* __restrict__ hist) {
    __shared__ float sh[6][REC_MAX_BINS];
    const int nb = a.num_bins;
    for (int i = threadIdx.x; i < 6 * REC_MAX_BINS; i += REC_THREADS) (&sh[0][0])[i] = 0.f;
    __syncthreads();
    const int b = blockIdx.x * REC_THREADS + threadIdx.x;
    if (b < a.num_variants) {
        const long long label = rec_col_at(a.labels, b), vt = rec_col_at(a.variant_types, b), src = rec_col_at(a.sources, b);
        const long long nr = rec_col_at(a.ref_counts, b), na = rec_col_at(a.alt_counts, b);
        const int rbin = (int)min(nr, (long long)a.max_ref_count) / a.count_bin_skip;
        const int abin = ((int)min(na, (long long)a.max_alt_count) - 1) / a.count_bin_skip;
        const int idx = (int)((((src * 3 + label) * a.num_variant_types + vt) * a.num_ref_bins + rbin) * a.num_alt_bins + abin);
        if (idx >= 0 && idx < nb) {
            const float is_labeled = label != 2 ? 1.f : 0.f;
            const float w = a.weights[b], sw = a.source_weights[b];
            const float lw = is_labeled * w, uw = (1.f - is_labeled) * w;
            atomicAdd(&sh[0][idx], a.supervised_b[b] * lw + a.unsupervised_b[b] * uw);  // primary totals
            atomicAdd(&sh[1][idx], lw + uw);                                              // primary counts
            atomicAdd(&sh[2][idx], a.alt_count_b[b] * w);                                 // alt-count totals
            atomicAdd(&sh[3][idx], w);                                                    // alt-count counts
            atomicAdd(&sh[4][idx], a.source_b[b] * sw);                                   // source totals
            atomicAdd(&sh[5][idx], sw);                                                   // source counts
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 6 * nb; i += REC_THREADS) {
        const int hsel = i / nb, j = i - hsel * nb;
        const float v = sh[hsel][j];
        if (v != 0.f) atomicAdd(&hist[(size_t)hsel * nb + j], v);
    }
}

extern "C" int pmt_record_losses(const PmtRecordArgs* args, float* histograms, void* stream) {
    if (!args || !histograms || args->num_variants < 0 || args->num_bins < 1 || args->num_bins > REC_MAX_BINS) return PMT_E_INVALID;
    if (args->num_variants == 0) return PMT_OK;
    for (const PmtIntColumn* c : {&args->labels, &args->variant_types, &args->sources, &args->ref_counts, &args->alt_counts})
        if (c->ptr != nullptr && c->elem_bytes != 4 && c->elem_bytes != 8) return PMT_E_INVALID;
    if (!args->labels.ptr || !args->variant_types.ptr || !args->ref_counts.ptr || !args->alt_counts.ptr || !args->weights || !args->source_weights ||
        !args->supervised_b || !args->unsupervised_b || !args->alt_count_b || !args->source_b || args->count_bin_skip < 1)
        return PMT_E_INVALID;
    hipLaunchKernelGGL(pmt_record_losses_kernel, dim3((args->num_variants + REC_THREADS - 1) / REC_THREADS), dim3(REC_THREADS), 0,
                       reinterpret_cast<hipStream_t>(stream), *args, histograms);
    return hipGetLastError() == hipSuccess ? PMT_OK : PMT_E_LAUNCH;
}

// ---- posterior hand-off: the float rows of a batch, scattered to dataset order (reference tools/filter_variants.py:302-320) --------
__global__ __launch_bounds__(256) void pmt_posterior_rows_kernel(const float* __restrict__ float_rows, long long float_stride, int n_scalars,
                                                                 int logit_col, const float* __restrict__ logits_b,
                                                                 const float* __restrict__ features_be, int e,
                                                                 const long long* __restrict__ dest_ids, int n, float* __restrict__ block,
                                                                 long long block_stride) {
    const int width = n_scalars + e;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < (long long)n * width; i += (long long)gridDim.x * 256) {
        const int v = (int)(i / width), c = (int)(i - (long long)v * width);
        const long long row = dest_ids ? dest_ids[v] : v;
        float x;
        if (c < n_scalars) {  // the scalar columns live in a float16 array (data/datum.py:25,76): round to nearest even through it
            const float raw = c == logit_col ? logits_b[v] : float_rows[(long long)v * float_stride + c];
            x = (float)(_Float16)raw;
        } else {
            x = features_be[(long long)v * e + (c - n_scalars)];
        }
        block[row * block_stride + c] = x;
    }
}

extern "C" int pmt_posterior_rows(const float* float_rows, int64_t float_stride, int32_t n_scalars, int32_t logit_col, const float* logits_b,
                                  const float* features_be, int32_t e, const int64_t* dest_ids, int32_t n, float* block, int64_t block_stride,
                                  void* stream) {
    if (!float_rows || !logits_b || !features_be || !block || n < 0 || n_scalars < 1 || e < 0 || logit_col < 0 || logit_col >= n_scalars ||
        block_stride < n_scalars + e)
        return PMT_E_INVALID;
    if (n == 0) return PMT_OK;
    const long long work = (long long)n * (n_scalars + e);
    const int grid = (int)((work + 255) / 256 < 4096 ? (work + 255) / 256 : 4096);
    hipLaunchKernelGGL(pmt_posterior_rows_kernel, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), float_rows, (long long)float_stride,
                       n_scalars, logit_col, logits_b, features_be, e, reinterpret_cast<const long long*>(dest_ids), n, block, (long long)block_stride);
    return hipGetLastError() == hipSuccess ? PMT_OK : PMT_E_LAUNCH;
}
