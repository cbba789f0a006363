// Read downsampling of a training batch on the device (reference data/batch.py:389-439 `DownsampledBatch` and
// training/downsampler.py:105-123 `Downsampler.calculate_downsampling_fractions`).
//
// The reference draws, per variant, a mixture component of four fixed Beta shapes and a keep fraction from it, then a
// Bernoulli keep decision per read (forcing one alt read per variant), and gathers the kept rows: ~35 small torch
// launches and a host sync per training step.  Here: one launch decides fractions and new counts, one (after the
// exclusive scans of the counts) writes the gather index the kernels consume; the decisions come from a counter-based
// generator keyed by (seed, row), so both launches agree without storing a mask and nothing returns to the host.
// The random STREAM differs from torch's, the distribution does not (tests/test_downsample_gpu.py).
#include <hip/hip_runtime.h>
#include <math.h>

#include "permutect_amd.h"

// splitmix64-style counter hash -> uniform in [0, 1)
__device__ __forceinline__ float uniform01(unsigned long long seed, unsigned long long stream, unsigned long long counter) {
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (counter + 1) + 0xD1B54A32D192ED03ull * (stream + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (float)(z >> 40) * (1.0f / 16777216.0f);
}

// Beta(a, b) for the reference's basis shapes (1,1), (1,5), (5,1), (5,5) (training/downsampler.py:27): for integer shapes
// it is the a-th smallest of a + b - 1 uniforms.
__device__ float beta_sample(int comp, unsigned long long seed, unsigned long long stream, unsigned long long counter) {
    const int a = (comp == 2 || comp == 3) ? 5 : 1, b = (comp == 1 || comp == 3) ? 5 : 1;
    const int n = a + b - 1;
    float u[9];
    for (int i = 0; i < n; ++i) u[i] = uniform01(seed, stream + 16 + i, counter);
    for (int i = 1; i < n; ++i) {  // insertion sort of at most 9 values
        const float x = u[i];
        int j = i - 1;
        for (; j >= 0 && u[j] > x; --j) u[j + 1] = u[j];
        u[j + 1] = x;
    }
    return u[a - 1];
}

__device__ int pick_component(const float* __restrict__ w4, float u) {
    if (w4 == nullptr) return min(3, (int)(u * 4.0f));
    const float tot = w4[0] + w4[1] + w4[2] + w4[3];
    float c = 0.f;
    for (int k = 0; k < 3; ++k) {
        c += w4[k];
        if (u * tot < c) return k;
    }
    return 3;
}

// stream ids: 0 ref component, 1 alt component, 2 ref keep decisions, 3 alt keep decisions, 16.. Beta order statistics
__device__ __forceinline__ bool keep_ref(const PmtDownsample& a, long long row, float frac) { return uniform01(a.seed, 2, row) < frac; }
__device__ __forceinline__ bool keep_alt(const PmtDownsample& a, long long alt_row, float frac, long long forced) {
    return alt_row == forced || uniform01(a.seed, 3, alt_row) < frac;
}
// the reference forces alt read  alt_end - (random_int % alt_count) - 1  of every variant (data/batch.py:418-421)
__device__ __forceinline__ long long forced_alt(const PmtDownsample& a, int a0, int na) { return na > 0 ? (long long)a0 + na - 1 - (a.force_random % na) : -1; }

__device__ __forceinline__ long long ds_col_at(const PmtIntColumn& c, int i) {
    if (c.ptr == nullptr) return 0;
    return c.elem_bytes == 8 ? reinterpret_cast<const long long*>(c.ptr)[(size_t)i * c.stride]
                             : (long long)reinterpret_cast<const int*>(c.ptr)[(size_t)i * c.stride];
}

// SIXTEEN lanes per variant (a WGS read set has at most 10 + 15 reads; deeper ones loop), sixteen variants per workgroup of 256.  The
// first version spent a whole 64-lane workgroup on every variant and drew the two Beta fractions in all 64 lanes: 119 us per
// 65 536-variant step for 850 K keep decisions.  The random STREAMS are unchanged (a decision is a function of (seed, stream, row)).
#define DS_THREADS 256
#define DS_LANES 16
__global__ __launch_bounds__(DS_THREADS) void pmt_downsample_counts_kernel(PmtDownsample a, float* __restrict__ ref_fracs,
                                                                           float* __restrict__ alt_fracs, int* __restrict__ new_ref,
                                                                           int* __restrict__ new_alt) {
    const int sub = threadIdx.x & (DS_LANES - 1);
    const int b = blockIdx.x * (DS_THREADS / DS_LANES) + threadIdx.x / DS_LANES;
    const bool live = b < a.num_variants;
    const int bb = live ? b : 0;
    const int r0 = a.ref_offsets[bb], nr = a.ref_offsets[bb + 1] - r0, a0 = a.alt_offsets[bb], na = a.alt_offsets[bb + 1] - a0;
    float fr = 0.f, fa = 0.f;
    if (a.ref_fracs_in != nullptr) {
        fr = live ? a.ref_fracs_in[bb] : 0.f;
        fa = live ? a.alt_fracs_in[bb] : 0.f;
    } else {
        // The two fractions: a Beta(a, b) of the basis shapes is the a-th smallest of a + b - 1 <= 9 uniforms.  Lanes 0 - 8 of the variant's
        // sixteen draw ONE uniform each (per side) and rank it among the nine with eight shuffles; the lane of rank a - 1 holds the
        // fraction.  The same uniforms (stream 32 / 64 + 16 + i) as beta_sample's sort, hence the same fractions bit for bit -- without two
        // lanes of sixteen sorting nine hashes each while fourteen wait (that was most of this kernel's 42 us).
        int comp[2] = {0, 0};
        if (sub < 2) {  // lane 0 picks the ref component, lane 1 the alt component
            const float* w4 = sub == 0 ? a.ref_weights_b4 : a.alt_weights_b4;
            if (w4 != nullptr) {
                w4 += 4 * (size_t)bb;
            } else if (a.ref_weight_table != nullptr) {  // the variant's cell of the Downsampler's tables: PARENT counts (data/batch.py:228-230)
                const PmtBinning& g = a.bins;
                const int rbin = min(nr, g.max_ref_count) / g.count_bin_skip, abin = (min(na, g.max_alt_count) - 1) / g.count_bin_skip;
                const long long cell = (((ds_col_at(a.sources, bb) * 3 + ds_col_at(a.labels, bb)) * g.num_variant_types + ds_col_at(a.variant_types, bb)) *
                                        g.num_ref_bins + rbin) * g.num_alt_bins + abin;
                w4 = (sub == 0 ? a.ref_weight_table : a.alt_weight_table) + 4 * cell;
            }
            comp[0] = pick_component(w4, uniform01(a.seed, sub, bb));
        }
        comp[1] = __shfl(comp[0], 1, DS_LANES);
        comp[0] = __shfl(comp[0], 0, DS_LANES);
#pragma unroll
        for (int side = 0; side < 2; ++side) {
            const int k = comp[side];
            const int sa = (k == 2 || k == 3) ? 5 : 1, sb = (k == 1 || k == 3) ? 5 : 1, n = sa + sb - 1;
            const float u = sub < n ? uniform01(a.seed, (side == 0 ? 32 : 64) + 16 + sub, bb) : 2.0f;  // (lanes beyond n: above every uniform)
            int rank = 0;  // how many of the n uniforms sort before this lane's (ties: the lower lane first, like the stable insertion sort)
#pragma unroll
            for (int o = 0; o < 9; ++o) {
                const float v = __shfl(u, o, DS_LANES);
                rank += (o < n && (v < u || (v == u && o < sub))) ? 1 : 0;
            }
            const unsigned long long holder = __ballot(sub < n && rank == sa - 1);  // one lane of each sub-group
            const int src = __ffsll((long long)((holder >> (threadIdx.x & 48)) & 0xFFFFull)) - 1;
            const float f = __shfl(u, src < 0 ? 0 : src, DS_LANES);
            if (side == 0) fr = f; else fa = f;
        }
    }
    const long long forced = forced_alt(a, a0, na);
    int cr = 0, ca = 0;
    if (live) {
        for (int i = sub; i < nr; i += DS_LANES) cr += keep_ref(a, (long long)r0 + i, fr) ? 1 : 0;
        for (int i = sub; i < na; i += DS_LANES) ca += keep_alt(a, (long long)a0 + i, fa, forced) ? 1 : 0;
    }
    for (int d = DS_LANES / 2; d > 0; d >>= 1) {
        cr += __shfl_xor(cr, d, DS_LANES);
        ca += __shfl_xor(ca, d, DS_LANES);
    }
    if (sub == 0 && live) {
        ref_fracs[b] = fr;
        alt_fracs[b] = fa;
        new_ref[b] = cr;
        new_alt[b] = ca;
    }
}

__global__ __launch_bounds__(64) void pmt_downsample_index_kernel(PmtDownsample a, const float* __restrict__ ref_fracs,
                                                                  const float* __restrict__ alt_fracs,
                                                                  const int* __restrict__ new_ref_off, const int* __restrict__ new_alt_off,
                                                                  long long* __restrict__ index) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const float fr = ref_fracs[b], fa = alt_fracs[b];
    const int r0 = a.ref_offsets[b], nr = a.ref_offsets[b + 1] - r0, a0 = a.alt_offsets[b], na = a.alt_offsets[b + 1] - a0;
    const long long total_ref = a.ref_offsets[a.num_variants], new_total_ref = new_ref_off[a.num_variants];
    const long long forced = forced_alt(a, a0, na);
    int base = new_ref_off[b];
    for (int i0 = 0; i0 < nr; i0 += 64) {  // kept rows in ascending order: rank = kept rows before this lane
        const int i = i0 + lane;
        const bool k = i < nr && keep_ref(a, (long long)r0 + i, fr);
        const unsigned long long m = __ballot(k);
        if (k) index[base + __popcll(m & ((1ull << lane) - 1ull))] = (long long)r0 + i;
        base += __popcll(m);
    }
    base = new_alt_off[b];
    // reference quirk: the kept alt indices index the ALT-ONLY mask but are used un-offset into the whole read
    // array; alt_row_offset = 0 reproduces it, total_ref gives the intended rows
    const long long alt_row_offset = a.reference_alt_gather ? 0 : total_ref;
    for (int i0 = 0; i0 < na; i0 += 64) {
        const int i = i0 + lane;
        const bool k = i < na && keep_alt(a, (long long)a0 + i, fa, forced);
        const unsigned long long m = __ballot(k);
        if (k) index[new_total_ref + base + __popcll(m & ((1ull << lane) - 1ull))] = alt_row_offset + a0 + i;
        base += __popcll(m);
    }
}

static int ds_check(const PmtDownsample* a) {
    if (!a || a->num_variants < 0) return PMT_E_INVALID;
    if (a->num_variants == 0) return PMT_OK;
    if (!a->ref_offsets || !a->alt_offsets || (a->ref_fracs_in != nullptr) != (a->alt_fracs_in != nullptr) || a->force_random < 0) return PMT_E_INVALID;
    return PMT_OK;
}

extern "C" int pmt_downsample_counts(const PmtDownsample* args, float* ref_fracs, float* alt_fracs, int32_t* new_ref_counts,
                                     int32_t* new_alt_counts, void* stream) {
    const int rc = ds_check(args);
    if (rc) return rc;
    if (!ref_fracs || !alt_fracs || !new_ref_counts || !new_alt_counts) return PMT_E_INVALID;
    if (args->num_variants == 0) return PMT_OK;
    if (args->ref_weight_table != nullptr || args->alt_weight_table != nullptr) {
        const PmtBinning& g = args->bins;
        if (!args->ref_weight_table || !args->alt_weight_table || !args->labels.ptr || !args->variant_types.ptr || g.count_bin_skip < 1 ||
            g.num_variant_types < 1 || g.num_ref_bins < 1 || g.num_alt_bins < 1)
            return PMT_E_INVALID;
        for (const PmtIntColumn* c : {&args->labels, &args->variant_types, &args->sources})
            if (c->ptr != nullptr && c->elem_bytes != 4 && c->elem_bytes != 8) return PMT_E_INVALID;
    }
    const int per_block = DS_THREADS / DS_LANES;
    hipLaunchKernelGGL(pmt_downsample_counts_kernel, dim3((args->num_variants + per_block - 1) / per_block), dim3(DS_THREADS), 0,
                       reinterpret_cast<hipStream_t>(stream), *args, ref_fracs, alt_fracs, new_ref_counts, new_alt_counts);
    return hipGetLastError() == hipSuccess ? PMT_OK : PMT_E_LAUNCH;
}

extern "C" int pmt_downsample_index(const PmtDownsample* args, const float* ref_fracs, const float* alt_fracs,
                                    const int32_t* new_ref_offsets, const int32_t* new_alt_offsets, int64_t* read_index, void* stream) {
    const int rc = ds_check(args);
    if (rc) return rc;
    if (!ref_fracs || !alt_fracs || !new_ref_offsets || !new_alt_offsets || !read_index) return PMT_E_INVALID;
    if (args->num_variants == 0) return PMT_OK;
    hipLaunchKernelGGL(pmt_downsample_index_kernel, dim3(args->num_variants), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), *args,
                       ref_fracs, alt_fracs, new_ref_offsets, new_alt_offsets, (long long*)read_index);
    return hipGetLastError() == hipSuccess ? PMT_OK : PMT_E_LAUNCH;
}
