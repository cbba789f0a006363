// Haplotype CNN, batched-column kernels: the production-shaped stack
//     Conv1d(10 -> C1, k1) -> MaxPool1d(2) -> act -> Conv1d(C1 -> C2, k2) -> act -> Flatten -> Linear(C2 * L2 -> O)
// (reference architecture/dna_sequence_convolution.py:31-111 on the one-hot of data/batch.py:115-130) as plain GEMMs on the
// fp32 matrix core (v_mfma_f32_16x16x4_f32: exact fp32, like the reference).
//
// pmt_cnn2.hip gives a wave ONE variant: a convolution's 7 or 19 output positions half-fill its 16-column tiles, every B
// operand is an im2col gather through a tap table (~10 instructions per element) and the input gradient is a col2im of LDS
// atomics: 18 vector instructions per MFMA, 6.7 % of the matrix peak.  Here a wave takes V = 8 variants at a time and a GEMM
// COLUMN is a (variant, position) pair, so the tiles are full.  Activations rest in the wave's private LDS region as
// [column][channel block of 16, in the register ("tile-position") order of pmt_device.hpp]:
//   * a convolution is one GEMM PER TAP, whose B operand is the input's column shifted by the tap: ONE ds_read_b128 per
//     16 channels, no index arithmetic, no tap table; its output registers are stored with one ds_write_b128 per 16 channels;
//   * the input gradient is the same gather with the transposed weights (out[q] = sum_k W_k^T dY[q - k]): plain stores, no
//     atomics;
//   * weight gradients contract over columns, their operands read from the same LDS arrays (a scalar per lane and k-step),
//     accumulated in registers over ALL the variants a wave sees (the kernels are persistent) and added to global memory once
//     per wave;
//   * the one-hot input is never materialised: a B operand of the first convolution is a byte compare on the haplotype;
//   * max-pooling pairs neighbouring columns of the first convolution's output with one DPP shift (its argmax goes into the
//     stash as one bit per channel); the first convolution's weight gradient uses the pooled gradient directly, split by argmax.
// No workgroup barrier after the prologue: a wave's LDS traffic completes in order (wave_sync).
//
// Covers exactly the layer pattern above with C1, C2 <= 32, O <= 16, kernels <= 7, stride 1, no padding / dilation, pool 2/2;
// everything else runs pmt_cnn2.hip / pmt_cnn.hip.  The backward needs the forward's stash (pooled activations, second
// convolution's output, pool argmax: P1 * 33 + L2 * 32 floats per variant).
#define PMT_OWN_WAVE_SHAPE
#define PMT_WAVES 8   // (pmt_device.hpp wants one; the kernels here take their wave count NW and batch size V as template parameters)
#define PMT_RT 1
#include "pmt_device.hpp"

// V variants per wave and batch, NW waves per workgroup: the forward runs <4, 8> (two waves per SIMD hide its LDS latencies), the
// backward <8, 4> (its register-resident weight gradients want the whole register file of a SIMD: measured 485 vs 633 us).
#ifndef C3_FWD_V
#define C3_FWD_V 4
#define C3_FWD_NW 8
#endif
#define C3_BWD_V 8
#define C3_BWD_NW 4
#define C3_MAXK 7
#define C3_NACC(K1, K2, L2) ((K1) * 2 + (K2) * 4 + (L2) * 2)   // 16x16 weight-gradient tiles a backward wave accumulates
#define C3_NBIAS 80                                             // d(bias1)[32], d(bias2)[32], d(linear bias)[16]

struct C3Cfg {
    int S, K1, L1, P1, K2, L2, C1, C2, F, O;   // sequence length, kernel 1, its output length, pooled length, kernel 2, its output length
    int st1, st2;                              // columns per variant of the conv1 / conv2 GEMMs (st1 even: pool pairs never straddle a tile)
    int n1t, n2t, nPt;                         // 16-column tiles of the conv1 / conv2 / pooled-gradient GEMMs over C3_V variants
    int act1, act2;                            // PMT_CNN_LEAKY_RELU | PMT_CNN_SELU
    int w1, b1, w2, b2, wl, bl;                // theta offsets
    int per_wave;                              // floats of LDS per wave
    int stash_per;                             // floats of stash per variant
};

DEV void c3_wave_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}
DEV int c3_row(int m) { return 4 * (m & 3) + (m >> 2); }  // A-fragment row m / block element e -> channel within the 16-block
DEV float c3_act(int kind, float x) { return kind == PMT_CNN_LEAKY_RELU ? fmaxf(x, 0.01f * x) : selu1(x); }  // (max: x for x > 0, 0.01 x below)
DEV float c3_act_grad(int kind, float y) { return kind == PMT_CNN_LEAKY_RELU ? (y > 0.f ? 1.f : 0.01f) : selu_grad_from_out(y); }
DEV f4 c3_act4(int kind, f4 v) { return f4{c3_act(kind, v[0]), c3_act(kind, v[1]), c3_act(kind, v[2]), c3_act(kind, v[3])}; }
DEV f4 c3_act_grad4(int kind, f4 y) { return f4{c3_act_grad(kind, y[0]), c3_act_grad(kind, y[1]), c3_act_grad(kind, y[2]), c3_act_grad(kind, y[3])}; }
DEV f4 c3_mfma4(f4 a, f4 b, f4 c) {
#pragma unroll
    for (int j = 0; j < 4; ++j) c = mfma16(a[j], b[j], c);
    return c;
}
DEV f4 c3_zero() { return f4{0.f, 0.f, 0.f, 0.f}; }
template <int CTRL>
DEV f4 c3_dpp4(f4 v) { return f4{dpp_mov<CTRL>(v[0]), dpp_mov<CTRL>(v[1]), dpp_mov<CTRL>(v[2]), dpp_mov<CTRL>(v[3])}; }

// ---- weights as A fragments in LDS (built once per workgroup from the natural layouts in theta) -----------------------------
// fragment element [lane = 16 g + m][j]: row = 16 mt + c3_row(m), k = 16 kt + 4 j + g   (pmt_device.hpp's convention)
struct C3Weights {
    const float *w1f, *w2f, *wlf;     // forward:  [K1][2][256], [K2][2 mt][2 kt][256], [L2][2][256]
    const float *w2tf, *wltf;         // backward: W2^T [K2][2 mt = ci][2 kt = co][256], Wl^T [L2][2][256] (rows = channels, k = outputs)
    const float *b1p, *b2p, *blp;     // biases in position order (element 4 g + j = feature 4 j + g): [2][16], [2][16], [16]
};
DEV int c3_weight_floats(const C3Cfg& c, bool backward) {
    const int fwd = c.K1 * 2 * 256 + c.K2 * 4 * 256 + c.L2 * 2 * 256 + 80;
    const int bwd = c.K2 * 4 * 256 + c.L2 * 2 * 256;
    return backward ? bwd : fwd;
}
DEV C3Weights c3_build_weights(const C3Cfg& c, const float* __restrict__ theta, float* __restrict__ lds, bool backward) {
    C3Weights W{};
    const int tid = threadIdx.x, nthr = blockDim.x;
    float* cur = lds;
    if (!backward) {
        float* w1f = cur; cur += c.K1 * 2 * 256;
        for (int i = tid; i < c.K1 * 2 * 256; i += nthr) {
            const int j = i & 3, lane = (i >> 2) & 63, mt = (i >> 8) & 1, tap = i >> 9;
            const int co = 16 * mt + c3_row(lane & 15), ci = 4 * j + (lane >> 4);
            w1f[i] = (co < c.C1 && ci < 10) ? theta[c.w1 + (co * 10 + ci) * c.K1 + tap] : 0.f;
        }
        float* w2f = cur; cur += c.K2 * 4 * 256;
        for (int i = tid; i < c.K2 * 4 * 256; i += nthr) {
            const int j = i & 3, lane = (i >> 2) & 63, kt = (i >> 8) & 1, mt = (i >> 9) & 1, tap = i >> 10;
            const int co = 16 * mt + c3_row(lane & 15), ci = 16 * kt + 4 * j + (lane >> 4);
            w2f[i] = (co < c.C2 && ci < c.C1) ? theta[c.w2 + (co * c.C1 + ci) * c.K2 + tap] : 0.f;
        }
        float* wlf = cur; cur += c.L2 * 2 * 256;
        for (int i = tid; i < c.L2 * 2 * 256; i += nthr) {
            const int j = i & 3, lane = (i >> 2) & 63, t = (i >> 8) & 1, p = i >> 9;
            const int o = c3_row(lane & 15), ch = 16 * t + 4 * j + (lane >> 4);
            wlf[i] = (o < c.O && ch < c.C2) ? theta[c.wl + o * c.F + ch * c.L2 + p] : 0.f;
        }
        float* bp = cur; cur += 80;
        for (int i = tid; i < 80; i += nthr) {
            const int e = i & 15, f = 16 * ((i >> 4) & 1) + c3_row(e);
            float v = 0.f;
            if (i < 32) v = f < c.C1 ? theta[c.b1 + f] : 0.f;
            else if (i < 64) v = f < c.C2 ? theta[c.b2 + f] : 0.f;
            else v = c3_row(e) < c.O ? theta[c.bl + c3_row(e)] : 0.f;
            bp[i] = v;
        }
        W.w1f = w1f; W.w2f = w2f; W.wlf = wlf; W.b1p = bp; W.b2p = bp + 32; W.blp = bp + 64;
    } else {
        float* w2tf = cur; cur += c.K2 * 4 * 256;
        for (int i = tid; i < c.K2 * 4 * 256; i += nthr) {
            const int j = i & 3, lane = (i >> 2) & 63, kt = (i >> 8) & 1, mt = (i >> 9) & 1, tap = i >> 10;
            const int ci = 16 * mt + c3_row(lane & 15), co = 16 * kt + 4 * j + (lane >> 4);
            w2tf[i] = (co < c.C2 && ci < c.C1) ? theta[c.w2 + (co * c.C1 + ci) * c.K2 + tap] : 0.f;
        }
        float* wltf = cur; cur += c.L2 * 2 * 256;
        for (int i = tid; i < c.L2 * 2 * 256; i += nthr) {
            const int j = i & 3, lane = (i >> 2) & 63, t = (i >> 8) & 1, p = i >> 9;
            const int ch = 16 * t + c3_row(lane & 15), o = 4 * j + (lane >> 4);
            wltf[i] = (o < c.O && ch < c.C2) ? theta[c.wl + o * c.F + ch * c.L2 + p] : 0.f;
        }
        W.w2tf = w2tf; W.wltf = wltf;
    }
    __syncthreads();
    return W;
}

// per-wave LDS region (floats): [V records][x: V * st2 * 32 (backward scratch)][dout: V * 16][hap: V * 2 S bytes].
// A RECORD is what the forward leaves of one variant and the backward needs: [a1: P1 * 32][a2: st2 * 32][pool argmax: P1 words],
// padded to a multiple of 4 floats -- in LDS exactly as in the stash, so a batch's stash moves with 16-byte copies whose
// loads are all issued before the first store (a load / store loop exposes one HBM latency per iteration).
#define C3_COPY_F4 20   // f4 per lane a batch copy may take: V * rec / 4 <= 64 * C3_COPY_F4
#define C3_HAP_LOADS 8  // V * 2 S <= 64 * C3_HAP_LOADS
struct C3Wave {
    float* rec;
    int rec_floats, a2_off, arg_off;
    float* x;
    float* dout;
    unsigned char* hap;
    int P1, st2;
    DEV float* a1(int v, int q) const { return rec + v * rec_floats + q * 32; }
    DEV float* a2(int v, int p) const { return rec + v * rec_floats + a2_off + p * 32; }
    DEV unsigned* argb(int v, int q) const { return reinterpret_cast<unsigned*>(rec + v * rec_floats + arg_off) + q; }
};
template <int V>
DEV C3Wave c3_wave_region(const C3Cfg& c, float* base) {
    C3Wave w;
    w.rec = base;
    w.rec_floats = c.stash_per;
    w.a2_off = c.P1 * 32;
    w.arg_off = c.P1 * 32 + c.st2 * 32;
    w.P1 = c.P1; w.st2 = c.st2;
    w.x = base + V * c.stash_per;
    w.dout = w.x + V * c.st2 * 32;
    w.hap = reinterpret_cast<unsigned char*>(w.dout + V * 16);
    return w;
}
template <int V>
DEV void c3_load_haplotypes(const C3Cfg& c, const C3Wave& w, const long long* __restrict__ hap, long long hap_stride, long long v0, int nv) {
    const int lane = threadIdx.x & 63, n2s = 2 * c.S;
    long long b[C3_HAP_LOADS];
#pragma unroll
    for (int k = 0; k < C3_HAP_LOADS; ++k) {  // every load is in flight before the first byte is stored
        const int i = lane + 64 * k, v = i / n2s, e = i - v * n2s;
        b[k] = 255;
        if (i < V * n2s && v < nv) b[k] = hap[(size_t)(v0 + v) * hap_stride + e];
    }
#pragma unroll
    for (int k = 0; k < C3_HAP_LOADS; ++k) {
        const int i = lane + 64 * k;
        if (i < V * n2s) w.hap[i] = (b[k] >= 0 && b[k] < 5) ? (unsigned char)b[k] : (unsigned char)255;  // anything else matches no one-hot channel
    }
}
// the records of a batch: stash -> LDS (zeros for variants beyond the end) / LDS -> stash
template <int V>
DEV void c3_load_records(const C3Wave& w, const float* __restrict__ src, int nv) {
    const int lane = threadIdx.x & 63, have = nv * w.rec_floats / 4, all = V * w.rec_floats / 4;
    f4 t[C3_COPY_F4];
#pragma unroll
    for (int k = 0; k < C3_COPY_F4; ++k) {
        const int i = lane + 64 * k;
        t[k] = f4{0.f, 0.f, 0.f, 0.f};
        if (i < have) t[k] = reinterpret_cast<const f4*>(src)[i];
    }
#pragma unroll
    for (int k = 0; k < C3_COPY_F4; ++k) {
        const int i = lane + 64 * k;
        if (i < all) reinterpret_cast<f4*>(w.rec)[i] = t[k];
    }
}
DEV void c3_store_records(const C3Wave& w, float* __restrict__ dst, int nv) {
    const int lane = threadIdx.x & 63, have = nv * w.rec_floats / 4;
#pragma unroll
    for (int k = 0; k < C3_COPY_F4; ++k) {
        const int i = lane + 64 * k;
        if (i < have) reinterpret_cast<f4*>(dst)[i] = reinterpret_cast<const f4*>(w.rec)[i];
    }
}
// one-hot B operand of the first convolution for (variant v, input position pos): channel ci = 4 j + g is base (ci >> 1) of the
// ref (ci even) / alt (ci odd) haplotype (reference data/batch.py:115-130)
DEV f4 c3_one_hot(int S, const C3Wave& w, int v, int pos, int g) {
    const int base = w.hap[v * 2 * S + (g & 1) * S + pos];
    const int b0 = g >> 1;
    return f4{base == b0 ? 1.f : 0.f, base == b0 + 2 ? 1.f : 0.f, (g < 2 && base == b0 + 4) ? 1.f : 0.f, 0.f};
}

// ================================================ forward ===========================================================
// Development: -DC3_TRACE=1 makes wave 0 of workgroup 0 log the cycle counter at every phase boundary over the stash rows of its
// own first batch (backward: dead once loaded; forward: from the second batch on, over what it stored for the first;
// scripts/cnn3_trace.py reads them back).  Timing only.
#ifndef C3_TRACE
#define C3_TRACE 0
#endif
#define C3_EV()                                                                                                      \
    do {                                                                                                             \
        if (C3_TRACE && tracing) { const unsigned long long t_ = __builtin_readcyclecounter(); if (lane == 0 && tr_n < 120) tr[tr_n] = t_; ++tr_n; } \
    } while (0)
// K1 / K2 / L2 (kernel sizes, second convolution's output length) are compile-time: the weight gradients are register arrays
// indexed by tap / position, and a run-time index would push them to scratch memory.
// K1 / K2 / L2 compile-time as in the backward: the tap loops unroll and their LDS reads are issued ahead of the matrix core.
// S1: the sequence length if known at compile time (0 = c.S): then the first convolution's tile loop unrolls too and its column ->
// (variant, position) split is integer arithmetic on constants.
template <int V, int NW, int K1, int K2, int L2, int S1>
__global__ __launch_bounds__(64 * NW, 1) void pmt_cnn3_forward_kernel(
    C3Cfg c, const float* __restrict__ theta, const long long* __restrict__ hap, long long hap_stride, int n, float* __restrict__ out,
    long long out_stride, float* __restrict__ stash) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const unsigned long long t_begin = C3_TRACE ? __builtin_readcyclecounter() : 0ull;
    const C3Weights W = c3_build_weights(c, theta, lds, false);
    const unsigned long long t_weights = C3_TRACE ? __builtin_readcyclecounter() : 0ull;
    const int lane = threadIdx.x & 63, g = lane >> 4, r = lane & 15, wave = uniform((int)(threadIdx.x >> 6));
    bool tracing = false;  // (switched on behind the first batch)
    unsigned long long* tr = reinterpret_cast<unsigned long long*>(stash);
    int tr_n = 2;
    const C3Wave w = c3_wave_region<V>(c, lds + ((c3_weight_floats(c, false) + 3) & ~3) + wave * c.per_wave);
    const int nbatches = (n + V - 1) / V;
    constexpr int P1 = L2 + K2 - 1, ST2 = L2 <= 4 ? 4 : 8, N2T = (V * ST2 + 15) / 16;
    const int S = S1 ? S1 : c.S, st1 = S1 ? ((S1 - K1 + 2) & ~1) : c.st1, n1t = S1 ? (V * st1 + 15) / 16 : c.n1t;  // (cnn3_config: st1 = (L1 + 1) & ~1)
    const float inv_st1 = 1.0f / (float)st1;
    const f4 b1v[2] = {*reinterpret_cast<const f4*>(W.b1p + 4 * g), *reinterpret_cast<const f4*>(W.b1p + 16 + 4 * g)};
    const f4 b2v[2] = {*reinterpret_cast<const f4*>(W.b2p + 4 * g), *reinterpret_cast<const f4*>(W.b2p + 16 + 4 * g)};
    const f4 blv = *reinterpret_cast<const f4*>(W.blp + 4 * g);
    for (int batch = blockIdx.x * NW + wave; batch < nbatches; batch += gridDim.x * NW) {
        const long long v0 = (long long)batch * V;
        const int nv = (int)min((long long)V, (long long)n - v0);
        C3_EV();  // batch begins
        c3_load_haplotypes<V>(c, w, hap, hap_stride, v0, nv);
        c3_wave_sync();
        C3_EV();  // haplotypes in LDS
        // ---- conv1 (+ bias) -> max-pool over column pairs -> activation -> a1 ------------------------------------------
#pragma unroll
        for (int T = 0; T < n1t; ++T) {
            const int col = 16 * T + r;
            int v = S1 ? col / st1 : (int)((float)col * inv_st1 + 1e-3f);
            int p = col - v * st1;
            const bool in_range = v < V;
            v = min(v, V - 1);
            f4 acc[2] = {b1v[0], b1v[1]};
#pragma unroll
            for (int tap = 0; tap < K1; ++tap) {
                const f4 b = c3_one_hot(S, w, v, min(p + tap, S - 1), g);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const f4 a = *reinterpret_cast<const f4*>(W.w1f + ((tap * 2 + mt) * 64 + lane) * 4);
#pragma unroll
                    for (int j = 0; j < 3; ++j) acc[mt] = mfma16(a[j], b[j], acc[mt]);  // ci = 4 j + g < 10: j = 3 is empty
                }
            }
            const int q = p >> 1;
            const bool store = in_range && !(p & 1) && q < P1;  // (p + 1 < L1 follows from q < P1 = L1 / 2)
            unsigned bits = 0;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const f4 nxt = c3_dpp4<0x101>(acc[mt]);  // row_shl:1 -- the column to the right (position p + 1 of the same variant)
                f4 m;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool second = nxt[j] > acc[mt][j];  // the first maximum wins, like ATen's max_pool1d
                    m[j] = second ? nxt[j] : acc[mt][j];
                    bits |= second ? 1u << (4 * mt + j) : 0u;
                }
                if (store) *reinterpret_cast<f4*>(w.a1(v, q) + 16 * mt + 4 * g) = c3_act4(c.act1, m);
            }
            if (store) reinterpret_cast<unsigned char*>(w.argb(v, q))[g] = (unsigned char)bits;
        }
        c3_wave_sync();
        C3_EV();  // conv1 done
        // ---- conv2 (+ bias) -> activation -> a2 --------------------------------------------------------------------------
#pragma unroll
        for (int T = 0; T < N2T; ++T) {
            const int col = 16 * T + r;
            int v = col / ST2;
            const int p = col - v * ST2;
            const bool valid = v < V && p < L2;
            v = min(v, V - 1);
            f4 acc[2] = {b2v[0], b2v[1]};
#pragma unroll
            for (int tap = 0; tap < K2; ++tap) {
                const float* src = w.a1(v, min(p + tap, P1 - 1)) + 4 * g;
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
                    const f4 b = *reinterpret_cast<const f4*>(src + 16 * kt);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
                        acc[mt] = c3_mfma4(*reinterpret_cast<const f4*>(W.w2f + (((tap * 2 + mt) * 2 + kt) * 64 + lane) * 4), b, acc[mt]);
                }
            }
            if (v < V && p < ST2 && col < V * ST2) {  // (padding columns hold zeros: they travel with the record)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) *reinterpret_cast<f4*>(w.a2(v, p) + 16 * mt + 4 * g) = valid ? c3_act4(c.act2, acc[mt]) : c3_zero();
            }
        }
        c3_wave_sync();
        C3_EV();  // conv2 done
        // ---- flatten + linear: a column per variant ------------------------------------------------------------------------
        {
            const int v = min(r, V - 1);
            f4 acc = blv;
#pragma unroll
            for (int p = 0; p < L2; ++p)
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    acc = c3_mfma4(*reinterpret_cast<const f4*>(W.wlf + ((p * 2 + t) * 64 + lane) * 4),
                                   *reinterpret_cast<const f4*>(w.a2(v, p) + 16 * t + 4 * g), acc);
            if (r < nv) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (4 * j + g < c.O) out[(size_t)(v0 + r) * out_stride + 4 * j + g] = acc[j];
            }
        }
        C3_EV();  // linear done
        // ---- stash for the backward: the records as they lie in LDS ----------------------------------------------------------
        if (stash) {
            c3_wave_sync();
            c3_store_records(w, stash + (size_t)v0 * c.stash_per, nv);
        }
        c3_wave_sync();
        if (C3_TRACE && stash != nullptr && blockIdx.x == 0 && wave == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (!tracing && lane == 0) { tr[0] = t_begin; tr[1] = t_weights; }
            tracing = true;
            C3_EV();  // records stored
            if (lane == 0) tr[127] = (unsigned long long)tr_n;
        }
    }
}

// ================================================ forward on the bf16 matrix pipe ======================================
// The same forward with every product as bf16 MFMAs on three-piece splits (pmt_device.hpp: an fp32 value IS hi + mid + lo in
// bf16; six v_mfma_f32_16x16x32_bf16 reproduce the fp32 product to a bit or two in 96 cycles per 32-deep block where eight
// v_mfma_f32_16x16x4_f32 take 256).  What changes against the kernel above:
//   * conv1's B operand is exact in ONE piece (a one-hot), and all of its K1 * 10 <= 32 taps x channels are one k block: 3 MFMAs
//     per 16 x 16 tile instead of 18.  The one-hot lies in LDS as bf16 [position][channel] per variant, so the operand of
//     column (v, p) -- k = tap * 10 + channel -- is the 64 contiguous bytes behind position p: four ds_read_b32, no compares;
//   * activations rest in LDS as bf16 pieces [column][piece][32 channels in record order]: a k block of conv2 / the linear is
//     one ds_read_b128 per piece; the producer splits its 4 channels per 16-block and stores 8 bytes per piece;
//   * conv1's and conv2's weights live in registers (96) for the whole kernel, the linear's in LDS; they are split once per
//     workgroup into a staging area that the waves' regions then reuse;
//   * a training forward writes its records (fp32 a1, a2, pool argmax: the backward's format) straight from registers.
// Selected for K1 * 10 <= 32 (PmtModel.cnn_debug bit 8 keeps the fp32 kernel above: A/B runs, and the parity tests run both).
#define C3B_OH_PAD 16
DEV int c3_chan_of_elem(int e) { return 16 * (e >> 4) + 4 * (e & 3) + ((e & 15) >> 2); }  // record element -> channel
typedef unsigned c3_u4 __attribute__((ext_vector_type(4)));
typedef unsigned c3_u2 __attribute__((ext_vector_type(2)));
// Round 4: the pieces are TWO f16 values per operand with the low one scaled by 2^12 (pmt_device.hpp, linear_acc_f16: fp32-equivalent
// like three bf16 pieces) -- three MFMAs per product instead of six (conv1: two instead of three), five vector operations per
// pair of activations instead of nine, 128 bytes of LDS per column instead of 192.  C3_F16 = 0 keeps the bf16 form (A/B runs).
#ifndef C3_F16
#define C3_F16 1
#endif
#if C3_F16
typedef h8 c3p8;
#define C3_NP 2
#define C3_ONE ((unsigned short)0x3C00)  // 1.0 as f16
DEV f4 c3p_mfma(c3p8 a, c3p8 b, f4 c) { return mfma_f16(a, b, c); }
DEV void c3p_split(float a, float b, unsigned (&pc)[3]) {
    float k = 4096.f;
    asm volatile("" : "+v"(k));
    split_pair_f16(a, b, pc[0], pc[1], k);
    pc[2] = 0u;
}
// acc += hi products, lo += the two first-order products (scaled by 2^12: the caller joins lo * 2^-12 once per accumulator)
DEV void c3p_mma(const c3p8 (&a)[C3_NP], const c3p8 (&b)[C3_NP], f4& acc, f4& lo) {
    lo = c3p_mfma(a[1], b[0], lo);
    lo = c3p_mfma(a[0], b[1], lo);
    acc = c3p_mfma(a[0], b[0], acc);
}
DEV f4 c3p_join(f4 acc, f4 lo) { return lo * (1.0f / 4096.f) + acc; }
#else
typedef bf8 c3p8;
#define C3_NP 3
#define C3_ONE ((unsigned short)0x3F80)  // 1.0 as bf16
DEV f4 c3p_mfma(c3p8 a, c3p8 b, f4 c) { return mfma_bf16(a, b, c); }
DEV void c3p_split(float a, float b, unsigned (&pc)[3]) { split_pair<3>(a, b, pc[0], pc[1], pc[2]); }
// acc += A B with both operands in three pieces (smallest terms first); lo unused
DEV void c3p_mma(const c3p8 (&a)[C3_NP], const c3p8 (&b)[C3_NP], f4& acc, f4& lo) {
    acc = mfma_bf16(a[2], b[0], acc);
    acc = mfma_bf16(a[0], b[2], acc);
    acc = mfma_bf16(a[1], b[1], acc);
    acc = mfma_bf16(a[1], b[0], acc);
    acc = mfma_bf16(a[0], b[1], acc);
    acc = mfma_bf16(a[0], b[0], acc);
}
DEV f4 c3p_join(f4 acc, f4 lo) { return acc; }
#endif
#define C3_COL_BYTES (64 * C3_NP)  // one column of activations in LDS: [piece][32 channels]
// the A fragment of lane `ln` (row m = ln & 15, k = 8 (ln >> 4) + i) in C3_NP pieces: dst[piece * 64 + ln]
template <typename F>
DEV void c3b_build_frag(c3p8* __restrict__ dst, int ln, F value) {
    unsigned pc[4][3];
#pragma unroll
    for (int q = 0; q < 4; ++q) c3p_split(value(2 * q), value(2 * q + 1), pc[q]);
#pragma unroll
    for (int k = 0; k < C3_NP; ++k) dst[64 * k + ln] = __builtin_bit_cast(c3p8, c3_u4{pc[0][k], pc[1][k], pc[2][k], pc[3][k]});
}
// four activations of one 16-block -> 8 bytes per piece at dst + piece * 64
DEV void c3b_store_pieces(unsigned char* dst, f4 v) {
    unsigned p0[3], p1[3];
    c3p_split(v[0], v[1], p0);
    c3p_split(v[2], v[3], p1);
#pragma unroll
    for (int k = 0; k < C3_NP; ++k) *reinterpret_cast<c3_u2*>(dst + 64 * k) = c3_u2{p0[k], p1[k]};
}
__host__ __device__ inline size_t c3b_wave_bytes(int V, int S, int P1, int ST2) {
    return (size_t)((V * (S * 10 + C3B_OH_PAD) * 2 + 15) & ~15) + (size_t)V * P1 * C3_COL_BYTES + (size_t)V * ST2 * C3_COL_BYTES;
}

// A1 / A2: the two activations at compile time (0 = read the configuration: both kinds are then evaluated per element);
// TRAIN: write the backward's records.
template <int V, int NW, int K1, int K2, int L2, int S1, int A1, int A2, bool TRAIN>
__global__ __launch_bounds__(64 * NW, 1) void pmt_cnn3_forward_bf_kernel(
    C3Cfg c, const float* __restrict__ theta, const long long* __restrict__ hap, long long hap_stride, int n, float* __restrict__ out,
    long long out_stride, float* __restrict__ stash) {
    static_assert(K1 * 10 <= 32, "conv1's taps x channels are one k block");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int P1 = L2 + K2 - 1, ST2 = L2 <= 4 ? 4 : 8, N2T = (V * ST2 + 15) / 16;
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, r = lane & 15, wave = uniform((int)(tid >> 6));
    const int S = S1 ? S1 : c.S, st1 = S1 ? ((S1 - K1 + 2) & ~1) : c.st1, n1t = S1 ? (V * st1 + 15) / 16 : c.n1t;
    const float inv_st1 = 1.0f / (float)st1;
    const int OHS = S * 10 + C3B_OH_PAD;  // bf16 elements of one variant's one-hot array
    // ---- LDS: [linear weights: L2 x 3 pieces x 1 KiB][biases: 80 floats][staging of conv1 / conv2 weights | the waves' regions]
    unsigned char* base = reinterpret_cast<unsigned char*>(lds);
    if (C3_F16) fp16_saturate_on();  // (f16 pieces beyond +-65504 saturate, never inf: pmt_device.hpp)
    c3p8* wlb = reinterpret_cast<c3p8*>(base);
    float* bp = reinterpret_cast<float*>(base + L2 * C3_NP * 1024);
    unsigned char* dyn = base + L2 * C3_NP * 1024 + 320;
    c3p8* w1s = reinterpret_cast<c3p8*>(dyn);
    c3p8* w2s = w1s + 2 * 64 * C3_NP;
    for (int e = tid; e < 2 * 64; e += 64 * NW) {
        const int ln = e & 63, mt = e >> 6, co = 16 * mt + c3_row(ln & 15), kg = ln >> 4;
        c3b_build_frag(w1s + mt * 64 * C3_NP, ln, [&](int i) {
            const int k = 8 * kg + i, tap = k / 10, ci = k - 10 * tap;
            return (k < 10 * K1 && co < c.C1) ? theta[c.w1 + (co * 10 + ci) * K1 + tap] : 0.f;
        });
    }
    for (int e = tid; e < K2 * 2 * 64; e += 64 * NW) {
        const int ln = e & 63, tm = e >> 6, tap = tm >> 1, mt = tm & 1, co = 16 * mt + c3_row(ln & 15), kg = ln >> 4;
        c3b_build_frag(w2s + tm * 64 * C3_NP, ln, [&](int i) {
            const int ci = c3_chan_of_elem(8 * kg + i);
            return (co < c.C2 && ci < c.C1) ? theta[c.w2 + (co * c.C1 + ci) * K2 + tap] : 0.f;
        });
    }
    for (int e = tid; e < L2 * 64; e += 64 * NW) {
        const int ln = e & 63, p = e >> 6, o = c3_row(ln & 15), kg = ln >> 4;
        c3b_build_frag(wlb + p * 64 * C3_NP, ln, [&](int i) {
            const int ch = c3_chan_of_elem(8 * kg + i);
            return (o < c.O && ch < c.C2) ? theta[c.wl + o * c.F + ch * L2 + p] : 0.f;
        });
    }
    for (int i = tid; i < 80; i += 64 * NW) {
        const int e = i & 15, f = 16 * ((i >> 4) & 1) + c3_row(e);
        float v = 0.f;
        if (i < 32) v = f < c.C1 ? theta[c.b1 + f] : 0.f;
        else if (i < 64) v = f < c.C2 ? theta[c.b2 + f] : 0.f;
        else v = c3_row(e) < c.O ? theta[c.bl + c3_row(e)] : 0.f;
        bp[i] = v;
    }
    __syncthreads();
    c3p8 w1r[2][C3_NP], w2r[K2][2][C3_NP];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int pc = 0; pc < C3_NP; ++pc) w1r[mt][pc] = w1s[(mt * C3_NP + pc) * 64 + lane];
#pragma unroll
    for (int tap = 0; tap < K2; ++tap)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int pc = 0; pc < C3_NP; ++pc) w2r[tap][mt][pc] = w2s[((tap * 2 + mt) * C3_NP + pc) * 64 + lane];
    const f4 b1v[2] = {*reinterpret_cast<const f4*>(bp + 4 * g), *reinterpret_cast<const f4*>(bp + 16 + 4 * g)};
    const f4 b2v[2] = {*reinterpret_cast<const f4*>(bp + 32 + 4 * g), *reinterpret_cast<const f4*>(bp + 48 + 4 * g)};
    const f4 blv = *reinterpret_cast<const f4*>(bp + 64 + 4 * g);
    __syncthreads();  // the staging area is dead: the waves' regions take its place
    unsigned char* oh = dyn + (size_t)wave * c3b_wave_bytes(V, S, P1, ST2);
    unsigned char* a1b = oh + ((V * OHS * 2 + 15) & ~15);
    unsigned char* a2b = a1b + V * P1 * C3_COL_BYTES;
    for (int i = lane; i < (V * OHS) / 2; i += 64) reinterpret_cast<unsigned*>(oh)[i] = 0u;  // (the pads stay zero for good)
    c3_wave_sync();
    const int nbatches = (n + V - 1) / V, n2s = 2 * S;
    const int rec_a2 = P1 * 32, rec_arg = P1 * 32 + ST2 * 32;
    // the haplotypes of a batch are requested one batch ahead: their HBM latency passes under the previous batch's arithmetic
    long long bnext[C3_HAP_LOADS];
    auto request = [&](int batch) {
        const long long v0 = (long long)batch * V;
#pragma unroll
        for (int k = 0; k < C3_HAP_LOADS; ++k) {
            const int i = lane + 64 * k, v = i / n2s, e = i - v * n2s;
            bnext[k] = 255;
            if (batch < nbatches && i < V * n2s && v0 + v < n) bnext[k] = hap[(size_t)(v0 + v) * hap_stride + e];
        }
    };
    request(blockIdx.x * NW + wave);
    for (int batch = blockIdx.x * NW + wave; batch < nbatches; batch += gridDim.x * NW) {
        const long long v0 = (long long)batch * V;
        const int nv = (int)min((long long)V, (long long)n - v0);
        // ---- haplotypes -> one-hot, bf16 [position][channel = 2 base + (0 ref | 1 alt)] (reference data/batch.py:115-130) ----
        {
            long long b[C3_HAP_LOADS];
#pragma unroll
            for (int k = 0; k < C3_HAP_LOADS; ++k) b[k] = bnext[k];
            request(batch + gridDim.x * NW);
#pragma unroll
            for (int k = 0; k < C3_HAP_LOADS; ++k) {
                const int i = lane + 64 * k, v = i / n2s, e = i - v * n2s, hs = e >= S ? 1 : 0, pos = e - hs * S;
                if (i < V * n2s) {
                    unsigned short* dst = reinterpret_cast<unsigned short*>(oh) + v * OHS + pos * 10 + hs;
#pragma unroll
                    for (int bs = 0; bs < 5; ++bs) dst[2 * bs] = b[k] == bs ? C3_ONE : (unsigned short)0;
                }
            }
        }
        c3_wave_sync();
        float* rec0 = TRAIN ? stash + (size_t)v0 * c.stash_per : nullptr;
        // ---- conv1 (+ bias) -> max-pool over column pairs -> activation -> a1 (bf16 pieces; fp32 record when training) --------
#pragma unroll
        for (int T = 0; T < n1t; ++T) {
            const int col = 16 * T + r;
            int v = S1 ? col / st1 : (int)((float)col * inv_st1 + 1e-3f);
            const int p = col - v * st1;
            const bool in_range = v < V;
            v = min(v, V - 1);
            const unsigned* src = reinterpret_cast<const unsigned*>(oh + 2 * (v * OHS + p * 10 + 8 * g));
            const c3p8 b = __builtin_bit_cast(c3p8, c3_u4{src[0], src[1], src[2], src[3]});
            f4 acc[2] = {b1v[0], b1v[1]};
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {  // (the one-hot is exact in one piece: the weights' pieces against it)
#if C3_F16
                const f4 lo = c3p_mfma(w1r[mt][1], b, c3_zero());
                acc[mt] = c3p_join(c3p_mfma(w1r[mt][0], b, acc[mt]), lo);
#else
                acc[mt] = mfma_bf16(w1r[mt][2], b, acc[mt]);
                acc[mt] = mfma_bf16(w1r[mt][1], b, acc[mt]);
                acc[mt] = mfma_bf16(w1r[mt][0], b, acc[mt]);
#endif
            }
            const int q = p >> 1;
            const bool store = in_range && !(p & 1) && q < P1;  // (p + 1 < L1 follows from q < P1 = L1 / 2)
            unsigned bits = 0;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const f4 nxt = c3_dpp4<0x101>(acc[mt]);  // row_shl:1 -- the column to the right (position p + 1 of the same variant)
                f4 m;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (TRAIN) {
                        const bool second = nxt[j] > acc[mt][j];  // the first maximum wins, like ATen's max_pool1d
                        m[j] = second ? nxt[j] : acc[mt][j];
                        bits |= second ? 1u << (4 * mt + j) : 0u;
                    } else {
                        m[j] = fmaxf(acc[mt][j], nxt[j]);  // (the value alone: no argmax to keep)
                    }
                }
                m = c3_act4(A1 ? A1 : c.act1, m);
                if (store) {
                    c3b_store_pieces(a1b + (v * P1 + q) * C3_COL_BYTES + (16 * mt + 4 * g) * 2, m);
                    if (TRAIN && v < nv) *reinterpret_cast<f4*>(rec0 + (size_t)v * c.stash_per + q * 32 + 16 * mt + 4 * g) = m;
                }
            }
            if (TRAIN && store && v < nv) reinterpret_cast<unsigned char*>(rec0 + (size_t)v * c.stash_per + rec_arg + q)[g] = (unsigned char)bits;
        }
        c3_wave_sync();
        // ---- conv2 (+ bias) -> activation -> a2 -----------------------------------------------------------------------------
#pragma unroll
        for (int T = 0; T < N2T; ++T) {
            const int col = 16 * T + r;
            int v = col / ST2;
            const int p = col - v * ST2;
            const bool valid = v < V && p < L2;
            const bool in_range = v < V;
            v = min(v, V - 1);
            f4 acc[2] = {b2v[0], b2v[1]}, lo2[2] = {c3_zero(), c3_zero()};
#pragma unroll
            for (int tap = 0; tap < K2; ++tap) {
                const unsigned char* src = a1b + (v * P1 + min(p + tap, P1 - 1)) * C3_COL_BYTES + 16 * g;
                c3p8 b[C3_NP];
#pragma unroll
                for (int k = 0; k < C3_NP; ++k) b[k] = *reinterpret_cast<const c3p8*>(src + 64 * k);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) c3p_mma(w2r[tap][mt], b, acc[mt], lo2[mt]);
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const f4 a = valid ? c3_act4(A2 ? A2 : c.act2, c3p_join(acc[mt], lo2[mt])) : c3_zero();
                if (valid) c3b_store_pieces(a2b + (v * ST2 + p) * C3_COL_BYTES + (16 * mt + 4 * g) * 2, a);
                // (the record's padding columns hold zeros, as the backward expects)
                if (TRAIN && in_range && v < nv) *reinterpret_cast<f4*>(rec0 + (size_t)v * c.stash_per + rec_a2 + p * 32 + 16 * mt + 4 * g) = a;
            }
        }
        c3_wave_sync();
        // ---- flatten + linear: a column per variant --------------------------------------------------------------------------
        {
            const int v = min(r, V - 1);
            f4 acc = blv, lol = c3_zero();
#pragma unroll
            for (int p = 0; p < L2; ++p) {
                const unsigned char* src = a2b + (v * ST2 + p) * C3_COL_BYTES + 16 * g;
                c3p8 b[C3_NP], a[C3_NP];
#pragma unroll
                for (int k = 0; k < C3_NP; ++k) {
                    b[k] = *reinterpret_cast<const c3p8*>(src + 64 * k);
                    a[k] = wlb[(p * C3_NP + k) * 64 + lane];
                }
                c3p_mma(a, b, acc, lol);
            }
            acc = c3p_join(acc, lol);
            if (r < nv) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (4 * j + g < c.O) out[(size_t)(v0 + r) * out_stride + 4 * j + g] = acc[j];
            }
        }
        c3_wave_sync();
    }
}

// ================================================ backward ==========================================================
template <int V, int NW, int K1, int K2, int L2>
__global__ __launch_bounds__(64 * NW, 1) void pmt_cnn3_backward_kernel(
    C3Cfg c, const float* __restrict__ theta, const long long* __restrict__ hap, long long hap_stride, int n, const float* __restrict__ d_out,
    long long d_out_stride, const float* __restrict__ stash, float* __restrict__ gtheta, float* __restrict__ ws, int ws_stride) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const unsigned long long t_begin = C3_TRACE ? __builtin_readcyclecounter() : 0ull;
    const C3Weights W = c3_build_weights(c, theta, lds, true);
    const unsigned long long t_weights = C3_TRACE ? __builtin_readcyclecounter() : 0ull;
    const int lane = threadIdx.x & 63, g = lane >> 4, r = lane & 15, wave = uniform((int)(threadIdx.x >> 6));
    const bool tracing = C3_TRACE && blockIdx.x == 0 && wave == 0;
    unsigned long long* tr = reinterpret_cast<unsigned long long*>(const_cast<float*>(stash));
    int tr_n = 2;
    const int m = r, kk = g;  // names of the same lane coordinates when the lane feeds an A / B operand: row (or column) m, k-slot kk
    const C3Wave w = c3_wave_region<V>(c, lds + ((c3_weight_floats(c, true) + 3) & ~3) + wave * c.per_wave);
    const int nbatches = (n + V - 1) / V;
    // the geometry behind the second convolution follows from the template parameters (cnn3_config checks the model against it):
    // compile-time trip counts let the compiler unroll the contraction loops and issue their LDS reads ahead of the matrix core
    constexpr int P1 = L2 + K2 - 1, ST2 = L2 <= 4 ? 4 : 8, N2T = (V * ST2) / 16, NPT = (V * P1 + 15) / 16;
    // register-resident weight gradients over every variant this wave sees.  C layout: acc[j] of lane (g, col) = row 4 g + j.
    f4 gw1[K1][2], gw2[K2][2][2], gwl[L2][2];  // rows = out channels (position order) | linear outputs; cols below
    f4 gb1[2] = {c3_zero(), c3_zero()}, gb2[2] = {c3_zero(), c3_zero()};  // per-lane partial sums over this lane's columns
    float gbl = 0.f;
#pragma unroll
    for (int k = 0; k < K1; ++k) gw1[k][0] = gw1[k][1] = c3_zero();
#pragma unroll
    for (int k = 0; k < K2; ++k) gw2[k][0][0] = gw2[k][0][1] = gw2[k][1][0] = gw2[k][1][1] = c3_zero();
#pragma unroll
    for (int k = 0; k < L2; ++k) gwl[k][0] = gwl[k][1] = c3_zero();
    for (int batch = blockIdx.x * NW + wave; batch < nbatches; batch += gridDim.x * NW) {
        const long long v0 = (long long)batch * V;
        const int nv = (int)min((long long)V, (long long)n - v0);
        // ---- this batch's inputs: haplotypes, upstream gradient, the forward's stash ------------------------------------
        c3_load_haplotypes<V>(c, w, hap, hap_stride, v0, nv);
        c3_load_records<V>(w, stash + (size_t)v0 * c.stash_per, nv);
#pragma unroll
        for (int k = 0; k < (V * 16) / 64; ++k) {
            const int i = lane + 64 * k, v = i >> 4, o = i & 15;
            w.dout[i] = (v < nv && o < c.O) ? d_out[(size_t)(v0 + v) * d_out_stride + o] : 0.f;
        }
        c3_wave_sync();
        if (C3_TRACE && tracing && lane == 0 && tr_n == 2) { tr[0] = t_begin; tr[1] = t_weights; }
        C3_EV();  // inputs in LDS
        // ---- 1. linear: d(a2) = Wl^T d(out), a column per variant; d(bias) -------------------------------------------------
        {
            const int v = min(r, V - 1);
            f4 bo;  // B operand: k = output o = 4 j + g of variant column r
#pragma unroll
            for (int j = 0; j < 4; ++j) bo[j] = (r < V) ? w.dout[v * 16 + 4 * j + g] : 0.f;
#pragma unroll
            for (int p = 0; p < L2; ++p)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const f4 d = c3_mfma4(*reinterpret_cast<const f4*>(W.wltf + ((p * 2 + t) * 64 + lane) * 4), bo, c3_zero());
                    if (r < V) *reinterpret_cast<f4*>(w.x + (v * ST2 + p) * 32 + 16 * t + 4 * g) = d;  // rows = channels 16 t + 4 j + g
                }
        }
        c3_wave_sync();
        C3_EV();  // 1 done
        // ---- 2. through the second activation: dY2 = d(a2) * act2'(a2), zero on padding columns; back to LDS for the
        //         transposed reads; linear weight gradient ------------------------------------------------------------------
        for (int T = 0; T < N2T; ++T) {
            const int col = 16 * T + r;
            int v = col / ST2;
            const int p = col - v * ST2;
            const bool valid = v < V && p < L2;
            v = min(v, V - 1);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float* px = w.x + (v * ST2 + p) * 32 + 16 * t + 4 * g;
                const f4 y = *reinterpret_cast<const f4*>(w.a2(v, p) + 16 * t + 4 * g);
                f4 d = *reinterpret_cast<const f4*>(px) * c3_act_grad4(c.act2, y);
                if (!valid) d = c3_zero();
                if (v < V && col < V * ST2) *reinterpret_cast<f4*>(px) = d;
                gb2[t] = gb2[t] + d;
            }
        }
        {   // dWl[o][(ch, p)] += sum_v d(out)[v][o] a2[v][p][ch]: k = variants
#pragma unroll
            for (int s = 0; s < V / 4; ++s) {
                const int v = 4 * s + kk;
                const float a = w.dout[v * 16 + m];  // A row m = output o (zero beyond O and beyond nv)
                gbl += a;
#pragma unroll
                for (int p = 0; p < L2; ++p)
#pragma unroll
                    for (int t = 0; t < 2; ++t) gwl[p][t] = mfma16(a, w.a2(v, p)[16 * t + m], gwl[p][t]);  // B col m = element m of the block
            }
        }
        c3_wave_sync();
        C3_EV();  // 2 done
        // ---- 3. second convolution's weight gradient: dW2_k[co][ci] += sum_cols dY2[co][col] a1[ci][col + k] ----------------
#pragma unroll
        for (int s = 0; s < (V * ST2) / 4; ++s) {
            const int col = 4 * s + kk;
            const int v = col / ST2, p = col - v * ST2;
            const float a0 = w.x[col * 32 + m], a1v = w.x[col * 32 + 16 + m];  // rows = element m of channel block 0 / 1
#pragma unroll
            for (int tap = 0; tap < K2; ++tap) {
                const float* src = w.a1(v, min(p + tap, P1 - 1)) + m;  // (padding columns: dY2 is zero there)
                const float b0 = src[0], b1 = src[16];
                gw2[tap][0][0] = mfma16(a0, b0, gw2[tap][0][0]);
                gw2[tap][0][1] = mfma16(a0, b1, gw2[tap][0][1]);
                gw2[tap][1][0] = mfma16(a1v, b0, gw2[tap][1][0]);
                gw2[tap][1][1] = mfma16(a1v, b1, gw2[tap][1][1]);
            }
        }
        c3_wave_sync();
        C3_EV();  // 3 done
        // ---- 4. second convolution's input gradient (gather with the transposed weights), through the first activation:
        //         d(pooled)[v][q] -> written over a1[v][q] (each lane reads its block of a1 before it overwrites it) ---------------
#pragma unroll
        for (int T = 0; T < NPT; ++T) {
            const int idx = 16 * T + r;  // flat (variant, pooled position)
            int v = idx / P1;
            const int q = idx - v * P1;
            const bool valid = v < V;
            v = min(v, V - 1);
            f4 acc[2] = {c3_zero(), c3_zero()};
#pragma unroll
            for (int tap = 0; tap < K2; ++tap) {
                const int p = q - tap;
                const bool ok = valid && p >= 0 && p < L2;
                const float* src = w.x + (v * ST2 + (ok ? p : 0)) * 32 + 4 * g;
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
                    f4 b = *reinterpret_cast<const f4*>(src + 16 * kt);
                    if (!ok) b = c3_zero();
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
                        acc[mt] = c3_mfma4(*reinterpret_cast<const f4*>(W.w2tf + (((tap * 2 + mt) * 2 + kt) * 64 + lane) * 4), b, acc[mt]);
                }
            }
            if (valid) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    float* pa = w.a1(v, q) + 16 * mt + 4 * g;
                    const f4 d = acc[mt] * c3_act_grad4(c.act1, *reinterpret_cast<const f4*>(pa));
                    *reinterpret_cast<f4*>(pa) = d;
                    gb1[mt] = gb1[mt] + d;  // the pool routes every pooled gradient to exactly one position: d(bias1) = its sum
                }
            }
        }
        c3_wave_sync();
        C3_EV();  // 4 done
        // ---- 5. first convolution's weight gradient from the pooled gradient, split by the pool's argmax:
        //         dW1_k[co][ci] += sum_(v,q) [arg = 0] dP one_hot[2 q + k] + [arg = 1] dP one_hot[2 q + 1 + k] -----------------------
#pragma unroll
        for (int s = 0; s < (V * P1 + 3) / 4; ++s) {
            const int idx = 4 * s + kk;
            const bool valid = idx < V * P1;
            const int ii = valid ? idx : 0;
            const int v = ii / P1, q = ii - v * P1;
            const unsigned word = *w.argb(v, q);
            float a[2][2];  // [argmax][channel block]
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const float d = valid ? w.a1(v, q)[16 * mt + m] : 0.f;
                const bool second = (word >> (8 * (m >> 2) + 4 * mt + (m & 3))) & 1u;  // byte g' = m >> 2, bit 4 mt + j' (j' = m & 3)
                a[0][mt] = second ? 0.f : d;
                a[1][mt] = second ? d : 0.f;
            }
            // one-hot B operands of input positions 2 q .. 2 q + K1: column m = input channel (base m >> 1 of the ref / alt haplotype)
            float oh[K1 + 1];
#pragma unroll
            for (int off = 0; off <= K1; ++off) {
                oh[off] = 0.f;
                if (m < 10) {
                    const int base = w.hap[v * 2 * c.S + (m & 1) * c.S + min(2 * q + off, c.S - 1)];
                    oh[off] = base == (m >> 1) ? 1.f : 0.f;
                }
            }
#pragma unroll
            for (int tap = 0; tap < K1; ++tap)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    gw1[tap][mt] = mfma16(a[0][mt], oh[tap], gw1[tap][mt]);
                    gw1[tap][mt] = mfma16(a[1][mt], oh[tap + 1], gw1[tap][mt]);
                }
        }
        c3_wave_sync();
        C3_EV();  // 5 done
    }
    // ---- the register-resident gradients leave the kernel once per workgroup ----------------------------------------------
    // Every wave holds the same C3_NACC accumulator tiles.  Global float atomics from here would be 64 * NW * gridDim.x adds to
    // EACH address, all issued within the same few microseconds: they serialise in the L2 (measured: 260 of the kernel's 505 us
    // were this epilogue).  With a workspace the waves of a workgroup sum their tiles through LDS and store ONE raw row per
    // workgroup ([tile][lane] f4 + the bias sums); pmt_cnn3_fold_kernel maps the rows to theta offsets and adds them up.
    if (ws != nullptr) {
        constexpr int NACC = C3_NACC(K1, K2, L2);
        __syncthreads();  // every wave is done with its LDS region (and with the weights)
        f4* red = reinterpret_cast<f4*>(lds);                   // [NW][NACC][64]
        float* redb = lds + NW * NACC * 256;                    // [NW][C3_NBIAS]
        f4* mine = red + wave * NACC * 64 + lane;
#pragma unroll
        for (int tap = 0; tap < K1; ++tap)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) mine[(tap * 2 + mt) * 64] = gw1[tap][mt];
#pragma unroll
        for (int tap = 0; tap < K2; ++tap)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) mine[(K1 * 2 + (tap * 2 + mt) * 2 + kt) * 64] = gw2[tap][mt][kt];
#pragma unroll
        for (int p = 0; p < L2; ++p)
#pragma unroll
            for (int t = 0; t < 2; ++t) mine[(K1 * 2 + K2 * 4 + p * 2 + t) * 64] = gwl[p][t];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float s1 = gb1[t][j], s2 = gb2[t][j];
                s1 += dpp_mov<0xB1>(s1); s1 += dpp_mov<0x4E>(s1); s1 += dpp_mov<0x124>(s1); s1 += dpp_mov<0x128>(s1);
                s2 += dpp_mov<0xB1>(s2); s2 += dpp_mov<0x4E>(s2); s2 += dpp_mov<0x124>(s2); s2 += dpp_mov<0x128>(s2);
                if (r == 0) {
                    redb[wave * C3_NBIAS + 16 * t + 4 * j + g] = s1;
                    redb[wave * C3_NBIAS + 32 + 16 * t + 4 * j + g] = s2;
                }
            }
        {
            const float tot = group_sum(gbl);
            if (g == 0) redb[wave * C3_NBIAS + 64 + m] = tot;
        }
        __syncthreads();
        float* row = ws + (size_t)blockIdx.x * ws_stride;
        for (int a = wave; a < NACC; a += NW) {
            f4 sum = red[a * 64 + lane];
#pragma unroll
            for (int w2 = 1; w2 < NW; ++w2) sum = sum + red[(w2 * NACC + a) * 64 + lane];
            reinterpret_cast<f4*>(row)[a * 64 + lane] = sum;
        }
        if ((int)threadIdx.x < C3_NBIAS) {
            float sum = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < NW; ++w2) sum += redb[w2 * C3_NBIAS + threadIdx.x];
            row[NACC * 256 + threadIdx.x] = sum;
        }
        if (C3_TRACE && tracing) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            C3_EV();  // gradients out
            if (lane == 0) tr[127] = (unsigned long long)tr_n;
        }
        return;
    }
    // no workspace: global float atomics straight from the registers.
    // C layout: acc[j] of lane (g, col) is row rho = 4 g + j; rows are in position order (channel 16 t + c3_row(rho)), columns too.
#pragma unroll
    for (int tap = 0; tap < K1; ++tap)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int co = 16 * mt + c3_row(4 * g + j), ci = r;  // conv1's columns are the input channels themselves
                if (co < c.C1 && ci < 10) atomicAdd(&gtheta[c.w1 + (co * 10 + ci) * K1 + tap], gw1[tap][mt][j]);
            }
#pragma unroll
    for (int tap = 0; tap < K2; ++tap)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int co = 16 * mt + c3_row(4 * g + j), ci = 16 * kt + c3_row(r);
                    if (co < c.C2 && ci < c.C1) atomicAdd(&gtheta[c.w2 + (co * c.C1 + ci) * K2 + tap], gw2[tap][mt][kt][j]);
                }
#pragma unroll
    for (int p = 0; p < L2; ++p)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int o = 4 * g + j, ch = 16 * t + c3_row(r);  // A rows were the outputs themselves
                if (o < c.O && ch < c.C2) atomicAdd(&gtheta[c.wl + o * c.F + ch * L2 + p], gwl[p][t][j]);
            }
    // biases: per-lane partial sums over columns (lanes r) of the C-layout gradients: rows 16 t + 4 j + g
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float s1 = gb1[t][j], s2 = gb2[t][j];
            s1 += dpp_mov<0xB1>(s1); s1 += dpp_mov<0x4E>(s1); s1 += dpp_mov<0x124>(s1); s1 += dpp_mov<0x128>(s1);
            s2 += dpp_mov<0xB1>(s2); s2 += dpp_mov<0x4E>(s2); s2 += dpp_mov<0x124>(s2); s2 += dpp_mov<0x128>(s2);
            const int ch = 16 * t + 4 * j + g;
            if (r == 0 && ch < c.C1) atomicAdd(&gtheta[c.b1 + ch], s1);
            if (r == 0 && ch < c.C2) atomicAdd(&gtheta[c.b2 + ch], s2);
        }
    {
        const float tot = group_sum(gbl);  // over the k-slots: lanes (m, *) hold d(bias)[o = m]
        if (g == 0 && m < c.O) atomicAdd(&gtheta[c.bl + m], tot);
    }
    if (C3_TRACE && tracing) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        C3_EV();  // gradients out
        if (lane == 0) tr[127] = (unsigned long long)tr_n;
    }
}

// ---- fold: the workgroups' raw rows -> theta offsets ------------------------------------------------------------------------
// element i < NACC * 256 of a row is accumulator tile a = i / 256, lane (i / 4) % 64, register j = i % 4 (the C layout: row 4 g + j,
// column r); the C3_NBIAS entries behind them are d(bias1)[32], d(bias2)[32], d(linear bias)[16] by channel.
#define C3_FOLD_SLICES 8
__global__ __launch_bounds__(256) void pmt_cnn3_fold_kernel(C3Cfg c, const float* __restrict__ ws, int rows, int ws_stride, float* __restrict__ gtheta) {
    const int nacc = C3_NACC(c.K1, c.K2, c.L2);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nacc * 256 + C3_NBIAS) return;
    int off = -1;
    if (i < nacc * 256) {
        const int j = i & 3, lane = (i >> 2) & 63, a = i >> 8, g = lane >> 4, r = lane & 15;
        if (a < c.K1 * 2) {
            const int tap = a >> 1, mt = a & 1, co = 16 * mt + c3_row(4 * g + j), ci = r;
            if (co < c.C1 && ci < 10) off = c.w1 + (co * 10 + ci) * c.K1 + tap;
        } else if (a < c.K1 * 2 + c.K2 * 4) {
            const int idx = a - c.K1 * 2, tap = idx >> 2, mt = (idx >> 1) & 1, kt = idx & 1;
            const int co = 16 * mt + c3_row(4 * g + j), ci = 16 * kt + c3_row(r);
            if (co < c.C2 && ci < c.C1) off = c.w2 + (co * c.C1 + ci) * c.K2 + tap;
        } else {
            const int idx = a - c.K1 * 2 - c.K2 * 4, p = idx >> 1, t = idx & 1;
            const int o = 4 * g + j, ch = 16 * t + c3_row(r);
            if (o < c.O && ch < c.C2) off = c.wl + o * c.F + ch * c.L2 + p;
        }
    } else {
        const int e = i - nacc * 256;
        if (e < 32) off = e < c.C1 ? c.b1 + e : -1;
        else if (e < 64) off = e - 32 < c.C2 ? c.b2 + e - 32 : -1;
        else off = e - 64 < c.O ? c.bl + e - 64 : -1;
    }
    if (off < 0) return;
    float sum = 0.f;
    const float* p = ws + i;
    int row = blockIdx.y;
    for (; row + 3 * C3_FOLD_SLICES < rows; row += 4 * C3_FOLD_SLICES) {  // four loads in flight
        const float v0 = p[(size_t)row * ws_stride], v1 = p[(size_t)(row + C3_FOLD_SLICES) * ws_stride];
        const float v2 = p[(size_t)(row + 2 * C3_FOLD_SLICES) * ws_stride], v3 = p[(size_t)(row + 3 * C3_FOLD_SLICES) * ws_stride];
        sum += (v0 + v1) + (v2 + v3);
    }
    for (; row < rows; row += C3_FOLD_SLICES) sum += p[(size_t)row * ws_stride];
    atomicAdd(gtheta + off, sum);  // (C3_FOLD_SLICES adds per address)
}

// ---- host side ---------------------------------------------------------------------------------------------------------
// `V`: variants per wave and batch of the kernel the configuration is for (tile counts and the wave's LDS region depend on it)
static bool cnn3_config(const PmtModel* m, C3Cfg* c, int V) {
    if (m->force_cnn == 1 || m->force_cnn == 2) return false;
    const PmtCnn* n = &m->cnn;
    if (n->n_layers != 7) return false;
    const PmtCnnLayer* L = n->layers;
    const bool act1 = L[2].kind == PMT_CNN_LEAKY_RELU || L[2].kind == PMT_CNN_SELU, act2 = L[4].kind == PMT_CNN_LEAKY_RELU || L[4].kind == PMT_CNN_SELU;
    if (L[0].kind != PMT_CNN_CONV || L[1].kind != PMT_CNN_POOL || !act1 || L[3].kind != PMT_CNN_CONV || !act2 ||
        L[5].kind != PMT_CNN_FLATTEN || L[6].kind != PMT_CNN_LINEAR)
        return false;
    for (int l = 0; l < 4; l += 3)
        if (L[l].stride != 1 || L[l].padding != 0 || L[l].dilation != 1 || L[l].kernel < 1 || L[l].kernel > C3_MAXK || L[l].out_ch > 32) return false;
    if (L[1].kernel != 2 || L[1].stride != 2 || L[1].padding != 0 || L[1].dilation != 1) return false;
    if (L[0].in_ch != 10 || L[6].out_ch > 16 || L[3].out_len > C3_MAXK || L[3].out_len < 1) return false;
    if (L[0].kernel != 3 || L[3].kernel != 3 || L[3].out_len != 7) return false;  // the backward instance compiled below: <K1 = 3, K2 = 3, L2 = 7>
    c->S = n->seq_len; c->K1 = L[0].kernel; c->L1 = L[0].out_len; c->P1 = L[1].out_len; c->K2 = L[3].kernel; c->L2 = L[3].out_len;
    c->C1 = L[0].out_ch; c->C2 = L[3].out_ch; c->F = c->C2 * c->L2; c->O = L[6].out_ch;
    if (c->L1 != c->S - c->K1 + 1 || c->P1 != c->L1 / 2 || c->L2 != c->P1 - c->K2 + 1 || L[3].in_ch != c->C1 || L[6].in_ch * L[6].in_len != c->F) return false;
    c->st1 = (c->L1 + 1) & ~1;
    c->st2 = c->L2 <= 4 ? 4 : 8;  // V * st2 is a multiple of 16: whole tiles, and the columns split evenly over the k-slots
    c->n1t = (V * c->st1 + 15) / 16;
    c->n2t = (V * c->st2 + 15) / 16;
    c->nPt = (V * c->P1 + 15) / 16;
    c->act1 = L[2].kind; c->act2 = L[4].kind;
    c->w1 = L[0].w_src; c->b1 = L[0].b_src; c->w2 = L[3].w_src; c->b2 = L[3].b_src; c->wl = L[6].w_src; c->bl = L[6].b_src;
    c->stash_per = (c->P1 * 32 + c->st2 * 32 + c->P1 + 3) & ~3;  // one record (C3Wave)
    if (V * c->stash_per > 4 * 64 * C3_COPY_F4 || V * 2 * c->S > 64 * C3_HAP_LOADS || (V * c->st2) % 16 != 0) return false;
    const int hap_floats = (V * 2 * c->S + 3) / 4;
    c->per_wave = (V * c->stash_per + V * c->st2 * 32 + V * 16 + hap_floats + 3) & ~3;
    return true;
}
static size_t cnn3_lds_bytes(const C3Cfg* c, bool backward, int nw) {
    const int wf = backward ? c->K2 * 4 * 256 + c->L2 * 2 * 256 : c->K1 * 2 * 256 + c->K2 * 4 * 256 + c->L2 * 2 * 256 + 80;
    return ((size_t)((wf + 3) & ~3) + (size_t)nw * c->per_wave) * sizeof(float);
}
static bool cnn3_covers(const PmtModel* m, C3Cfg* fwd, C3Cfg* bwd) {
    return cnn3_config(m, fwd, C3_FWD_V) && cnn3_config(m, bwd, C3_BWD_V) && cnn3_lds_bytes(fwd, false, C3_FWD_NW) <= 160 * 1024 &&
           cnn3_lds_bytes(bwd, true, C3_BWD_NW) <= 160 * 1024;
}

// floats of stash per variant, 0 when these kernels do not cover the model (pmt_cnn_stash_floats asks here first)
extern "C" size_t pmt_cnn3_stash_floats(const PmtModel* m) {
    C3Cfg f, b;
    return (m && cnn3_covers(m, &f, &b)) ? (size_t)b.stash_per : 0;
}

// the device a launch on `stream` runs on (NOT the calling thread's current device: a caller may hold another one current)
static int cnn3_stream_device(hipStream_t stream) {
    int dev = 0;
    if (hipStreamGetDevice(stream, &dev) != hipSuccess && hipGetDevice(&dev) != hipSuccess) dev = 0;
    return dev;
}
static int cnn3_grid(int n, int v, int nw, int dev) {
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const long long batches = ((long long)n + v - 1) / v, wgs = (batches + nw - 1) / nw;
    return (int)(wgs < cus ? wgs : cus);
}

// More than 64 KiB of dynamic LDS needs the function attribute.  It is a property of the loaded code object on a device, not
// library state that results depend on: raised (never lowered) under a mutex, per device and kernel, remembered so that the
// steady state makes no runtime call.  The call is not a stream operation and must stay out of a stream capture
// (tests/graph_capture.py warms up eagerly first).  Devices beyond the table are simply set every time.
#include <mutex>
static bool cnn3_allow_lds(const void* kernel, size_t bytes, int which, int dev) {
    static std::mutex mu;
    static size_t allowed[64][3] = {};
    std::lock_guard<std::mutex> lock(mu);
    const bool tabled = dev >= 0 && dev < 64;
    if (tabled && bytes <= allowed[dev][which]) return true;
    int cur = -1;
    const bool switched = hipGetDevice(&cur) == hipSuccess && cur != dev && hipSetDevice(dev) == hipSuccess;
    const bool ok = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess;
    if (switched) (void)hipSetDevice(cur);
    if (ok && tabled) allowed[dev][which] = bytes;
    return ok;
}

// 0 = done, 1 = configuration not covered (the caller runs the other kernels), < 0 = error
extern "C" int pmt_cnn3_try_forward(const PmtModel* model_host, const float* theta, const int64_t* haplotypes, int64_t hap_stride, int32_t n,
                                    float* out, int64_t out_stride, float* stash, void* stream) {
    C3Cfg c, cb;
    if (!cnn3_covers(model_host, &c, &cb)) return 1;
    const bool bf = (model_host->cnn_debug & 256) == 0;  // the bf16-pipe forward (bit 8: the fp32-MFMA kernel, for A/B runs and the parity tests)
    size_t lds = cnn3_lds_bytes(&c, false, C3_FWD_NW);
    auto kernel = c.S == 21 ? pmt_cnn3_forward_kernel<C3_FWD_V, C3_FWD_NW, 3, 3, 7, 21>   // the reference's 20 + 1 bases of context
                            : pmt_cnn3_forward_kernel<C3_FWD_V, C3_FWD_NW, 3, 3, 7, 0>;  // (cnn3_config admits exactly these instances)
    if (bf) {
        const bool leaky = c.act1 == PMT_CNN_LEAKY_RELU && c.act2 == PMT_CNN_LEAKY_RELU, s21 = c.S == 21;  // the production stack's instance
        if (leaky && s21) kernel = stash ? pmt_cnn3_forward_bf_kernel<C3_FWD_V, C3_FWD_NW, 3, 3, 7, 21, PMT_CNN_LEAKY_RELU, PMT_CNN_LEAKY_RELU, true>
                                         : pmt_cnn3_forward_bf_kernel<C3_FWD_V, C3_FWD_NW, 3, 3, 7, 21, PMT_CNN_LEAKY_RELU, PMT_CNN_LEAKY_RELU, false>;
        else kernel = stash ? pmt_cnn3_forward_bf_kernel<C3_FWD_V, C3_FWD_NW, 3, 3, 7, 0, 0, 0, true> : pmt_cnn3_forward_bf_kernel<C3_FWD_V, C3_FWD_NW, 3, 3, 7, 0, 0, 0, false>;
        const size_t regions = (size_t)C3_FWD_NW * c3b_wave_bytes(C3_FWD_V, c.S, c.P1, c.st2), staging = (size_t)(2 + 2 * c.K2) * C3_NP * 1024;
        lds = (size_t)c.L2 * C3_NP * 1024 + 320 + (regions > staging ? regions : staging);
        if (lds > 160 * 1024 || C3_FWD_V * 2 * c.S > 64 * C3_HAP_LOADS) return 1;
    }
    const int dev = cnn3_stream_device(reinterpret_cast<hipStream_t>(stream));
    if (!cnn3_allow_lds(reinterpret_cast<const void*>(kernel), lds, bf ? 2 : 0, dev)) return PMT_E_LAUNCH;
    hipLaunchKernelGGL(kernel, dim3(cnn3_grid(n, C3_FWD_V, C3_FWD_NW, dev)), dim3(64 * C3_FWD_NW), lds, reinterpret_cast<hipStream_t>(stream), c, theta,
                       (const long long*)haplotypes, (long long)hap_stride, n, out, (long long)out_stride, stash);
    return hipGetLastError() == hipSuccess ? PMT_OK : PMT_E_LAUNCH;
}

// floats of one workgroup's row of raw weight-gradient sums (16-byte multiple)
static int cnn3_ws_stride(const C3Cfg* c) { return C3_NACC(c->K1, c->K2, c->L2) * 256 + ((C3_NBIAS + 3) & ~3); }
// the workgroup's LDS has to hold its waves' accumulator tiles at the end of the kernel
static bool cnn3_ws_fits(const C3Cfg* c) {
    return cnn3_lds_bytes(c, true, C3_BWD_NW) >= (size_t)C3_BWD_NW * (C3_NACC(c->K1, c->K2, c->L2) * 256 + C3_NBIAS) * sizeof(float);
}

// floats of workspace with which pmt_cnn3_try_backward avoids gradient atomics (one row per workgroup it can launch); 0 = not used
extern "C" size_t pmt_cnn3_workspace_floats(const PmtModel* m) {
    C3Cfg f, b;
    if (!m || !cnn3_covers(m, &f, &b) || !cnn3_ws_fits(&b)) return 0;
    // One row per workgroup the backward can launch = per compute unit of the device it runs on.  The call has no stream to
    // name that device and the calling thread's current device may be another one (a rank that holds a different card
    // current), so the size covers the LARGEST device this process can see: right whichever card the stream belongs to.
    int ndev = 0, grid = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) ndev = 1;
    for (int dev = 0; dev < ndev; ++dev) {
        const int gd = cnn3_grid(1 << 30, C3_BWD_V, C3_BWD_NW, dev);
        grid = gd > grid ? gd : grid;
    }
    return (size_t)grid * cnn3_ws_stride(&b);
}

extern "C" int pmt_cnn3_try_backward(const PmtModel* model_host, const float* theta, const int64_t* haplotypes, int64_t hap_stride, int32_t n,
                                     const float* d_out, int64_t d_out_stride, const float* stash, float* grad_theta, float* workspace,
                                     size_t workspace_floats, void* stream) {
    C3Cfg cf, c;
    if (!stash || !cnn3_covers(model_host, &cf, &c)) return 1;
    const size_t lds = cnn3_lds_bytes(&c, true, C3_BWD_NW);
    auto kernel = pmt_cnn3_backward_kernel<C3_BWD_V, C3_BWD_NW, 3, 3, 7>;  // (cnn3_config admits exactly the instances compiled here)
    const int dev = cnn3_stream_device(reinterpret_cast<hipStream_t>(stream));
    if (!cnn3_allow_lds(reinterpret_cast<const void*>(kernel), lds, 1, dev)) return PMT_E_LAUNCH;
    const int grid = cnn3_grid(n, C3_BWD_V, C3_BWD_NW, dev), stride = cnn3_ws_stride(&c);
    // (a few workgroups: their atomics do not queue, and the fold would be one more launch on a latency-bound step)
    const bool rows = workspace != nullptr && grid >= 8 && cnn3_ws_fits(&c) && workspace_floats >= (size_t)grid * stride;  // else: global atomics
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(64 * C3_BWD_NW), lds, reinterpret_cast<hipStream_t>(stream), c, theta,
                       (const long long*)haplotypes, (long long)hap_stride, n, d_out, (long long)d_out_stride, stash, grad_theta,
                       rows ? workspace : nullptr, stride);
    if (rows) {
        const int elems = C3_NACC(c.K1, c.K2, c.L2) * 256 + C3_NBIAS;
        hipLaunchKernelGGL(pmt_cnn3_fold_kernel, dim3((elems + 255) / 256, C3_FOLD_SLICES), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), c,
                           workspace, grid, stride, grad_theta);
    }
    return hipGetLastError() == hipSuccess ? PMT_OK : PMT_E_LAUNCH;
}
