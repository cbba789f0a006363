// Haplotype CNN, wave-per-variant kernels (the fast path of pmt_cnn_forward / pmt_cnn_backward).
//
// pmt_cnn.hip gives a workgroup a few variants and separates every layer by a workgroup barrier: ~25 short,
// latency-bound phases per chunk (58 % of its wave cycles are spent parked).  Here ONE WAVE owns one variant at a time:
// its activations live in a private LDS region, LDS operations of one wave complete in order, so no barrier separates
// the layers and the waves of a CU drift apart and hide each other's latencies.  The kernels are persistent (a wave
// walks over many variants) and keep every convolution weight gradient in registers, contracted over the output
// positions directly from LDS in the MFMA operand layout (no transposes, no exchange); they reach global memory once per
// workgroup.  Only the final linear layer's dW is cooperative: after each round (one variant per wave, two workgroup
// barriers) every wave accumulates its slice of input columns over all the round's variants.
//
// Covers: <= 2 convolutions with out_ch <= 64 (instances for <= 32 and <= 64 output channels) and in_ch * kernel <= 96, pooling / activations, FLATTEN + one final LINEAR
// with <= 16 outputs (the P0 and T0 configurations); anything else runs the general kernels of pmt_cnn.hip.
#define PMT_OWN_WAVE_SHAPE
#define PMT_WAVES 8
#define PMT_RT 1
#include <stdlib.h>
#include <string.h>

#include "pmt_device.hpp"
#include "pmt_bwd_device.hpp"

#define C2_MAX_CONVS 2
#define C2_MAX_NTO 4  // out-channel tiles of a convolution: the kernels are instantiated for 2 (<= 32 channels) and 4 (<= 64: the reference's
                      // test configuration T0, one convolution 10 -> 64, which ran the general kernels at 13 x the time until round 5)
#define C2_NTI 6
#define C2_MAX_LIN_OUT 16
#define C2_LIN_REGS 16
#define C2_LEAKY 0.01f

DEV void wave_sync() {  // orders this wave's LDS traffic across lanes (the LDS pipe itself is in order per wave)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}
DEV int c2_div(int a, int b, float inv_b) {  // a / b for small non-negative a without the integer-division sequence
    int q = (int)((float)a * inv_b);
    if (q * b > a) --q;
    if ((q + 1) * b <= a) ++q;
    return q;
}
DEV float c2_act(int kind, float x) {
    if (kind == PMT_CNN_LEAKY_RELU) return x > 0.f ? x : C2_LEAKY * x;
    return selu1(x);
}
DEV float c2_act_grad_from_out(int kind, float y) {  // activations are in place: the derivative comes from the OUTPUT
    if (kind == PMT_CNN_LEAKY_RELU) return y > 0.f ? 1.f : C2_LEAKY;
    return selu_grad_from_out(y);
}

DEV void c2_build_taps(int* __restrict__ tap, const PmtCnnLayer& L) {
    const int K = L.in_ch * L.kernel;
    for (int f = threadIdx.x; f < PMT_MAX_ROW_INPUT; f += blockDim.x) {
        int v = -1;
        if (f < K) {
            const int ci = f / L.kernel, k = f - ci * L.kernel;
            v = (ci * L.in_len) | ((k * L.dilation - L.padding + 64) << 16);
        }
        tap[f] = v;
    }
}

// one-hot of variant v: channel 2*base + (0 ref | 1 alt), position s (reference data/batch.py:115-130)
DEV void c2_one_hot(float* __restrict__ dst, const long long* __restrict__ hap_row, int seq_len) {
    const int lane = threadIdx.x & 63;
    for (int i = lane; i < 2 * seq_len; i += 64) {
        const int side = i >= seq_len ? 1 : 0, s = i - side * seq_len;
        const long long base = hap_row[i];
#pragma unroll
        for (int b = 0; b < 5; ++b) dst[(2 * b + side) * seq_len + s] = base == b ? 1.f : 0.f;
    }
}

// B operand of a convolution in the tile-position layout: x[t][j] = tap feat_of(t, j, g) of output position so
DEV void c2_gather(f4 (&x)[1][C2_NTI], const float* __restrict__ in, const int* __restrict__ tap, const PmtCnnLayer& L, int so,
                   bool valid, int nkt, int g) {
#pragma unroll
    for (int t = 0; t < C2_NTI; ++t) {
        x[0][t] = f4{0.f, 0.f, 0.f, 0.f};
        if (t < nkt) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int tp = tap[feat_of(t, j, g)];
                const int s = so * L.stride + (tp >> 16) - 64;
                if (valid && tp >= 0 && s >= 0 && s < L.in_len) x[0][t][j] = in[(tp & 0xFFFF) + s];
            }
        }
    }
}

// Where the kernels read their weights.  The kernels are persistent and every variant uses the same weights, so they
// are copied into LDS once per workgroup (c2_stage_weights) when the host found room: a fragment then arrives in ~100
// cycles instead of an L2 round trip per fragment with a one-deep prefetch (~36 such loads per variant before).
struct C2Weights {
    const float* w[C2_MAX_CONVS];    // A fragments of the convolution weight viewed as [out_ch][in_ch * kernel]
    const float* wt[C2_MAX_CONVS];   // ... of its transpose (input gradient)
    const float* b[C2_MAX_CONVS];    // bias in tile-position order
    const float* lin_w;              // final linear [out][in], row major
};

DEV void c2_copy(float* __restrict__ dst, const float* __restrict__ src, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
}
// `stage`: LDS region of `stage_floats` floats (0: read everything from global memory).  Ends with a workgroup barrier.
DEV C2Weights c2_stage_weights(const PmtModel* __restrict__ M, const float* __restrict__ theta, const float* __restrict__ packed,
                               float* __restrict__ stage, int stage_floats, bool with_transposes) {
    const PmtCnn& C = M->cnn;
    C2Weights W;
    int conv = 0;
    float* cur = stage;
    W.lin_w = nullptr;
    for (int c = 0; c < C2_MAX_CONVS; ++c) W.w[c] = W.wt[c] = W.b[c] = nullptr;
    for (int l = 0; l < C.n_layers; ++l) {
        const PmtCnnLayer& L = C.layers[l];
        if (L.kind == PMT_CNN_CONV) {
            const PmtLinear& Wl = M->lin[L.lin];
            const int nfl = frag_floats_dev(Wl), nb = 16 * ((Wl.out_dim + 15) >> 4);
            W.w[conv] = packed + Wl.w_frag;
            W.wt[conv] = packed + Wl.wt_frag;
            W.b[conv] = packed + Wl.b_pvec;
            if (stage_floats > 0) {  // (+256: linear_acc prefetches one fragment past the end)
                c2_copy(cur, W.w[conv], nfl); W.w[conv] = cur; cur += nfl + 256;
                c2_copy(cur, W.b[conv], nb); W.b[conv] = cur; cur += nb;
                if (with_transposes && L.in_off != 0) { c2_copy(cur, W.wt[conv], nfl); W.wt[conv] = cur; cur += nfl + 256; }
            }
            ++conv;
        } else if (L.kind == PMT_CNN_LINEAR) {
            W.lin_w = theta + L.w_src;
            if (stage_floats > 0) {
                const int nfl = (L.out_ch * L.in_ch * L.in_len + 3) & ~3;
                c2_copy(cur, W.lin_w, L.out_ch * L.in_ch * L.in_len); W.lin_w = cur; cur += nfl;
            }
        }
    }
    __syncthreads();
    return W;
}

template <int NTO>
DEV void c2_conv_forward(const PmtModel* __restrict__ M, const PmtCnnLayer& L, const float* __restrict__ w_frag, const float* __restrict__ b_pvec,
                         const float* __restrict__ in, float* __restrict__ out, const int* __restrict__ tap) {
    const int lane = threadIdx.x & 63, g = lane >> 4, r = lane & 15;
    const PmtLinear& W = M->lin[uniform(L.lin)];
    const int K = uniform(W.in_dim), OC = uniform(W.out_dim), out_len = uniform(L.out_len), nkt = (K + 15) >> 4;
    for (int tile = 0; tile * 16 < out_len; ++tile) {
        const int so = tile * 16 + r;
        const bool valid = so < out_len;
        f4 x[1][C2_NTI], y[1][NTO];
        c2_gather(x, in, tap, L, so, valid, nkt, g);
        init_bias<NTO>(y, b_pvec, OC, g);
        linear_acc<C2_NTI, NTO, false>(y, x, w_frag, K, OC);
        if (valid) {
#pragma unroll
            for (int t = 0; t < NTO; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int co = feat_of(t, j, g);
                    if (co < OC) out[co * out_len + so] = y[0][t][j];
                }
        }
    }
}

// every layer of one variant, activations at acts + in_off / out_off (this wave's LDS region)
template <int NTO>
DEV void c2_forward_variant(const PmtModel* __restrict__ M, const float* __restrict__ theta, const C2Weights& cw,
                            float* __restrict__ acts, const long long* __restrict__ hap_row, const int (*taps)[PMT_MAX_ROW_INPUT]) {
    const PmtCnn& C = M->cnn;
    const int lane = threadIdx.x & 63, nl = uniform(C.n_layers);
    c2_one_hot(acts, hap_row, uniform(C.seq_len));
    wave_sync();
    int conv = 0;
    for (int l = 0; l < nl; ++l) {
        const PmtCnnLayer& L = C.layers[l];
        const int kind = uniform(L.kind);
        if (kind == PMT_CNN_FLATTEN) continue;
        const float* in = acts + uniform(L.in_off);
        float* out = acts + uniform(L.out_off);
        if (kind == PMT_CNN_CONV) {
            c2_conv_forward<NTO>(M, L, conv == 0 ? cw.w[0] : cw.w[1], conv == 0 ? cw.b[0] : cw.b[1], in, out, taps[conv]);
            ++conv;
        } else if (kind == PMT_CNN_POOL) {
            const int per = L.out_ch * L.out_len;
            const float inv = 1.0f / (float)L.out_len;
            for (int i = lane; i < per; i += 64) {
                const int c = c2_div(i, L.out_len, inv), so = i - c * L.out_len;
                float m = -INFINITY;
                for (int k = 0; k < L.kernel; ++k) {
                    const int s = so * L.stride + k;
                    if (s < L.in_len) m = fmaxf(m, in[c * L.in_len + s]);
                }
                out[i] = m;
            }
        } else if (kind == PMT_CNN_LEAKY_RELU || kind == PMT_CNN_SELU) {
            const int per = L.out_ch * L.out_len;
            for (int i = lane; i < per; i += 64) out[i] = c2_act(kind, in[i]);
        } else {  // LINEAR: the lanes split every dot product
            const float* Wt = cw.lin_w;
            const int nin = L.in_ch * L.in_len;
            for (int o = 0; o < L.out_ch; ++o) {
                float acc = 0.f;
                for (int k = lane; k < nin; k += 64) acc += Wt[(size_t)o * nin + k] * in[k];
                acc = wave_sum(acc);
                if (lane == 0) out[o] = acc + theta[L.b_src + o];
            }
        }
        wave_sync();
    }
}

template <int NTO>
__global__ __launch_bounds__(PMT_THREADS, 2) void pmt_cnn2_forward_kernel(
    const PmtModel* __restrict__ M, const float* __restrict__ theta, const float* __restrict__ packed,
    const long long* __restrict__ hap, long long hap_stride, int n, int per_wave, int stage_floats, float* __restrict__ out,
    long long out_stride, float* __restrict__ stash) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ int taps[C2_MAX_CONVS][PMT_MAX_ROW_INPUT];
    const PmtCnn& C = M->cnn;
    const int lane = threadIdx.x & 63, wave = uniform((int)(threadIdx.x >> 6)), nw = blockDim.x >> 6;
    int conv = 0;
    for (int l = 0; l < C.n_layers; ++l)
        if (C.layers[l].kind == PMT_CNN_CONV) c2_build_taps(taps[conv++], C.layers[l]);
    const C2Weights cw = c2_stage_weights(M, theta, packed, lds + (((size_t)nw * per_wave + 3) & ~(size_t)3), stage_floats, false);
    float* acts = lds + (size_t)wave * per_wave;
    const int od = uniform(C.out_dim);
    // the last layer's output offset
    int last_off = 0;
    for (int l = 0; l < C.n_layers; ++l) last_off = C.layers[l].out_off;
    for (long long v = (long long)blockIdx.x * nw + wave; v < n; v += (long long)gridDim.x * nw) {
        c2_forward_variant<NTO>(M, theta, cw, acts, hap + (size_t)v * hap_stride, taps);
        for (int o = lane; o < od; o += 64) out[(size_t)v * out_stride + o] = acts[last_off + o];
        if (stash) {  // every layer output (the one-hot input is rebuilt by the backward: 10 S floats it need not read)
            const int first = 10 * uniform(C.seq_len), per = uniform(C.sum_act) - first;
            float* dst = stash + (size_t)v * per;
            for (int i = lane; i < per; i += 64) dst[i] = acts[first + i];
        }
        wave_sync();
    }
}

// dW of one convolution for one variant, straight from LDS in the MFMA operand layout: for k-step s of tile `tile`, lane
// (m = lane & 15, kg = lane >> 4) supplies A = dY[16 ot + m][col] and B = im2col[16 it + m][col], col = 16 tile + 4 s + kg.
template <int NTO, int NTI>
DEV void c2_conv_wgrad(f4 (&acc)[NTO][NTI], float (&bsum)[NTO], const PmtCnnLayer& L, const float* __restrict__ gout,
                       const float* __restrict__ xin, const int* __restrict__ tap, int K, int OC) {
    const int lane = threadIdx.x & 63, m = lane & 15, kg = lane >> 4;
    const int out_len = L.out_len, nmt = (OC + 15) >> 4, nkt = (K + 15) >> 4;
    int tp[NTI];
#pragma unroll
    for (int it = 0; it < NTI; ++it) tp[it] = it < nkt ? tap[16 * it + m] : -1;
    for (int tile = 0; tile * 16 < out_len; ++tile) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int col = tile * 16 + 4 * s + kg;
            const bool cv = col < out_len;
            float a[NTO];
#pragma unroll
            for (int ot = 0; ot < NTO; ++ot) {
                const int oc = 16 * ot + m;
                a[ot] = (cv && ot < nmt && oc < OC) ? gout[oc * out_len + col] : 0.f;
                bsum[ot] += a[ot];
            }
#pragma unroll
            for (int it = 0; it < NTI; ++it) {
                if (it < nkt) {
                    const int pos = col * L.stride + (tp[it] >> 16) - 64;
                    const float b = (cv && tp[it] >= 0 && pos >= 0 && pos < L.in_len) ? xin[(tp[it] & 0xFFFF) + pos] : 0.f;
#pragma unroll
                    for (int ot = 0; ot < NTO; ++ot)
                        if (ot < nmt) acc[ot][it] = mfma16(a[ot], b, acc[ot][it]);
                }
            }
        }
    }
}

// NTI0 / NTI1: 16-wide k-tiles (in_ch * kernel) of the first / second convolution.  Their weight gradients live in
// registers for the whole kernel, so the instance is sized to the model (P0: 30 and 96 inputs = 2 and 6 tiles; sizing
// both for 6 cost 153 spilled VGPRs).
template <int NTO, int NTI0, int NTI1>
__global__ __launch_bounds__(PMT_THREADS, 2) void pmt_cnn2_backward_kernel(
    const PmtModel* __restrict__ M, const float* __restrict__ theta, const float* __restrict__ packed,
    const long long* __restrict__ hap, long long hap_stride, int n, int per_wave, int stage_floats, const float* __restrict__ d_out,
    long long d_out_stride, const float* __restrict__ stash, float* __restrict__ gtheta, int dbg) {
    // dbg (development, PMT_CNN_DBG; 0 in production, results are wrong otherwise): 1 skip the forward recompute, 2 skip the
    // convolutions' input gradients, 4 skip their weight gradients, 8 skip the final linear's dW, 16 skip pool / activation /
    // linear input gradients
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ int taps[C2_MAX_CONVS][PMT_MAX_ROW_INPUT];
    __shared__ float dout_sh[PMT_WAVES][C2_MAX_LIN_OUT];
    const PmtCnn& C = M->cnn;
    const int lane = threadIdx.x & 63, g = lane >> 4, r = lane & 15, wave = uniform((int)(threadIdx.x >> 6)), nw = blockDim.x >> 6;
    const int sa = uniform(C.sum_act), ma = uniform(C.max_act), nl = uniform(C.n_layers), od = uniform(C.out_dim);
    int conv_layer[C2_MAX_CONVS] = {-1, -1}, lin_layer = -1, nconv = 0;
    for (int l = 0; l < nl; ++l) {
        if (C.layers[l].kind == PMT_CNN_CONV) {
            c2_build_taps(taps[nconv], C.layers[l]);
            if (nconv == 0) conv_layer[0] = l; else conv_layer[1] = l;
            ++nconv;
        } else if (C.layers[l].kind == PMT_CNN_LINEAR) {
            lin_layer = l;
        }
    }
    const C2Weights cw = c2_stage_weights(M, theta, packed, lds + (((size_t)nw * per_wave + 3) & ~(size_t)3), stage_floats, true);
    float* acts = lds + (size_t)wave * per_wave;
    float* gA = acts + sa;
    float* gB = gA + ma;

    f4 cacc0[NTO][NTI0], cacc1[NTO][NTI1];
    float cb[C2_MAX_CONVS][NTO];
#pragma unroll
    for (int ot = 0; ot < NTO; ++ot) {
        cb[0][ot] = cb[1][ot] = 0.f;
#pragma unroll
        for (int it = 0; it < NTI0; ++it) cacc0[ot][it] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int it = 0; it < NTI1; ++it) cacc1[ot][it] = f4{0.f, 0.f, 0.f, 0.f};
    }
    // final linear: this wave owns the input columns [k0, k0 + klen) of dW for every variant of the workgroup
    float lw[C2_LIN_REGS], lb = 0.f;
#pragma unroll
    for (int q = 0; q < C2_LIN_REGS; ++q) lw[q] = 0.f;
    int lin_nin = 0, lin_oc = 0, klen = 0, k0 = 0;
    if (lin_layer >= 0) {
        lin_nin = C.layers[lin_layer].in_ch * C.layers[lin_layer].in_len;
        lin_oc = C.layers[lin_layer].out_ch;
        klen = (lin_nin + nw - 1) / nw;
        k0 = wave * klen;
    }
    const float inv_klen = klen > 0 ? 1.0f / (float)klen : 0.f;

    const long long per_round = (long long)gridDim.x * nw;
    for (long long base = (long long)blockIdx.x * nw; base < n; base += per_round) {
        const long long v = base + wave;
        if (v < n) {
            if (stash) {  // the forward kept its layer outputs: load them instead of recomputing
                const int first = 10 * uniform(C.seq_len), per = sa - first;
                const float* src = stash + (size_t)v * per;
                c2_one_hot(acts, hap + (size_t)v * hap_stride, uniform(C.seq_len));
                for (int i = lane; i < per; i += 64) acts[first + i] = src[i];
                wave_sync();
            } else if (!(dbg & 1)) {
                c2_forward_variant<NTO>(M, theta, cw, acts, hap + (size_t)v * hap_stride, taps);
            }
            for (int o = lane; o < od; o += 64) {
                const float d = d_out[(size_t)v * d_out_stride + o];
                gA[o] = d;
                dout_sh[wave][o] = d;
            }
            wave_sync();
            float* gout = gA;
            float* gin = gB;
            int conv = nconv;
            for (int l = nl - 1; l >= 0; --l) {
                const PmtCnnLayer& L = C.layers[l];
                const int kind = uniform(L.kind);
                if (kind == PMT_CNN_FLATTEN) continue;
                const float* xin = acts + uniform(L.in_off);
                const float* yout = acts + uniform(L.out_off);
                const int nin = uniform(L.in_ch) * uniform(L.in_len), nout = uniform(L.out_ch) * uniform(L.out_len);
                const bool need_din = uniform(L.in_off) != 0;  // the one-hot input needs no gradient
                if ((dbg & 16) && kind != PMT_CNN_CONV) {
                } else if (kind == PMT_CNN_LEAKY_RELU || kind == PMT_CNN_SELU) {
                    for (int i = lane; i < nout; i += 64) gin[i] = gout[i] * c2_act_grad_from_out(kind, yout[i]);
                } else if (kind == PMT_CNN_POOL) {
                    for (int i = lane; i < nin; i += 64) gin[i] = 0.f;
                    wave_sync();
                    const float inv = 1.0f / (float)L.out_len;
                    for (int i = lane; i < nout; i += 64) {
                        const int ch = c2_div(i, L.out_len, inv), so = i - ch * L.out_len;
                        int arg = so * L.stride;
                        float mx = -INFINITY;
                        for (int k = 0; k < L.kernel; ++k) {  // first maximum wins, like ATen's max_pool backward
                            const int s = so * L.stride + k;
                            if (s < L.in_len) {
                                const float val = xin[ch * L.in_len + s];
                                if (val > mx) { mx = val; arg = s; }
                            }
                        }
                        float* dst = &gin[ch * L.in_len + arg];
                        if (L.stride >= L.kernel) *dst = gout[i]; else atomicAdd(dst, gout[i]);
                    }
                } else if (kind == PMT_CNN_LINEAR) {
                    if (need_din) {
                        const float* Wt = cw.lin_w;
                        for (int k = lane; k < nin; k += 64) {
                            float acc = 0.f;
                            for (int o = 0; o < L.out_ch; ++o) acc += Wt[(size_t)o * nin + k] * gout[o];
                            gin[k] = acc;
                        }
                    }
                } else {  // CONV
                    --conv;
                    const PmtLinear& Wl = M->lin[uniform(L.lin)];
                    const int K = uniform(Wl.in_dim), OC = uniform(Wl.out_dim), out_len = uniform(L.out_len);
                    if (dbg & 4) {
                    } else if (conv == 0) c2_conv_wgrad<NTO, NTI0>(cacc0, cb[0], L, gout, xin, taps[0], K, OC);
                    else c2_conv_wgrad<NTO, NTI1>(cacc1, cb[1], L, gout, xin, taps[1], K, OC);
                    if (need_din && !(dbg & 2)) {
                        for (int i = lane; i < nin; i += 64) gin[i] = 0.f;
                        wave_sync();
                        const int nkt = (K + 15) >> 4;
                        const int* tap = taps[conv];
                        for (int tile = 0; tile * 16 < out_len; ++tile) {
                            const int so = tile * 16 + r;
                            const bool valid = so < out_len;
                            f4 dy[1][NTO], dx[1][C2_NTI];
#pragma unroll
                            for (int t = 0; t < NTO; ++t)
#pragma unroll
                                for (int j = 0; j < 4; ++j) {
                                    const int co = feat_of(t, j, g);
                                    dy[0][t][j] = (valid && co < OC) ? gout[co * out_len + so] : 0.f;
                                }
                            init_bias<C2_NTI>(dx, nullptr, K, g);
                            linear_acc<NTO, C2_NTI, false>(dx, dy, conv == 0 ? cw.wt[0] : cw.wt[1], OC, K);
#pragma unroll
                            for (int t = 0; t < C2_NTI; ++t)
                                if (t < nkt) {
#pragma unroll
                                    for (int j = 0; j < 4; ++j) {
                                        const int tp = tap[feat_of(t, j, g)];
                                        const int s = so * L.stride + (tp >> 16) - 64;
                                        if (valid && tp >= 0 && s >= 0 && s < L.in_len) atomicAdd(&gin[(tp & 0xFFFF) + s], dx[0][t][j]);  // col2im
                                    }
                                }
                        }
                    }
                }
                wave_sync();
                float* t = gout; gout = gin; gin = t;
            }
        }
        // ---- final linear dW: every wave adds its column slice over the variants of this round ----
        __syncthreads();
        if (lin_layer >= 0 && !(dbg & 8)) {
            const int in_off = C.layers[lin_layer].in_off;
            const int nvar = (int)min((long long)nw, (long long)n - base);
#pragma unroll
            for (int q = 0; q < C2_LIN_REGS; ++q) {
                const int e = lane + 64 * q;
                if (e < lin_oc * klen) {
                    const int o = c2_div(e, klen, inv_klen), k = k0 + (e - o * klen);
                    if (k < lin_nin) {
                        float acc = 0.f;
                        for (int u = 0; u < nvar; ++u) acc += dout_sh[u][o] * lds[(size_t)u * per_wave + in_off + k];
                        lw[q] += acc;
                    }
                }
            }
            if (wave == 0 && lane < lin_oc)
                for (int u = 0; u < nvar; ++u) lb += dout_sh[u][lane];
        }
        __syncthreads();
    }

    // ---- add the register-resident weight gradients to global memory, once per workgroup ----
#pragma unroll
    for (int c = 0; c < C2_MAX_CONVS; ++c) {
        if (conv_layer[c] < 0) continue;
        const PmtCnnLayer& L = C.layers[conv_layer[c]];
        const PmtLinear& Wl = M->lin[uniform(L.lin)];
        const int K = uniform(Wl.in_dim), OC = uniform(Wl.out_dim);
        float* gw = gtheta + uniform(Wl.w_src);
        float* gb = gtheta + uniform(Wl.b_src);
#pragma unroll
        for (int ot = 0; ot < NTO; ++ot) {
#pragma unroll
            for (int it = 0; it < C2_NTI; ++it)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int oc = 16 * ot + 4 * g + j, f = 16 * it + r;  // plain C layout: no feature permutation here
                    float v = 0.f;
                    if (c == 0 && it < NTI0) v = cacc0[ot][it][j];
                    if (c == 1 && it < NTI1) v = cacc1[ot][it][j];
                    if (oc < OC && f < K && ((c == 0 && it < NTI0) || (c == 1 && it < NTI1))) atomicAdd(&gw[(size_t)oc * K + f], v);
                }
            const float tot = group_sum(cb[c][ot]);  // lanes (m, *) hold the sum over all positions for output channel m
            if (g == 0 && 16 * ot + r < OC) atomicAdd(&gb[16 * ot + r], tot);
        }
    }
    if (lin_layer >= 0) {
        const PmtCnnLayer& L = C.layers[lin_layer];
#pragma unroll
        for (int q = 0; q < C2_LIN_REGS; ++q) {
            const int e = lane + 64 * q;
            if (e < lin_oc * klen) {
                const int o = c2_div(e, klen, inv_klen), k = k0 + (e - o * klen);
                if (k < lin_nin) atomicAdd(&gtheta[L.w_src + (size_t)o * lin_nin + k], lw[q]);
            }
        }
        if (wave == 0 && lane < lin_oc) atomicAdd(&gtheta[L.b_src + lane], lb);
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------------
static int cnn2_supported(const PmtModel* m) {
    if (m->force_cnn == 1) return 0;  // always the general kernels (the parity tests cover every instance)
    const PmtCnn* c = &m->cnn;
    int nconv = 0, nlin = 0;
    for (int l = 0; l < c->n_layers; ++l) {
        const PmtCnnLayer* L = &c->layers[l];
        if (L->kind == PMT_CNN_CONV) {
            const PmtLinear* w = &m->lin[L->lin];
            if (++nconv > C2_MAX_CONVS || w->out_dim > 16 * C2_MAX_NTO || w->in_dim > 16 * C2_NTI) return 0;
            if (L->in_ch * L->in_len >= 65536 || L->kernel * L->dilation >= 64 || L->padding >= 64) return 0;
        } else if (L->kind == PMT_CNN_LINEAR) {
            if (++nlin > 1 || l != c->n_layers - 1 || L->out_ch > C2_MAX_LIN_OUT) return 0;
        } else if (L->kind == PMT_CNN_POOL) {
            if (L->padding != 0 || L->dilation != 1) return 0;
        }
    }
    // more than 32 output channels: ONE convolution of at most 32 im2col columns (the four-out-tile instance holds 4 x 2 tiles of dW)
    for (int l = 0; l < c->n_layers; ++l) {
        const PmtCnnLayer* L = &c->layers[l];
        if (L->kind == PMT_CNN_CONV && m->lin[L->lin].out_dim > 32 && (nconv != 1 || m->lin[L->lin].in_dim > 32)) return 0;
    }
    return 1;
}

// out-channel tiles of the widest convolution (the kernel instance: 2 or C2_MAX_NTO)
static int cnn2_out_tiles(const PmtModel* m) {
    int t = 1;
    for (int l = 0; l < m->cnn.n_layers; ++l)
        if (m->cnn.layers[l].kind == PMT_CNN_CONV) {
            const int ot = (m->lin[m->cnn.layers[l].lin].out_dim + 15) / 16;
            t = ot > t ? ot : t;
        }
    return t;
}

// floats of LDS the weights take when staged (c2_stage_weights)
static size_t cnn2_stage_floats(const PmtModel* m, bool with_transposes) {
    size_t n = 0;
    for (int l = 0; l < m->cnn.n_layers; ++l) {
        const PmtCnnLayer* L = &m->cnn.layers[l];
        if (L->kind == PMT_CNN_CONV) {
            const PmtLinear* w = &m->lin[L->lin];
            const size_t nfl = (size_t)((w->out_dim + 15) / 16) * ((w->in_dim + 15) / 16) * 256;
            n += nfl + 256 + 16 * (size_t)((w->out_dim + 15) / 16);
            if (with_transposes && L->in_off != 0) n += nfl + 256;
        } else if (L->kind == PMT_CNN_LINEAR) {
            n += ((size_t)L->out_ch * L->in_ch * L->in_len + 3) & ~(size_t)3;
        }
    }
    return n + 4;
}

// waves per workgroup so that `floats_per_wave` floats of LDS fit (8 at most, 0 = does not fit).  The kernels are bound by
// the latency of their short dependent phases, so residency matters more than workgroup size: when two workgroups of at
// least 4 waves fit a CU, plan for two (*per_cu = 2).
static int cnn2_waves(size_t floats_per_wave, size_t static_bytes, int* per_cu) {
    const size_t bytes = floats_per_wave * sizeof(float);
    int half = 78 * 1024 > static_bytes ? (int)((78 * 1024 - static_bytes) / bytes) : 0;
    if (half > PMT_WAVES) half = PMT_WAVES;
    if (half >= 4) {
        *per_cu = 2;
        return half;
    }
    *per_cu = 1;
    const int nw = 156 * 1024 > static_bytes ? (int)((156 * 1024 - static_bytes) / bytes) : 0;
    return nw > PMT_WAVES ? PMT_WAVES : nw;
}

extern "C" size_t pmt_cnn3_stash_floats(const PmtModel* m);
extern "C" size_t pmt_cnn_stash_floats(const PmtModel* m) {
    if (m) {
        const size_t per3 = pmt_cnn3_stash_floats(m);  // the batched-column kernels take the model when they cover it
        if (per3 > 0) return per3;
    }
    if (!m || !cnn2_supported(m)) return 0;
    const int per = m->cnn.sum_act - 10 * m->cnn.seq_len;
    return per > 0 ? (size_t)per : 0;
}

extern "C" int pmt_cnn2_try_forward(const PmtModel* model_host, const PmtModel* model_dev, const float* theta, const float* packed,
                                    const int64_t* haplotypes, int64_t hap_stride, int32_t n, float* out, int64_t out_stride,
                                    float* stash, void* stream) {
    if (!cnn2_supported(model_host)) return 1;  // not an error: the caller runs the general kernels
    const size_t per = (size_t)model_host->cnn.sum_act;
    int per_cu = 1;
    const size_t static_fwd = sizeof(int) * C2_MAX_CONVS * PMT_MAX_ROW_INPUT;
    size_t stage = cnn2_stage_floats(model_host, false);
    int nw = cnn2_waves(per, static_fwd + stage * sizeof(float), &per_cu);
    if (nw < 4) {  // no room for the weights next to a useful number of waves: read them from global memory
        stage = 0;
        nw = cnn2_waves(per, static_fwd, &per_cu);
    }
    if (nw < 2) return 1;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    cus *= per_cu;
    const int blocks = (int)(((long long)n + nw - 1) / nw < cus ? ((long long)n + nw - 1) / nw : cus);
    const size_t lds_bytes = ((size_t)nw * per + 4 + stage) * sizeof(float);  // (+4: the weights start 16-byte aligned)
    auto kernel = cnn2_out_tiles(model_host) <= 2 ? pmt_cnn2_forward_kernel<2> : pmt_cnn2_forward_kernel<C2_MAX_NTO>;
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(64 * nw), lds_bytes, reinterpret_cast<hipStream_t>(stream), model_dev,
                       theta, packed, (const long long*)haplotypes, (long long)hap_stride, n, (int)per, (int)stage, out, (long long)out_stride, stash);
    return hipGetLastError() == hipSuccess ? PMT_OK : PMT_E_LAUNCH;
}

extern "C" int pmt_cnn2_try_backward(const PmtModel* model_host, const PmtModel* model_dev, const float* theta, const float* packed,
                                     const int64_t* haplotypes, int64_t hap_stride, int32_t n, const float* d_out,
                                     int64_t d_out_stride, const float* stash, float* grad_theta, void* stream) {
    if (!cnn2_supported(model_host)) return 1;
    const size_t per = (size_t)model_host->cnn.sum_act + 2 * (size_t)model_host->cnn.max_act;
    int per_cu = 1;
    const size_t static_bwd = sizeof(int) * C2_MAX_CONVS * PMT_MAX_ROW_INPUT + sizeof(float) * PMT_WAVES * C2_MAX_LIN_OUT;
    size_t stage = cnn2_stage_floats(model_host, true);
    int nw = cnn2_waves(per, static_bwd + stage * sizeof(float), &per_cu);
    if (nw < 4) {
        stage = 0;
        nw = cnn2_waves(per, static_bwd, &per_cu);
    }
    if (nw < 2) return 1;
    // the final linear's dW slice of a wave must fit its registers
    for (int l = 0; l < model_host->cnn.n_layers; ++l) {
        const PmtCnnLayer* L = &model_host->cnn.layers[l];
        if (L->kind == PMT_CNN_LINEAR) {
            const int nin = L->in_ch * L->in_len, klen = (nin + nw - 1) / nw;
            if (L->out_ch * klen > 64 * C2_LIN_REGS) return 1;
        }
    }
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    cus *= per_cu;
    const int blocks = (int)(((long long)n + nw - 1) / nw < cus ? ((long long)n + nw - 1) / nw : cus);
    const size_t lds_bytes = ((size_t)nw * per + 4 + stage) * sizeof(float);  // (+4: the weights start 16-byte aligned)
    int kt[C2_MAX_CONVS] = {0, 0}, nc = 0;  // k-tiles of the convolutions: the instance that holds their dW in registers
    for (int l = 0; l < model_host->cnn.n_layers; ++l)
        if (model_host->cnn.layers[l].kind == PMT_CNN_CONV) kt[nc++] = (model_host->lin[model_host->cnn.layers[l].lin].in_dim + 15) / 16;
    // (more than 32 output channels: the instance with four out tiles -- only for a SINGLE convolution of at most 32 im2col columns, the
    //  reference's test configuration T0: its dW alone is 4 x 2 tiles of registers; wider two-convolution stacks stay with the general kernels)
    auto kernel = (kt[0] <= 2) ? pmt_cnn2_backward_kernel<2, 2, C2_NTI> : pmt_cnn2_backward_kernel<2, C2_NTI, C2_NTI>;
    if (cnn2_out_tiles(model_host) > 2) kernel = pmt_cnn2_backward_kernel<C2_MAX_NTO, 2, 1>;
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(64 * nw), lds_bytes, reinterpret_cast<hipStream_t>(stream), model_dev,
                       theta, packed, (const long long*)haplotypes, (long long)hap_stride, n, (int)per, (int)stage, d_out, (long long)d_out_stride,
                       stash, grad_theta, model_host->cnn_debug);
    return hipGetLastError() == hipSuccess ? PMT_OK : PMT_E_LAUNCH;
}
