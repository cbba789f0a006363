// Haplotype CNN of the artifact model (reference architecture/dna_sequence_convolution.py:31-111 applied to
// data/batch.py:115-130's one-hot haplotypes), forward and backward.
//
// A workgroup takes VPB variants, builds their one-hot input straight from the int64 haplotype rows, and walks the layer
// list with every activation resident in LDS.  Convolutions run on the matrix cores as implicit GEMMs: a column is one
// (variant, output position) pair, its im2col vector (in_ch * kernel taps, <= PMT_MAX_CNN_TAPS) is gathered from LDS directly into
// the B-operand register layout of pmt_device.hpp, and the convolution weight [out_ch][in_ch * kernel] is an ordinary
// packed PmtLinear -- so the forward is linear_acc, the weight gradient is wgrad_exchange and the input gradient is
// linear_acc with the transposed fragments followed by a col2im scatter-add in LDS.  Pooling, activations and the final
// (wide) linear layer are small and stay on the vector ALU.  The backward kernel recomputes the forward keeping every layer
// output in LDS.
// Wave shape of this translation unit: 4 waves x 2 tiles.  Every layer is a short, latency-bound phase between two
// workgroup barriers, so the kernels want SEVERAL small workgroups per CU (each with its own barriers) rather than one
// large one: 256 threads and <= ~78 KB of LDS let two be resident.
#define PMT_OWN_WAVE_SHAPE
#define PMT_WAVES 4
#define PMT_RT 2
#define PMT_STAGE_PLANES 32  // 16 planes per tile (4 of dy + 12 of im2col): 2 tiles per exchange pass; leaves LDS for the activations
#include <stdlib.h>
#include <string.h>

#include "pmt_device.hpp"
#include "pmt_bwd_device.hpp"

#define CNN_NTIN (PMT_MAX_CNN_TAPS / 16)
#define LEAKY_SLOPE 0.01f

// a / b for 0 <= a < 2^22 without the ~35-instruction integer division sequence (inv_b = 1.0f / b)
DEV int fast_div(int a, int b, float inv_b) {
    int q = (int)((float)a * inv_b);
    if (q * b > a) --q;
    if ((q + 1) * b <= a) ++q;
    return q;
}

DEV float act_fwd(int kind, float x) {
    if (kind == PMT_CNN_LEAKY_RELU) return x > 0.f ? x : LEAKY_SLOPE * x;
    return x > 0.f ? PMT_SELU_SCALE * x : (PMT_SELU_ALPHA * PMT_SELU_SCALE) * expm1f(x);
}
DEV float act_bwd(int kind, float x_in, float y_out) {  // d(out)/d(in)
    if (kind == PMT_CNN_LEAKY_RELU) return x_in > 0.f ? 1.f : LEAKY_SLOPE;
    return x_in > 0.f ? PMT_SELU_SCALE : y_out + PMT_SELU_ALPHA * PMT_SELU_SCALE;
}

// one-hot input of variant v: channel 2*base + (0 ref | 1 alt), position s  (reference data/batch.py:115-130)
DEV void build_one_hot(float* __restrict__ dst, int dst_stride, const long long* __restrict__ hap, int seq_len, int nv,
                       long long hap_stride, int v0) {
    const int per = 10 * seq_len;
    const float inv_len = 1.0f / (float)seq_len;
    for (int v = 0; v < nv; ++v)
        for (int rem = threadIdx.x; rem < per; rem += PMT_THREADS) {
            const int c = fast_div(rem, seq_len, inv_len), s = rem - c * seq_len;
            const long long base = hap[(size_t)(v0 + v) * hap_stride + (c & 1) * seq_len + s];
            dst[v * dst_stride + rem] = (base == (c >> 1)) ? 1.f : 0.f;
        }
}

// per-layer tap table: for im2col feature f = ci * kernel + k :  tap[f] = (ci * in_len) | ((k * dilation - padding + 64) << 16)
DEV void build_taps(int* __restrict__ tap, const PmtCnnLayer& L) {
    const int K = L.in_ch * L.kernel;
    for (int f = threadIdx.x; f < PMT_MAX_CNN_TAPS; f += PMT_THREADS) {
        int v = -1;
        if (f < K) {
            const int ci = f / L.kernel, k = f - ci * L.kernel;
            v = (ci * L.in_len) | ((k * L.dilation - L.padding + 64) << 16);
        }
        tap[f] = v;
    }
}

struct ColMeta {
    int v, so;     // variant within the block, output position
    bool valid;
};
DEV ColMeta col_meta(int tile, int ncol, int out_len) {
    ColMeta m;
    const int col = tile * 16 + (threadIdx.x & 15);
    m.valid = col < ncol;
    m.v = m.valid ? fast_div(col, out_len, 1.0f / (float)out_len) : 0;
    m.so = m.valid ? col - m.v * out_len : 0;
    return m;
}

// im2col columns of this wave's tiles, in the B-operand register layout: x[rt][t][j] = tap feat_of(t, j, g) of column r
DEV void gather_im2col(f4 (&x)[PMT_RT][CNN_NTIN], const float* __restrict__ in, int in_stride, const int* __restrict__ tap,
                       const PmtCnnLayer& L, const ColMeta (&cm)[PMT_RT], int g) {
    const int nkt = (L.in_ch * L.kernel + 15) >> 4;
#pragma unroll
    for (int t = 0; t < CNN_NTIN; ++t) {
        if (t < nkt) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int tp = tap[feat_of(t, j, g)];
                const int base = tp & 0xFFFF, ks = (tp >> 16) - 64;
#pragma unroll
                for (int rt = 0; rt < PMT_RT; ++rt) {
                    const int s = cm[rt].so * L.stride + ks;
                    const bool ok = cm[rt].valid && tp >= 0 && s >= 0 && s < L.in_len;
                    x[rt][t][j] = ok ? in[cm[rt].v * in_stride + base + s] : 0.f;
                }
            }
        } else {
#pragma unroll
            for (int rt = 0; rt < PMT_RT; ++rt) x[rt][t] = f4{0.f, 0.f, 0.f, 0.f};
        }
    }
}

// convolution forward for nv variants (LDS -> LDS) on the matrix cores
DEV void conv_forward(const PmtModel* __restrict__ M, const PmtCnnLayer& L, const float* __restrict__ packed,
                      const float* __restrict__ in, int in_stride, float* __restrict__ out, int out_stride, int nv,
                      const int* __restrict__ tap) {
    const int lane = threadIdx.x & 63, g = lane >> 4, wave = uniform((int)(threadIdx.x >> 6));
    const PmtLinear& W = M->lin[uniform(L.lin)];
    const int K = uniform(W.in_dim), OC = uniform(W.out_dim), out_len = uniform(L.out_len);
    const int ncol = nv * out_len, ntiles = (ncol + 15) >> 4;
    for (int tile0 = 0; tile0 < ntiles; tile0 += PMT_WAVES * PMT_RT) {
        ColMeta cm[PMT_RT];
#pragma unroll
        for (int rt = 0; rt < PMT_RT; ++rt) cm[rt] = col_meta(tile0 + wave * PMT_RT + rt, ncol, out_len);
        if (tile0 + wave * PMT_RT < ntiles) {  // wave-uniform
            f4 x[PMT_RT][CNN_NTIN], y[PMT_RT][PMT_NT];
            gather_im2col(x, in, in_stride, tap, L, cm, g);
            init_bias<PMT_NT>(y, packed + uniform(W.b_pvec), OC, g);
            linear_acc<CNN_NTIN, PMT_NT, false>(y, x, packed + uniform(W.w_frag), K, OC);
#pragma unroll
            for (int rt = 0; rt < PMT_RT; ++rt)
                if (cm[rt].valid) {
#pragma unroll
                    for (int t = 0; t < PMT_NT; ++t)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int co = feat_of(t, j, g);
                            if (co < OC) out[cm[rt].v * out_stride + co * out_len + cm[rt].so] = y[rt][t][j];
                        }
                }
        }
    }
}

// the vector-ALU layers (LDS -> LDS)
DEV void small_layer_forward(const PmtCnnLayer& L, const float* __restrict__ theta, const float* __restrict__ in,
                             float* __restrict__ out, int nv, int in_stride, int out_stride) {
    const int kind = L.kind;
    if (kind == PMT_CNN_POOL) {
        const int per = L.out_ch * L.out_len;
        const float inv_len = 1.0f / (float)L.out_len;
        for (int v = 0; v < nv; ++v)
            for (int rem = threadIdx.x; rem < per; rem += PMT_THREADS) {
                const int c = fast_div(rem, L.out_len, inv_len), so = rem - c * L.out_len;
                float m = -INFINITY;
                for (int k = 0; k < L.kernel; ++k) {
                    const int s = so * L.stride + k;
                    if (s < L.in_len) m = fmaxf(m, in[v * in_stride + c * L.in_len + s]);
                }
                out[v * out_stride + rem] = m;
            }
    } else if (kind == PMT_CNN_LEAKY_RELU || kind == PMT_CNN_SELU) {
        const int per = L.out_ch * L.out_len;
        for (int v = 0; v < nv; ++v)
            for (int rem = threadIdx.x; rem < per; rem += PMT_THREADS) out[v * out_stride + rem] = act_fwd(kind, in[v * in_stride + rem]);
    } else if (kind == PMT_CNN_LINEAR) {
        // out[v][o] = b[o] + W[o][:] . in[v][:] : the 16 lanes of a lane-group split the dot product, coalesced weight reads
        const float* W = theta + L.w_src;
        const float* b = theta + L.b_src;
        const int nin = L.in_ch * L.in_len;
        const int sub = threadIdx.x & 15, slot = threadIdx.x >> 4, nslots = PMT_THREADS >> 4;
        const float inv_oc = 1.0f / (float)L.out_ch;
        for (int i = slot; i < nv * L.out_ch; i += nslots) {
            const int v = fast_div(i, L.out_ch, inv_oc), o = i - v * L.out_ch;
            float acc = 0.f;
            for (int k = sub; k < nin; k += 16) acc += W[(size_t)o * nin + k] * in[v * in_stride + k];
            acc += __shfl_xor(acc, 1);
            acc += __shfl_xor(acc, 2);
            acc += __shfl_xor(acc, 4);
            acc += __shfl_xor(acc, 8);
            if (sub == 0) out[v * out_stride + o] = acc + b[o];
        }
    }
}

struct CnnFwdShared {
    int tap[PMT_MAX_CNN_TAPS];
};

extern "C" __global__ __launch_bounds__(PMT_THREADS, 2) void pmt_cnn_forward_kernel(
    const PmtModel* __restrict__ M, const float* __restrict__ theta, const float* __restrict__ packed,
    const long long* __restrict__ hap, long long hap_stride, int n, int vpb, float* __restrict__ out, long long out_stride) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ CnnFwdShared sh;
    const PmtCnn& C = M->cnn;
    const int v0 = blockIdx.x * vpb;
    const int nv = min(vpb, n - v0);
    const int ma = uniform(C.max_act);
    float* a = lds;
    float* b = lds + (size_t)vpb * ma;
    build_one_hot(a, ma, hap, uniform(C.seq_len), nv, hap_stride, v0);
    __syncthreads();
    const int nl = uniform(C.n_layers);
    for (int l = 0; l < nl; ++l) {
        const PmtCnnLayer& L = C.layers[l];
        const int kind = uniform(L.kind);
        if (kind == PMT_CNN_FLATTEN) continue;
        if (kind == PMT_CNN_CONV) {
            build_taps(sh.tap, L);
            __syncthreads();
            conv_forward(M, L, packed, a, ma, b, ma, nv, sh.tap);
        } else if (kind == PMT_CNN_LEAKY_RELU || kind == PMT_CNN_SELU) {
            small_layer_forward(L, theta, a, a, nv, ma, ma);
            __syncthreads();
            continue;
        } else {
            small_layer_forward(L, theta, a, b, nv, ma, ma);
        }
        __syncthreads();
        float* t = a; a = b; b = t;
    }
    const int od = uniform(C.out_dim);
    for (int i = threadIdx.x; i < nv * od; i += PMT_THREADS) {
        const int v = i / od, o = i - v * od;
        out[(size_t)(v0 + v) * out_stride + o] = a[v * ma + o];
    }
}

struct CnnBwdShared {
    int tap[PMT_MAX_CNN_TAPS];
    float aux[PMT_WAVES][PMT_AUX_CAP];
    int aux_dst[PMT_AUX_CAP];
    f4 stage[PMT_STAGE_PLANES * 64];
};

// Backward: dynamic LDS holds, per variant, the one-hot input and the output of every layer (stride sum_act) plus two
// gradient buffers of max_act floats.
extern "C" __global__ __launch_bounds__(PMT_THREADS, 2) void pmt_cnn_backward_kernel(
    const PmtModel* __restrict__ M, const float* __restrict__ theta, const float* __restrict__ packed,
    const long long* __restrict__ hap, long long hap_stride, int n, int vpb, const float* __restrict__ d_out,
    long long d_out_stride, float* __restrict__ gtheta) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ __attribute__((aligned(16))) CnnBwdShared sh;
    const PmtCnn& C = M->cnn;
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, wave = uniform((int)(tid >> 6));
    const int v0 = blockIdx.x * vpb;
    const int nv = min(vpb, n - v0);
    const int ma = uniform(C.max_act), sa = uniform(C.sum_act), nl = uniform(C.n_layers);
    float* acts = lds;
    float* g0 = lds + (size_t)vpb * sa;
    float* g1 = g0 + (size_t)vpb * ma;
    // ---- recompute the forward, keeping everything ----
    build_one_hot(acts, sa, hap, uniform(C.seq_len), nv, hap_stride, v0);
    __syncthreads();
    for (int l = 0; l < nl; ++l) {
        const PmtCnnLayer& L = C.layers[l];
        const int kind = uniform(L.kind);
        if (kind == PMT_CNN_FLATTEN) continue;
        if (kind == PMT_CNN_CONV) {
            build_taps(sh.tap, L);
            __syncthreads();
            conv_forward(M, L, packed, acts + uniform(L.in_off), sa, acts + uniform(L.out_off), sa, nv, sh.tap);
        } else {
            small_layer_forward(L, theta, acts + uniform(L.in_off), acts + uniform(L.out_off), nv, sa, sa);
        }
        __syncthreads();
    }
    // ---- d(out) -> g0 ----
    const int od = uniform(C.out_dim);
    for (int i = tid; i < nv * od; i += PMT_THREADS) {
        const int v = i / od, o = i - v * od;
        g0[v * ma + o] = d_out[(size_t)(v0 + v) * d_out_stride + o];
    }
    __syncthreads();
    BwdCtx c{M, theta, theta, packed, gtheta, gtheta, &sh.stage[0], &sh.aux[0][0], &sh.aux_dst[0], g, 0u,
             wave * PMT_RT, 0, 0, 0, 0, nullptr};
    float* gout = g0;
    float* gin = g1;
    for (int l = nl - 1; l >= 0; --l) {
        const PmtCnnLayer& L = C.layers[l];
        const int kind = uniform(L.kind);
        if (kind == PMT_CNN_FLATTEN) continue;
        const float* xin = acts + uniform(L.in_off);
        const float* yout = acts + uniform(L.out_off);
        const int nin = uniform(L.in_ch) * uniform(L.in_len), nout = uniform(L.out_ch) * uniform(L.out_len);
        const bool need_din = uniform(L.in_off) != 0;  // the one-hot input needs no gradient
        if (kind == PMT_CNN_LEAKY_RELU || kind == PMT_CNN_SELU) {
            for (int v = 0; v < nv; ++v)
                for (int rem = tid; rem < nout; rem += PMT_THREADS)
                    gin[v * ma + rem] = gout[v * ma + rem] * act_bwd(kind, xin[v * sa + rem], yout[v * sa + rem]);
        } else if (kind == PMT_CNN_POOL) {
            for (int v = 0; v < nv; ++v)
                for (int rem = tid; rem < nin; rem += PMT_THREADS) gin[v * ma + rem] = 0.f;
            __syncthreads();
            const float inv_len = 1.0f / (float)L.out_len;
            for (int i = tid; i < nv * nout; i += PMT_THREADS) {
                const int v = fast_div(i, nout, 1.0f / (float)nout), rem = i - v * nout, ch = fast_div(rem, L.out_len, inv_len), so = rem - ch * L.out_len;
                int arg = so * L.stride;
                float m = -INFINITY;
                for (int k = 0; k < L.kernel; ++k) {  // first maximum wins, like ATen's max_pool backward
                    const int s = so * L.stride + k;
                    if (s < L.in_len) {
                        const float val = xin[v * sa + ch * L.in_len + s];
                        if (val > m) { m = val; arg = s; }
                    }
                }
                float* dst = &gin[v * ma + ch * L.in_len + arg];
                if (L.stride >= L.kernel) *dst = gout[v * ma + rem];  // disjoint windows: one writer per input element
                else atomicAdd(dst, gout[v * ma + rem]);
            }
        } else if (kind == PMT_CNN_LINEAR) {
            const float* W = theta + L.w_src;
            for (int o = 0; o < L.out_ch; ++o)  // dW[o][k] += sum_v dout[v][o] x[v][k]
                for (int k = tid; k < nin; k += PMT_THREADS) {
                    float acc = 0.f;
                    for (int v = 0; v < nv; ++v) acc += gout[v * ma + o] * xin[v * sa + k];
                    atomicAdd(&gtheta[L.w_src + o * nin + k], acc);
                }
            for (int o = tid; o < L.out_ch; o += PMT_THREADS) {
                float acc = 0.f;
                for (int v = 0; v < nv; ++v) acc += gout[v * ma + o];
                atomicAdd(&gtheta[L.b_src + o], acc);
            }
            if (need_din)
                for (int v = 0; v < nv; ++v)
                    for (int k = tid; k < nin; k += PMT_THREADS) {
                        float acc = 0.f;
                        for (int o = 0; o < L.out_ch; ++o) acc += W[(size_t)o * nin + k] * gout[v * ma + o];
                        gin[v * ma + k] = acc;
                    }
        } else if (kind == PMT_CNN_CONV) {
            const PmtLinear& Wl = M->lin[uniform(L.lin)];
            const int K = uniform(Wl.in_dim), OC = uniform(Wl.out_dim), out_len = uniform(L.out_len);
            const int ncol = nv * out_len, ntiles = (ncol + 15) >> 4;
            build_taps(sh.tap, L);
            if (need_din)
                for (int v = 0; v < nv; ++v)
                    for (int rem = tid; rem < nin; rem += PMT_THREADS) gin[v * ma + rem] = 0.f;
            __syncthreads();
            for (int tile0 = 0; tile0 < ntiles; tile0 += PMT_WAVES * PMT_RT) {
                ColMeta cm[PMT_RT];
                unsigned present = 0;
#pragma unroll
                for (int rt = 0; rt < PMT_RT; ++rt) {
                    cm[rt] = col_meta(tile0 + wave * PMT_RT + rt, ncol, out_len);
                    if (tile0 + wave * PMT_RT + rt < ntiles) present |= 1u << rt;
                }
                f4 x[PMT_RT][CNN_NTIN], dy[PMT_RT][PMT_NT];
                gather_im2col(x, xin, sa, sh.tap, L, cm, g);
#pragma unroll
                for (int rt = 0; rt < PMT_RT; ++rt)
#pragma unroll
                    for (int t = 0; t < PMT_NT; ++t)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int co = feat_of(t, j, g);
                            dy[rt][t][j] = (cm[rt].valid && co < OC) ? gout[cm[rt].v * ma + co * out_len + cm[rt].so] : 0.f;
                        }
                c.mask_all = present;
                c.ntiles = c.tiles_ref = min(PMT_WG_TILES, ntiles - tile0);
                linear_wgrad<PMT_NT, CNN_NTIN>(c, Wl, dy, x);  // workgroup barriers inside
                if (need_din) {
                    f4 dx[PMT_RT][CNN_NTIN];
                    init_bias<CNN_NTIN>(dx, nullptr, K, g);
                    linear_acc<PMT_NT, CNN_NTIN, false>(dx, dy, packed + uniform(Wl.wt_frag), OC, K);
                    const int nkt = (K + 15) >> 4;
#pragma unroll
                    for (int t = 0; t < CNN_NTIN; ++t)
                        if (t < nkt) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const int tp = sh.tap[feat_of(t, j, g)];
                                const int base = tp & 0xFFFF, ks = (tp >> 16) - 64;
#pragma unroll
                                for (int rt = 0; rt < PMT_RT; ++rt) {
                                    const int s = cm[rt].so * L.stride + ks;
                                    if (cm[rt].valid && tp >= 0 && s >= 0 && s < L.in_len)
                                        atomicAdd(&gin[cm[rt].v * ma + base + s], dx[rt][t][j]);  // col2im
                                }
                            }
                        }
                }
            }
        }
        __syncthreads();
        float* t = gout; gout = gin; gin = t;
    }
}

// pmt_cnn2.hip: 0 = done, 1 = configuration not covered (run the general kernels below), < 0 = error
extern "C" int pmt_cnn2_try_forward(const PmtModel* model_host, const PmtModel* model_dev, const float* theta, const float* packed,
                                    const int64_t* haplotypes, int64_t hap_stride, int32_t n, float* out, int64_t out_stride,
                                    float* stash, void* stream);
extern "C" int pmt_cnn2_try_backward(const PmtModel* model_host, const PmtModel* model_dev, const float* theta, const float* packed,
                                     const int64_t* haplotypes, int64_t hap_stride, int32_t n, const float* d_out,
                                     int64_t d_out_stride, const float* stash, float* grad_theta, void* stream);

// pmt_cnn3.hip (batched-column kernels for the production-shaped stack): same return convention
extern "C" int pmt_cnn3_try_forward(const PmtModel* model_host, const float* theta, const int64_t* haplotypes, int64_t hap_stride, int32_t n,
                                    float* out, int64_t out_stride, float* stash, void* stream);
extern "C" int pmt_cnn3_try_backward(const PmtModel* model_host, const float* theta, const int64_t* haplotypes, int64_t hap_stride, int32_t n,
                                     const float* d_out, int64_t d_out_stride, const float* stash, float* grad_theta, float* workspace,
                                     size_t workspace_floats, void* stream);
extern "C" size_t pmt_cnn3_stash_floats(const PmtModel* m);
extern "C" size_t pmt_cnn3_workspace_floats(const PmtModel* m);

static int cnn_check(const PmtModel* m) {
    if (!m) return PMT_E_INVALID;
    const PmtCnn* c = &m->cnn;
    if (c->n_layers < 1 || c->n_layers > PMT_MAX_CNN_LAYERS || c->seq_len < 1 || c->max_act < 1 || c->sum_act < c->max_act)
        return PMT_E_INVALID;
    for (int l = 0; l < c->n_layers; ++l) {
        const PmtCnnLayer* L = &c->layers[l];
        if (L->kind < 0 || L->kind > PMT_CNN_LINEAR) return PMT_E_UNSUPPORTED;
        if (L->kind == PMT_CNN_POOL && (L->padding != 0 || L->dilation != 1)) return PMT_E_UNSUPPORTED;
        if (L->kind == PMT_CNN_CONV) {
            if (L->lin < 0 || L->lin >= m->n_linear) return PMT_E_INVALID;
            const PmtLinear* w = &m->lin[L->lin];
            if (w->in_dim != L->in_ch * L->kernel || w->out_dim != L->out_ch || w->b_pvec < 0) return PMT_E_INVALID;
            if (w->in_dim > PMT_MAX_CNN_TAPS || w->out_dim > PMT_MAX_WIDTH) return PMT_E_UNSUPPORTED;
            if (L->in_ch * L->in_len >= 65536 || L->kernel * L->dilation >= 64 || L->padding >= 64) return PMT_E_UNSUPPORTED;
        }
    }
    return PMT_OK;
}

static int pick_vpb(size_t floats_per_variant, size_t static_bytes, int blocks_per_cu) {
    const size_t lds = 156 * 1024 / blocks_per_cu;
    if (lds <= static_bytes) return 0;
    const size_t budget = (lds - static_bytes) / sizeof(float);
    int vpb = (int)(budget / floats_per_variant);
    if (vpb > 16) vpb = 16;
    return vpb;
}

extern "C" int pmt_cnn_forward(const PmtModel* model_host, const PmtModel* model_dev, const float* theta, const float* packed,
                               const int64_t* haplotypes, int64_t hap_stride, int32_t n, float* out, int64_t out_stride,
                               float* stash, void* stream) {
    const int rc = cnn_check(model_host);
    if (rc) return rc;
    if (!model_dev || !theta || !packed || !haplotypes || !out || n < 0) return PMT_E_INVALID;
    if (n == 0) return PMT_OK;
    {   // the batched-column kernels where they cover the model (with `stash`: in the layout their backward reads)
        const int rc3 = pmt_cnn3_try_forward(model_host, theta, haplotypes, hap_stride, n, out, out_stride, stash, stream);
        if (rc3 <= 0) return rc3;
        if (model_host->force_cnn == 3) return PMT_E_UNSUPPORTED;  // a kernel family asked for by name never falls back silently
    }
    if (stash) {  // a training forward that keeps its layer outputs: the wave-per-variant kernel writes them in the backward's layout
        const int rc2 = pmt_cnn2_try_forward(model_host, model_dev, theta, packed, haplotypes, hap_stride, n, out, out_stride, stash, stream);
        return rc2 <= 0 ? rc2 : PMT_E_UNSUPPORTED;  // (pmt_cnn_stash_floats said 0 for such a model)
    }
    // The wave-per-variant forward (pmt_cnn2.hip) measures slower than the kernel below (0.54 vs 0.47 ms at 65 536 variants,
    // P0): the forward has no weight gradients to keep resident, which is what the wave-per-variant backward wins with.  It
    // stays selectable (PMT_CNN=wave) and parity-tested.
    if (model_host->force_cnn == 2) {
        const int rc2 = pmt_cnn2_try_forward(model_host, model_dev, theta, packed, haplotypes, hap_stride, n, out, out_stride, nullptr, stream);
        if (rc2 <= 0) return rc2;
        return PMT_E_UNSUPPORTED;  // PMT_CNN=wave on a stack the wave-per-variant kernels do not cover
    }
    const size_t per = 2 * (size_t)model_host->cnn.max_act;
    const int vpb = pick_vpb(per, sizeof(CnnFwdShared), 2);
    if (vpb < 1) return PMT_E_UNSUPPORTED;
    hipLaunchKernelGGL(pmt_cnn_forward_kernel, dim3((n + vpb - 1) / vpb), dim3(PMT_THREADS), vpb * per * sizeof(float),
                       reinterpret_cast<hipStream_t>(stream), model_dev, theta, packed, (const long long*)haplotypes,
                       (long long)hap_stride, n, vpb, out, (long long)out_stride);
    return hipGetLastError() == hipSuccess ? PMT_OK : PMT_E_LAUNCH;
}

extern "C" size_t pmt_cnn_workspace_floats(const PmtModel* model_host) {
    if (cnn_check(model_host)) return 0;
    return pmt_cnn3_workspace_floats(model_host);
}

extern "C" int pmt_cnn_backward(const PmtModel* model_host, const PmtModel* model_dev, const float* theta, const float* packed,
                                const int64_t* haplotypes, int64_t hap_stride, int32_t n, const float* d_out,
                                int64_t d_out_stride, const float* stash, float* grad_theta, float* workspace, size_t workspace_floats,
                                void* stream) {
    const int rc = cnn_check(model_host);
    if (rc) return rc;
    if (!model_dev || !theta || !packed || !haplotypes || !d_out || !grad_theta || n < 0) return PMT_E_INVALID;
    if (n == 0) return PMT_OK;
    {
        const int rc3 = pmt_cnn3_try_backward(model_host, theta, haplotypes, hap_stride, n, d_out, d_out_stride, stash, grad_theta, workspace,
                                                  workspace_floats, stream);
        if (rc3 <= 0) return rc3;
        if (stash && pmt_cnn3_stash_floats(model_host) > 0) return PMT_E_INVALID;  // (the stash is in pmt_cnn3's layout)
        if (model_host->force_cnn == 3) return PMT_E_UNSUPPORTED;
    }
    {
        const int rc2 = pmt_cnn2_try_backward(model_host, model_dev, theta, packed, haplotypes, hap_stride, n, d_out, d_out_stride,
                                              stash, grad_theta, stream);
        if (rc2 <= 0) return rc2;
        if (stash || model_host->force_cnn == 2) return PMT_E_UNSUPPORTED;
    }
    const size_t per = (size_t)model_host->cnn.sum_act + 2 * (size_t)model_host->cnn.max_act;
    const int vpb = pick_vpb(per, sizeof(CnnBwdShared), 2);
    if (vpb < 1) return PMT_E_UNSUPPORTED;
    hipLaunchKernelGGL(pmt_cnn_backward_kernel, dim3((n + vpb - 1) / vpb), dim3(PMT_THREADS), vpb * per * sizeof(float),
                       reinterpret_cast<hipStream_t>(stream), model_dev, theta, packed, (const long long*)haplotypes,
                       (long long)hap_stride, n, vpb, d_out, (long long)d_out_stride, grad_theta);
    return hipGetLastError() == hipSuccess ? PMT_OK : PMT_E_LAUNCH;
}
