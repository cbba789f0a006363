// Row-wise MLPs over independent rows: the per-variant branches of the artifact model (info embedding,
// alt-count adversary, source adversary; reference artifact_model.py:244, :180-183/:276-279, :267-274, mlp.py:8-76).
// Same register layout, MFMA linear, layer-program interpreter and weight-gradient machinery as the read-set kernels
// (pmt_device.hpp); a "tile" is 16 rows, there are no sets.  The first linear may read up to 128 input features (the
// info vector has 71), held in an 8-tile input array.
#include "pmt_device.hpp"
#include "pmt_mlp_device.hpp"
#include "pmt_bwd_device.hpp"

#ifndef PMT_ROWS_TRACE
#define PMT_ROWS_TRACE 0  // development: cycle stamps of one wave of the forward into its first output row (timing only)
#endif
#define ROWS_NTIN (PMT_MAX_ROW_INPUT / 16)
#define ROWS_PER_BLOCK (PMT_WAVES * PMT_RT * 16)

extern "C" int pmt_stash_slots(const PmtModel* m);

// rows -> input registers (tile-position layout), zero beyond in_dim / n_rows
template <int NTIN>
DEV void load_rows(f4 (&x)[PMT_RT][NTIN], const float* __restrict__ in, long long stride, int n_rows, int in_dim, int tile0,
                   int g) {
    const int r = threadIdx.x & 15;
#pragma unroll
    for (int rt = 0; rt < PMT_RT; ++rt) {
        const int row = (tile0 + rt) * 16 + r;
        const float* p = in + (size_t)row * stride;
#pragma unroll
        for (int t = 0; t < NTIN; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int f = feat_of(t, j, g);
                x[rt][t][j] = (row < n_rows && f < in_dim) ? p[f] : 0.f;
            }
    }
}

// WLDS: the MLP's weight fragments ([span_lo, span_lo + span_len) of `packed`: every linear of a row MLP lies in one run) are
// copied into LDS once per workgroup and every layer reads them from there.  A workgroup walks ONE chain of ~8 small layers over
// its 256 rows; with the fragments in global memory every layer started with an L2 round trip that nothing could hide (33 us for
// 0.5 GFLOP over 65 536 rows).
// NT: tiles of 16 features the activations take (2 when every layer behind the input is at most 32 wide: half the registers, half the
// element-wise work of the 64-wide generic layout).
template <bool TRAIN, bool WLDS, int NT>
__global__ __launch_bounds__(PMT_THREADS, 2) void pmt_rows_forward_kernel(const PmtModel* __restrict__ M, int which,
                                                                           const float* __restrict__ theta,
                                                                           const float* __restrict__ packed_g,
                                                                           const float* __restrict__ in, long long in_stride,
                                                                           int n_rows, float* __restrict__ out,
                                                                           long long out_stride, float* __restrict__ stash,
                                                                           unsigned long long dropout_seed, int span_lo, int span_len) {
    extern __shared__ __attribute__((aligned(16))) float rows_lds[];
#if PMT_ROWS_TRACE
    unsigned long long tr[8];
    tr[0] = __builtin_readcyclecounter();
#define ROWS_EV(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); tr[k] = __builtin_readcyclecounter(); } while (0)
#else
#define ROWS_EV(k) do {} while (0)
#endif
    const float* packed = packed_g;
    if constexpr (WLDS) {
        const f4* __restrict__ src = reinterpret_cast<const f4*>(packed_g + span_lo);
        for (int i = threadIdx.x; i < span_len / 4; i += PMT_THREADS) reinterpret_cast<f4*>(rows_lds)[i] = src[i];
        __syncthreads();
        packed = rows_lds - span_lo;
    }
    ROWS_EV(1);
    const PmtMlp& mlp = M->row_mlp[which];
    const int lane = threadIdx.x & 63, g = lane >> 4, wave = uniform((int)(threadIdx.x >> 6));
    const int tile0 = (blockIdx.x * PMT_WAVES + wave) * PMT_RT;
    const int in_dim = uniform(mlp.in_dim), out_dim = uniform(mlp.out_dim), n_ops = uniform(mlp.n_ops);
    float* stash_tile[PMT_RT];
    unsigned present = 0;
#pragma unroll
    for (int rt = 0; rt < PMT_RT; ++rt) {
        if ((tile0 + rt) * 16 < n_rows) present |= 1u << rt;
        stash_tile[rt] = TRAIN ? stash + (size_t)(tile0 + rt) * (size_t)((n_ops - 1) * PMT_SLOT_FLOATS) : nullptr;
    }
    PmtDrop drop = drop_setup(M, dropout_seed, uniform(mlp.dropout));
#pragma unroll
    for (int rt = 0; rt < PMT_RT; ++rt) drop.row[rt] = (tile0 + rt) * 16 + (lane & 15);
    f4 x[PMT_RT][NT];
    int slot = 0, op_begin = 0;
    if (in_dim > PMT_MAX_WIDTH) {  // wide first linear (checked on the host: op 0 is LINEAR)
        f4 xin[PMT_RT][ROWS_NTIN];
        load_rows<ROWS_NTIN>(xin, in, in_stride, n_rows, in_dim, tile0, g);
        ROWS_EV(2);
        const PmtOp& o = mlp.ops[0];
        const PmtLinear& L = M->lin[uniform(o.lin[0])];
        const int b_pvec = uniform(L.b_pvec);
        init_bias<NT>(x, b_pvec >= 0 ? packed + b_pvec : nullptr, uniform(L.out_dim), g);
        linear_acc<ROWS_NTIN, NT, false>(x, xin, packed + uniform(L.w_frag), in_dim, uniform(L.out_dim));
        if (drop.on) drop_apply<NT>(drop, uniform(o.lin[0]), x, g);
        if (uniform(o.selu_after) != 0) {
#pragma unroll
            for (int rt = 0; rt < PMT_RT; ++rt)
#pragma unroll
                for (int t = 0; t < NT; ++t) x[rt][t] = selu4(x[rt][t]);
        }
        op_begin = 1;
    } else {
        load_rows<NT>(x, in, in_stride, n_rows, in_dim, tile0, g);
    }
    ROWS_EV(3);
    run_mlp<TRAIN, NT, false>(M, mlp, x, theta, g, present, stash_tile, slot, 1, packed, op_begin, uniform(mlp.n_ops), &drop);
    ROWS_EV(4);
    const int r = lane & 15;
#pragma unroll
    for (int rt = 0; rt < PMT_RT; ++rt) {
        const int row = (tile0 + rt) * 16 + r;
        if (row < n_rows) {
            float* p = out + (size_t)row * out_stride;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (feat_of(t, j, g) < out_dim) p[feat_of(t, j, g)] = x[rt][t][j];
        }
    }
#if PMT_ROWS_TRACE
    ROWS_EV(5);
    if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0)
        for (int k = 1; k < 6; ++k) out[(size_t)(tile0 * 16) * out_stride + k - 1] = (float)(tr[k] - tr[0]);
#endif
}

#ifndef PMT_ROWS_DBG
#define PMT_ROWS_DBG 0  // development: BwdCtx.dbg knock-out bits for the row kernels (wrong results; timing only)
#endif
struct RowsBwdShared {
    float aux[PMT_WAVES][PMT_AUX_CAP];
    int aux_dst[PMT_AUX_CAP];
    f4 stage[PMT_STAGE_PLANES * 64];
};

template <int NT>
__global__ __launch_bounds__(PMT_THREADS, 2) void pmt_rows_backward_kernel(
    const PmtModel* __restrict__ M, int which, const float* __restrict__ theta, const float* __restrict__ packed,
    const float* __restrict__ in, long long in_stride, int n_rows, const float* __restrict__ d_out, long long d_out_stride,
    const float* __restrict__ stash, float* __restrict__ gtheta, float* __restrict__ d_in, long long d_in_stride,
    float d_in_scale, float* __restrict__ replicas, int rep_lo, int rep_span, int rep_count, unsigned long long dropout_seed) {
    __shared__ __attribute__((aligned(16))) RowsBwdShared sh;
    // Every workgroup adds its weight-gradient blocks to the SAME addresses at the same point of the same program: with
    // hundreds of workgroups the L2 serialises those float atomics (78 of 164 us at 65 536 rows).  With `replicas` workgroup b
    // adds into copy b % rep_count of the MLP's span [rep_lo, rep_lo + rep_span) of the gradient buffer instead
    // (pmt_rows_fold_kernel sums the copies).
    if (replicas != nullptr) gtheta = replicas + (size_t)(blockIdx.x % rep_count) * rep_span - rep_lo;
    const PmtMlp& mlp = M->row_mlp[which];
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, r = lane & 15, wave = uniform((int)(tid >> 6));
    const int tile0 = (blockIdx.x * PMT_WAVES + wave) * PMT_RT;
    const int in_dim = uniform(mlp.in_dim), out_dim = uniform(mlp.out_dim), n_ops = uniform(mlp.n_ops);
    unsigned present = 0;
    const float* stash_tile[PMT_RT];
#pragma unroll
    for (int rt = 0; rt < PMT_RT; ++rt) {
        if ((tile0 + rt) * 16 < n_rows) present |= 1u << rt;
        stash_tile[rt] = stash + (size_t)(tile0 + rt) * (size_t)((n_ops - 1) * PMT_SLOT_FLOATS);
    }
    // (phi / gphi are never dereferenced for row MLPs -- all their leaves are direct -- but passing nullptr constants here
    //  makes hipcc 7.2's SimplifyCFG crash while folding grad_ptr(), so theta / gtheta stand in)
    const int wg_tiles = min(PMT_GROUP_TILES, ((n_rows + 15) >> 4) - (int)blockIdx.x * PMT_GROUP_TILES);
    BwdCtx c{M, theta, theta, packed, gtheta, gtheta, &sh.stage[0], &sh.aux[0][0], &sh.aux_dst[0], g, present,
             wave * PMT_RT, wg_tiles, wg_tiles, 0, PMT_ROWS_DBG, nullptr};
    PmtDrop drop = drop_setup(M, dropout_seed, uniform(mlp.dropout));
#pragma unroll
    for (int rt = 0; rt < PMT_RT; ++rt) drop.row[rt] = (tile0 + rt) * 16 + r;
    c.drop = &drop;
    // d(out) -> registers (zero for padding rows: they then contribute nothing to any weight gradient)
    f4 dy[PMT_RT][NT];
#pragma unroll
    for (int rt = 0; rt < PMT_RT; ++rt) {
        const int row = (tile0 + rt) * 16 + r;
        const float* p = d_out + (size_t)row * d_out_stride;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int f = feat_of(t, j, g);
                dy[rt][t][j] = (row < n_rows && f < out_dim) ? p[f] : 0.f;
            }
    }
    auto load_input = [&](int op, f4 (&x)[PMT_RT][NT]) {
        if (op == 0) {
            load_rows<NT>(x, in, in_stride, n_rows, in_dim, tile0, g);
        } else {
#pragma unroll
            for (int rt = 0; rt < PMT_RT; ++rt) {
#pragma unroll
                for (int t = 0; t < NT; ++t) x[rt][t] = f4{0.f, 0.f, 0.f, 0.f};
                if (present & (1u << rt)) stash_load<NT>(stash_tile[rt] + (op - 1) * PMT_SLOT_FLOATS, x[rt]);
            }
        }
    };
    const bool wide = in_dim > PMT_MAX_WIDTH;
    const bool want_d_in = d_in != nullptr;
    const int first_op = wide ? 1 : 0;
    mlp_backward<NT, false>(c, mlp, dy, want_d_in || wide, load_input, first_op, n_ops);
    if (wide) {  // op 0 is a LINEAR with up to 128 inputs: weight gradient only (its input needs no gradient)
        const PmtOp& o = mlp.ops[0];
        const PmtLinear& L = M->lin[uniform(o.lin[0])];
        f4 xin[PMT_RT][ROWS_NTIN];
        load_rows<ROWS_NTIN>(xin, in, in_stride, n_rows, in_dim, tile0, g);
        if (uniform(o.selu_after) != 0) {
            f4 y[PMT_RT][NT];
            const int b_pvec = uniform(L.b_pvec);
            init_bias<NT>(y, b_pvec >= 0 ? packed + b_pvec : nullptr, uniform(L.out_dim), g);
            linear_acc<ROWS_NTIN, NT, false>(y, xin, packed + uniform(L.w_frag), in_dim, uniform(L.out_dim));
            if (drop.on) drop_apply<NT>(drop, uniform(o.lin[0]), y, g);
#pragma unroll
            for (int rt = 0; rt < PMT_RT; ++rt)
#pragma unroll
                for (int t = 0; t < NT; ++t) dy[rt][t] = selu_bwd4(dy[rt][t], selu4(y[rt][t]));
        }
        if (drop.on) drop_apply<NT>(drop, uniform(o.lin[0]), dy, g);
        linear_wgrad<NT, ROWS_NTIN>(c, L, dy, xin);
    }
    aux_flush(c);  // skip-block alphas
    if (!wide && want_d_in) {
#pragma unroll
        for (int rt = 0; rt < PMT_RT; ++rt) {
            const int row = (tile0 + rt) * 16 + r;
            if (row < n_rows) {
                float* p = d_in + (size_t)row * d_in_stride;
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (feat_of(t, j, g) < in_dim) p[feat_of(t, j, g)] = d_in_scale * dy[rt][t][j];
            }
        }
    }
}

// gradient replicas -> grad_theta[lo, lo + span); leaves the replicas zero for the next launch
#define ROWS_FOLD_SLICES 4
#define ROWS_REPLICAS 256
__global__ __launch_bounds__(256) void pmt_rows_fold_kernel(float* __restrict__ replicas, int used, int span, float* __restrict__ dst) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= span) return;
    float sum = 0.f;
    int r = blockIdx.y;
    for (; r + 3 * ROWS_FOLD_SLICES < used; r += 4 * ROWS_FOLD_SLICES) {  // four loads in flight
        float* p0 = replicas + (size_t)r * span + i;
        float* p1 = p0 + (size_t)ROWS_FOLD_SLICES * span;
        float* p2 = p1 + (size_t)ROWS_FOLD_SLICES * span;
        float* p3 = p2 + (size_t)ROWS_FOLD_SLICES * span;
        const float v0 = *p0, v1 = *p1, v2 = *p2, v3 = *p3;
        *p0 = 0.f; *p1 = 0.f; *p2 = 0.f; *p3 = 0.f;
        sum += (v0 + v1) + (v2 + v3);
    }
    for (; r < used; r += ROWS_FOLD_SLICES) {
        float* p = replicas + (size_t)r * span + i;
        sum += *p;
        *p = 0.f;
    }
    if (sum != 0.f) atomicAdd(dst + i, sum);
}

// [lo, hi) of theta that holds every parameter of a row MLP (its leaves are direct: offsets into theta)
static bool rows_param_span(const PmtModel* m, int which, int* lo, int* hi) {
    const PmtMlp* mlp = &m->row_mlp[which];
    int a = INT32_MAX, b = -1;
    for (int op = 0; op < mlp->n_ops; ++op) {
        const PmtOp* o = &mlp->ops[op];
        const int nl = o->kind == PMT_OP_SKIP ? o->n_layers : 1;
        if (o->kind == PMT_OP_SKIP) {
            if (o->alpha_src < 0) return false;
            a = o->alpha_src < a ? o->alpha_src : a;
            b = o->alpha_src + 1 > b ? o->alpha_src + 1 : b;
        }
        for (int l = 0; l < nl; ++l) {
            const PmtLinear* L = &m->lin[o->lin[l]];
            if (L->w_src < 0 || L->b_src < -1) return false;
            a = L->w_src < a ? L->w_src : a;
            b = L->w_src + L->in_dim * L->out_dim > b ? L->w_src + L->in_dim * L->out_dim : b;
            if (L->b_src >= 0) {
                a = L->b_src < a ? L->b_src : a;
                b = L->b_src + L->out_dim > b ? L->b_src + L->out_dim : b;
            }
        }
    }
    if (b <= a || b - a > (1 << 16)) return false;  // (scattered leaves: replicas of the whole range would not pay)
    *lo = a; *hi = b;
    return true;
}

extern "C" size_t pmt_rows_workspace_floats(const PmtModel* m, int which) {
    int lo, hi;
    if (!m || which < 0 || which > 2 || m->row_mlp[which].n_ops < 1 || !rows_param_span(m, which, &lo, &hi)) return 0;
    return (size_t)ROWS_REPLICAS * (size_t)(hi - lo);
}

static int rows_check(const PmtModel* m, int which, int n_rows) {
    if (!m || which < 0 || which > 2 || n_rows < 0) return PMT_E_INVALID;
    if (m->row_mlp[which].n_ops < 1) return PMT_E_INVALID;
    return pmt_model_check(m);
}

extern "C" size_t pmt_rows_stash_bytes(const PmtModel* m, int which, int32_t n_rows) {
    if (!m || which < 0 || which > 2) return 0;
    const size_t tiles = ((size_t)n_rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK * (PMT_WAVES * PMT_RT);
    const int slots = m->row_mlp[which].n_ops > 1 ? m->row_mlp[which].n_ops - 1 : 1;
    return tiles * slots * PMT_SLOT_FLOATS * sizeof(float);
}

// more than 64 KiB of dynamic LDS needs the function attribute (raised once per device and kernel; see cnn3_allow_lds)
#include <mutex>
static bool rows_allow_lds(const void* kernel, size_t bytes, int which, hipStream_t stream) {
    static std::mutex mu;
    static size_t allowed[64][4] = {};
    int dev = 0;
    if (hipStreamGetDevice(stream, &dev) != hipSuccess && hipGetDevice(&dev) != hipSuccess) dev = 0;
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
// [lo, lo + len) of `packed` that holds every fragment and bias of a row MLP (its linears are lowered one after the other)
static bool rows_packed_span(const PmtModel* m, int which, int* lo, int* len) {
    const PmtMlp* mlp = &m->row_mlp[which];
    long long a = INT32_MAX, b = -1;
    for (int op = 0; op < mlp->n_ops; ++op) {
        const PmtOp* o = &mlp->ops[op];
        const int nl = o->kind == PMT_OP_SKIP ? o->n_layers : 1;
        for (int l = 0; l < nl; ++l) {
            const PmtLinear* L = &m->lin[o->lin[l]];
            const long long frag = (long long)((L->out_dim + 15) / 16) * ((L->in_dim + 15) / 16) * 256;
            if (L->w_frag < 0 || L->wt_frag < 0) return false;
            a = L->w_frag < a ? L->w_frag : a;
            b = L->wt_frag + frag > b ? L->wt_frag + frag : b;
            if (L->b_pvec >= 0 && (L->b_pvec < L->w_frag || L->b_pvec >= L->wt_frag)) return false;  // (the bias lies between the two)
        }
    }
    if (b <= a || (a & 3) || ((b - a) & 3)) return false;
    *lo = (int)a; *len = (int)(b - a);
    return true;
}
#define ROWS_LDS_MAX_FLOATS (19 * 1024)  // 76 KiB: two workgroups per compute unit
// 2 when every activation behind the input fits two tiles (the input itself may take the wide first-linear path), else PMT_NT
static int rows_nt(const PmtModel* m, int which) {
    const PmtMlp* mlp = &m->row_mlp[which];
    if (mlp->out_dim > 32 || (mlp->in_dim > 32 && mlp->in_dim <= PMT_MAX_WIDTH)) return PMT_NT;
    for (int op = 0; op < mlp->n_ops; ++op) {
        const PmtOp* o = &mlp->ops[op];
        const int nl = o->kind == PMT_OP_SKIP ? o->n_layers : 1;
        for (int l = 0; l < nl; ++l) {
            const PmtLinear* L = &m->lin[o->lin[l]];
            const bool wide_first = op == 0 && l == 0 && mlp->in_dim > PMT_MAX_WIDTH;
            if (L->out_dim > 32 || (L->in_dim > 32 && !wide_first)) return PMT_NT;
        }
    }
    return 2;
}

extern "C" int pmt_rows_forward(const PmtModel* model_host, const PmtModel* model_dev, int which, const float* theta,
                                const float* packed, const float* in, int64_t in_stride, int32_t n_rows, float* out,
                                int64_t out_stride, float* stash, uint64_t dropout_seed, void* stream) {
    const int rc = rows_check(model_host, which, n_rows);
    if (rc) return rc;
    if (!model_dev || !theta || !packed || !in || !out) return PMT_E_INVALID;
    if (n_rows == 0) return PMT_OK;
    const int grid = (n_rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    int lo = 0, len = 0;
    const bool wlds = rows_packed_span(model_host, which, &lo, &len) && len <= ROWS_LDS_MAX_FLOATS;
    const bool narrow = rows_nt(model_host, which) == 2;
    auto kernel = stash ? (wlds ? pmt_rows_forward_kernel<true, true, PMT_NT> : pmt_rows_forward_kernel<true, false, PMT_NT>)
                        : (wlds ? pmt_rows_forward_kernel<false, true, PMT_NT> : pmt_rows_forward_kernel<false, false, PMT_NT>);
    if (narrow)
        kernel = stash ? (wlds ? pmt_rows_forward_kernel<true, true, 2> : pmt_rows_forward_kernel<true, false, 2>)
                       : (wlds ? pmt_rows_forward_kernel<false, true, 2> : pmt_rows_forward_kernel<false, false, 2>);
    const size_t lds = wlds ? (size_t)len * sizeof(float) : 0;
    if (lds > 64 * 1024 && !rows_allow_lds(reinterpret_cast<const void*>(kernel), lds, (stash ? 1 : 0) + (narrow ? 2 : 0), s)) return PMT_E_LAUNCH;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(PMT_THREADS), lds, s, model_dev, which, theta, packed, in, (long long)in_stride, n_rows, out,
                       (long long)out_stride, stash, (unsigned long long)dropout_seed, lo, len);
    return hipGetLastError() == hipSuccess ? PMT_OK : PMT_E_LAUNCH;
}

extern "C" int pmt_rows_backward(const PmtModel* model_host, const PmtModel* model_dev, int which, const float* theta,
                                 const float* packed, const float* in, int64_t in_stride, int32_t n_rows, const float* d_out,
                                 int64_t d_out_stride, const float* stash, float* grad_theta, float* d_in,
                                 int64_t d_in_stride, float d_in_scale, float* workspace, size_t workspace_floats, uint64_t dropout_seed,
                                 void* stream) {
    const int rc = rows_check(model_host, which, n_rows);
    if (rc) return rc;
    if (!model_dev || !theta || !packed || !in || !d_out || !stash || !grad_theta) return PMT_E_INVALID;
    if (model_host->row_mlp[which].in_dim > PMT_MAX_WIDTH && d_in) return PMT_E_UNSUPPORTED;
    if (n_rows == 0) return PMT_OK;
    const int grid = (n_rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
    // gradient replicas pay once several workgroups collide on an address
    int lo = 0, hi = 0;
    const bool rep = workspace != nullptr && grid >= 8 && rows_param_span(model_host, which, &lo, &hi) &&
                     workspace_floats >= (size_t)ROWS_REPLICAS * (size_t)(hi - lo);
    const int used = grid < ROWS_REPLICAS ? grid : ROWS_REPLICAS;
    auto kernel = rows_nt(model_host, which) == 2 ? pmt_rows_backward_kernel<2> : pmt_rows_backward_kernel<PMT_NT>;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(PMT_THREADS), 0, reinterpret_cast<hipStream_t>(stream),
                       model_dev, which, theta, packed, in, (long long)in_stride, n_rows, d_out, (long long)d_out_stride, stash,
                       grad_theta, d_in, (long long)d_in_stride, d_in_scale, rep ? workspace : nullptr, lo, hi - lo, used,
                       (unsigned long long)dropout_seed);
    if (rep)
        hipLaunchKernelGGL(pmt_rows_fold_kernel, dim3((hi - lo + 255) / 256, ROWS_FOLD_SLICES), dim3(256), 0,
                           reinterpret_cast<hipStream_t>(stream), workspace, used, hi - lo, grad_theta + lo);
    return hipGetLastError() == hipSuccess ? PMT_OK : PMT_E_LAUNCH;
}
