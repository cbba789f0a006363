// Device functions shared by every backward kernel (read-set backward, per-variant row MLPs): weight-gradient
// accumulation through per-wave LDS transposes, LayerNorm / SELU backward, the MLP-program backward interpreter.
#pragma once
#include "pmt_device.hpp"

#define TR_STRIDE 20  // floats per row of the per-wave transpose tile (16 + 4 pad: conflict-free b32 writes, 16B-aligned b128 reads)
#ifndef PMT_WG_COLS
#define PMT_WG_COLS PMT_MAX_WIDTH  // columns (input features) of the LDS weight-gradient tile; a TU may widen it
#endif
#define WG_TILE (PMT_MAX_WIDTH * PMT_WG_COLS)

DEV float read_lanes_sum(float v) {  // sum over the 16 reads of a tile (lanes with equal lane >> 4)
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 8);
    return v;
}
DEV float wave_sum(float v) { return group_sum(read_lanes_sum(v)); }

// C-layout registers of one 16x16 (feature x read) tile -> "reads on k" operand: element ks = value of feature position
// p = lane & 15 for read 4 * (lane >> 4) + ks.
//
// Done on the matrix core, with no LDS round trip: a C-layout register is also a valid A operand (m = read = lane & 15,
// k = lane >> 4 <-> position 4*(lane>>4) + j at k-step j), so  D = X^T * I  with the 16x16 identity as B lands the tile
// transposed in C layout (row 4G+J = read, column = position).  Multiplying by exact 0 / 1 makes it bit-exact.  Four
// MFMAs per tile on a pipe that is mostly idle in this kernel, instead of 4 ds_write + 1 ds_read and two LDS latencies.
DEV f4 transpose_tile(float* __restrict__ /*unused LDS scratch*/, f4 v) {
    const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
    f4 o = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) o = mfma16(v[j], (n == 4 * g + j) ? 1.0f : 0.0f, o);
    return o;
}

// Accumulate this wave's contribution to dW (and db) of one linear into the shared tile buffer.
//   dy: [out_v] gradient w.r.t. the linear's output, x: [in_v] its input; rows of padding reads carry dy = 0.
template <int NTO, int NTI>
DEV void wgrad_accumulate(float* __restrict__ wgbuf, float* __restrict__ tr, const f4 (&dy)[PMT_RT][NTO],
                          const f4 (&x)[PMT_RT][NTI], int out_v, int in_v, unsigned tile_mask, bool with_bias) {
    const int lane = threadIdx.x & 63, g = lane >> 4;
    const int nmt = (out_v + 15) >> 4, nkt = (in_v + 15) >> 4;
    f4 xT[PMT_RT][NTI];
#pragma unroll
    for (int rt = 0; rt < PMT_RT; ++rt)
#pragma unroll
        for (int it = 0; it < NTI; ++it)
            if ((tile_mask & (1u << rt)) && it < nkt) xT[rt][it] = transpose_tile(tr, x[rt][it]);
#pragma unroll
    for (int ot = 0; ot < NTO; ++ot) {
        if (ot < nmt && tile_mask) {
            f4 acc[NTI];
#pragma unroll
            for (int it = 0; it < NTI; ++it) acc[it] = f4{0.f, 0.f, 0.f, 0.f};
            f4 bsum = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int rt = 0; rt < PMT_RT; ++rt) {
                if (tile_mask & (1u << rt)) {
                    const f4 dT = transpose_tile(tr, dy[rt][ot]);
                    bsum = bsum + dy[rt][ot];
#pragma unroll
                    for (int it = 0; it < NTI; ++it)
                        if (it < nkt) {
#pragma unroll
                            for (int ks = 0; ks < 4; ++ks) acc[it] = mfma16(dT[ks], xT[rt][it][ks], acc[it]);
                        }
                }
            }
#pragma unroll
            for (int it = 0; it < NTI; ++it)
                if (it < nkt) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        atomicAdd(&wgbuf[(16 * ot + 4 * g + j) * PMT_WG_COLS + 16 * it + (lane & 15)], acc[it][j]);
                }
            if (with_bias) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float s = read_lanes_sum(bsum[j]);
                    if ((lane & 15) == 0) atomicAdd(&wgbuf[WG_TILE + 16 * ot + 4 * g + j], s);
                }
            }
        }
    }
}

DEV int pos_to_feat(int p) { return 16 * (p >> 4) + 4 * (p & 3) + ((p & 15) >> 2); }
DEV int split_row_dev(int v, int h) {
    if (h <= 0) return v;
    if (v < 16) return v < h ? v : -1;
    return (v - 16) < h ? h + (v - 16) : -1;
}

// After a workgroup barrier: add the shared tile into the flat gradient buffers with global float atomics and clear it.
DEV void wgrad_flush(float* __restrict__ wgbuf, const PmtLinear& L, float scale, float* __restrict__ gtheta,
                     float* __restrict__ gphi) {
    const int h = L.out_split, out_dim = L.out_dim, in_dim = L.in_dim;
    const int out_v = h > 0 ? 16 + h : out_dim;
    const int nmt = (out_v + 15) >> 4, nkt = (in_dim + 15) >> 4;
    float* gw = grad_ptr(L.w_src, gtheta, gphi);
    for (int i = threadIdx.x; i < nmt * 16 * nkt * 16; i += PMT_THREADS) {
        const int po = i / (nkt * 16), pi = i - po * (nkt * 16);
        float* cell = &wgbuf[po * PMT_WG_COLS + pi];
        const float v = *cell;
        *cell = 0.f;
        const int o = split_row_dev(pos_to_feat(po), h), c = pos_to_feat(pi);
        if (o >= 0 && o < out_dim && c < in_dim && pos_to_feat(po) < out_v) atomicAdd(&gw[(size_t)o * in_dim + c], scale * v);
    }
    if (L.b_src != -1) {
        float* gb = grad_ptr(L.b_src, gtheta, gphi);
        for (int p = threadIdx.x; p < nmt * 16; p += PMT_THREADS) {
            float* cell = &wgbuf[WG_TILE + p];
            const float v = *cell;
            *cell = 0.f;
            const int o = split_row_dev(pos_to_feat(p), h);
            if (o >= 0 && o < out_dim && pos_to_feat(p) < out_v) atomicAdd(&gb[o], scale * v);
        }
    }
}

// per-feature parameter gradient (tile-position registers summed over this wave's reads) -> global atomics
template <int NT>
DEV void vec_grad_atomic(float* __restrict__ dst, const f4 (&v)[NT], int dim, int g) {
    const int nt = (dim + 15) >> 4;
#pragma unroll
    for (int t = 0; t < NT; ++t)
        if (t < nt) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float s = read_lanes_sum(v[t][j]);
                const int f = feat_of(t, j, g);
                if ((threadIdx.x & 15) == 0 && f < dim) atomicAdd(dst + f, s);
            }
        }
}
DEV void scalar_grad_atomic(float* __restrict__ dst, float v) {
    const float s = wave_sum(v);
    if ((threadIdx.x & 63) == 0) atomicAdd(dst, s);
}

// LayerNorm backward for one read tile: given d(y) with y = xhat*w + b, returns d(x); accumulates dw, db partials.
template <int NT>
DEV void layernorm_bwd_tile(f4 (&dx)[NT], const f4 (&dyv)[NT], const f4 (&xhat)[NT], float rstd, int dim,
                            const f4 (&w)[NT], f4 (&dw)[NT], f4 (&db)[NT], int g) {
    const int nt = (dim + 15) >> 4;
    float s1 = 0.f, s2 = 0.f;
    f4 dxh[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        dxh[t] = f4{0.f, 0.f, 0.f, 0.f};
        if (t < nt) {
            dw[t] = dw[t] + dyv[t] * xhat[t];
            db[t] = db[t] + dyv[t];
            dxh[t] = dyv[t] * w[t];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (feat_of(t, j, g) < dim) {
                    s1 += dxh[t][j];
                    s2 += dxh[t][j] * xhat[t][j];
                } else {
                    dxh[t][j] = 0.f;
                }
        }
    }
    const float m1 = group_sum(s1) / (float)dim, m2 = group_sum(s2) / (float)dim;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            dx[t][j] = (t < nt && feat_of(t, j, g) < dim) ? rstd * (dxh[t][j] - m1 - xhat[t][j] * m2) : 0.f;
}

// xhat = (x - mean) * rstd without the affine part
template <int NT>
DEV void layernorm_stats_tile(f4 (&xhat)[NT], float& rstd, const f4 (&x)[NT], int dim, int g) {
    const int nt = (dim + 15) >> 4;
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
        if (t < nt) s += (x[t][0] + x[t][1]) + (x[t][2] + x[t][3]);
    const float mean = group_sum(s) / (float)dim;
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float d = (t < nt && feat_of(t, j, g) < dim) ? x[t][j] - mean : 0.f;
            xhat[t][j] = d;
            q += d * d;
        }
    rstd = rsqrtf(group_sum(q) / (float)dim + PMT_LN_EPS);
#pragma unroll
    for (int t = 0; t < NT; ++t) xhat[t] = xhat[t] * rstd;
}

// acc += LayerNorm backward of d(y) (y = xhat * w + b); also accumulates dw, db.  No temporaries of width D besides xhat.
template <int NT>
DEV void layernorm_bwd_inplace_tile(f4 (&acc)[NT], const f4 (&dyv)[NT], const f4 (&xhat)[NT], float rstd, int dim,
                                    const f4 (&w)[NT], f4 (&dw)[NT], f4 (&db)[NT], int g) {
    const int nt = (dim + 15) >> 4;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
        if (t < nt) {
            dw[t] = dw[t] + dyv[t] * xhat[t];
            db[t] = db[t] + dyv[t];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (feat_of(t, j, g) < dim) {
                    const float d = dyv[t][j] * w[t][j];
                    s1 += d;
                    s2 += d * xhat[t][j];
                }
        }
    const float m1 = group_sum(s1) / (float)dim, m2 = group_sum(s2) / (float)dim;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (t < nt && feat_of(t, j, g) < dim) acc[t][j] += rstd * (dyv[t][j] * w[t][j] - m1 - xhat[t][j] * m2);
}

DEV f4 selu_bwd4(f4 d, f4 s) {
    return f4{d[0] * selu_grad_from_out(s[0]), d[1] * selu_grad_from_out(s[1]), d[2] * selu_grad_from_out(s[2]),
              d[3] * selu_grad_from_out(s[3])};
}

// d/dz of the reference's logerfc (exponentially_modified_gaussian.py:30-55)
DEV float dlogerfc_dev(float z) {
    if (z > 5.f) {
        const float z2 = z * z, z3 = z2 * z, z4 = z2 * z2, z5 = z4 * z, z6 = z4 * z2, z7 = z6 * z;
        const float q = -1.f / (2.f * z2) + 3.f / (4.f * z4) - 15.f / (8.f * z6);
        const float dq = 1.f / z3 - 3.f / z5 + 45.f / (4.f * z7);
        return -2.f * z - 1.f / z + dq / (1.f + q);
    }
    const float e = erfcf(z);
    return e > 1.0e-12f ? -1.1283791670955126f * expf(-z * z) / e : 0.f;
}

struct BwdCtx {
    const PmtModel* M;
    const float* theta;
    const float* phi;
    const float* packed;
    float* gtheta;
    float* gphi;
    float* wg;      // LDS: two CONTIGUOUS weight(+bias)-gradient tile buffers of WG_TILE + PMT_MAX_WIDTH floats each.
                    // One base pointer + offset (not an array of pointers): a runtime-indexed pointer array defeats
                    // address-space inference and turns every LDS atomic into a slow flat_atomic_add_f32.
    float* tr;      // this wave's transpose tile
    int g;
    unsigned mask_all;
    int wg_flip;    // which weight-gradient buffer the next linear uses
    int dbg;        // profiling aid (PmtBatch.debug_flags[1]): bit 0 skip weight gradients, bit 1 skip their flush only,
                    // bit 3 accumulate per-section cycle counts into prof[]
    unsigned long long* prof;  // device, 8 counters (see scripts/bwd_ablate.py); only touched when dbg bit 3 is set
};
DEV unsigned long long prof_now() { return __builtin_readcyclecounter(); }
DEV void prof_add(const BwdCtx& c, int slot, unsigned long long t0) {
    if ((c.dbg & 8) && c.prof != nullptr && (threadIdx.x & 63) == 0) atomicAdd(c.prof + slot, prof_now() - t0);
}

// One linear's weight/bias gradient: accumulate, workgroup barrier, flush.  Every wave of the group must call it.
template <int NTO, int NTI>
DEV void linear_wgrad(BwdCtx& c, const PmtLinear& L, const f4 (&dy)[PMT_RT][NTO], const f4 (&x)[PMT_RT][NTI], unsigned mask,
                      float scale = 1.0f) {
    if (c.dbg & 1) return;
    float* buf = c.wg + c.wg_flip * (WG_TILE + PMT_MAX_WIDTH);
    const int h = uniform(L.out_split);
    const int out_v = h > 0 ? 16 + h : uniform(L.out_dim);
    unsigned long long t0 = prof_now();
    wgrad_accumulate<NTO, NTI>(buf, c.tr, dy, x, out_v, uniform(L.in_dim), mask, uniform(L.b_src) != -1);
    prof_add(c, 0, t0);
    t0 = prof_now();
    __syncthreads();
    prof_add(c, 1, t0);
    t0 = prof_now();
    if (!(c.dbg & 2)) wgrad_flush(buf, L, scale, c.gtheta, c.gphi);
    prof_add(c, 2, t0);
    c.wg_flip ^= 1;
}

// backward of one MLP program.  dy (in/out): gradient w.r.t. the MLP output on entry, w.r.t. its input on exit
// (not computed for op 0 when need_input_grad is false).  in_slot(op) gives the stash slot of op's input.
template <int NT = PMT_NT, typename LoadInput>
DEV void mlp_backward(BwdCtx& c, const PmtMlp& mlp, f4 (&dy)[PMT_RT][NT], bool need_input_grad, LoadInput load_input,
                      int op_begin = 0) {
    const PmtModel* M = c.M;
    const int n_ops = uniform(mlp.n_ops);
    for (int op = n_ops - 1; op >= op_begin; --op) {
        const PmtOp& o = mlp.ops[op];
        f4 x[PMT_RT][NT];
        load_input(op, x);
        if (uniform(o.kind) == PMT_OP_LINEAR) {
            const PmtLinear& L = M->lin[uniform(o.lin[0])];
            const int in_dim = uniform(L.in_dim), out_dim = uniform(L.out_dim);
            if (uniform(o.selu_after) != 0) {  // recompute s = selu(Wx + b); dy <- dy * selu'(s)
                f4 y[PMT_RT][NT];
                init_bias<NT>(y, uniform(L.b_pvec) >= 0 ? c.packed + uniform(L.b_pvec) : nullptr, out_dim, c.g);
                linear_acc<NT, NT, false>(y, x, c.packed + uniform(L.w_frag), in_dim, out_dim, PMT_FULL_MASK);
#pragma unroll
                for (int rt = 0; rt < PMT_RT; ++rt)
#pragma unroll
                    for (int t = 0; t < NT; ++t) dy[rt][t] = selu_bwd4(dy[rt][t], selu4(y[rt][t]));
            }
            linear_wgrad<NT, NT>(c, L, dy, x, c.mask_all);
            if (op > 0 || need_input_grad) {
                f4 dx[PMT_RT][NT];
                init_bias<NT>(dx, nullptr, in_dim, c.g);
                linear_acc<NT, NT, false>(dx, dy, c.packed + uniform(L.wt_frag), out_dim, in_dim, PMT_FULL_MASK);
#pragma unroll
                for (int rt = 0; rt < PMT_RT; ++rt)
#pragma unroll
                    for (int t = 0; t < NT; ++t) dy[rt][t] = dx[rt][t];
            }
        } else {
            // y = x + alpha * f(x); f = L2(selu(L1(selu(x))))  (n = 2)   or   f = L1(selu(x))  (n = 1)
            // Register discipline: at most three width-sized arrays are live at once (dy + two); s0 = selu(x) is
            // recomputed from a second read of the stashed x instead of being kept across the L2 phase.
            const int nl = uniform(o.n_layers);
            const PmtLinear& L1 = M->lin[uniform(o.lin[0])];
            const PmtLinear& L2 = M->lin[uniform(o.lin[nl - 1])];
            const int width = uniform(L1.in_dim);
            const float alpha = uniform(c.theta[uniform(o.alpha_src)]);
            f4 s1[PMT_RT][NT];
            if (nl == 2) {  // s1 = selu(L1 selu(x) + b1)
                init_bias<NT>(s1, c.packed + uniform(L1.b_pvec), width, c.g);
                linear_acc<NT, NT, true>(s1, x, c.packed + uniform(L1.w_frag), width, width, PMT_FULL_MASK);
            }
#pragma unroll
            for (int rt = 0; rt < PMT_RT; ++rt)
#pragma unroll
                for (int t = 0; t < NT; ++t) s1[rt][t] = selu4(nl == 2 ? s1[rt][t] : x[rt][t]);
            {   // d(alpha) = sum dy . f,  f = L2 s1 + b2   (x's registers are free from here on)
                f4 f[PMT_RT][NT];
                init_bias<NT>(f, c.packed + uniform(L2.b_pvec), width, c.g);
                linear_acc<NT, NT, false>(f, s1, c.packed + uniform(L2.w_frag), width, width, PMT_FULL_MASK);
                float da = 0.f;
#pragma unroll
                for (int rt = 0; rt < PMT_RT; ++rt)
#pragma unroll
                    for (int t = 0; t < NT; ++t) da += (dy[rt][t][0] * f[rt][t][0] + dy[rt][t][1] * f[rt][t][1]) + (dy[rt][t][2] * f[rt][t][2] + dy[rt][t][3] * f[rt][t][3]);
                scalar_grad_atomic(c.gtheta + uniform(o.alpha_src), da);
            }
            __builtin_amdgcn_sched_barrier(0);
            // last layer: d(f) = alpha * dy
            linear_wgrad<NT, NT>(c, L2, dy, s1, c.mask_all, alpha);
            f4 d1[PMT_RT][NT];
            init_bias<NT>(d1, nullptr, width, c.g);
            linear_acc<NT, NT, false>(d1, dy, c.packed + uniform(L2.wt_frag), width, width, PMT_FULL_MASK);
#pragma unroll
            for (int rt = 0; rt < PMT_RT; ++rt)
#pragma unroll
                for (int t = 0; t < NT; ++t) d1[rt][t] = alpha * selu_bwd4(d1[rt][t], s1[rt][t]);  // d(h1) (n=2) or d(x) part (n=1)
            __builtin_amdgcn_sched_barrier(0);
            if (nl == 2) {
                f4 s0[PMT_RT][NT];  // second read of x (s1's registers are free now)
                load_input(op, s0);
#pragma unroll
                for (int rt = 0; rt < PMT_RT; ++rt)
#pragma unroll
                    for (int t = 0; t < NT; ++t) s0[rt][t] = selu4(s0[rt][t]);
                linear_wgrad<NT, NT>(c, L1, d1, s0, c.mask_all);
                f4 d0[PMT_RT][NT];
                init_bias<NT>(d0, nullptr, width, c.g);
                linear_acc<NT, NT, false>(d0, d1, c.packed + uniform(L1.wt_frag), width, width, PMT_FULL_MASK);
#pragma unroll
                for (int rt = 0; rt < PMT_RT; ++rt)
#pragma unroll
                    for (int t = 0; t < NT; ++t) dy[rt][t] = dy[rt][t] + selu_bwd4(d0[rt][t], s0[rt][t]);
            } else {
#pragma unroll
                for (int rt = 0; rt < PMT_RT; ++rt)
#pragma unroll
                    for (int t = 0; t < NT; ++t) dy[rt][t] = dy[rt][t] + d1[rt][t];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

