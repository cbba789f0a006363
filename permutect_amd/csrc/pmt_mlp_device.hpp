// The MLP layer-program interpreter (forward), shared by the read-set forward and the per-variant row kernels.
#pragma once
#include "pmt_device.hpp"

// One LINEAR op between register arrays of different tile counts (the first op of the read MLP, the last op of the
// reducer, the wide first op of a row MLP): y = act(W x + b).
// WI / WO / W (compile time, 0 = read the descriptor): the widths, for the instances that have them compiled in.
// BF: the matrix products on the bf16 pipe (linear_acc_bf16; exact instances, direct weight path only).
// DROP (+ drop): dropout behind the Linear, before the SELU (reference mlp.py:57-58).
// X1 (f16 instances): the input is exact in one f16 piece (linear_acc_f16).
template <int NTI, int NTO, bool EXACT, int WI = 0, int WO = 0, int BF = 0, bool DROP = false, bool X1 = false>
DEV void run_linear_op(const PmtModel* __restrict__ M, const PmtOp& o, f4 (&y)[PMT_RT][NTO], const f4 (&x)[PMT_RT][NTI],
                       int g, const float* __restrict__ packed, const PmtDrop* drop = nullptr) {
    const PmtLinear& L = M->lin[uniform(o.lin[0])];
    const int b_pvec = uniform(L.b_pvec), base = uniform(L.w_frag);
    const float* st = packed + base;  // [fragments | bias]
    const int in_dim = WI ? WI : uniform(L.in_dim), out_dim = WO ? WO : uniform(L.out_dim);
    init_bias<NTO>(y, b_pvec >= 0 ? st + (b_pvec - base) : nullptr, out_dim, g);
    if constexpr (BF) linear_acc_mx<NTI, NTO, false, BF, X1>(y, x, packed, L);
    else linear_acc<NTI, NTO, false, EXACT, WI>(y, x, st, in_dim, out_dim);
    if constexpr (DROP) {
        if (drop != nullptr && drop->on != 0) drop_apply<NTO>(*drop, uniform(o.lin[0]), y, g);
    }
    if (uniform(o.selu_after) != 0) {
#pragma unroll
        for (int rt = 0; rt < PMT_RT; ++rt)
#pragma unroll
            for (int t = 0; t < NTO; ++t) y[rt][t] = selu4(y[rt][t]);
    }
}

// ---- MLP program interpreter (reference architecture/mlp.py) ------------------------------------------------------
// Runs ops [op_begin, op_end) on x in place; every op in the range maps NT tiles to NT tiles.  With TRAIN the INPUT of
// every op with index >= first_stashed_op is written to consecutive stash slots starting at `slot` (tiles in
// store_mask only).
// drop (generic instances only, nullptr = none): dropout behind every Linear (reference mlp.py:57-58), BEFORE the SELU that
// follows it; inside a skip block both Linears are followed by one, so f(x) itself is masked before alpha scales it.
template <bool TRAIN, int NT, bool EXACT, int W = 0, int BF = 0, bool DROP = !EXACT>
DEV void run_mlp(const PmtModel* __restrict__ M, const PmtMlp& mlp, f4 (&x)[PMT_RT][NT], const float* __restrict__ theta,
                 int g, unsigned store_mask, float* const (&stash_tile)[PMT_RT], int& slot, int first_stashed_op,
                 const float* __restrict__ packed, int op_begin, int op_end, const PmtDrop* drop = nullptr) {
    const bool dropping = DROP && drop != nullptr && drop->on != 0;
    for (int op = op_begin; op < op_end; ++op) {
        const PmtOp& o = mlp.ops[op];
        if (TRAIN && op >= first_stashed_op) {
#pragma unroll
            for (int rt = 0; rt < PMT_RT; ++rt)
                if (store_mask & (1u << rt)) stash_store<NT>(stash_tile[rt] + slot * PMT_SLOT_FLOATS, x[rt]);
            ++slot;
        }
        f4 y[PMT_RT][NT];
        if (uniform(o.kind) == PMT_OP_LINEAR) {
            if (dropping) {
                const PmtLinear& L = M->lin[uniform(o.lin[0])];
                const int b_pvec = uniform(L.b_pvec);
                init_bias<NT>(y, b_pvec >= 0 ? packed + b_pvec : nullptr, W ? W : uniform(L.out_dim), g);
                if constexpr (BF) linear_acc_mx<NT, NT, false, BF>(y, x, packed, L);
                else linear_acc<NT, NT, false, EXACT, W>(y, x, packed + uniform(L.w_frag), W ? W : uniform(L.in_dim), W ? W : uniform(L.out_dim));
                drop_apply<NT>(*drop, uniform(o.lin[0]), y, g);
                const bool act = uniform(o.selu_after) != 0;
#pragma unroll
                for (int rt = 0; rt < PMT_RT; ++rt)
#pragma unroll
                    for (int t = 0; t < NT; ++t) x[rt][t] = act ? selu4(y[rt][t]) : y[rt][t];
                continue;
            }
            run_linear_op<NT, NT, EXACT, W, W, BF>(M, o, y, x, g, packed);
#pragma unroll
            for (int rt = 0; rt < PMT_RT; ++rt)
#pragma unroll
                for (int t = 0; t < NT; ++t) x[rt][t] = y[rt][t];
        } else if (dropping) {
            // x + alpha * D2(L2(selu(D1(L1(selu(x))))))   (n = 2)   or   x + alpha * D1(L1(selu(x)))   (n = 1)
            const int nl = uniform(o.n_layers);
            const int width = W ? W : uniform(M->lin[uniform(o.lin[0])].in_dim);
            const float alpha = uniform(theta[uniform(o.alpha_src)]);
            f4 f[PMT_RT][NT];
            if (nl >= 2) {
                const PmtLinear& L1 = M->lin[uniform(o.lin[0])];
                init_bias<NT>(y, packed + uniform(L1.b_pvec), width, g);
                if constexpr (BF) linear_acc_mx<NT, NT, true, BF>(y, x, packed, L1);
                else linear_acc<NT, NT, true, EXACT, W>(y, x, packed + uniform(L1.w_frag), width, width);
                drop_apply<NT>(*drop, uniform(o.lin[0]), y, g);
                if constexpr (!EXACT) {  // blocks of three and four layers (generic instances only: pmt_shape_id): the layers in between
                    for (int k = 1; k < nl - 1; ++k) {
                        const PmtLinear& Lk = M->lin[uniform(o.lin[k])];
                        init_bias<NT>(f, packed + uniform(Lk.b_pvec), width, g);
                        linear_acc<NT, NT, true, false>(f, y, packed + uniform(Lk.w_frag), width, width);
                        drop_apply<NT>(*drop, uniform(o.lin[k]), f, g);
#pragma unroll
                        for (int rt = 0; rt < PMT_RT; ++rt)
#pragma unroll
                            for (int t = 0; t < NT; ++t) y[rt][t] = f[rt][t];
                    }
                }
            } else {
#pragma unroll
                for (int rt = 0; rt < PMT_RT; ++rt)
#pragma unroll
                    for (int t = 0; t < NT; ++t) y[rt][t] = x[rt][t];
            }
            const PmtLinear& L2 = M->lin[uniform(o.lin[nl - 1])];
            init_bias<NT>(f, packed + uniform(L2.b_pvec), width, g);
            if constexpr (BF) linear_acc_mx<NT, NT, true, BF>(f, y, packed, L2);
            else linear_acc<NT, NT, true, EXACT, W>(f, y, packed + uniform(L2.w_frag), width, width);
            drop_apply<NT>(*drop, uniform(o.lin[nl - 1]), f, g);
#pragma unroll
            for (int rt = 0; rt < PMT_RT; ++rt)
#pragma unroll
                for (int t = 0; t < NT; ++t) x[rt][t] = x[rt][t] + alpha * f[rt][t];
        } else {
            // x + alpha * f(x) with one or two (generic instances: up to four) (SELU, Linear) layers.  Two register arrays are live: the last
            // layer accumulates straight into x, with alpha folded into its B operand and bias.
            const int nl = uniform(o.n_layers);
            const int width = W ? W : uniform(M->lin[uniform(o.lin[0])].in_dim);
            if (nl == 1) {
#pragma unroll
                for (int rt = 0; rt < PMT_RT; ++rt)
#pragma unroll
                    for (int t = 0; t < NT; ++t) y[rt][t] = x[rt][t];
            } else {
                const PmtLinear& L1 = M->lin[uniform(o.lin[0])];
                const float* st1 = packed + uniform(L1.w_frag);
                init_bias<NT>(y, st1 + (uniform(L1.b_pvec) - uniform(L1.w_frag)), width, g);
                if constexpr (BF) linear_acc_mx<NT, NT, true, BF>(y, x, packed, L1);
                else linear_acc<NT, NT, true, EXACT, W>(y, x, st1, width, width);
                if constexpr (!EXACT) {
                    // Blocks of three and four (SELU, Linear) layers (reference mlp.py:15-22 takes any depth): the layers between the first
                    // and the last, one more register array while they run.  Generic instances only -- a model with such a block has
                    // pmt_shape_id 0 -- so that the exact instances keep their two live arrays (and their register budget).
                    for (int k = 1; k < nl - 1; ++k) {
                        const PmtLinear& Lk = M->lin[uniform(o.lin[k])];
                        f4 y2[PMT_RT][NT];
                        init_bias<NT>(y2, packed + uniform(Lk.b_pvec), width, g);
                        linear_acc<NT, NT, true, false>(y2, y, packed + uniform(Lk.w_frag), width, width);
#pragma unroll
                        for (int rt = 0; rt < PMT_RT; ++rt)
#pragma unroll
                            for (int t = 0; t < NT; ++t) y[rt][t] = y2[rt][t];
                    }
                }
            }
            const PmtLinear& L2 = M->lin[uniform(o.lin[nl - 1])];
            const float alpha = uniform(theta[uniform(o.alpha_src)]);
            const int nmt = (width + 15) >> 4;
            const float* st2 = packed + uniform(L2.w_frag);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (EXACT || t < nmt) {
                    const f4 b = alpha * load_pvec(st2 + (uniform(L2.b_pvec) - uniform(L2.w_frag)), t, g);
#pragma unroll
                    for (int rt = 0; rt < PMT_RT; ++rt) x[rt][t] = x[rt][t] + b;
                }
            }
            if constexpr (BF) linear_acc_mx<NT, NT, true, BF>(x, y, packed, L2, alpha);
            else linear_acc<NT, NT, true, EXACT, W>(x, y, st2, width, width, alpha);
        }
    }
}
