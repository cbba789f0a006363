// Device functions shared by every backward kernel (read-set backward, per-variant row MLPs): weight-gradient
// accumulation through per-wave LDS transposes, LayerNorm / SELU backward, the MLP-program backward interpreter.
#pragma once
#include "pmt_device.hpp"

// Operand precision of the backward's input-gradient products dx = W^T dy and of the layers it recomputes (pre-activations behind a
// SELU, the inner layer of a skip block).  The forward keeps every product fp32-equivalent (three bf16 pieces per operand, six
// MFMAs); in the backward the GRADIENT contract is 1e-4 relative, and the reference's own fp32 arithmetic sits 8.6e-6 from an fp64
// evaluation of the same step.  Two pieces per operand (16 significant bits) and the three first-order MFMAs put the HIP gradients
// at 7.3e-6 from fp64 -- the size of the reference's own error (7.9e-6 against 7.0e-6 at 4 096 read sets) -- where the six-MFMA
// form gives 1.1e-6, for 5.8 % of the kernel (scripts/grad_vs_fp64.py, tests/test_scale_gpu.py).  The weight gradients have contracted two-piece operands since round 2.
// -DPMT_DGRAD_PIECES=3 -DPMT_RECOMPUTE_PIECES=3 restores the six-MFMA form; PMT_TWO_PIECE_MFMAS=5 keeps the second-order terms.
#ifndef PMT_DGRAD_PIECES
#define PMT_DGRAD_PIECES 2
#endif
// The BF template argument threaded through the backward's helpers: bits 0-2 = the bf16 pieces of the activations (0: the fp32 instances),
// bit 3 (PMT_BF_PRIV) = the weight-gradient exchange adds into the workgroup's PRIVATE row (pmt_backward: partials), known at COMPILE time: the
// choice between the row's one store and the atomics' forest of lane branches used to be a run-time branch in every exchange, and a
// branch ends a scheduling region (2.10 -> 2.05 ms).
#define PMT_BF_PRIV 8
#define PMT_BF_PIECES(BF) ((BF) & 7)
#define PMT_DG(BF) (PMT_BF_PIECES(BF) == 3 ? PMT_DGRAD_PIECES : PMT_BF_PIECES(BF))
#ifndef PMT_RECOMPUTE_PIECES
#define PMT_RECOMPUTE_PIECES 2
#endif
#define PMT_RC(BF) (PMT_BF_PIECES(BF) == 3 ? PMT_RECOMPUTE_PIECES : PMT_BF_PIECES(BF))
#ifndef PMT_STAGE_PLANES
#define PMT_STAGE_PLANES 96  // LDS operand-exchange capacity in planes of 64 x float4 (1 KiB each); a TU may shrink it
#endif
#ifndef PMT_AUX_CAP
#define PMT_AUX_CAP 256      // floats per wave of the small-parameter gradient slab
#endif
#ifndef PMT_BWD_ABLATE
#define PMT_BWD_ABLATE 0  // development build (scripts/bwd_ablate.py): knock-outs of the exchange's parts through PmtBatch.debug_flags[1]
#endif

DEV float read_lanes_sum(float v) {  // sum over the 16 reads of a tile (lanes with equal lane >> 4)
    v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_mov<0x124>(v);  // row_ror 4
    v += dpp_mov<0x128>(v);  // row_ror 8
    return v;
}
DEV float wave_sum(float v) { return group_sum(read_lanes_sum(v)); }

// Touch one 4 KiB stash slot of a tile (64 lanes x one dword, 64 bytes apart) so that the lines are in L2 when the real load
// comes an op later.  The dwords land in a 256-byte LDS sink by LDS-DMA: no VGPR is written, nothing has to be waited for.
DEV void stash_prefetch(const float* __restrict__ slot, float* __restrict__ sink) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(slot + (pmt_tid() & 63) * 16),
                                     (__attribute__((address_space(3))) void*)sink, 4, 0, 0);
}

// The weight-gradient contraction runs over READS, so its operands are the transposes of the C-layout tiles: element ks of
// lane (G, n) = value of feature position n for read 4 G + ks (position n = 4 g + j is register j of lane group g).
// The first version transposed in registers on the matrix core (D = X^T * I, bit-exact, 4 MFMAs per tile); once the
// fp32 pipe became the bound that was a quarter of the block loop's MFMA work.
// The wgrad exchange needs the transposed tile in LDS, not in registers: there the transposition is free, it is only a
// matter of WHERE each lane stores its four values.  A plane of the stage is 64 slots of 16 bytes; slot sigma(G, n) holds
// X[position n][reads 4G .. 4G + 3], exactly what the lane (G, n) of the consuming wave feeds to the matrix core.
// sigma = 16 (n & 3) + 8 (n >> 3) + 2 (G ^ (n & 3)) + ((n >> 2) & 1)  is chosen so that the producer's four ds_write_b32
// (32-lane halves, bank = dword address mod 32) and the consumer's ds_read_b128 (lane groups {0-3, 12-15, 20-27}, ...;
// 16-byte slot mod 16) are both conflict-free (MI355X_MICROARCH.md, LDS).  Four stores on the LDS pipe replace four
// MFMAs on the fp32 pipe, which bounds this kernel.
DEV int stage_slot(int G, int n) { return 16 * (n & 3) + 8 * (n >> 3) + 2 * (G ^ (n & 3)) + ((n >> 2) & 1); }
DEV void stage_store_transposed(f4* __restrict__ plane, f4 v) {
    const int lane = pmt_tid() & 63, g = lane >> 4, r = lane & 15, G = r >> 2, J = r & 3;
    float* p = reinterpret_cast<float*>(plane) + 32 * (g >> 1) + 4 * (g & 1) + J;
#pragma unroll
    for (int j = 0; j < 4; ++j) p[64 * j + 8 * (G ^ j)] = v[j];
}

DEV int pos_to_feat(int p) { return 16 * (p >> 4) + 4 * (p & 3) + ((p & 15) >> 2); }
DEV int split_row_dev(int v, int h) {
    if (h <= 0) return v;
    if (v < PMT_SPLIT0) return v < h ? v : -1;
    return (v - PMT_SPLIT0) < h ? h + (v - PMT_SPLIT0) : -1;
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
    const float inv_dim = fast_rcp((float)dim);
    const float m1 = group_sum(s1) * inv_dim, m2 = group_sum(s2) * inv_dim;
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
    const float inv_dim = fast_rcp((float)dim);
    const float mean = group_sum(s) * inv_dim;
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float d = (t < nt && feat_of(t, j, g) < dim) ? x[t][j] - mean : 0.f;
            xhat[t][j] = d;
            q += d * d;
        }
    rstd = __builtin_amdgcn_rsqf(group_sum(q) * inv_dim + PMT_LN_EPS);
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
    const float inv_dim = fast_rcp((float)dim);
    const float m1 = group_sum(s1) * inv_dim, m2 = group_sum(s2) * inv_dim;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (t < nt && feat_of(t, j, g) < dim) acc[t][j] += rstd * (dyv[t][j] * w[t][j] - m1 - xhat[t][j] * m2);
}

DEV f4 selu_bwd4(f4 d, f4 s) {  // d * selu'(a) through the output s = selu(a); packed add / multiply
    const f4 neg = s + PMT_SELU_ALPHA * PMT_SELU_SCALE;
    const f4 gr = f4{s[0] > 0.f ? PMT_SELU_SCALE : neg[0], s[1] > 0.f ? PMT_SELU_SCALE : neg[1],
                     s[2] > 0.f ? PMT_SELU_SCALE : neg[2], s[3] > 0.f ? PMT_SELU_SCALE : neg[3]};
    return d * gr;
}

// d/dz of the reference's logerfc (exponentially_modified_gaussian.py:30-55).  Divisions are v_rcp_f32 multiplies (1 ulp: this
// feeds gradients, whose tolerance is four orders of magnitude wider); an IEEE division is ~10 instructions and the asymptotic
// branch alone had eight of them, evaluated by every lane whenever one read of the wave is beyond z = 5.
DEV float dlogerfc_dev(float z) {
    if (z > 5.f) {
        const float r = fast_rcp(z), r2 = r * r, r3 = r2 * r, r4 = r2 * r2, r5 = r4 * r, r6 = r4 * r2, r7 = r6 * r;
        const float q = -0.5f * r2 + 0.75f * r4 - 1.875f * r6;
        const float dq = r3 - 3.f * r5 + 11.25f * r7;
        return -2.f * z - r + dq * fast_rcp(1.f + q);
    }
    const float e = erfcf(z);
    return e > 1.0e-12f ? -1.1283791670955126f * expf(-z * z) * fast_rcp(e) : 0.f;
}

// Weight gradients contract over READS, and the reads of a workgroup are spread over its waves.  Summing per-wave
// partial dW tiles with LDS float atomics measured ~100 cycles per ds_add_f32 instruction (4.4 ms of a 16 ms kernel), so
// the waves exchange OPERANDS instead ("owner computes"): every wave writes the transposed tiles of dy and x of its
// reads into the LDS stage (plain 16-byte stores), and after one barrier each wave owns distinct 16x16 blocks of dW,
// which it contracts over ALL reads of the workgroup on the matrix core and adds straight from registers into the flat
// gradient buffer: no LDS atomics, no dW tile in LDS, no flush pass.
struct BwdCtx {
    const PmtModel* M;
    const float* theta;
    const float* phi;
    const float* packed;
    float* gtheta;
    float* gphi;
    f4* stage;      // LDS [PMT_STAGE_PLANES][64] operand exchange
    float* aux;     // LDS [PMT_WAVES][PMT_AUX_CAP]: per-wave sums of small-parameter gradients, reduced over the waves and
                    // added to global memory once per workgroup inside the next exchange round (or by aux_flush)
    int* aux_dst;   // LDS [PMT_AUX_CAP] destination of every slab entry (encoded like PmtLinear.w_src; -1 = none)
    int g;
    unsigned mask_all;  // which of this wave's tiles exist
    int slot0;      // tile slot (0 .. PMT_WG_TILES-1) of this wave's tile 0; the existing tiles of the workgroup are
    int ntiles;     //   exactly the slots [0, ntiles)
    int tiles_ref;  // slots below this belong to side 0 (ref) of a two-sided linear pair
    int aux_n;      // slab entries in use (wave-uniform, identical in every wave)
    int dbg;        // development switches (PmtBatch.debug_flags[1]): bit 0 skip weight gradients, bit 1 skip only their
                    // global atomics, bit 2 skip the gated blocks, bit 3 collect cycle counters, bit 4 drop small-parameter
                    // gradients.  0 in production.
    unsigned long long* prof;  // device, 24 counters (see scripts/bwd_ablate.py); only touched when dbg bit 3 is set
    // bf16 weight-gradient exchange (wgrad_exchange_bf; read-set kernel, exact-width instance only)
    int wr;         // waves [0, wr) hold ref tiles, [wr, PMT_WAVES) alt tiles (GroupGeom.wr)
    int wbase;      // this lane's byte offset inside a 1 KiB operand plane when it STORES its reads (stage_pair_bf16)
    int rbase;      // ... when it LOADS its MFMA operand (16 bytes)
    float* pf_sink; // LDS, 64 floats: where stash_prefetch drops its dwords
    float* priv = nullptr; // this workgroup's private row of weight-gradient partial sums, biased by -emit_base (pmt_backward.hip); nullptr = global atomics
    const PmtDrop* drop = nullptr;  // dropout of the MLP being differentiated (generic instances; nullptr / on = 0: none)
    int* trace = nullptr;  // development: this wave's event log (PmtBatch.debug_flags[2] selects ONE workgroup; scripts/bwd_trace.py)
    int trace_n = 0;
};
// one (event id, clock) pair per call; 250 events per wave at most.  Compiled in with -DPMT_BWD_TRACE=1 only (the log
// pointer and counter cost scalar registers the kernel does not have to spare).
#ifndef PMT_BWD_TRACE
#define PMT_BWD_TRACE 0
#endif
DEV void trace_ev(BwdCtx& c, int id) {
    if (!PMT_BWD_TRACE || c.trace == nullptr) return;
    if (c.trace_n < 250 && (pmt_tid() & 63) == 0) {
        c.trace[2 * c.trace_n] = id;
        c.trace[2 * c.trace_n + 1] = (int)__builtin_readcyclecounter();
    }
    ++c.trace_n;
}
// cycle counters per phase (scripts/bwd_ablate.py): compiled in with -DPMT_BWD_PROF=1 only -- the start stamps are live
// scalar registers through every phase, and the kernel has none to spare
#ifndef PMT_BWD_PROF
#define PMT_BWD_PROF 0
#endif
DEV unsigned long long prof_now() { return PMT_BWD_PROF ? __builtin_readcyclecounter() : 0ull; }
DEV void prof_add(const BwdCtx& c, int slot, unsigned long long t0) {
    if (PMT_BWD_PROF && (c.dbg & 8) && c.prof != nullptr && (pmt_tid() & 63) == 0) atomicAdd(c.prof + slot, prof_now() - t0);
}

// ---- small-parameter gradients ---------------------------------------------------------------------------------------
DEV int enc_phi(int off) { return -(off + 2); }
DEV int enc_at(int enc, int f) { return enc >= 0 ? enc + f : enc - f; }

// sum the slabs over the waves and add to global memory; callers bracket it with workgroup barriers
DEV void aux_reduce(BwdCtx& c) {
    static_assert(PMT_AUX_CAP <= PMT_THREADS, "one slab entry per thread");
    if (const int i = pmt_tid(); i < c.aux_n) {
        const int d = c.aux_dst[i];
        if (d != -1) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < PMT_WAVES; ++w) s += c.aux[w * PMT_AUX_CAP + i];
            if (!(PMT_BWD_ABLATE && (c.dbg & 16384))) atomicAdd(grad_ptr(d, c.gtheta, c.gphi), s);
        }
    }
    c.aux_n = 0;
}
DEV void aux_flush(BwdCtx& c) {
    lds_barrier();
    aux_reduce(c);
    lds_barrier();
}
// per-feature parameter gradient (tile-position registers, summed here over this wave's reads)
// CHECK = false (every push helper): the caller has shown at compile time that the slab has room -- the check is a uniform branch around two
// barriers and a loop, inlined at every push, and each one ends a scheduling region.
template <int NT, bool CHECK = true>
DEV void aux_push_vec(BwdCtx& c, int enc, const f4 (&v)[NT], int dim) {
    const int nt = (dim + 15) >> 4, lane = pmt_tid() & 63, wave = pmt_tid() >> 6;
    if (CHECK && c.aux_n + 16 * nt > PMT_AUX_CAP) aux_flush(c);
    if (c.dbg & 16) return;
    // all the cross-lane sums first, then ONE region with a sixteenth of the lanes enabled that stores them
    // (a region per element costs an exec save / restore and a branch each, and runs at full instruction cost)
    f4 sums[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) sums[t][j] = t < nt ? read_lanes_sum(v[t][j]) : 0.f;
    if ((lane & 15) == 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
            if (t < nt) {
                const int p = c.aux_n + 16 * t + 4 * c.g;
#pragma unroll
                for (int j = 0; j < 4; ++j) c.aux[wave * PMT_AUX_CAP + p + j] = sums[t][j];  // (p need not be 16-byte aligned)
                if (wave == 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int f = feat_of(t, j, c.g);
                        c.aux_dst[p + j] = f < dim ? enc_at(enc, f) : -1;
                    }
                }
            }
    }
    c.aux_n += 16 * nt;
}
// The same for arrays that fill their NT tiles (the exact instances): a HALVING butterfly over the 16 lanes of a row instead of
// 4 NT independent 4-step reductions.  At each step a lane keeps one half of its values and hands the other half to its
// partner (DPP: rotate by 4, rotate by 8, quad swaps), so the work halves every step: 15 exchange-adds for 16 values
// instead of 64, and at the end every lane holds ONE finished sum -- value index (b2 b3 b1 b0) of its lane number's bits
// -- which it stores itself: no 1-in-16 store region.
#define PMT_DPP_ADD(NAME, CTRL)                                                                                                 \
    DEV float NAME(float keep, float send) { /* keep + (send of the partner lane); s_nop: 2 wait states VALU write -> DPP read */ \
        float r;                                                                                                                \
        asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %2 " CTRL " row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(send), "v"(keep));      \
        return r;                                                                                                               \
    }
PMT_DPP_ADD(dpp_add_ror4, "row_ror:4")
PMT_DPP_ADD(dpp_add_ror8, "row_ror:8")
PMT_DPP_ADD(dpp_add_x2, "quad_perm:[2,3,0,1]")
PMT_DPP_ADD(dpp_add_x1, "quad_perm:[1,0,3,2]")
template <int STEP>
DEV float dpp_add_step(float keep, float send) {
    return STEP == 0 ? dpp_add_ror4(keep, send) : STEP == 1 ? dpp_add_ror8(keep, send) : STEP == 2 ? dpp_add_x2(keep, send) : dpp_add_x1(keep, send);
}
template <int N, int STEP>
DEV float row_halving_sum(const float (&v)[N], int lane) {  // returns the total of value index idx over the row's 16 lanes
    constexpr int BIT[4] = {4, 8, 2, 1};
    if constexpr (N == 1) {
        float r = v[0];
        if constexpr (STEP < 4) {
            const float one[1] = {dpp_add_step<STEP>(r, r)};
            return row_halving_sum<1, STEP + 1>(one, lane);
        }
        return r;
    } else {
        const bool up = (lane & BIT[STEP]) != 0;
        float h[N / 2];
#pragma unroll
        for (int k = 0; k < N / 2; ++k) h[k] = dpp_add_step<STEP>(up ? v[N / 2 + k] : v[k], up ? v[k] : v[N / 2 + k]);
        return row_halving_sum<N / 2, STEP + 1>(h, lane);
    }
}
template <int NT, bool CHECK = true>
DEV void aux_push_vec_full(BwdCtx& c, int enc, const f4 (&v)[NT], int dim) {
    static_assert(NT == 1 || NT == 2 || NT == 4, "4, 8 or 16 values per lane");
    const int lane = pmt_tid() & 63, wave = pmt_tid() >> 6;
    if (CHECK && c.aux_n + 16 * NT > PMT_AUX_CAP) aux_flush(c);
    if (c.dbg & 16) return;
    float vals[4 * NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) vals[4 * t + j] = v[t][j];
    const float total = row_halving_sum<4 * NT, 0>(vals, lane);
    // which value this lane ended up with: the halving steps took bits 2, 3, 1, 0 of the lane number, most significant first
    constexpr int STEPS = NT == 4 ? 4 : NT == 2 ? 3 : 2;
    const int bits[4] = {(lane >> 2) & 1, (lane >> 3) & 1, (lane >> 1) & 1, lane & 1};
    int idx = 0, spare = 0;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
        if (st < STEPS) idx = 2 * idx + bits[st];
        else spare |= bits[st];  // lanes that differ only in a bit summed AFTER the halving hold the same total: one of them stores
    }
    const int p = c.aux_n + 16 * (idx >> 2) + 4 * c.g + (idx & 3);
    if (spare == 0) c.aux[wave * PMT_AUX_CAP + p] = total;
    if (wave == 0 && spare == 0) {
        const int f = feat_of(idx >> 2, idx & 3, c.g);
        c.aux_dst[p] = f < dim ? enc_at(enc, f) : -1;
    }
    c.aux_n += 16 * NT;
}
template <int NT, bool FULL, bool CHECK = true>
DEV void aux_push_vec_x(BwdCtx& c, int enc, const f4 (&v)[NT], int dim) {
    if constexpr (FULL && (NT == 1 || NT == 2 || NT == 4)) aux_push_vec_full<NT, CHECK>(c, enc, v, dim);
    else aux_push_vec<NT, CHECK>(c, enc, v, dim);
}
// one 16-position row (tile 0) whose per-position totals already sit in every lane group: lane (g, p) holds position p
template <bool CHECK = true>
DEV void aux_push_row16(BwdCtx& c, int enc, float v, int dim) {
    const int lane = pmt_tid() & 63, wave = pmt_tid() >> 6;
    if (CHECK && c.aux_n + 16 > PMT_AUX_CAP) aux_flush(c);
    if (c.dbg & 16) return;
    if (lane < 16) {
        const int f = pos_to_feat(lane);
        c.aux[wave * PMT_AUX_CAP + c.aux_n + lane] = v;
        if (wave == 0) c.aux_dst[c.aux_n + lane] = f < dim ? enc_at(enc, f) : -1;
    }
    c.aux_n += 16;
}
// N scalar gradients at once (4 or 8; enc -1 = none): the halving butterfly of aux_push_vec_full over the 16 lanes of a row, then the
// four rows -- N + 5 exchange-adds instead of 6 N (a scalar's wave_sum is six exchange steps; the head and the gated blocks push
// ~60 scalars per group, 6 % of the kernel's vector instructions before this)
template <int N, bool CHECK = true>
DEV void aux_push_scalars(BwdCtx& c, const int (&enc)[N], const float (&v)[N]) {
    static_assert(N == 4 || N == 8, "4 or 8 scalars");
    const int lane = pmt_tid() & 63, wave = pmt_tid() >> 6;
    if (CHECK && c.aux_n + N > PMT_AUX_CAP) aux_flush(c);
    if (c.dbg & 16) return;
    const float total = group_sum(row_halving_sum<N, 0>(v, lane));
    constexpr int STEPS = N == 8 ? 3 : 2;
    const int bits[4] = {(lane >> 2) & 1, (lane >> 3) & 1, (lane >> 1) & 1, lane & 1};
    int idx = 0, spare = lane >> 4;  // (every row holds the totals: row 0 stores)
#pragma unroll
    for (int st = 0; st < 4; ++st) {
        if (st < STEPS) idx = 2 * idx + bits[st];
        else spare |= bits[st];
    }
    if (spare == 0) {
        c.aux[wave * PMT_AUX_CAP + c.aux_n + idx] = total;
        if (wave == 0) {
            int d = enc[0];
#pragma unroll
            for (int k = 1; k < N; ++k) d = idx == k ? enc[k] : d;
            c.aux_dst[c.aux_n + idx] = d;
        }
    }
    c.aux_n += N;
}
template <bool CHECK = true>
DEV void aux_push_scalar(BwdCtx& c, int enc, float v) {
    if (CHECK && c.aux_n + 1 > PMT_AUX_CAP) aux_flush(c);
    if (c.dbg & 16) return;
    const float s = wave_sum(v);
    if ((pmt_tid() & 63) == 0) {
        c.aux[(pmt_tid() >> 6) * PMT_AUX_CAP + c.aux_n] = s;
        if (pmt_tid() == 0) c.aux_dst[c.aux_n] = enc;
    }
    c.aux_n += 1;
}

// ---- weight gradients ------------------------------------------------------------------------------------------------
// dW (+ db) of one linear (SIDES = 1) or of a ref / alt pair applied to the two sides of the group (SIDES = 2).
// A WgradAcc holds this wave's blocks of dW in registers: init once, accumulate any number of exchange rounds (a
// persistent kernel keeps it across workgroup iterations), emit once with global float atomics.
template <int NTO, int NTI, int SIDES>
struct WgradAcc {
    static constexpr int TPW = (SIDES * NTO * NTI + PMT_WAVES - 1) / PMT_WAVES;  // blocks per wave
    f4 acc[TPW];
    float bs[TPW];                              // bias partials (blocks with it == 0)
    int t_side[TPW], t_ot[TPW], t_it[TPW];      // wave-uniform block coordinates; t_side < 0: no block
    int h, out_dim, in_dim, out_v, nmt, nkt;
    bool any[2];                                // a round with tiles on that side has been accumulated
};

template <int NTO, int NTI, int SIDES>
DEV void wgrad_init(WgradAcc<NTO, NTI, SIDES>& a, const PmtLinear& L0) {
    const int wave = uniform((int)(pmt_tid() >> 6));
    a.h = uniform(L0.out_split); a.out_dim = uniform(L0.out_dim); a.in_dim = uniform(L0.in_dim);
    a.out_v = a.h > 0 ? PMT_SPLIT0 + a.h : a.out_dim;
    a.nmt = (a.out_v + 15) >> 4; a.nkt = (a.in_dim + 15) >> 4;
    const int per_side = a.nmt * a.nkt, ntask = SIDES * per_side;
    a.any[0] = a.any[1] = false;
#pragma unroll
    for (int k = 0; k < a.TPW; ++k) {
        a.acc[k] = f4{0.f, 0.f, 0.f, 0.f};
        a.bs[k] = 0.f;
        const int q = wave + PMT_WAVES * k;
        a.t_side[k] = (SIDES == 2 && q >= per_side) ? 1 : 0;
        const int rem = q - a.t_side[k] * per_side;
        a.t_ot[k] = rem / a.nkt;
        a.t_it[k] = rem - a.t_ot[k] * a.nkt;
        if (q >= ntask) a.t_side[k] = -1;
    }
}

// One exchange: dy = gradient w.r.t. the linear's output (rows of padding reads are zero), x = its input, for the
// tiles of this wave.  Every wave of the workgroup must call it (two barriers per pass; one pass unless
// (NTO + NTI) * c.ntiles planes exceed the stage).
template <int NTO, int NTI, int SIDES>
DEV void wgrad_accumulate(WgradAcc<NTO, NTI, SIDES>& a, BwdCtx& c, const f4 (&dy)[PMT_RT][NTO], const f4 (&x)[PMT_RT][NTI]) {
    constexpr int P = NTO + NTI;
    constexpr int TP = (PMT_STAGE_PLANES / P) < PMT_WG_TILES ? (PMT_STAGE_PLANES / P) : PMT_WG_TILES;
    static_assert(TP >= 1, "stage too small");
    const int lane = pmt_tid() & 63;
    if (SIDES == 2) { a.any[0] |= c.tiles_ref > 0; a.any[1] |= c.ntiles > c.tiles_ref; } else { a.any[0] |= c.ntiles > 0; }
    for (int t0 = 0; t0 < c.ntiles; t0 += TP) {
        __syncthreads();  // the stage (and the slabs) of the previous round have been consumed
        if (t0 == 0) aux_reduce(c);
#pragma unroll
        for (int rt = 0; rt < PMT_RT; ++rt) {
            const int tau = c.slot0 + rt - t0;
            if (((c.mask_all >> rt) & 1u) && tau >= 0 && tau < TP) {
                f4* pl = c.stage + (size_t)(tau * P) * 64;
#pragma unroll
                for (int ot = 0; ot < NTO; ++ot)
                    if (ot < a.nmt) stage_store_transposed(pl + ot * 64, dy[rt][ot]);
#pragma unroll
                for (int it = 0; it < NTI; ++it)
                    if (it < a.nkt) stage_store_transposed(pl + (NTO + it) * 64, x[rt][it]);
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < a.TPW; ++k) {
            if (a.t_side[k] < 0) continue;
            int lo = (SIDES == 2 && a.t_side[k] == 1) ? c.tiles_ref : 0;
            int hi = (SIDES == 2 && a.t_side[k] == 0) ? c.tiles_ref : c.ntiles;
            lo = max(lo, t0) - t0;
            hi = min(hi, t0 + TP) - t0;
            const int slot = stage_slot(lane >> 4, lane & 15);
            const f4* pa = c.stage + (size_t)a.t_ot[k] * 64 + slot;
            const f4* pb = c.stage + (size_t)(NTO + a.t_it[k]) * 64 + slot;
            const bool with_bias = a.t_it[k] == 0;
            for (int tau = lo; tau < hi; ++tau) {
                const f4 va = pa[tau * P * 64], vb = pb[tau * P * 64];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) a.acc[k] = mfma16(va[ks], vb[ks], a.acc[k]);
                if (with_bias) a.bs[k] += (va[0] + va[1]) + (va[2] + va[3]);
            }
        }
    }
}

template <int NTO, int NTI, int SIDES>
DEV void wgrad_emit(WgradAcc<NTO, NTI, SIDES>& a, BwdCtx& c, const PmtLinear& L0, const PmtLinear& L1, float scale) {
    const int lane = pmt_tid() & 63, g = lane >> 4;
#pragma unroll
    for (int k = 0; k < a.TPW; ++k) {
        if (a.t_side[k] < 0) continue;
        const bool side1 = SIDES == 2 && a.t_side[k] == 1;
        if (!a.any[side1 ? 1 : 0]) continue;
        const PmtLinear& L = side1 ? L1 : L0;
        float* gw = grad_ptr(uniform(L.w_src), c.gtheta, c.gphi);
        const int col = pos_to_feat(16 * a.t_it[k] + (lane & 15));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int pf = feat_of(a.t_ot[k], j, g), o = split_row_dev(pf, a.h);
            if (o >= 0 && o < a.out_dim && pf < a.out_v && col < a.in_dim) atomicAdd(&gw[(size_t)o * a.in_dim + col], scale * a.acc[k][j]);
        }
        if (a.t_it[k] == 0 && uniform(L.b_src) != -1) {
            const float tot = group_sum(a.bs[k]);  // lanes (m, *) now hold the sum over all reads for output position m
            const int pf = pos_to_feat(16 * a.t_ot[k] + (lane & 15)), o = split_row_dev(pf, a.h);
            if (g == 0 && o >= 0 && o < a.out_dim && pf < a.out_v) atomicAdd(&grad_ptr(uniform(L.b_src), c.gtheta, c.gphi)[o], scale * tot);
        }
    }
}

template <int NTO, int NTI, int SIDES>
DEV void wgrad_exchange(BwdCtx& c, const PmtLinear& L0, const PmtLinear& L1, const f4 (&dy)[PMT_RT][NTO],
                        const f4 (&x)[PMT_RT][NTI], float scale) {
    if (c.dbg & 1) return;
    WgradAcc<NTO, NTI, SIDES> a;
    wgrad_init(a, L0);
    unsigned long long t0c = prof_now();
    wgrad_accumulate(a, c, dy, x);
    prof_add(c, 0, t0c);
    if (c.dbg & 2) return;
    t0c = prof_now();
    wgrad_emit(a, c, L0, L1, scale);
    prof_add(c, 2, t0c);
}

// ---- weight gradients on the bf16 matrix pipe ------------------------------------------------------------------------
// dW = sum over reads of dy x^T is a LEAF result: its rounding error goes no further (unlike the dgrad chain, which keeps its
// three-piece products).  Here both operands are split into TWO bf16 pieces (hi + mid: 16 significant bits, relative error
// <= 2^-17 per element, random in sign over the reads of a sum) and a 16 x 16 block of dW over 32 reads is THREE
// v_mfma_f32_16x16x32_bf16 (mid.hi + hi.mid + hi.hi, fp32 accumulation): 48 matrix-pipe cycles instead of the 8 x 32 of the
// exact fp32 MFMA -- which also holds the vector ALU while it runs, and the vector ALU is what bounds this kernel.
//
// Exchange layout.  The contraction index of one MFMA is 32 reads = the two tiles of ONE wave, and any order of the reads
// inside it will do as long as both operands use the same one.  Taking  k = 8 kg + 2 q + tile  for read 4 kg + q of the
// wave's tile `tile`, the two values a lane holds for one feature position (one per tile) are NEIGHBOURS in k: one
// v_cvt_pk_bf16_f32 packs them and one ds_write_b32 stores them -- no cross-lane shuffle at all.  A plane (16 positions x
// the wave's 32 reads, one piece) is 64 slots of 16 bytes; slot(kg, m) = 16 kg + (m ^ kg) holds position m, reads
// 4 kg .. 4 kg + 3 of both tiles: the consumer lane (m, kg) loads exactly its MFMA operand with one ds_read_b128, and the
// XOR keeps the producers' ds_write_b32 (banks = dword address mod 32) and the consumers' ds_read_b128 conflict-free.
// Stage: [wave][plane][piece hi | mid][1 KiB].
DEV unsigned pack_bf16x2(float lo, float hi) {  // v_cvt_pk_bf16_f32: round to nearest even
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    const bf2 p = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned, p);
}
DEV int stage_wbase(int lane) {  // byte offset of (kg = r >> 2, position 4 g, dword q = r & 3); position 4 g + j: XOR 16 j
    const int g = lane >> 4, r = lane & 15, kg = r >> 2, q = r & 3;
    return 256 * kg + 64 * g + 16 * kg + 4 * q;
}
DEV int stage_rbase(int lane) {  // lane (m = lane & 15, kg = lane >> 4) -> slot 16 kg + (m ^ kg)
    const int m = lane & 15, kg = lane >> 4;
    return 16 * (16 * kg + (m ^ kg));
}
// one operand plane of this wave: v0 / v1 = the plane's registers of tile 0 / tile 1 (zero for padding reads and absent tiles);
// pj[j] = the wave's stage + (wbase ^ 16 j), off = the plane's byte offset (compile time: it lands in the DS offset fields).
// Six VALU operations and one ds_write2st64_b32 per pair of values.  The two instructions are spelled out because the
// optimizer otherwise (a) re-converts v0 alone to get hi << 16 and (b) moves v0, v1 into an aligned register pair to use one
// v_pk_add_f32 for the two subtractions: nine operations instead of six, a tenth of the backward kernel's VALU work.
DEV unsigned cvt_pk_bf16(float lo, float hi) {  // round to nearest even
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
DEV float sub_f32(float a, float b) {
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
DEV char* lds_ptr(unsigned byte_address) {  // an LDS address back as a pointer (address space 3 -> generic)
    return (char*)((__attribute__((address_space(3))) char*)(size_t)byte_address);
}
#ifndef PMT_STAGE_BLOCK
#define PMT_STAGE_BLOCK 1
#endif
// (Round 5: forming the residuals of one TILE's two neighbouring values with one v_pk_add_f32 -- ten operations per four values instead of
//  twelve -- was built and measured EQUAL, 2.16 ms: the staging is not bound by its vector instructions.)
// Two pairs as ONE instruction block: the same six operations per pair, but interleaved two deep (every result is consumed two
// instructions later) and without the wait state the compiler puts behind every separate
// inline-asm instruction whose result the next one reads (`s_nop 0`, twice per pair: a fifth of the staging's issue slots).
DEV void split_pairs_block(float a0, float a1, float b0, float b1, unsigned& h0, unsigned& h1, unsigned& m0, unsigned& m1) {
    unsigned t0, t1;
    asm("v_cvt_pk_bf16_f32 %[h0], %[a0], %[b0]\n\t"
        "v_cvt_pk_bf16_f32 %[h1], %[a1], %[b1]\n\t"
        "v_lshlrev_b32 %[m0], 16, %[h0]\n\t"
        "v_lshlrev_b32 %[m1], 16, %[h1]\n\t"
        "v_and_b32 %[t0], 0xffff0000, %[h0]\n\t"
        "v_and_b32 %[t1], 0xffff0000, %[h1]\n\t"
        "v_sub_f32 %[m0], %[a0], %[m0]\n\t"
        "v_sub_f32 %[m1], %[a1], %[m1]\n\t"
        "v_sub_f32 %[t0], %[b0], %[t0]\n\t"
        "v_sub_f32 %[t1], %[b1], %[t1]\n\t"
        "v_cvt_pk_bf16_f32 %[m0], %[m0], %[t0]\n\t"
        "v_cvt_pk_bf16_f32 %[m1], %[m1], %[t1]"
        : [h0] "=&v"(h0), [h1] "=&v"(h1), [m0] "=&v"(m0), [m1] "=&v"(m1), [t0] "=&v"(t0), [t1] "=&v"(t1)
        : [a0] "v"(a0), [a1] "v"(a1), [b0] "v"(b0), [b1] "v"(b1));
}
template <int PIECES = 3>
DEV void stage_pair_bf16(char* const (&pj)[4], int off, f4 v0, f4 v1) {
    if constexpr (PIECES != 1 && PMT_STAGE_BLOCK) {
#pragma unroll
        for (int j = 0; j < 4; j += 2) {
            unsigned h0, h1, m0, m1;
            split_pairs_block(v0[j], v0[j + 1], v1[j], v1[j + 1], h0, h1, m0, m1);
            *reinterpret_cast<unsigned*>(pj[j] + off) = h0;
            *reinterpret_cast<unsigned*>(pj[j] + off + 1024) = m0;
            *reinterpret_cast<unsigned*>(pj[j + 1] + off) = h1;
            *reinterpret_cast<unsigned*>(pj[j + 1] + off + 1024) = m1;
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        char* p = pj[j] + off;
        if constexpr (PIECES == 1) {
            *reinterpret_cast<unsigned*>(p) = pack_bf16x2(v0[j], v1[j]);
        } else {
            const unsigned hi = cvt_pk_bf16(v0[j], v1[j]);
            const float h0 = __builtin_bit_cast(float, hi << 16), h1 = __builtin_bit_cast(float, hi & 0xFFFF0000u);
            *reinterpret_cast<unsigned*>(p) = hi;
            *reinterpret_cast<unsigned*>(p + 1024) = cvt_pk_bf16(sub_f32(v0[j], h0), sub_f32(v1[j], h1));
        }
    }
}

// One tile per wave (PMT_RT == 1): the wave writes ITS half of every dword of the pair's plane -- the same addresses, 16-bit stores (pj
// carries the + 2 bytes of the odd wave).  Two positions share a conversion: the packed pair's low half goes out with ds_write_b16, its
// high half with ds_write_b16_d16_hi.
DEV void lds_store16(char* p, unsigned v) { *reinterpret_cast<unsigned short*>(p) = (unsigned short)v; }
template <int PIECES = 3>
DEV void stage_half_bf16(char* const (&pj)[4], int off, f4 v) {
#pragma unroll
    for (int j = 0; j < 4; j += 2) {
        const unsigned hi = cvt_pk_bf16(v[j], v[j + 1]);
        lds_store16(pj[j] + off, hi);
        lds_store16(pj[j + 1] + off, hi >> 16);
        if constexpr (PIECES != 1) {
            const float h0 = __builtin_bit_cast(float, hi << 16), h1 = __builtin_bit_cast(float, hi & 0xFFFF0000u);
            const unsigned mid = cvt_pk_bf16(sub_f32(v[j], h0), sub_f32(v[j + 1], h1));
            lds_store16(pj[j] + off + 1024, mid);
            lds_store16(pj[j + 1] + off + 1024, mid >> 16);
        }
    }
}

#define PMT_BF_PLANE_BYTES 2048  // hi + mid piece of one (wave, plane)
#define PMT_ABL(c, bit) (PMT_BWD_ABLATE && ((c).dbg & (bit)))
template <int NTO, int NTI, int SIDES, int BF = 3>
DEV void wgrad_exchange_bf(BwdCtx& c, const PmtLinear& L0, const PmtLinear& L1, const f4 (&dy)[PMT_RT][NTO],
                           const f4 (&x)[PMT_RT][NTI], float scale) {
    static_assert(PMT_RT == 2 || PMT_RT == 1, "a wave's two tiles, or the tiles of a pair of waves, are the 32 reads of one MFMA");
    if (c.dbg & 1) return;
    constexpr int PIECES = PMT_BF_PIECES(BF);
    constexpr bool PRIV = (BF & PMT_BF_PRIV) != 0;  // c.priv != nullptr, as a compile-time fact
    constexpr int P = NTO + NTI, NB = NTO * NTI;
    constexpr int NSLOT = PMT_WG_TILES / 2;  // operand slots of the stage: one per 32 reads (a wave; with one tile per wave a pair of waves)
    constexpr bool COLS = SIDES == 1 && PMT_WAVES % NTI == 0;  // one linear whose columns of blocks divide the waves (1, 2, 4, 8 input tiles)
    constexpr int PW_CAP = (PMT_STAGE_PLANES * 1024) / (P * PMT_BF_PLANE_BYTES);          // waves whose operands fit the stage
    constexpr int PW = PW_CAP < NSLOT ? PW_CAP : NSLOT;
    static_assert(PW >= 1, "stage too small");
    // Blocks of this wave.  One linear (COLS): the wave owns column `it` = wave % NTI and the rows wave / NTI + k * (waves / NTI): its
    // blocks share the x operand, loaded once per pair of tiles.  A ref / alt pair: task q = wave + waves * k of the
    // 2 NB blocks, side-major (a wave's k-th blocks of the two sides have equal coordinates: the work is balanced
    // whatever the split of the group between the sides).  One linear with 3, 5, 6 or 7 input tiles (the wide build's shapes): the
    // same task list over its NB blocks.
    constexpr int ROWS = COLS ? PMT_WAVES / NTI : 1;
    constexpr int TPW = COLS ? (NTO + ROWS - 1) / ROWS : (SIDES * NB + PMT_WAVES - 1) / PMT_WAVES;
    constexpr bool ALL_TASKS = COLS ? NTO % ROWS == 0 : (SIDES * NB) % PMT_WAVES == 0;  // every wave has TPW blocks: no "is there a block" branches
    const int lane = pmt_tid() & 63, g = lane >> 4, wave = uniform((int)(pmt_tid() >> 6));
    const int my_slot = PMT_RT == 2 ? wave : wave >> 1, wr_s = PMT_RT == 2 ? c.wr : c.wr >> 1;  // (GroupGeom.wr is even with one tile per wave)
    int t_ot[TPW], t_it[TPW], t_side[TPW];
    f4 acc[TPW], accb[TPW];
#pragma unroll
    for (int k = 0; k < TPW; ++k) {
        acc[k] = accb[k] = f4{0.f, 0.f, 0.f, 0.f};
        if (COLS) {
            t_it[k] = wave % NTI;
            t_ot[k] = wave / NTI + ROWS * k;
            t_side[k] = t_ot[k] < NTO ? 0 : -1;
        } else {
            const int q = wave + PMT_WAVES * k;
            t_side[k] = q >= SIDES * NB ? -1 : (q >= NB ? 1 : 0);
            const int rem = q - (q >= NB ? NB : 0);
            t_ot[k] = rem / NTI;
            t_it[k] = rem - t_ot[k] * NTI;
        }
    }
    typedef unsigned u4v __attribute__((ext_vector_type(4)));
    const bf8 ones = __builtin_bit_cast(bf8, u4v{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u});  // 1.0 everywhere
    // destinations of this wave's blocks (PmtLinear.emit_tab), fetched now: the loads are long back when the sums are ready
    typedef int i4 __attribute__((ext_vector_type(4)));
    i4 e[TPW], eb[TPW];
    bool with_bias[TPW];
#pragma unroll
    for (int k = 0; k < TPW; ++k) {
        e[k] = eb[k] = i4{-1, -1, -1, -1};
        with_bias[k] = false;
        if (!ALL_TASKS && t_side[k] < 0) continue;
        const PmtLinear& L = (SIDES == 2 && t_side[k] == 1) ? L1 : L0;
        // (private partial sums: the same 16 bytes are this lane's running sums of the block instead of its destinations)
        const int* tab = reinterpret_cast<const int*>(PRIV ? c.priv + uniform(L.emit_tab) : c.packed + uniform(L.emit_tab));
        e[k] = *reinterpret_cast<const i4*>(tab + ((t_ot[k] * NTI + t_it[k]) * 64 + lane) * 4);
        with_bias[k] = t_it[k] == t_ot[k] % NTI;  // the one block of row ot that also sums the bias gradient (dy x ones)
        if (with_bias[k]) eb[k] = *reinterpret_cast<const i4*>(tab + NB * 256 + t_ot[k] * 16 + 4 * g);
    }
    char* stage = reinterpret_cast<char*>(c.stage);
    unsigned long long t0c = prof_now();
    for (int w0 = 0; w0 < NSLOT; w0 += PW) {
        unsigned long long t1 = prof_now();
        trace_ev(c, 100);
        if (PMT_ABL(c, 8192)) {} else if (c.dbg & 128) __syncthreads(); else lds_barrier();  // the stage (and the slabs) of the previous round have been consumed
        trace_ev(c, 101);
        prof_add(c, 17, t1);
        if (w0 == 0) aux_reduce(c);
        t1 = prof_now();
        if (!PMT_ABL(c, 1024) && (PW == NSLOT || (my_slot >= w0 && my_slot < w0 + PW))) {
            char* mine = stage + (my_slot - w0) * (P * PMT_BF_PLANE_BYTES);
            // (through an empty asm: computed ONCE per exchange; the compiler otherwise rebuilds each address at every store)
            unsigned a0 = (unsigned)(size_t)(mine + c.wbase), a1 = (unsigned)(size_t)(mine + (c.wbase ^ 16)),
                     a2 = (unsigned)(size_t)(mine + (c.wbase ^ 32)), a3 = (unsigned)(size_t)(mine + (c.wbase ^ 48));
            asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
            char* const pj[4] = {lds_ptr(a0), lds_ptr(a1), lds_ptr(a2), lds_ptr(a3)};
            if constexpr (PMT_RT == 2) {
#pragma unroll
                for (int ot = 0; ot < NTO; ++ot) stage_pair_bf16<PIECES>(pj, ot * PMT_BF_PLANE_BYTES, dy[0][ot], dy[PMT_RT - 1][ot]);
#pragma unroll
                for (int it = 0; it < NTI; ++it) stage_pair_bf16<PIECES>(pj, (NTO + it) * PMT_BF_PLANE_BYTES, x[0][it], x[PMT_RT - 1][it]);
            } else {
#pragma unroll
                for (int ot = 0; ot < NTO; ++ot) stage_half_bf16<PIECES>(pj, ot * PMT_BF_PLANE_BYTES, dy[0][ot]);
#pragma unroll
                for (int it = 0; it < NTI; ++it) stage_half_bf16<PIECES>(pj, (NTO + it) * PMT_BF_PLANE_BYTES, x[0][it]);
            }
        }
        prof_add(c, 18, t1);
        t1 = prof_now();
        trace_ev(c, 102);
        if (PMT_ABL(c, 8192)) {} else if (c.dbg & 128) __syncthreads(); else lds_barrier();
        trace_ev(c, 103);
        prof_add(c, 19, t1);
        const int whi_all = min(w0 + PW, NSLOT);
        const char* rd = stage + c.rbase;
        if (PMT_ABL(c, 512)) continue;
        if (COLS) {
            if (ALL_TASKS || t_side[0] >= 0) {
#pragma unroll
                for (int w = 0; w < PW; ++w) {  // branch-free body: the scheduler overlaps the reads of one pair with the MFMAs of another
                    if (w0 + w >= NSLOT) break;
                    const char* pw = rd + w * (P * PMT_BF_PLANE_BYTES);
                    const bf8 bh = *reinterpret_cast<const bf8*>(pw + (NTO + t_it[0]) * PMT_BF_PLANE_BYTES);
                    const bf8 bm = *reinterpret_cast<const bf8*>(pw + (NTO + t_it[0]) * PMT_BF_PLANE_BYTES + 1024);
#pragma unroll
                    for (int k = 0; k < TPW; ++k) {
                        const int ot = t_side[k] >= 0 ? t_ot[k] : 0;  // (a missing block reads row 0 and is never emitted)
                        const bf8 ah = *reinterpret_cast<const bf8*>(pw + ot * PMT_BF_PLANE_BYTES);
                        const bf8 am = *reinterpret_cast<const bf8*>(pw + ot * PMT_BF_PLANE_BYTES + 1024);
                        if constexpr (PIECES != 1) {
                            acc[k] = mfma_bf16(am, bh, acc[k]);
                            acc[k] = mfma_bf16(ah, bm, acc[k]);
                            accb[k] = mfma_bf16(am, ones, accb[k]);
                        }
                        acc[k] = mfma_bf16(ah, bh, acc[k]);
                        accb[k] = mfma_bf16(ah, ones, accb[k]);
                    }
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < TPW; ++k) {
                if (!ALL_TASKS && t_side[k] < 0) continue;
                const int lo = SIDES == 1 ? w0 : max(t_side[k] == 1 ? wr_s : 0, w0), hi = SIDES == 1 ? whi_all : min(t_side[k] == 0 ? wr_s : NSLOT, whi_all);
                const char* pa = rd + t_ot[k] * PMT_BF_PLANE_BYTES;
                const char* pb = rd + (NTO + t_it[k]) * PMT_BF_PLANE_BYTES;
#pragma unroll 2
                for (int w = lo; w < hi; ++w) {
                    const int o = (w - w0) * (P * PMT_BF_PLANE_BYTES);
                    const bf8 ah = *reinterpret_cast<const bf8*>(pa + o), am = *reinterpret_cast<const bf8*>(pa + o + 1024);
                    const bf8 bh = *reinterpret_cast<const bf8*>(pb + o), bm = *reinterpret_cast<const bf8*>(pb + o + 1024);
                    if constexpr (PIECES != 1) {
                        acc[k] = mfma_bf16(am, bh, acc[k]);
                        acc[k] = mfma_bf16(ah, bm, acc[k]);
                        accb[k] = mfma_bf16(am, ones, accb[k]);
                    }
                    acc[k] = mfma_bf16(ah, bh, acc[k]);
                    accb[k] = mfma_bf16(ah, ones, accb[k]);
                }
            }
        }
        // (bias gradients = the row sums of dy: EVERY block multiplies its A operand once more against a plane of ones -- the matrix pipe has
        //  the room -- and the one block per row that owns the bias emits them: no second pass over the stage by half the waves)
    }
    prof_add(c, 0, t0c);
    trace_ev(c, 104);
    if (c.dbg & 2) return;
    t0c = prof_now();
    // emit.  ONE branch on the destination kind around the blocks (not one per block: a branch ends a scheduling region)
    if constexpr (PRIV) {  // running sums + this group's block, back to the private row (one 16-byte store per lane)
#pragma unroll
        for (int k = 0; k < TPW; ++k) {
            if (!ALL_TASKS && t_side[k] < 0) continue;
            const bool side1 = SIDES == 2 && t_side[k] == 1;
            // (a side without tiles: its blocks' sums are zero, the row gets its own value back)
            const PmtLinear& L = side1 ? L1 : L0;
            float* row = c.priv + uniform(L.emit_tab);
            *reinterpret_cast<f4*>(row + ((t_ot[k] * NTI + t_it[k]) * 64 + lane) * 4) = __builtin_bit_cast(f4, e[k]) + scale * acc[k];
            if (with_bias[k] && (lane & 15) == 0)
                *reinterpret_cast<f4*>(row + NB * 256 + t_ot[k] * 16 + 4 * g) = __builtin_bit_cast(f4, eb[k]) + scale * accb[k];
        }
    } else {  // four atomics per block at the tabulated offsets; no index arithmetic here
#pragma unroll
        for (int k = 0; k < TPW; ++k) {
            if (t_side[k] < 0) continue;
            const bool side1 = SIDES == 2 && t_side[k] == 1;
            if (SIDES == 2 && (side1 ? c.ntiles <= c.tiles_ref : c.tiles_ref <= 0)) continue;
            const PmtLinear& L = side1 ? L1 : L0;
            float* gw = uniform(L.w_src) >= 0 ? c.gtheta : c.gphi;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (e[k][j] >= 0) atomicAdd(gw + e[k][j], scale * acc[k][j]);
            if (with_bias[k] && (lane & 15) == 0) {  // every column of accb holds the row sums: the lanes of column 0 add them
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (eb[k][j] >= 0) atomicAdd(c.gtheta + eb[k][j], scale * accb[k][j]);
            }
        }
    }
    prof_add(c, 2, t0c);
    trace_ev(c, 105);
}

template <int NTO, int NTI, int BF = 0>
DEV void linear_wgrad(BwdCtx& c, const PmtLinear& L, const f4 (&dy)[PMT_RT][NTO], const f4 (&x)[PMT_RT][NTI],
                      float scale = 1.0f) {
    if constexpr (BF != 0) wgrad_exchange_bf<NTO, NTI, 1, BF>(c, L, L, dy, x, scale);
    else wgrad_exchange<NTO, NTI, 1>(c, L, L, dy, x, scale);
}

// Backward of one LINEAR op between register arrays of different tile counts (see run_linear_op): dy is the gradient
// w.r.t. the op's output (modified: multiplied by the activation derivative), x its input; dx (if wanted) = W^T dy.
template <int NTI, int NTO, bool EXACT, int WI = 0, int WO = 0, int BF = 0, bool DROP = false>
DEV void linear_op_backward(BwdCtx& c, const PmtOp& o, f4 (&dy)[PMT_RT][NTO], const f4 (&x)[PMT_RT][NTI],
                            f4 (&dx)[PMT_RT][NTI], bool want_dx) {
    const PmtLinear& L = c.M->lin[uniform(o.lin[0])];
    const int in_dim = WI ? WI : uniform(L.in_dim), out_dim = WO ? WO : uniform(L.out_dim);
    const bool dropping = DROP && c.drop != nullptr && c.drop->on != 0;
    if (uniform(o.selu_after) != 0) {  // recompute s = selu(Wx + b); dy <- dy * selu'(s)
        f4 y[PMT_RT][NTO];
        init_bias<NTO>(y, uniform(L.b_pvec) >= 0 ? c.packed + uniform(L.b_pvec) : nullptr, out_dim, c.g);
        if constexpr (BF) linear_acc_bf16<NTI, NTO, false, PMT_RC(BF)>(y, x, c.packed + uniform(L.wb_frag));
        else linear_acc<NTI, NTO, false, EXACT, WI>(y, x, c.packed + uniform(L.w_frag), in_dim, out_dim);
        if (dropping) drop_apply<NTO>(*c.drop, uniform(o.lin[0]), y, c.g);
#pragma unroll
        for (int rt = 0; rt < PMT_RT; ++rt)
#pragma unroll
            for (int t = 0; t < NTO; ++t) dy[rt][t] = selu_bwd4(dy[rt][t], selu4(y[rt][t]));
    }
    if (dropping) drop_apply<NTO>(*c.drop, uniform(o.lin[0]), dy, c.g);
    linear_wgrad<NTO, NTI, BF>(c, L, dy, x);
    if (want_dx) {
        init_bias<NTI>(dx, nullptr, in_dim, c.g);
        if constexpr (BF) linear_acc_bf16<NTO, NTI, false, PMT_DG(BF)>(dx, dy, c.packed + uniform(L.wtb_frag));
        else linear_acc<NTO, NTI, false, EXACT, WO>(dx, dy, c.packed + uniform(L.wt_frag), out_dim, in_dim);
    }
}

// Backward of a skip block of THREE or FOUR layers (generic instances; the reference takes any depth, mlp.py:15-22):
//   p_0 = x,  p_k = D_k(b_k + W_k selu(p_{k-1})),  y = x + alpha p_n        (D_k: the step's dropout mask behind Linear k, if any)
// walked from the last layer to the first.  Layer k needs s_{k-1} = selu(p_{k-1}) twice -- as the input of its weight gradient and for
// the SELU derivative that carries d(s_{k-1}) on to layer k - 1 -- and gets it by re-running layers 1 .. k - 1 from the stashed x:
// n (n - 1) / 2 recomputed products per block (the two-layer path's one), no intermediate of the block in the stash.  Four register
// arrays at most: dy, the running d(p_k), the chain's two.
template <int NT, bool DROP, typename LoadInput>
DEV void skip_block_backward_deep(BwdCtx& c, const PmtOp& o, int op, f4 (&dy)[PMT_RT][NT], LoadInput load_input) {
    const PmtModel* M = c.M;
    const int nl = uniform(o.n_layers);
    const bool dropping = DROP && c.drop != nullptr && c.drop->on != 0;
    const int width = uniform(M->lin[uniform(o.lin[0])].in_dim);
    const float alpha = uniform(c.theta[uniform(o.alpha_src)]);
    f4 d[PMT_RT][NT];  // gradient w.r.t. Linear k's own output: for k = n per unit of alpha (the weight gradient takes alpha as its scale)
#pragma unroll
    for (int rt = 0; rt < PMT_RT; ++rt)
#pragma unroll
        for (int t = 0; t < NT; ++t) d[rt][t] = dy[rt][t];
    if (dropping) drop_apply<NT>(*c.drop, uniform(o.lin[nl - 1]), d, c.g);
    for (int k = nl; k >= 1; --k) {
        const PmtLinear& Lk = M->lin[uniform(o.lin[k - 1])];
        f4 p[PMT_RT][NT];
        load_input(op, p);
        for (int j = 1; j < k; ++j) {
            const PmtLinear& Lj = M->lin[uniform(o.lin[j - 1])];
            f4 q[PMT_RT][NT];
            init_bias<NT>(q, c.packed + uniform(Lj.b_pvec), width, c.g);
            linear_acc<NT, NT, true, false>(q, p, c.packed + uniform(Lj.w_frag), width, width);
            if (dropping) drop_apply<NT>(*c.drop, uniform(o.lin[j - 1]), q, c.g);
#pragma unroll
            for (int rt = 0; rt < PMT_RT; ++rt)
#pragma unroll
                for (int t = 0; t < NT; ++t) p[rt][t] = q[rt][t];
        }
#pragma unroll
        for (int rt = 0; rt < PMT_RT; ++rt)
#pragma unroll
            for (int t = 0; t < NT; ++t) p[rt][t] = selu4(p[rt][t]);  // s_{k-1}
        linear_wgrad<NT, NT, 0>(c, Lk, d, p, k == nl ? alpha : 1.0f);
        f4 ds[PMT_RT][NT];
        init_bias<NT>(ds, nullptr, width, c.g);
        linear_acc<NT, NT, false, false>(ds, d, c.packed + uniform(Lk.wt_frag), width, width);
        if (k == nl) {  // d(alpha) = sum dy . p_n = sum (W_n^T d) . s_{n-1} + sum d . b_n  (the forward product is never formed)
            float da = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const f4 bn = load_pvec(c.packed + uniform(Lk.b_pvec), t, c.g);
#pragma unroll
                for (int rt = 0; rt < PMT_RT; ++rt) {
                    const f4 v = ds[rt][t] * p[rt][t] + d[rt][t] * bn;
                    da += (v[0] + v[1]) + (v[2] + v[3]);
                }
            }
            aux_push_scalar(c, uniform(o.alpha_src), da);
        }
        const float sc = k == nl ? alpha : 1.0f;
#pragma unroll
        for (int rt = 0; rt < PMT_RT; ++rt)
#pragma unroll
            for (int t = 0; t < NT; ++t) ds[rt][t] = sc * selu_bwd4(ds[rt][t], p[rt][t]);  // d(p_{k-1})
        if (k > 1) {
            if (dropping) drop_apply<NT>(*c.drop, uniform(o.lin[k - 2]), ds, c.g);
#pragma unroll
            for (int rt = 0; rt < PMT_RT; ++rt)
#pragma unroll
                for (int t = 0; t < NT; ++t) d[rt][t] = ds[rt][t];
        } else {
#pragma unroll
            for (int rt = 0; rt < PMT_RT; ++rt)
#pragma unroll
                for (int t = 0; t < NT; ++t) dy[rt][t] = dy[rt][t] + ds[rt][t];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// backward of one MLP program.  dy (in/out): gradient w.r.t. the MLP output on entry, w.r.t. its input on exit
// (not computed for op 0 when need_input_grad is false).  in_slot(op) gives the stash slot of op's input.
template <int NT, bool EXACT, int W = 0, int BF = 0, bool DROP = !EXACT, typename LoadInput>
DEV void mlp_backward(BwdCtx& c, const PmtMlp& mlp, f4 (&dy)[PMT_RT][NT], bool need_input_grad, LoadInput load_input,
                      int op_begin, int op_end) {
    const PmtModel* M = c.M;
    const bool dropping = DROP && c.drop != nullptr && c.drop->on != 0;
    for (int op = op_end - 1; op >= op_begin; --op) {
        const PmtOp& o = mlp.ops[op];
        f4 x[PMT_RT][NT];
        trace_ev(c, 200 + op);
        load_input(op, x);
        if (uniform(o.kind) == PMT_OP_LINEAR) {
            const PmtLinear& L = M->lin[uniform(o.lin[0])];
            const int in_dim = W ? W : uniform(L.in_dim), out_dim = W ? W : uniform(L.out_dim);
            if (uniform(o.selu_after) != 0) {  // recompute s = selu(Wx + b); dy <- dy * selu'(s)
                f4 y[PMT_RT][NT];
                init_bias<NT>(y, uniform(L.b_pvec) >= 0 ? c.packed + uniform(L.b_pvec) : nullptr, out_dim, c.g);
                if constexpr (BF) linear_acc_bf16<NT, NT, false, PMT_RC(BF)>(y, x, c.packed + uniform(L.wb_frag));
                else linear_acc<NT, NT, false, EXACT, W>(y, x, c.packed + uniform(L.w_frag), in_dim, out_dim);
                if (dropping) drop_apply<NT>(*c.drop, uniform(o.lin[0]), y, c.g);  // s = selu(mask * (Wx + b))
#pragma unroll
                for (int rt = 0; rt < PMT_RT; ++rt)
#pragma unroll
                    for (int t = 0; t < NT; ++t) dy[rt][t] = selu_bwd4(dy[rt][t], selu4(y[rt][t]));
            }
            if (dropping) drop_apply<NT>(*c.drop, uniform(o.lin[0]), dy, c.g);  // d(Wx + b) = mask * d(masked)
            linear_wgrad<NT, NT, BF>(c, L, dy, x);
            if (op > op_begin || need_input_grad) {
                f4 dx[PMT_RT][NT];
                init_bias<NT>(dx, nullptr, in_dim, c.g);
                if constexpr (BF) linear_acc_bf16<NT, NT, false, PMT_DG(BF)>(dx, dy, c.packed + uniform(L.wtb_frag));
                else linear_acc<NT, NT, false, EXACT, W>(dx, dy, c.packed + uniform(L.wt_frag), out_dim, in_dim);
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
            if constexpr (!EXACT) {
                if (nl > 2) {  // three and four layers: generic instances only (pmt_shape_id)
                    skip_block_backward_deep<NT, DROP>(c, o, op, dy, load_input);
                    continue;
                }
            }
            const PmtLinear& L1 = M->lin[uniform(o.lin[0])];
            const PmtLinear& L2 = M->lin[uniform(o.lin[nl - 1])];
            const int width = W ? W : uniform(L1.in_dim);
            const float alpha = uniform(c.theta[uniform(o.alpha_src)]);
            f4 s1[PMT_RT][NT];
            if (nl == 2) {  // s1 = selu(L1 selu(x) + b1)
                init_bias<NT>(s1, c.packed + uniform(L1.b_pvec), width, c.g);
                if (c.dbg & 4096) {}  // knock-out (wrong results): the most that stashing s1 could save
                else if constexpr (BF) linear_acc_bf16<NT, NT, true, PMT_RC(BF)>(s1, x, c.packed + uniform(L1.wb_frag));
                else linear_acc<NT, NT, true, EXACT, W>(s1, x, c.packed + uniform(L1.w_frag), width, width);
                if (dropping) drop_apply<NT>(*c.drop, uniform(o.lin[0]), s1, c.g);
            }
#pragma unroll
            for (int rt = 0; rt < PMT_RT; ++rt)
#pragma unroll
                for (int t = 0; t < NT; ++t) s1[rt][t] = selu4(nl == 2 ? s1[rt][t] : x[rt][t]);
            // last layer: d(f) = alpha * dy
            trace_ev(c, 220);
            f4 d1[PMT_RT][NT];
            // dyl: the gradient w.r.t. the last Linear's own output (per unit of alpha) -- dy itself, or with dropout mask2 * dy
            // (a copy: dy is still needed whole for the residual path)
            auto last_layer = [&](const f4 (&dyl)[PMT_RT][NT]) {
                linear_wgrad<NT, NT, BF>(c, L2, dyl, s1, alpha);
                init_bias<NT>(d1, nullptr, width, c.g);
                if constexpr (BF) linear_acc_bf16<NT, NT, false, PMT_DG(BF)>(d1, dyl, c.packed + uniform(L2.wtb_frag));
                else linear_acc<NT, NT, false, EXACT, W>(d1, dyl, c.packed + uniform(L2.wt_frag), width, width);
                // d(alpha) = sum dy . f with f = W2 s1 + b2, i.e. sum (W2^T dy) . s1 + sum dy . b2: the first factor is d1 as it
                // stands here, so the forward product f is never formed (it was a quarter of this op's matrix work)
                float da = 0.f;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const f4 b2 = load_pvec(c.packed + uniform(L2.b_pvec), t, c.g);
#pragma unroll
                    for (int rt = 0; rt < PMT_RT; ++rt) {
                        const f4 p = d1[rt][t] * s1[rt][t] + dyl[rt][t] * b2;
                        da += (p[0] + p[1]) + (p[2] + p[3]);
                    }
                }
                aux_push_scalar<!EXACT>(c, uniform(o.alpha_src), da);  // (right behind an exchange, which empties the slab)
            };
            if (dropping) {
                f4 dm[PMT_RT][NT];
#pragma unroll
                for (int rt = 0; rt < PMT_RT; ++rt)
#pragma unroll
                    for (int t = 0; t < NT; ++t) dm[rt][t] = dy[rt][t];
                drop_apply<NT>(*c.drop, uniform(o.lin[nl - 1]), dm, c.g);
                last_layer(dm);
            } else {
                last_layer(dy);
            }
#pragma unroll
            for (int rt = 0; rt < PMT_RT; ++rt)
#pragma unroll
                for (int t = 0; t < NT; ++t) d1[rt][t] = alpha * selu_bwd4(d1[rt][t], s1[rt][t]);  // d(h1) (n=2) or d(x) part (n=1)
            if (dropping && nl == 2) drop_apply<NT>(*c.drop, uniform(o.lin[0]), d1, c.g);  // d(L1's own output)
            __builtin_amdgcn_sched_barrier(0);
            trace_ev(c, 221);
            if (nl == 2) {
                f4 s0[PMT_RT][NT];  // second read of x (s1's registers are free now)
                load_input(op, s0);
#pragma unroll
                for (int rt = 0; rt < PMT_RT; ++rt)
#pragma unroll
                    for (int t = 0; t < NT; ++t) s0[rt][t] = selu4(s0[rt][t]);
                trace_ev(c, 222);
                linear_wgrad<NT, NT, BF>(c, L1, d1, s0);
                f4 d0[PMT_RT][NT];
                init_bias<NT>(d0, nullptr, width, c.g);
                if constexpr (BF) linear_acc_bf16<NT, NT, false, PMT_DG(BF)>(d0, d1, c.packed + uniform(L1.wtb_frag));
                else linear_acc<NT, NT, false, EXACT, W>(d0, d1, c.packed + uniform(L1.wt_frag), width, width);
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

