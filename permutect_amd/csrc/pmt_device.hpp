// Device-side building blocks shared by the forward and backward read-set kernels (gfx950 / CDNA4 only).
//
// Data layout in registers ("tile-position layout")
// -------------------------------------------------
// A wave owns up to PMT_RT tiles of 16 reads.  An activation of width d (<= 64) for one tile is NT = 4 float4
// registers per lane.  With lane = 16*g + r (r = read within the tile, g = lane group 0..3), register [t][j] holds
//        feature f = 16*t + 4*j + g          of read r.
// This is exactly the C/D layout of v_mfma_f32_16x16x4_f32 when features are the M (row) dimension and reads the
// N (column) dimension  (C: col = lane & 15, row = 4*(lane >> 4) + reg), with the rows of each 16-row tile
// permuted so that k-step j of a following MFMA contracts features {16t + 4j + g : g = 0..3}.  A layer output is
// therefore directly the B operand of the next layer: activations never leave registers between layers.
// Weights are pre-packed (pmt_pack.hip) into A-fragment order with the same permutation:
//        frag[((mt * nkt + kt) * 64 + lane) * 4 + j] = W[16*mt + 4*(m & 3) + (m >> 2)][16*kt + 4*j + g],  m = lane & 15
// so one 16-byte load per lane (1 KiB per wave, fully coalesced) feeds four MFMAs per read tile.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "permutect_amd.h"
#include "pmt_dropout.hpp"

typedef float f4 __attribute__((ext_vector_type(4)));

#define PMT_NT (PMT_MAX_WIDTH / 16)  // feature tiles per activation: 4 (the wide build of the library: 8)
#define PMT_HT (PMT_MAX_HALF_FFN / 16)  // tiles of ONE half of a gated block's hidden layer (z1 | z2): 1 (a build for d_ffn / 2 in 17 .. 32: 2)
#define PMT_SPLIT0 (16 * PMT_HT)        // positions of the first half in proj1's virtual output (out_split): the second half starts here
#define PMT_ZW (32 * PMT_HT)            // floats of a read set's per-block sums (z2 / d(gate)): [ref | alt][16 PMT_HT]
#ifndef PMT_RT
#define PMT_RT 2                 // read tiles per wave (a translation unit may choose its own wave shape)
#endif
#ifndef PMT_WAVES
#define PMT_WAVES PMT_GROUP_WAVES  // waves per workgroup; PMT_WAVES * PMT_RT must equal the planned group capacity
#endif
#define PMT_THREADS (PMT_WAVES * 64)

#define PMT_SELU_ALPHA 1.6732632423543772848170429916717f
#define PMT_SELU_SCALE 1.0507009873554804934193349852946f
#define PMT_LN_EPS 1e-5f
#define PMT_LOG2PI 1.8378770664093453f
#define PMT_MAX_LOGIT_F 20.0f

#define PMT_WG_TILES (PMT_WAVES * PMT_RT)  // tiles a workgroup processes at once
#ifndef PMT_OWN_WAVE_SHAPE  // the read-set kernels must match the planner; other kernels may choose their own shape
static_assert(PMT_GROUP_TILES == PMT_WG_TILES, "group capacity");
#endif

#define DEV __device__ __forceinline__
// threadIdx.x; a translation unit whose kernel LOOPS over groups sets PMT_OPAQUE_TID so that nothing derived from the thread id
// is loop invariant (hoisted values would be live across the whole loop body, in registers the body does not have)
#ifndef PMT_OPAQUE_TID
#define PMT_OPAQUE_TID 0
#endif
DEV int pmt_tid() {
    int t = threadIdx.x;
#if PMT_OPAQUE_TID
    asm volatile("" : "+v"(t));
#endif
    return t;
}

// Register-array shapes of the read-set kernels, in 16-feature tiles: F read features, R read-MLP widths (after its first
// linear), D d_model and reducer widths, E feature_dim.  EXACT: every layer fills its arrays completely, the read MLP
// starts and the reducer ends with a LINEAR op (the two ops that change the tile count); the host picks the instance.
// XF .. XE (only with EXACT): the EXACT widths of the model, known at compile time (0 = read from the descriptor).  With them
// the padding masks (feature < width), the k-steps that hold nothing but padding and most width bookkeeping fold away:
// the masks alone cost ~30 SGPR pairs that the generic code keeps (and spills) across the block loop.
// XBF: 0 = exact-fp32 MFMAs; 3 = the layers' matrix products as SIX bf16 MFMAs on three-piece splits of both operands
// (linear_acc_bf16: fp32-equivalent); 1 = ONE bf16 MFMA per product on single bf16 roundings of both operands -- the plain
// bf16 mode BASELINE.json's training configuration names: no parity claim, measured and labelled as such (bench.py --dtype bf16).
//      16 (PMT_F16X2) = three f16 MFMAs on two-piece splits with a scaled low piece (linear_acc_f16: fp32-equivalent).
// XDROP: the instance carries the dropout masks of a training step (pmt_dropout.hpp); the generic instances always do.
template <int F, int R, int D, int E, bool EXACT_, int XF = 0, int XR = 0, int XD = 0, int XH = 0, int XE = 0, int XBF = 0, bool XDROP = false>
struct Shape {
    static constexpr int NTF = F, NTR = R, NTD = D, NTE = E;
    static constexpr bool EXACT = EXACT_;
    static constexpr int DIM_F = XF, DIM_R = XR, DIM_D = XD, DIM_H = XH, DIM_E = XE;  // read features, read width, d_model, d_ffn / 2, feature_dim
    static constexpr int BF16 = XBF;  // pieces per operand (0: fp32 MFMA)
    static constexpr bool DROP = XDROP || !EXACT_;
    static_assert(!XBF || EXACT_, "the bf16 path has no tile guards");
};
using ShapeAny = Shape<PMT_NT, PMT_NT, PMT_NT, PMT_NT, false>;   // any supported model
// The exact-width instances are compiled for ONE model shape per build of the library: the production hyperparameters (SURVEY:
// P0) by default; `make SHAPE="ntf,ntr,ntd,nte,F,R,D,H,E" LIB=...` builds the same library around another shape (tile counts of
// the read features / read widths / d_model / feature_dim, then the widths themselves: read features, read-MLP width, d_model,
// d_ffn / 2, feature_dim).  permutect_amd/engine/instances.py builds and loads such a library for a model whose shape is not
// the default's (T0: 4,1,2,2,61,10,30,10,20), so every model that meets the EXACT conditions runs exact-width kernels.
#ifndef PMT_SH_NTF
#define PMT_SH_NTF 4
#define PMT_SH_NTR 2
#define PMT_SH_NTD 4
#define PMT_SH_NTE 1
#define PMT_SH_F 61
#define PMT_SH_R 30
#define PMT_SH_D 60
#define PMT_SH_H 10
#define PMT_SH_E 10
#endif
#ifdef PMT_GENERIC_ONLY
#define PMT_GENERIC_ONLY_BUILD PMT_GENERIC_ONLY  // (a build without exact instances: the shape above is not used)
#else
#define PMT_GENERIC_ONLY_BUILD 0
#endif
#define PMT_SH_TILES PMT_SH_NTF, PMT_SH_NTR, PMT_SH_NTD, PMT_SH_NTE
#define PMT_SH_DIMS PMT_SH_F, PMT_SH_R, PMT_SH_D, PMT_SH_H, PMT_SH_E
static_assert(PMT_SH_F <= 16 * PMT_SH_NTF && PMT_SH_F > 16 * (PMT_SH_NTF - 1) && PMT_SH_R <= 16 * PMT_SH_NTR && PMT_SH_R > 16 * (PMT_SH_NTR - 1) &&
              PMT_SH_D <= 16 * PMT_SH_NTD && PMT_SH_D > 16 * (PMT_SH_NTD - 1) && PMT_SH_E <= 16 * PMT_SH_NTE && PMT_SH_E > 16 * (PMT_SH_NTE - 1) &&
              PMT_SH_H >= 1 && PMT_SH_H <= PMT_MAX_HALF_FFN && (PMT_SH_H > 16 * (PMT_HT - 1) || PMT_GENERIC_ONLY_BUILD), "SHAPE: the widths must fill exactly the tile counts given");
#ifndef PMT_GENERIC_ONLY
#define PMT_GENERIC_ONLY 0  // 1 (the WIDE build, csrc/Makefile): no exact instances -- every shape below is the generic one, pmt_shape_id is 0
#endif
#if PMT_GENERIC_ONLY
using ShapeP0 = ShapeAny; using ShapeP0X = ShapeAny; using ShapeP0XB = ShapeAny; using ShapeP0XD = ShapeAny; using ShapeP0XH = ShapeAny;
using ShapeP0XHD = ShapeAny; using ShapeP0T = ShapeAny; using ShapeP0TH = ShapeAny;
#else
using ShapeP0 = Shape<PMT_SH_TILES, true>;     // the shape's TILE counts (P0: F in 49..64, read widths 17..32, d_model / reducer widths 49..64, E <= 16), widths at run time
using ShapeP0X = Shape<PMT_SH_TILES, true, PMT_SH_DIMS, 3>;  // exactly the shape's widths (default: the production hyperparameters, SURVEY: P0)
using ShapeP0XB = Shape<PMT_SH_TILES, true, PMT_SH_DIMS, 1>; // the same widths, plain bf16 products (not a parity mode)
using ShapeP0XD = Shape<PMT_SH_TILES, true, PMT_SH_DIMS, 3, true>;  // the shape in a training step WITH dropout
// The FORWARD instances of the shape since round 4: the same widths with the products as THREE f16 MFMAs on two-piece
// splits (XBF = 16 = PMT_F16X2, linear_acc_f16 below; fp32-equivalent like the six bf16 MFMAs).  The backward keeps its bf16
// pieces (gradients need bf16's exponent range), so ShapeP0X / ShapeP0XD above remain its instances and, for the forward, the
// round-3 form that PmtModel.force_shape = 5 asks for.
using ShapeP0XH = Shape<PMT_SH_TILES, true, PMT_SH_DIMS, 16>;
using ShapeP0XHD = Shape<PMT_SH_TILES, true, PMT_SH_DIMS, 16, true>;
// A model that fills the shape's TILE counts but not its widths (other widths inside the same tiles, or layers of different widths
// within one MLP -- the reference's test configuration T0: reducer 30 -> 20 -> 20 -> 20) still runs on the 16-bit matrix pipes:
// the tile-exact instances with the widths read from the descriptor (pmt_shape_id: 6).  Padding positions hold zeros in the
// packed weights and in every activation, so whole tiles are multiplied without guards.
using ShapeP0T = Shape<PMT_SH_TILES, true, 0, 0, 0, 0, 0, 3>;    // backward (bf16 pieces)
using ShapeP0TH = Shape<PMT_SH_TILES, true, 0, 0, 0, 0, 0, 16>;  // forward (f16 pieces)
#endif

DEV float uniform(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); }
DEV int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// Cross-lane sums without the LDS pipe (__shfl_xor lowers to ds_bpermute_b32): DPP-modified adds inside a row of 16
// lanes, and the gfx950 row-swap instructions across rows.
template <int CTRL>
DEV float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a workgroup-scope fence + barrier and drains this
// wave's GLOBAL memory queue too (s_waitcnt vmcnt(0)): every stash store, gradient atomic and load in flight would have to
// come home before each of a workgroup's barriers.  The kernels' barriers hand over LDS data only.
DEV void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- wavefront segmented reduce over the reads of a tile (reference sets/ragged_sets.py:144-158: sums over sets) -----------
// The 16 reads of a tile sit in the 16 lanes of a DPP row, the reads of one set in CONTIGUOUS lanes.  A Hillis-Steele scan with
// row shifts 1, 2, 4, 8 that only adds across equal set ids leaves every set's total in the LAST lane of its run; that lane
// alone adds to the per-set LDS accumulator.  One ds_add per (set, tile, value) without address conflicts inside the
// instruction, where every read used to add for itself and up to 16 lanes hit one address (serialised, ~100 cycles).
template <int CTRL>
DEV int dpp_mov_int(int v, int absent) {  // `absent`: what lanes without a source lane read
    return __builtin_amdgcn_update_dpp(absent, v, CTRL, 0xF, 0xF, false);
}
#ifndef PMT_SEG_FMAC
#define PMT_SEG_FMAC 1
#endif
struct SegPlan {
    bool t1, t2, t4, t8;  // lane r - d holds the same set
    bool last;            // the run ends in this lane
    float m1, m2, m4, m8; // the same as multipliers (1.0 / 0.0): a scan step is ONE v_fmac_f32 with a DPP-shifted operand
};
DEV SegPlan seg_plan(int key) {  // key: the read's set, < 0 for lanes without a read
    SegPlan p;
    const bool ok = key >= 0;
    int k1 = dpp_mov_int<0x111>(key, -2), k2 = dpp_mov_int<0x112>(key, -2), k4 = dpp_mov_int<0x114>(key, -2),  // row_shr:1, 2, 4, 8
        k8 = dpp_mov_int<0x118>(key, -2), kn = dpp_mov_int<0x101>(key, -2);                                     // row_shl:1: the next lane's set
    asm volatile("" : "+v"(k1), "+v"(k2), "+v"(k4), "+v"(k8), "+v"(kn));  // (all lanes shift: see dpp_mov_all)
    p.t1 = ok && k1 == key;
    p.t2 = ok && k2 == key;
    p.t4 = ok && k4 == key;
    p.t8 = ok && k8 == key;
    p.last = ok && kn != key;
    p.m1 = p.t1 ? 1.f : 0.f; p.m2 = p.t2 ? 1.f : 0.f; p.m4 = p.t4 ? 1.f : 0.f; p.m8 = p.t8 ? 1.f : 0.f;
    return p;
}
template <int CTRL>
DEV float dpp_mov_all(float v) {
    // every lane of the row must execute the shift (a lane that sits it out cannot be a SOURCE either): the empty asm keeps
    // hipcc 7.2 from sinking the cross-lane move into the select that consumes it
    float t = dpp_mov<CTRL>(v);
    asm volatile("" : "+v"(t));
    return t;
}
// A step of the scan is v += m * (v of lane r - d) with m = 1.0 / 0.0: ONE v_fmac_f32 with a DPP-shifted operand.  0 * inf is NaN, so
// a non-finite value in one read set would leak into the sets that share its row of 16 lanes, where the reference (and a select)
// confines it to the offending set.  A GUARDed scan therefore first asks whether the WAVE holds a non-finite input at
// all (one v_cmp_class per value, a scalar OR and a branch) and only then takes the select form below, six issue slots a step.
// The guard costs the production forward 6 % (0.548 -> 0.582 ms: eight more spilled registers at its 128), and exactly that
// instance cannot meet a non-finite activation unless its WEIGHTS are non-finite (every set is NaN then, nothing to confine): its
// inputs are bytes, its matrix operands f16 pieces that saturate at +-65504 (MODE.FP16_OVFL clamps even an inf input), and
// nothing between two LayerNorms can reach 3e38 from there.  So the f16 instances (and the backward that consumes their stash)
// scan unguarded; every other instance -- fp32 or bf16 operands, where an inf input does propagate -- keeps the guard
// (tests/test_forward_gpu.py: test_a_non_finite_read_set_does_not_leak_into_its_neighbours runs all of them).
DEV bool seg_all_finite(float v) { return !__builtin_amdgcn_classf(v, 0x203); }  // not (signalling | quiet NaN | -inf | +inf)
DEV bool wave_any(bool b) { return __builtin_amdgcn_ballot_w64(b) != 0ull; }
DEV float seg_sum_select(float v, const SegPlan& p) {
    float t = dpp_mov_all<0x111>(v);
    v += p.t1 ? t : 0.f;
    t = dpp_mov_all<0x112>(v);
    v += p.t2 ? t : 0.f;
    t = dpp_mov_all<0x114>(v);
    v += p.t4 ? t : 0.f;
    t = dpp_mov_all<0x118>(v);
    v += p.t8 ? t : 0.f;
    return v;
}
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "the inline-asm DPP scans below hard-code gfx950's wave64 rows of 16 lanes and its VALU -> DPP wait states"
#endif
template <bool GUARD = true>
DEV float seg_sum(float v, const SegPlan& p) {  // v must be 0 in lanes without a read
    if (PMT_SEG_FMAC && (!GUARD || !wave_any(!seg_all_finite(v)))) {
        // v += m_d * (v of lane r - d), d = 1, 2, 4, 8: four fused multiply-adds whose first operand comes through the DPP
        // row shift (lanes without a source lane keep their value; their multiplier is 0 anyway).  Was per step: a move, the
        // DPP move, a select on a scalar-register mask and an add, with two wait states in between -- six issue slots, and
        // four scalar register pairs per plan.  s_nop 1 = the two wait states between a VALU write and a DPP read of it.
        asm("s_nop 1\n\tv_fmac_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 1\n\tv_fmac_f32_dpp %0, %0, %2 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 1\n\tv_fmac_f32_dpp %0, %0, %3 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 1\n\tv_fmac_f32_dpp %0, %0, %4 row_shr:8 row_mask:0xf bank_mask:0xf"
            : "+v"(v) : "v"(p.m1), "v"(p.m2), "v"(p.m4), "v"(p.m8));
        return v;
    }
    return seg_sum_select(v, p);
}
// four independent values at once: the steps of the four scans interleave, so no wait state is needed between them
template <bool GUARD = true>
DEV f4 seg_sum4(f4 v, const SegPlan& p) {
    const bool finite = !GUARD || (seg_all_finite(v[0]) && seg_all_finite(v[1]) && seg_all_finite(v[2]) && seg_all_finite(v[3]));
    if (!PMT_SEG_FMAC || (GUARD && wave_any(!finite))) return f4{seg_sum_select(v[0], p), seg_sum_select(v[1], p), seg_sum_select(v[2], p), seg_sum_select(v[3], p)};
    float a = v[0], b = v[1], c = v[2], d = v[3];
#define PMT_SEG_STEP(M, SH)                                                            \
    "v_fmac_f32_dpp %0, %0, %" #M " row_shr:" #SH " row_mask:0xf bank_mask:0xf\n\t"  \
    "v_fmac_f32_dpp %1, %1, %" #M " row_shr:" #SH " row_mask:0xf bank_mask:0xf\n\t"  \
    "v_fmac_f32_dpp %2, %2, %" #M " row_shr:" #SH " row_mask:0xf bank_mask:0xf\n\t"  \
    "v_fmac_f32_dpp %3, %3, %" #M " row_shr:" #SH " row_mask:0xf bank_mask:0xf\n\t"
    asm("s_nop 1\n\t" PMT_SEG_STEP(4, 1) PMT_SEG_STEP(5, 2) PMT_SEG_STEP(6, 4) PMT_SEG_STEP(7, 8) "s_nop 0"
        : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(p.m1), "v"(p.m2), "v"(p.m4), "v"(p.m8));
#undef PMT_SEG_STEP
    return f4{a, b, c, d};
}

// sum over the 4 lane groups (lanes r, r+16, r+32, r+48): completes a per-read reduction over features
DEV float group_sum(float v) {
    // Inline asm, not __builtin_amdgcn_permlane{16,32}_swap: hipcc 7.2 folds the builtin's two results into one register
    // when they are added (it emits r[0] + r[0]).  s_nop 1 = the two wait states between a VALU write and the swap.
    float a = v, b = v;
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));  // a = rows {0,0,2,2}, b = rows {1,1,3,3}
    a += b;
    b = a;
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));  // a = lower half twice, b = upper half twice
    return a + b;
}

DEV float fast_rcp(float v) { return __builtin_amdgcn_rcpf(v); }

DEV f4 mfma16(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

DEV float selu1(float x) {
    // scale * (x > 0 ? x : alpha * (exp(x) - 1)).  The negative branch uses the hardware exp2 (v_exp_f32, ~1 ulp):
    // absolute error <= ~1.2e-7 on a value in (-1.76, 0], the same size as one fp32 rounding of an O(1) activation.
    // (OCML expm1f costs ~35 VALU instructions per element and made the kernels VALU-bound.)
    const float neg = (PMT_SELU_ALPHA * PMT_SELU_SCALE) * (__builtin_amdgcn_exp2f(x * 1.4426950408889634f) - 1.0f);
    return x > 0.f ? PMT_SELU_SCALE * x : neg;
}
// Four at a time with the packed fp32 instructions (v_pk_mul_f32 / v_pk_fma_f32 take two lanes' worth per issue slot):
// 4.5 instead of 7 VALU slots per element; SELU is a third of the non-matrix instructions of the MLP layers.  Same
// formula and roundings as selu1 except that alpha*scale*(e - 1) is one fused multiply-add (e * c - c).
typedef float f2 __attribute__((ext_vector_type(2)));
#ifndef PMT_SELU_MINMAX
#define PMT_SELU_MINMAX 0
#endif
#ifndef PMT_SELU_SCALAR
#define PMT_SELU_SCALAR 0
#endif
DEV f4 selu4(f4 v) {
    if (PMT_SELU_MINMAX) {
        // branch-free form: scale * max(x, 0) + min(c e^x - c, 0), c = alpha * scale -- bit for bit the two-branch result (for
        // x > 0 the second term is 0 and the fma rounds scale * x once; for x <= 0 the first is 0), without the compare /
        // select pair per element and the wait state between them
        constexpr float c = PMT_SELU_ALPHA * PMT_SELU_SCALE;
        f4 r;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float e = __builtin_amdgcn_exp2f(v[i] * 1.4426950408889634f);
            const float neg = fminf(__builtin_fmaf(e, c, -c), 0.f);
            r[i] = __builtin_fmaf(PMT_SELU_SCALE, fmaxf(v[i], 0.f), neg);
        }
        return r;
    }
    if (PMT_SELU_SCALAR) {  // the same arithmetic without the packed fp32 instructions (A/B switch)
        f4 r;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float t, e, n, p;
            asm("v_mul_f32 %0, 0x3fb8aa3b, %1" : "=v"(t) : "v"(v[i]));
            e = __builtin_amdgcn_exp2f(t);
            constexpr float c = PMT_SELU_ALPHA * PMT_SELU_SCALE;
            n = __builtin_fmaf(e, c, -c);
            p = v[i] * PMT_SELU_SCALE;
            asm volatile("" : "+v"(n), "+v"(p));
            r[i] = v[i] > 0.f ? p : n;
        }
        return r;
    }
    const f2 lo = f2{v[0], v[1]}, hi = f2{v[2], v[3]};
    const f2 tl = lo * 1.4426950408889634f, th = hi * 1.4426950408889634f;
    const f2 el = f2{__builtin_amdgcn_exp2f(tl[0]), __builtin_amdgcn_exp2f(tl[1])};
    const f2 eh = f2{__builtin_amdgcn_exp2f(th[0]), __builtin_amdgcn_exp2f(th[1])};
    constexpr float c = PMT_SELU_ALPHA * PMT_SELU_SCALE;
    const f2 nl = __builtin_elementwise_fma(el, f2{c, c}, f2{-c, -c}), nh = __builtin_elementwise_fma(eh, f2{c, c}, f2{-c, -c});
    const f2 pl = lo * PMT_SELU_SCALE, ph = hi * PMT_SELU_SCALE;
    return f4{v[0] > 0.f ? pl[0] : nl[0], v[1] > 0.f ? pl[1] : nl[1], v[2] > 0.f ? ph[0] : nh[0], v[3] > 0.f ? ph[1] : nh[1]};
}

// d selu(a) / da expressed through the OUTPUT s = selu(a):  s > 0 ? scale : s + alpha*scale
DEV float selu_grad_from_out(float s) { return s > 0.f ? PMT_SELU_SCALE : s + PMT_SELU_ALPHA * PMT_SELU_SCALE; }

// feature index held by register j of tile t for this lane's group g
DEV int feat_of(int t, int j, int g) { return 16 * t + 4 * j + g; }

// load a tile-position vector ("pvec"): 4 consecutive floats for (tile t, group g)
DEV f4 load_pvec(const float* __restrict__ p, int t, int g) { return *reinterpret_cast<const f4*>(p + 16 * t + 4 * g); }

// ---------------------------------------------------------------------------------------------------------------
// acc[rt][mt] += sum_k W[m][k] * in[rt][k]   for every tile of the wave.
// `frag` = packed A fragments of W ([out_dim][in_dim]).  SELU_IN applies SELU to the input on the fly.
// ---------------------------------------------------------------------------------------------------------------
// KDIM / KSPLIT (compile time, 0 = unknown): the input width, or for the split layout of a block's hidden state the width
// of each 16-position half.  A k-step covers 4 consecutive features; with the width known, the steps that hold nothing
// but padding are not issued (d_model 60: 15 of 16; d_ffn / 2 = 10: 3 of 4).
template <int KDIM, int KSPLIT>
DEV constexpr bool kstep_live(int kt, int j) {
    return KSPLIT > 0 ? 16 * (kt % PMT_HT) + 4 * j < KSPLIT : (KDIM > 0 ? 16 * kt + 4 * j < KDIM : true);
}
template <int NTI, int NTO, bool SELU_IN, bool EXACT, int KDIM = 0, int KSPLIT = 0>
DEV void linear_acc_impl(f4 (&acc)[PMT_RT][NTO], const f4 (&in)[PMT_RT][NTI], const float* __restrict__ frag, int in_dim,
                         int out_dim, float in_scale) {
    // Fragments are stored kt-major ((kt * nmt + mt) * 256 floats), i.e. in exactly the order this loop nest consumes
    // them, so the NEXT fragment is one fixed stride away and is fetched before the current fragment's MFMAs: the LDS
    // (or L2) latency hides behind 4 * PMT_RT MFMAs.  All 4 k-steps of a tile always run: the fragment rows / columns
    // beyond the layer's true dimensions are zero, and so are the activations there.
    const int nkt = (in_dim + 15) >> 4, nmt = (out_dim + 15) >> 4;
    const f4* __restrict__ fp = reinterpret_cast<const f4*>(frag) + (pmt_tid() & 63);
    f4 a_next = fp[0];
#pragma unroll
    for (int kt = 0; kt < NTI; ++kt) {
        if (EXACT || kt < nkt) {
            f4 b[PMT_RT];
#pragma unroll
            for (int rt = 0; rt < PMT_RT; ++rt) b[rt] = SELU_IN ? selu4(in[rt][kt]) * in_scale : in[rt][kt];
#pragma unroll
            for (int mt = 0; mt < NTO; ++mt) {
                if (EXACT || mt < nmt) {
                    const f4 a = a_next;
                    fp += 64;
                    a_next = fp[0];  // one fragment past the end on the last step: still inside the padded region
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (kstep_live<KDIM, KSPLIT>(kt, j)) {
#pragma unroll
                            for (int rt = 0; rt < PMT_RT; ++rt)
                                acc[rt][mt] = mfma16(a[j], b[rt][j], acc[rt][mt]);
                        }
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The same product on the bf16 matrix pipe.  v_mfma_f32_16x16x32_bf16 does 8x the work of v_mfma_f32_16x16x4_f32 in half
// the cycles, and an fp32 value is EXACTLY the sum of three bf16 pieces (8 + 8 + 8 significant bits), so
//     w x = (wh + wm + wl)(xh + xm + xl) = wh xh + (wh xm + wm xh) + (wh xl + wm xm + wl xh) + O(2^-24 |w x|):
// six bf16 MFMAs (fp32 accumulation inside the matrix core) reproduce the fp32 product to the last bit or two, in
// 6 x 16 cycles per 32-wide k block instead of 8 x 32.  The weights are split once per step by pmt_pack_params
// (PmtLinear.wb_frag); the activations are split here, ~5 VALU operations per element.  One k block = two activation
// tiles in their fp32 register order (see the pack kernel), so layers chain exactly as in linear_acc.
// ---------------------------------------------------------------------------------------------------------------
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
DEV f4 mfma_bf16(bf8 a, bf8 b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
// the 16-deep variant on elements 0..3 of both operands (lane (m, kg) supplies k = 4 kg + i: one activation tile)
typedef short s4v __attribute__((ext_vector_type(4)));
DEV f4 mfma_bf16_k16(bf8 a, bf8 b, f4 c) {
    typedef short s8v __attribute__((ext_vector_type(8)));
    const s8v sa = __builtin_bit_cast(s8v, a), sb = __builtin_bit_cast(s8v, b);
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(s4v{sa[0], sa[1], sa[2], sa[3]}, s4v{sb[0], sb[1], sb[2], sb[3]}, c, 0, 0, 0);
}
DEV void split_bf16x3(float x, __bf16& hi, __bf16& mid, __bf16& lo) {
    hi = (__bf16)x;
    const float r1 = x - (float)hi;
    mid = (__bf16)r1;
    lo = (__bf16)(r1 - (float)mid);
}
// Splitting a pair of values into bf16 pieces: pack (v_cvt_pk_bf16_f32), subtract each half back (shift / mask the half
// into a float, one v_pk_add_f32 for both), pack the residuals, ...: 9 VALU operations for three pieces, 6 for two.
// PMT_SPLIT_DOT2 = 1 forms the residuals on the dot-product unit instead (v_dot2c_f32_bf16 with the multiplier pair (-1, 0) or
// (0, -1): acc - one half, exact, 7 / 4 operations): measured equal in the backward and 6 % SLOWER in the filter forward
// (0.81 -> 0.86 ms) -- the instruction is not full rate.  Kept as a switch.
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
DEV unsigned pack_bf16_pair(float lo, float hi) {  // v_cvt_pk_bf16_f32, round to nearest even
    const bf2 p = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned, p);
}
// (the multiplier pairs travel in registers: as a compile-time constant hipcc 7.2 encodes the pair (-1, 0) = 0x0000BF80 as the
//  INLINE constant -1.0, which the instruction reads as 0xBF800000 = the pair (0, -1))
DEV unsigned opaque_bits(unsigned v) {
    asm volatile("" : "+s"(v));
    return v;
}
#ifndef PMT_SPLIT_DOT2
#define PMT_SPLIT_DOT2 0
#endif
DEV float minus_lo_half(unsigned pk, float acc) {
    if (!PMT_SPLIT_DOT2) return acc - __builtin_bit_cast(float, pk << 16);
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, pk), __builtin_bit_cast(bf2, opaque_bits(0x0000BF80u)), acc, false);
}
DEV float minus_hi_half(unsigned pk, float acc) {
    if (!PMT_SPLIT_DOT2) return acc - __builtin_bit_cast(float, pk & 0xFFFF0000u);
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, pk), __builtin_bit_cast(bf2, opaque_bits(0xBF800000u)), acc, false);
}
template <int PIECES>
DEV void split_pair(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
    h = pack_bf16_pair(a, b);
    if constexpr (PIECES >= 2) {
        const float ra = minus_lo_half(h, a), rb = minus_hi_half(h, b);
        m = pack_bf16_pair(ra, rb);
        if constexpr (PIECES >= 3) l = pack_bf16_pair(minus_lo_half(m, ra), minus_hi_half(m, rb));
    }
}
// the bf16 pieces of k block `kb` of the input (two activation tiles in register order), SELU applied on the way if asked
template <int NTI, bool SELU_IN, int PIECES>
DEV void split_kblock(const f4 (&in)[PMT_RT][NTI], int kb, float in_scale, bf8 (&bh)[PMT_RT], bf8 (&bm)[PMT_RT], bf8 (&bl)[PMT_RT]) {
    typedef unsigned u4v __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int rt = 0; rt < PMT_RT; ++rt) {
        const f4 zero = f4{0.f, 0.f, 0.f, 0.f};
        f4 v0 = in[rt][2 * kb], v1 = (2 * kb + 1 < NTI) ? in[rt][(2 * kb + 1 < NTI) ? 2 * kb + 1 : 0] : zero;
        if (SELU_IN) {
            v0 = selu4(v0) * in_scale;
            if (2 * kb + 1 < NTI) v1 = selu4(v1) * in_scale;
        }
        unsigned hh[4] = {0u, 0u, 0u, 0u}, mm[4] = {0u, 0u, 0u, 0u}, ll[4] = {0u, 0u, 0u, 0u};
        split_pair<PIECES>(v0[0], v0[1], hh[0], mm[0], ll[0]);
        split_pair<PIECES>(v0[2], v0[3], hh[1], mm[1], ll[1]);
        if (2 * kb + 1 < NTI) {
            split_pair<PIECES>(v1[0], v1[1], hh[2], mm[2], ll[2]);
            split_pair<PIECES>(v1[2], v1[3], hh[3], mm[3], ll[3]);
        }
        const u4v h = {hh[0], hh[1], hh[2], hh[3]}, m = {mm[0], mm[1], mm[2], mm[3]}, l = {ll[0], ll[1], ll[2], ll[3]};
        bh[rt] = __builtin_bit_cast(bf8, h);
        if constexpr (PIECES >= 2) bm[rt] = __builtin_bit_cast(bf8, m);
        if constexpr (PIECES >= 3) bl[rt] = __builtin_bit_cast(bf8, l);
    }
}
#ifndef PMT_TWO_PIECE_MFMAS
#define PMT_TWO_PIECE_MFMAS 3  // MFMAs of a product with two-piece activations: 5 (all but a_hi b_lo) or 3 (first order only)
#endif
#ifndef PMT_BF16_K16_TAIL
#define PMT_BF16_K16_TAIL 0  // 1 (development: reproduces the hazard): the 16-deep MFMA for the half-filled last k block of 3, 5, 7 input tiles
#endif
#ifndef PMT_FRAG_AHEAD
#define PMT_FRAG_AHEAD 0  // the forward (4 waves per SIMD hide the latency; no registers to spare): 1.15 ms -> 1.18 with 1
#endif
template <int NTI, int NTO, bool SELU_IN, int PIECES = 3>
DEV void linear_acc_bf16(f4 (&acc)[PMT_RT][NTO], const f4 (&in)[PMT_RT][NTI], const float* __restrict__ fragb, float in_scale = 1.0f) {
    // PIECES = pieces of the ACTIVATION: 3 fp32-equivalent (six MFMAs), 2 hi + mid (16 significant bits; five MFMAs against the
    // three-piece weights), 1 plain bf16.
    // The weight fragments of step (kb, mt) lie 3 KiB apart in consumption order and come from L2 (a step's weights are
    // ~0.7 MB: far beyond the 32 KiB L1), ~2 x the 12 MFMAs of a step away: they are fetched PMT_FRAG_AHEAD steps ahead, the
    // first ones before the input is split, so that no MFMA group waits for its own loads.
    static_assert(PIECES >= 1 && PIECES <= 3, "pieces");
    constexpr int NKB = (NTI + 1) / 2, NSTEP = NKB * NTO, AH = PMT_FRAG_AHEAD < NSTEP ? PMT_FRAG_AHEAD : NSTEP;
    constexpr int NP = PIECES == 1 ? 1 : 3;  // (the weights keep their three-piece layout; the hi piece alone is the bf16 rounding)
    const bf8* __restrict__ fp = reinterpret_cast<const bf8*>(fragb) + (pmt_tid() & 63);
    bf8 q[AH + 1][NP];
#pragma unroll
    for (int s = 0; s < AH; ++s)
#pragma unroll
        for (int p = 0; p < NP; ++p) q[s][p] = fp[192 * s + 64 * p];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        bf8 bh[PMT_RT], bm[PMT_RT], bl[PMT_RT];
        split_kblock<NTI, SELU_IN, PIECES>(in, kb, in_scale, bh, bm, bl);
        // A k block with one tile only: the 16-deep MFMA on the lower halves -- when it is the ONLY k block (NTI == 1).  The half-filled
        // LAST block of 3, 5 or 7 input tiles takes the 32-deep MFMA on its zero-padded operands (the fragments' upper halves are zeros,
        // pmt_pack_kernel): `v_mfma_f32_16x16x16_bf16 D, a, b, D` straight behind a `v_mfma_f32_16x16x32_bf16` that writes D reads the
        // accumulator before the longer instruction has written it back -- the hardware interlocks an accumulation chain of ONE opcode
        // only, and hipcc 7.2 puts no wait state between the two (scripts/microbench/mfma_chain_hazard.hip).  Wrong sums whenever
        // the two issue back to back, i.e. depending on what the SIMD's other wave does: the "race" of round 4's split read sets.
        const bool half_block = (NKB == 1 || PMT_BF16_K16_TAIL) && 2 * kb + 1 >= NTI;
#pragma unroll
        for (int mt = 0; mt < NTO; ++mt) {
            const int step = kb * NTO + mt;
            if (step + AH < NSTEP) {
#pragma unroll
                for (int p = 0; p < NP; ++p) q[(step + AH) % (AH + 1)][p] = fp[192 * (step + AH) + 64 * p];
            }
            const bf8 ah = q[step % (AH + 1)][0];
            if constexpr (PIECES == 1) {
#pragma unroll
                for (int rt = 0; rt < PMT_RT; ++rt) acc[rt][mt] = half_block ? mfma_bf16_k16(ah, bh[rt], acc[rt][mt]) : mfma_bf16(ah, bh[rt], acc[rt][mt]);
            } else {
                const bf8 am = q[step % (AH + 1)][NP > 1 ? 1 : 0], al = q[step % (AH + 1)][NP > 2 ? 2 : 0];
#pragma unroll
                for (int rt = 0; rt < PMT_RT; ++rt) {  // smallest terms first
                    // (two-piece activations drop the a_hi b_lo term, 2^-16 of the product: with PMT_TWO_PIECE_MFMAS = 3 the two
                    //  other terms of that order, a_lo b_hi and a_mid b_mid, go as well -- three MFMAs, the same error order)
                    constexpr bool SECOND_ORDER = PIECES == 3 || PMT_TWO_PIECE_MFMAS != 3;
                    if (half_block) {
                        if constexpr (SECOND_ORDER) acc[rt][mt] = mfma_bf16_k16(al, bh[rt], acc[rt][mt]);
                        if constexpr (PIECES == 3) acc[rt][mt] = mfma_bf16_k16(ah, bl[rt], acc[rt][mt]);
                        if constexpr (SECOND_ORDER) acc[rt][mt] = mfma_bf16_k16(am, bm[rt], acc[rt][mt]);
                        acc[rt][mt] = mfma_bf16_k16(am, bh[rt], acc[rt][mt]);
                        acc[rt][mt] = mfma_bf16_k16(ah, bm[rt], acc[rt][mt]);
                        acc[rt][mt] = mfma_bf16_k16(ah, bh[rt], acc[rt][mt]);
                    } else {
                        if constexpr (SECOND_ORDER) acc[rt][mt] = mfma_bf16(al, bh[rt], acc[rt][mt]);
                        if constexpr (PIECES == 3) acc[rt][mt] = mfma_bf16(ah, bl[rt], acc[rt][mt]);
                        if constexpr (SECOND_ORDER) acc[rt][mt] = mfma_bf16(am, bm[rt], acc[rt][mt]);
                        acc[rt][mt] = mfma_bf16(am, bh[rt], acc[rt][mt]);
                        acc[rt][mt] = mfma_bf16(ah, bm[rt], acc[rt][mt]);
                        acc[rt][mt] = mfma_bf16(ah, bh[rt], acc[rt][mt]);
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The same product on the f16 matrix pipe with TWO pieces per operand: three MFMAs instead of six, five vector operations per
// pair of activation values instead of nine, two KiB of weight fragments per (out tile, k block) instead of three.
//   x = xh + 2^-12 xl',  xh = f16(x),  xl' = f16(2^12 (x - xh));   w = wh + 2^-12 wl' likewise (pmt_pack_params, wh_frag)
// x - xh is exact in fp32 and at most 2^-11 |x| (round to nearest), so the scaled low piece is a normal f16 with its full 11
// bits whenever xh is normal: the pair carries x to 2^-23 relative from |x| = 6e-5 to 65504, over the whole range the
// hardware's f16 has -- unscaled, the low piece of any |x| < 1/4 would be a denormal (measured: an activation vector of size
// 1e-3 lost half its digits, scripts/microbench/f16x2.hip).  Then
//   w x = wh xh + 2^-12 (wh xl' + wl' xh) + 2^-24 wl' xl':
// the first product accumulates straight into `acc`, the two first-order ones into an accumulator of their own per out tile
// that joins `acc` with one fused multiply-add, the last term (at most 2^-22 |w x|, 2^-25 rms) is dropped.  Measured against
// fp64 on 60-wide dot products: rms error 1.9e-8 of the sum of |terms| (three bf16 pieces / six MFMAs: 1.7e-8; sequential fp32
// fused multiply-adds: 2.9e-8).  Range: the kernels run with MODE.FP16_OVFL set, so an activation beyond +-65504 saturates
// (finite, wrong) instead of turning into inf - inf; no activation of a LayerNorm'ed network is near it, and the limit is
// stated in DESIGN.md.  Loop order: out tile outermost -- the low-order accumulator is 2 x 4 registers, the pieces of the whole
// input (NKB x 2 tiles x 2 pieces x 4 registers) are made once up front, after which the input registers are dead.
// ---------------------------------------------------------------------------------------------------------------
#define PMT_F16X2 16  // Shape::BF16 value of the instances whose forward products run this way
#define PMT_F16_OPERAND_MAX 65504.0f
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4v __attribute__((ext_vector_type(4)));
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
DEV f4 mfma_f16(h8 a, h8 b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
DEV f4 mfma_f16_k16(h8 a, h8 b, f4 c) {  // the 16-deep variant on elements 0..3 of both operands (one activation tile)
    return __builtin_amdgcn_mfma_f32_16x16x16f16(h4v{a[0], a[1], a[2], a[3]}, h4v{b[0], b[1], b[2], b[3]}, c, 0, 0, 0);
}
DEV void fp16_saturate_on() { asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1"); }  // MODE.FP16_OVFL
#ifndef PMT_F16_SPLIT_ASM
#define PMT_F16_SPLIT_ASM 1
#endif
DEV void split_pair_f16(float a, float b, unsigned& h, unsigned& l, float k4096) {
    if (PMT_F16_SPLIT_ASM) {
        // v_cvt_pk_f16_f32 (round to nearest even); the residuals straight from the packed halves (v_fma_mix_f32 reads an f16
        // half as an operand: x - h, exact); scaled and rounded by v_fma_mixlo / mixhi_f16 (fp32 product, one rounding)
        float ra, rb;
        asm("v_cvt_pk_f16_f32 %0, %3, %4\n\t"
            "v_fma_mix_f32 %1, %0, -1.0, %3 op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %2, %0, -1.0, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
            : "=&v"(h), "=&v"(ra), "=&v"(rb) : "v"(a), "v"(b));
        // (s_nop 1: the packed low pieces go straight into an MFMA as its B operand, and v_fma_mixhi_f16 -- a 16-bit write that keeps the
        //  other half -- is not home when an MFMA issued right behind it reads the register: wrong low pieces, a 1e-4-relative
        //  error, in ONE instance (filter forward, 7 + 2 x 2 tiles) until the wait states went in; the compiler does not look for
        //  hazards behind inline asm.  One wait state was enough on the hardware; two are issued.)
        asm("v_fma_mixlo_f16 %0, %1, %3, 0\n\t"
            "v_fma_mixhi_f16 %0, %2, %3, 0\n\t"
            "s_nop 1"
            : "=&v"(l) : "v"(ra), "v"(rb), "v"(k4096));
        return;
    }
    const h2v hh = {(_Float16)a, (_Float16)b};
    const h2v ll = {(_Float16)((a - (float)hh[0]) * 4096.f), (_Float16)((b - (float)hh[1]) * 4096.f)};
    h = __builtin_bit_cast(unsigned, hh);
    l = __builtin_bit_cast(unsigned, ll);
}
#ifndef PMT_F16_K16_TAIL
#define PMT_F16_K16_TAIL 0  // 1: the 16-deep MFMA for the half-filled last k block of 3, 5, 7 input tiles too (see linear_acc_f16)
#endif
#ifndef PMT_F16_AHEAD
#define PMT_F16_AHEAD 0  // weight fragments of the next (out tile, k block) step requested one step ahead (8 more registers)
#endif
// X1: the input is exact in ONE f16 piece (packed read rows: bits and k / 32 quantiles; float16 read rows): no low piece is made and
// the wh * xl' product is skipped -- two MFMAs per product.
template <int NTI, int NTO, bool SELU_IN, bool X1 = false>
DEV void linear_acc_f16(f4 (&acc)[PMT_RT][NTO], const f4 (&in)[PMT_RT][NTI], const float* __restrict__ fragh, float in_scale = 1.0f) {
    typedef unsigned u4v __attribute__((ext_vector_type(4)));
    constexpr int NKB = (NTI + 1) / 2;
    const h8* __restrict__ fp = reinterpret_cast<const h8*>(fragh) + (pmt_tid() & 63);
    h8 xh[NKB][PMT_RT], xl[NKB][PMT_RT];
    float k4096 = 4096.f;
    asm volatile("" : "+v"(k4096));  // (one register for the whole kernel, not a literal per instruction)
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int rt = 0; rt < PMT_RT; ++rt) {
            const f4 zero = f4{0.f, 0.f, 0.f, 0.f};
            f4 v0 = in[rt][2 * kb], v1 = (2 * kb + 1 < NTI) ? in[rt][(2 * kb + 1 < NTI) ? 2 * kb + 1 : 0] : zero;
            if (SELU_IN) {
                v0 = selu4(v0) * in_scale;
                if (2 * kb + 1 < NTI) v1 = selu4(v1) * in_scale;
            }
            unsigned hh[4] = {0u, 0u, 0u, 0u}, ll[4] = {0u, 0u, 0u, 0u};
            if constexpr (X1) {
                const h2v p0 = {(_Float16)v0[0], (_Float16)v0[1]}, p1 = {(_Float16)v0[2], (_Float16)v0[3]};
                hh[0] = __builtin_bit_cast(unsigned, p0);
                hh[1] = __builtin_bit_cast(unsigned, p1);
                if (2 * kb + 1 < NTI) {
                    const h2v p2 = {(_Float16)v1[0], (_Float16)v1[1]}, p3 = {(_Float16)v1[2], (_Float16)v1[3]};
                    hh[2] = __builtin_bit_cast(unsigned, p2);
                    hh[3] = __builtin_bit_cast(unsigned, p3);
                }
            } else {
            split_pair_f16(v0[0], v0[1], hh[0], ll[0], k4096);
            split_pair_f16(v0[2], v0[3], hh[1], ll[1], k4096);
            if (2 * kb + 1 < NTI) {
                split_pair_f16(v1[0], v1[1], hh[2], ll[2], k4096);
                split_pair_f16(v1[2], v1[3], hh[3], ll[3], k4096);
            }
            }
            xh[kb][rt] = __builtin_bit_cast(h8, u4v{hh[0], hh[1], hh[2], hh[3]});
            xl[kb][rt] = __builtin_bit_cast(h8, u4v{ll[0], ll[1], ll[2], ll[3]});
        }
    constexpr int NSTEP = NTO * NKB;
    h8 qh[2], ql[2];
    if (PMT_F16_AHEAD) { qh[0] = fp[0]; ql[0] = fp[64]; }
#pragma unroll
    for (int mt = 0; mt < NTO; ++mt) {
        f4 lo[PMT_RT];
#pragma unroll
        for (int rt = 0; rt < PMT_RT; ++rt) lo[rt] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            const int step = mt * NKB + kb;
            h8 ah, al;
            if (PMT_F16_AHEAD) {
                ah = qh[step & 1]; al = ql[step & 1];
                if (step + 1 < NSTEP) { qh[(step + 1) & 1] = fp[128 * (step + 1)]; ql[(step + 1) & 1] = fp[128 * (step + 1) + 64]; }
            } else {
                ah = fp[128 * step]; al = fp[128 * step + 64];
            }
            // a k block with one tile only (NTI == 1): the 16-deep MFMA on the lower halves.  The half-filled LAST block of 3, 5 or 7 input
            // tiles (the wide build's shapes) takes the 32-deep one on its zero-padded operands instead: a 16-deep f16 MFMA chained
            // behind 32-deep ones on the same accumulator gave wrong sums (measured, scripts/wide_debug.py)
            const bool half_block = (NKB == 1 || PMT_F16_K16_TAIL) && 2 * kb + 1 >= NTI;
#pragma unroll
            for (int rt = 0; rt < PMT_RT; ++rt) {
                if (half_block) {
                    lo[rt] = mfma_f16_k16(al, xh[kb][rt], lo[rt]);
                    if constexpr (!X1) lo[rt] = mfma_f16_k16(ah, xl[kb][rt], lo[rt]);
                    acc[rt][mt] = mfma_f16_k16(ah, xh[kb][rt], acc[rt][mt]);
                } else {
                    lo[rt] = mfma_f16(al, xh[kb][rt], lo[rt]);
                    if constexpr (!X1) lo[rt] = mfma_f16(ah, xl[kb][rt], lo[rt]);
                    acc[rt][mt] = mfma_f16(ah, xh[kb][rt], acc[rt][mt]);
                }
            }
        }
#pragma unroll
        for (int rt = 0; rt < PMT_RT; ++rt) acc[rt][mt] = lo[rt] * (1.0f / 4096.f) + acc[rt][mt];
    }
}
// the layers' products of an exact-width instance: BF = PMT_F16X2 two f16 pieces (wh_frag), else bf16 pieces (wb_frag)
template <int NTI, int NTO, bool SELU_IN, int BF, bool X1 = false>
DEV void linear_acc_mx(f4 (&acc)[PMT_RT][NTO], const f4 (&in)[PMT_RT][NTI], const float* __restrict__ packed, const PmtLinear& L,
                       float in_scale = 1.0f) {
    if constexpr (BF == PMT_F16X2) linear_acc_f16<NTI, NTO, SELU_IN, X1>(acc, in, packed + uniform(L.wh_frag), in_scale);
    else linear_acc_bf16<NTI, NTO, SELU_IN, BF>(acc, in, packed + uniform(L.wb_frag), in_scale);
}

// A wave's tiles are all on one side of the ref / alt boundary (group_geometry), so no per-tile masks exist.
// EXACT (compile time): the caller's Shape guarantees that the layer fills every tile of both register arrays -> one
// straight-line MFMA chain with no per-tile guards (guards turn every accumulator into a web of PHI copies; they were
// the source of thousands of VGPR spills).  Otherwise the same test is made at run time.
template <int NTI, int NTO, bool SELU_IN, bool EXACT = false, int KDIM = 0, int KSPLIT = 0>
DEV void linear_acc(f4 (&acc)[PMT_RT][NTO], const f4 (&in)[PMT_RT][NTI], const float* __restrict__ frag, int in_dim,
                    int out_dim, float in_scale = 1.0f) {
    static_assert(EXACT || (KDIM == 0 && KSPLIT == 0), "compile-time widths belong to the exact instances");
    if (EXACT || (((in_dim + 15) >> 4) == NTI && ((out_dim + 15) >> 4) == NTO))
        linear_acc_impl<NTI, NTO, SELU_IN, true, KDIM, KSPLIT>(acc, in, frag, in_dim, out_dim, in_scale);
    else
        linear_acc_impl<NTI, NTO, SELU_IN, false>(acc, in, frag, in_dim, out_dim, in_scale);
}

// acc[rt][mt] = bias (tile-position order) for every tile; rows beyond out_dim are zero in the packed bias
template <int NTO>
DEV void init_bias(f4 (&acc)[PMT_RT][NTO], const float* __restrict__ bias_pvec, int out_dim, int g) {
    const int nmt = (out_dim + 15) >> 4;
#pragma unroll
    for (int mt = 0; mt < NTO; ++mt) {
        f4 b = f4{0.f, 0.f, 0.f, 0.f};
        if (bias_pvec != nullptr && mt < nmt) b = load_pvec(bias_pvec, mt, g);
#pragma unroll
        for (int rt = 0; rt < PMT_RT; ++rt) acc[rt][mt] = b;
    }
}

// LayerNorm over the feature axis of one read tile.  Returns the normalised value BEFORE the affine in `xhat`
// (needed by the backward pass) and y = xhat * w + b.  Positions >= dim produce 0.
template <int NT>
DEV void layernorm_tile(f4 (&y)[NT], f4 (&xhat)[NT], float& rstd, const f4 (&x)[NT], int dim, const f4 (&w)[NT],
                        const f4 (&b)[NT], int g) {
    const int nt = (dim + 15) >> 4;
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
        if (t < nt) s += (x[t][0] + x[t][1]) + (x[t][2] + x[t][3]);
    const float inv_dim = fast_rcp((float)dim);  // v_rcp_f32 (1 ulp) instead of a ~10-instruction IEEE division per lane
    const float mean = group_sum(s) * inv_dim;
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float d = (t < nt && feat_of(t, j, g) < dim) ? x[t][j] - mean : 0.f;
            xhat[t][j] = d;
            q += d * d;
        }
    }
    rstd = __builtin_amdgcn_rsqf(group_sum(q) * inv_dim + PMT_LN_EPS);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        xhat[t] = xhat[t] * rstd;
        y[t] = xhat[t] * w[t] + b[t];
    }
}

// ---- model-descriptor access helpers (all wave-uniform scalar loads) -------------------------------------------
DEV const float* src_ptr(int src, const float* theta, const float* phi) {
    // >= 0: theta offset; <= -2: phi offset -(src + 2)
    return src >= 0 ? theta + src : phi + (-(src + 2));
}
DEV float* grad_ptr(int src, float* gtheta, float* gphi) { return src >= 0 ? gtheta + src : gphi + (-(src + 2)); }

// ---- stash layout (floats per 16-read tile) ---------------------------------------------------------------------
// slots: [read-MLP op boundaries 1..n-1][x_0 .. x_L][reducer op boundaries 1..n-1], each slot NT*256 floats
#ifndef PMT_STASH_Z
#define PMT_STASH_Z 1  // 1: the blocks' z in the stash; 0: the backward recomputes it.  Measured again in round 3, when the training forward had
                       // turned out to be bound by its stash WRITES (3.0 GB per 65 536-set launch at ~3.5 TB/s) and the layers'
                       // recomputation had dropped to three MFMAs per product: without z (22 % of the stash) the forward gains 40 us,
                       // but the gate multiplies by z, and recomputed on two bf16 pieces it puts the gradients 5e-5 from fp64 (7e-6
                       // with z stashed); recomputed on three pieces the backward pays 70 us.  z stays in the stash.
#endif
// stash slots per tile: the inputs of the read MLP's ops 1.., xhat_0 .. xhat_{L-1} and x_L, the inputs of the reducer's ops 1..,
// then every block's z (after SELU; half a slot used)
DEV int stash_num_slots(const PmtModel* M) { return (M->read_mlp.n_ops - 1) + (M->num_blocks + 1) + (M->reducer.n_ops - 1) + M->num_blocks; }
#define PMT_SLOT_FLOATS (PMT_NT * 256)

#ifndef PMT_STASH_NT
#define PMT_STASH_NT 1
#endif
#ifndef PMT_STASH_EXPERIMENT
#define PMT_STASH_EXPERIMENT 0
#endif
template <int NT>
DEV void stash_store(float* __restrict__ base, const f4 (&v)[NT]) {
    if (PMT_STASH_EXPERIMENT == 1) return;  // development knock-out (timing only): no stash stores at all
    f4* p = reinterpret_cast<f4*>(base) + (pmt_tid() & 63);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (PMT_STASH_NT) __builtin_nontemporal_store(v[t], p + t * 64);  // written once, read by the backward long after
        else p[t * 64] = v[t];
    }
}
template <int NT>
DEV void stash_load(const float* __restrict__ base, f4 (&v)[NT]) {
    const f4* p = reinterpret_cast<const f4*>(base) + (pmt_tid() & 63);
#pragma unroll
    for (int t = 0; t < NT; ++t) v[t] = p[t * 64];
}

// ---- packed read rows ------------------------------------------------------------------------------------------------
// One row of the reference's packed read layout with five quantile bytes (data/batch.py:51-56, data/datum.py:35: seven
// MSB-first bit bytes, then (uint8 + 128) / 32 with the uint8 wrap of plain_text_data.py:510-511) = 12 bytes = F 61: three
// dword loads per row, after which each of the lane's 16 features (16 t + 4 j + g) is a bit field of a register -- instead of
// up to ten byte loads per row and lane.
DEV void decode_packed12(f4 (&x)[4], const unsigned char* __restrict__ row, int g) {
    const unsigned* p = reinterpret_cast<const unsigned*>(row);
    const unsigned d0 = p[0], d1 = p[1], d2 = p[2];
    const unsigned long long hi = ((unsigned long long)d2 << 32) | d1;  // bytes 4 .. 11
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (16 * t + 4 * j + 3 < 56) {  // bit features: byte 2 t + (j >> 1), bit 7 - (4 (j & 1) + g)
                const unsigned d = (t >> 1) ? d1 : d0;
                const int sh = 8 * ((2 * t + (j >> 1)) & 3) + 7 - 4 * (j & 1) - g;
                x[t][j] = (float)((d >> sh) & 1u);
            } else {                        // quantile features 56 .. 60: byte 7 + idx
                const int idx = 4 * (j - 2) + g;
                const unsigned u = (unsigned)(hi >> (8 * (3 + (idx < 5 ? idx : 0)))) & 0xFFu;
                x[t][j] = idx < 5 ? (float)((u + 128u) & 0xFFu) * (1.0f / 32.0f) : 0.f;
            }
        }
}

// ---- group geometry ----------------------------------------------------------------------------------------------
struct GroupGeom {
    int v0, nsets;           // first variant, number of sets
    int ref_base, nref;      // first ref row (in the ref region) and ref rows of the group
    int alt_base, nalt;      // first alt row (absolute row = total_ref + alt_base) and alt rows
    int total_ref;
    int tiles_ref, tiles_alt, ntiles;
    int tile_begin, tile_count;  // this wave's contiguous tile range
    int side;                    // the side (0 ref / 1 alt) of ALL of this wave's tiles
    int wr;                      // waves [0, wr) hold the ref tiles, waves [wr, PMT_WAVES) the alt tiles
};

// A group fits a workgroup when its ref tiles and its alt tiles can be dealt to disjoint sets of waves, PMT_RT per wave
// (the planner, pmt_plan_groups, packs with the same rule).
DEV bool group_fits(int tiles_ref, int tiles_alt) {
    return (tiles_ref + PMT_RT - 1) / PMT_RT + (tiles_alt + PMT_RT - 1) / PMT_RT <= PMT_WAVES;
}

DEV GroupGeom group_geometry(const PmtBatch& bt, int group) {
    GroupGeom gg;
    if (bt.group_span != nullptr) {  // explicit row ranges: the group may cover only part of a read set
        const int* sp = bt.group_span + 6 * (size_t)group;
        gg.v0 = sp[0];
        gg.nsets = sp[1] - sp[0];
        gg.ref_base = sp[2];
        gg.nref = sp[3] - sp[2];
        gg.alt_base = sp[4];
        gg.nalt = sp[5] - sp[4];
    } else {
        gg.v0 = bt.group_start[group];
        const int v1 = bt.group_start[group + 1];
        gg.nsets = v1 - gg.v0;
        gg.ref_base = bt.ref_offsets[gg.v0];
        gg.nref = bt.ref_offsets[v1] - gg.ref_base;
        gg.alt_base = bt.alt_offsets[gg.v0];
        gg.nalt = bt.alt_offsets[v1] - gg.alt_base;
    }
    gg.total_ref = bt.ref_offsets[bt.num_variants];
    gg.tiles_ref = (gg.nref + 15) >> 4;
    gg.tiles_alt = (gg.nalt + 15) >> 4;
    gg.ntiles = gg.tiles_ref + gg.tiles_alt;
    // Waves are side-homogeneous: the first wr waves share the ref tiles, the others the alt tiles, each side dealt
    // contiguously and evenly; wr is the proportional share clamped so that no wave gets more than PMT_RT tiles.
    const int wave = uniform((int)(pmt_tid() >> 6));
    const int need_r = (gg.tiles_ref + PMT_RT - 1) / PMT_RT, need_a = (gg.tiles_alt + PMT_RT - 1) / PMT_RT;
    int wr = gg.ntiles > 0 ? (PMT_WAVES * gg.tiles_ref + gg.ntiles / 2) / gg.ntiles : 0;
    wr = min(max(wr, need_r), PMT_WAVES - need_a);
    if (PMT_RT == 1) {  // one tile per wave: PAIRS of waves are side-homogeneous (a pair's two tiles are the 32 reads of one weight-gradient
                        // MFMA, pmt_bwd_device.hpp); the planner's rule ceil(ref / 2) + ceil(alt / 2) <= waves / 2 leaves room for the rounding
        const int pr = (gg.tiles_ref + 1) >> 1, pa = (gg.tiles_alt + 1) >> 1;
        wr = 2 * min(max((wr + 1) >> 1, pr), PMT_WAVES / 2 - pa);
    }
    gg.wr = wr;
    gg.side = wave < wr ? 0 : 1;
    const int nw = gg.side == 0 ? wr : PMT_WAVES - wr, i = gg.side == 0 ? wave : wave - wr;
    const int nt = gg.side == 0 ? gg.tiles_ref : gg.tiles_alt;
    const int q = nw > 0 ? nt / nw : 0, rem = nw > 0 ? nt % nw : 0;
    gg.tile_begin = (gg.side == 0 ? 0 : gg.tiles_ref) + i * q + min(i, rem);
    gg.tile_count = q + (i < rem ? 1 : 0);
    return gg;
}

// per-tile metadata for this lane
struct TileMeta {
    int side;       // 0 = ref, 1 = alt: the wave's side (GroupGeom.side).  Tiles the wave does not have are computed like
                    // any other tile (all lanes are padding) and never stored.
    bool present;   // tile exists in this group (wave-uniform)
    int row;        // global row in the batch's read order (before the optional gather), -1 = padding lane
    int set;        // local set index within the group (0 if padding)
    bool valid;     // this lane holds a real read
};

// s_off: LDS array [2][PMT_GROUP_MAX_SETS + 1] of group-local exclusive offsets per side
DEV TileMeta tile_meta(const GroupGeom& gg, int rt, const int* s_off) {
    TileMeta tm;
    const int r = pmt_tid() & 15;
    tm.present = rt < gg.tile_count;
    tm.side = gg.side;
    if (!tm.present) {
        tm.row = -1; tm.set = 0; tm.valid = false;
        return tm;
    }
    const int tau = gg.tile_begin + rt;
    const int local = (tm.side == 0 ? tau : tau - gg.tiles_ref) * 16 + r;
    const int n_side = tm.side == 0 ? gg.nref : gg.nalt;
    tm.valid = local < n_side;
    tm.row = tm.valid ? (tm.side == 0 ? gg.ref_base + local : gg.total_ref + gg.alt_base + local) : -1;
    // binary search: largest s with off[s] <= local
    const int* off = s_off + tm.side * (PMT_GROUP_MAX_SETS + 1);
    int lo = 0, hi = gg.nsets;  // invariant off[lo] <= local < off[hi] (for valid lanes)
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (off[mid] <= local) lo = mid; else hi = mid;
    }
    tm.set = tm.valid ? lo : 0;
    return tm;
}

// ---- read sets that span several workgroups, joined INSIDE one launch ---------------------------------------------------
// A read set beyond one workgroup's capacity is split over consecutive groups (pmt_plan_groups_split).  Its per-set sums (the
// gated blocks' z2 sums in the forward, their d(gate) sums in the backward: reference gated_mlp.py:236-239, sets/
// ragged_sets.py:144-158) then need every group's part.  The layered launches exchange them through kernel boundaries (L + 1
// launches each way, every activation parked in HBM in between); here the groups of ONE launch exchange them through HBM while
// their activations stay in registers:
//   publish: every group adds its partial sums to the set's global row with device-scope float atomics (they execute at the
//            memory side: nothing to write back), waits until they are acknowledged (vmcnt(0) in every wave, then the workgroup
//            barrier), and only then adds 1 to the set's arrival counter;
//   wait:    one lane per set polls the counter (relaxed device-scope loads, s_sleep in between) until all PmtBatch.set_groups[set]
//            groups have arrived; the complete sums are then read back with returning atomics (adds of 0: memory-side reads, no
//            stale cache line on any XCD).
// Progress: groups are handed out by a device-side TICKET (pmt_join_ticket), so the groups that have started always form a
// prefix 0 .. T-1; a group waits only for groups within a few indices of its own, which have started (or start as soon as any
// lower group finishes), provided a few dozen workgroups can be resident at once.  Every wait is bounded: after ~0.2 s it gives
// up, raises the fault word and carries on with whatever arrived (wrong numbers, no hang); the host checks the word.
struct PmtJoin {
    int on;          // 0: not joined
    int* ticket;     // device, [1]: the next group to hand out (zero at launch)
    int* arrivals;   // device, [B][L]: groups that have published (variant, block) (zero at launch)
    int* fault;      // device, [1]: set to 1 when a wait gave up
};
DEV int pmt_join_ticket(const PmtJoin& j, int* lds_slot) {
    if (pmt_tid() == 0) *lds_slot = __hip_atomic_fetch_add(j.ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int t = uniform(*lds_slot);
    __syncthreads();
    return t;
}
// lds: the group's partial sums [nsets][32]; glob / arrivals: the rows / counters of the group's FIRST set for this block, one
// set apart by set_stride floats / arr_stride ints; expected: PmtBatch.set_groups + v0.  On return `lds` holds the COMPLETE sums
// of every set (the caller's next LDS barrier publishes them to the workgroup).
DEV void pmt_join_sets(const PmtJoin& j, float* lds, float* glob, int set_stride, int* arrivals, int arr_stride, const int* expected, int nsets) {
    const int tid = pmt_tid();
    for (int i = tid; i < nsets * PMT_ZW; i += PMT_THREADS) {
        const float v = lds[i];
        if (v != 0.f) atomicAdd(&glob[(size_t)(i / PMT_ZW) * set_stride + (i % PMT_ZW)], v);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's atomics have been acknowledged ...
    __syncthreads();                                  // ... and every other wave's
    if (tid < nsets) {
        int* a = arrivals + (size_t)tid * arr_stride;
        const int need = expected[tid];
        __hip_atomic_fetch_add(a, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (need > 1) {
            int spins = 0;
            while (__hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1 << 17)) {  // ~0.2 s of polls: a legitimate wait is over within a group's run time (~0.2 ms)
                    __hip_atomic_fetch_or(j.fault, PMT_FAULT_JOIN, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < nsets * PMT_ZW; i += PMT_THREADS)
        if (expected[i / PMT_ZW] > 1)
            lds[i] = __hip_atomic_fetch_add(&glob[(size_t)(i / PMT_ZW) * set_stride + (i % PMT_ZW)], 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- dropout in training (pmt_dropout.hpp: the mask is a function of (seed, linear, row, feature)) --------------------------
// Only the generic instances (ShapeAny read-set kernels, the row kernels) carry it; the host selects them when the model has
// dropout_p > 0 AND the batch brings a seed (train mode).  row[rt]: the batch row of this lane's read (or variant) in tile rt.
struct PmtDrop {
    unsigned on, s0, s1, thresh;
    float scale;     // 1 / (1 - p)
    int row[PMT_RT];
};
DEV PmtDrop drop_setup(const PmtModel* __restrict__ M, unsigned long long seed, int mlp_flag) {
    PmtDrop d;
    const float p = uniform(M->dropout_p);
    d.on = (seed != 0ull && p > 0.f && mlp_flag != 0) ? 1u : 0u;
    d.s0 = (unsigned)seed; d.s1 = (unsigned)(seed >> 32);
    d.thresh = pmt_drop_threshold(p);
    d.scale = 1.0f / (1.0f - p);
#pragma unroll
    for (int rt = 0; rt < PMT_RT; ++rt) d.row[rt] = 0;
    return d;
}
template <int NT>
DEV void drop_apply(const PmtDrop& d, int lin, f4 (&y)[PMT_RT][NT], int g) {
#pragma unroll
    for (int rt = 0; rt < PMT_RT; ++rt) {
        const unsigned key = pmt_drop_row_key(d.s0, d.s1, lin, d.row[rt]);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) y[rt][t][j] = pmt_drop_keep(key, feat_of(t, j, g), d.thresh) ? y[rt][t][j] * d.scale : 0.f;
    }
}

extern "C" int pmt_stash_slots(const PmtModel* m);  // host helper (pmt_host.hip)
extern "C" int pmt_shape_id(const PmtModel* m);     // host: 2 = ShapeP0X (exact widths), 6 = ShapeP0T (exact tiles, 16-bit pipes), 1 = ShapeP0 (exact tiles, fp32 MFMAs: asked for only), 0 = ShapeAny
// A batch that brings a dropout seed to a model with dropout_p > 0 runs an instance that carries the masks: 4 = ShapeP0XD (the
// production shape, one-launch path only), else 0 = the generic instance.
static inline int pmt_shape_for(const PmtModel* m, const PmtBatch* b, bool layered = false) {
    int shape = pmt_shape_id(m);
    if (shape == 6 && layered) shape = 1;  // (split read sets of a tile-exact model: the fp32 tile-exact instances)
    if (!(m->dropout_p > 0.f && b->dropout_seed != 0)) return shape;
    return (shape == 2 && !layered) ? 4 : 0;
}

DEV int frag_floats_dev(const PmtLinear& L) {
    const int h = uniform(L.out_split);
    const int out_v = h > 0 ? PMT_SPLIT0 + h : uniform(L.out_dim);
    return ((out_v + 15) >> 4) * ((uniform(L.in_dim) + 15) >> 4) * 256;
}
