// Fused read-set backward for gfx950.  Same group / tile / register layout as the forward (pmt_device.hpp).  The
// kernel walks the network in reverse; inside each layer group it RECOMPUTES the forward from the stashed layer-group
// input (pmt_forward<TRAIN> wrote those), so only ~0.7 K floats per read cross HBM between the two passes.
//
//   dgrad  dx = W^T dy      : MFMA with the transposed A fragments (wt_frag; in the exact-width instances their bf16 pieces
//                             wtb_frag: two pieces of dy against two of W, three MFMAs -- pmt_bwd_device.hpp, PMT_DGRAD_PIECES);
//                             dy is already the B operand.
//   wgrad  dW += dy x^T     : contraction over READS.  Both operands are written transposed into an LDS stage (reads move
//                             from the lane axis to the MFMA k axis purely by the store addressing) as two bf16 pieces each and
//                             exchanged; each wave then owns distinct 16x16 blocks of dW, contracts them over all reads of the
//                             workgroup (three bf16 MFMAs per 32 reads; fp32 MFMAs in the generic instances) and adds them to
//                             its workgroup's PRIVATE row of partial sums, which pmt_grad_fold_kernel folds into the flat
//                             gradient buffer (without the workspace: global float atomics).  No LDS atomics.
//   set-coupled terms       : per-set sums of d(gate) go through LDS exactly like the forward's z2 sums.
//
// Replaces autograd over reference artifact_model.py:239-297 (misc_utils.py:127 `loss.backward()`).
// Wave shape: 8 waves x 2 read tiles (2 waves per SIMD, 256 VGPRs each).  VALU instructions only address the 256
// architectural VGPRs, so a 512-register "fat wave" (4 x 4) just shuffles values through AGPRs: measured slower.
// PMT_BWD_RT = 1 (experimental build, `make EXTRA=-DPMT_BWD_RT=1`): the same 16-tile groups on SIXTEEN waves of one tile each, 128 registers,
// four waves per SIMD -- twice the waves to hide a phase's latency behind; the weight-gradient exchange pairs neighbouring waves' tiles
// into the 32 reads of one MFMA (16-bit stores into the halves of the dwords the two-tile form writes whole).
#ifndef PMT_BWD_RT
#define PMT_BWD_RT 2
#endif
#if PMT_BWD_RT == 1
#define PMT_WAVES (2 * PMT_GROUP_WAVES)
#define PMT_RT 1
#define PMT_AUX_CAP 160  // (16 slabs: the LDS budget)
#define PMT_BWD_WAVES_PER_SIMD 4
#ifndef PMT_BWD_FRAG_AHEAD
#define PMT_BWD_FRAG_AHEAD 2
#endif
#else
#define PMT_WAVES PMT_GROUP_WAVES
#define PMT_RT 2
#define PMT_BWD_WAVES_PER_SIMD 2
#endif
#ifndef PMT_BWD_PIECES
#define PMT_BWD_PIECES 3  // pieces of the products that keep the forward's precision (the head's recomputation); the input-gradient and
                          // recomputation products of the layers take PMT_DGRAD_PIECES / PMT_RECOMPUTE_PIECES (pmt_bwd_device.hpp)
#endif
#include "permutect_amd.h"
#define PMT_STAGE_PLANES ((16 - 2 * (PMT_MAX_HALF_FFN / 16 - 1)) * PMT_GROUP_WAVES)  // every wave's operands of a 4 + 4 tile linear at once (8 waves x 8 planes x (hi + mid)); a build with two-tile gate halves gives 16 KiB of it to the wider per-set tables
#include "permutect_amd.h"
#define PMT_OPAQUE_TID 1  // the kernel loops over groups (persistent launch): see pmt_tid
#ifndef PMT_BWD_XH4_AT_P3
#define PMT_BWD_XH4_AT_P3 0  // where phase 4's stash read of xhat_l is requested: 0 in phase 4 itself; 1 / 2 in phase 3 (round 2: 2.94 -> 3.04 ms: the
                             // in-order memory counter makes the phase's small loads wait for it); 3 together with z in phase 1, pinned by
                             // scheduling barriers (round 5, on the kernel with run-time debug branches: 2.42 -> 2.39 ms).  Measured again on the
                             // kernel WITHOUT those branches (the scheduler now moves loads across whole phases by itself): 0 / 2: 2.03 - 2.04 ms,
                             // 3: 2.06, 1: 2.09 -- the 32 registers an early request holds cost more than the overlap gives; back to 0
#endif
#ifndef PMT_BWD_PRIO
#define PMT_BWD_PRIO 0
#endif
#ifndef PMT_BWD_FRAG_AHEAD
#define PMT_BWD_FRAG_AHEAD 6
#endif
#define PMT_FRAG_AHEAD PMT_BWD_FRAG_AHEAD  // weight fragments six MFMA groups ahead (2 waves per SIMD do not hide an L2 round trip): 0 -> 1: 3.59 -> 3.52 ms; 1 -> 3: 3.31 -> 3.27 ms; round 4, 3 -> 6: 2.43 -> 2.40 ms (8: the same)
#include "pmt_device.hpp"
#include "pmt_mlp_device.hpp"
#include "pmt_bwd_device.hpp"
// The development switches of PmtBatch.debug_flags[1] (BwdCtx.dbg) are compiled OUT of a production build: the word is a dependent global
// load at the top of every group, ~25 uniform branches on its bits cut the scheduler's regions (one in front of every exchange), and it
// holds a scalar register the kernel spills for.  The trace / profile / knock-out builds (scripts/bwd_trace.py, bwd_ablate.py) turn it on.
#ifndef PMT_BWD_DEBUG
#define PMT_BWD_DEBUG (PMT_BWD_PROF || PMT_BWD_ABLATE)  // (the event log of -DPMT_BWD_TRACE=1 has its own word, debug_flags[2])
#endif

struct BwdShared {
    int off[2][PMT_GROUP_MAX_SETS + 1];
    float gsum[PMT_GROUP_MAX_SETS][2][16 * (PMT_MAX_HALF_FFN / 16)];   // per-set sums of d(gate), current block
    float dmean[PMT_GROUP_MAX_SETS][2][16 * (PMT_MAX_HALF_FFN / 16)];                     // d(m_ref), d(m_alt) already divided by (n + w)
    float dl[PMT_GROUP_MAX_SETS][PMT_MAX_CLUSTERS + 2];         // d(loss)/d(Lambda[b][j]) incl. the logit path
    float aux[PMT_WAVES][PMT_AUX_CAP];                          // small-parameter gradient slabs (BwdCtx.aux)
    int aux_dst[PMT_AUX_CAP];
    // Weight-gradient operand exchange (BwdCtx.stage).  Its last 32 KiB double as two short-lived tables that are never
    // alive while an exchange runs: dfeat (read once, before the first exchange) and dv (filled and written out between the
    // blocks' last exchange and the read MLP's first).
    f4 stage[PMT_STAGE_PLANES * 64];
    float pf_sink[64];                                          // where stash_prefetch's LDS-DMA drops its dwords (never read)
    int ticket;                                                 // joined execution: the group this workgroup drew (pmt_join_ticket)
    typedef float DFeat[2][PMT_MAX_WIDTH];                      // d(loss)/d(set mean) / (n + 1e-4), position order
    typedef float DVar[PMT_MAX_WIDTH];                          // per-set sum of d(x_0) (variant-embedding part)
    static constexpr int ALIAS_F4 = (PMT_STAGE_PLANES * 1024 - (int)sizeof(DFeat) * PMT_GROUP_MAX_SETS) / 16;
    __device__ DFeat* dfeat() { return reinterpret_cast<DFeat*>(&stage[ALIAS_F4]); }
    __device__ DVar* dv() { return reinterpret_cast<DVar*>(&stage[ALIAS_F4]); }
};
static_assert(sizeof(BwdShared::DFeat) * PMT_GROUP_MAX_SETS <= PMT_STAGE_PLANES * 1024, "the aliased tables fit the stage");

static_assert(sizeof(BwdShared) <= 160 * 1024, "LDS budget");

DEV float read_feature_b(const unsigned char* __restrict__ row, int fmt, int f, int F) {
    if (f >= F) return 0.f;
    if (fmt == PMT_READS_PACKED_U8) {
        if (f < 56) return (float)((row[f >> 3] >> (7 - (f & 7))) & 1);
        const unsigned u = row[7 + (f - 56)];
        return (float)((u + 128u) & 0xFFu) * (1.0f / 32.0f);
    } else if (fmt == PMT_READS_F16) {
        return (float)reinterpret_cast<const _Float16*>(row)[f];
    }
    return reinterpret_cast<const float*>(row)[f];
}

// The backward walks a tile's stash slots from the last to the first, each slot an HBM miss (the forward wrote them gigabytes
// ago).  `sink` != nullptr: while slot `slot` is loaded, slot - 1 -- the next one the walk needs -- is touched into L2.
template <int NT>
DEV void load_slot_tiles(const float* const (&stash_tile)[PMT_RT], unsigned mask, int slot, f4 (&v)[PMT_RT][NT], float* sink = nullptr) {
#pragma unroll
    for (int rt = 0; rt < PMT_RT; ++rt) {
        // (a tile the wave does not have reads the group's FIRST tile -- stash_tile[] points there: finite values under a gradient that
        //  is zero in every lane, like the padding reads of a tile it has; no zero fill, no branch around the load)
        stash_load<NT>(stash_tile[rt] + slot * PMT_SLOT_FLOATS, v[rt]);
        if (sink != nullptr && slot > 0 && (mask & (1u << rt))) stash_prefetch(stash_tile[rt] + (slot - 1) * PMT_SLOT_FLOATS, sink);
    }
}

// Layered execution (pmt_backward_layered; read sets split over several workgroups): launch `slice` finishes block
// L - slice from the COMPLETE per-set sums of d(gate) and starts block L - slice - 1; the sums accumulate in HBM, the
// running gradient and the half-finished block's per-read state rest in scratch between launches.
#ifndef PMT_BWD_Z_F16
#define PMT_BWD_Z_F16 1
#endif
#define PMT_BWD_PARK_TILES (5 * PMT_HT + 1)  // z1, z2, z2hat, d(gate), d(u) (PMT_HT tiles each) and the rstd of LayerNorm(h)
struct PmtBwdLayered {
    int slice;
    float* dy_scratch;  // [total_tiles][PMT_SLOT_FLOATS] running gradient
    float* park;        // [total_tiles][PMT_BWD_PARK_TILES][256]: z1, z2 (after SELU), z2hat, d(gate), d(u), rstd of LayerNorm(h)
    float* gsum_g;      // [B][L][PMT_ZW] per-set sums of d(gate)
    PmtJoin join;       // join.on: ONE launch; the groups of a split read set join their d(gate) sums through HBM (pmt_device.hpp)
};

// One group (blockIdx.x of the one-group-per-workgroup launch; `grp` of the persistent one).  priv: this workgroup's private
// row of weight-gradient partial sums, biased so that a PmtLinear.emit_tab offset indexes it directly; nullptr = atomics.
template <typename S, bool LAYERED, bool PRIV>
DEV void backward_group(
    const PmtModel* __restrict__ M, const float* __restrict__ theta, const float* __restrict__ phi,
    const float* __restrict__ packed, const PmtBatch& bt, const PmtOutputs& out, const PmtOutputGrads& dout, const float* __restrict__ stash,
    const float* __restrict__ zsum_stash, const float* __restrict__ rstd_stash, float* __restrict__ gtheta, float* __restrict__ gphi,
    float* __restrict__ gvar, const PmtBwdLayered& lay, const int grp, BwdShared& sh, float* __restrict__ priv) {
    constexpr int NTF = S::NTF, NTR = S::NTR, NTD = S::NTD, NTE = S::NTE;
    constexpr bool EX = S::EXACT;
    // Small-parameter pushes of a gated block between two exchanges (an exchange empties the slab): the set coupling's rows, LayerNorm(h)'s
    // two vectors and the gate scalars in front of the proj1 exchange; LayerNorm(D)'s two vectors behind it (+ 8: a skip block's alpha
    // left by the MLP before).  When both fit, the block's pushes skip the capacity check (pmt_bwd_device.hpp: CHECK).
    constexpr bool AUXCHK = !(EX && 3 * 16 * PMT_HT + 8 + 8 <= PMT_AUX_CAP && 2 * 16 * NTD + 8 <= PMT_AUX_CAP);
    // Pieces of the activations / gradients in the backward's products: BFB (three, like the forward) where PMT_DG / PMT_RC do not
    // apply; the layers' input-gradient and recomputation products run on two (pmt_bwd_device.hpp, with the fp64 yardstick).
    // BFP: the pieces, for the products called from here; BFB: the same + the PRIV bit, for everything that reaches an exchange.
    constexpr int BFP = S::BF16 == 3 ? PMT_BWD_PIECES : S::BF16;
    constexpr int BFB = BFP | ((PRIV && S::BF16 != 0) ? PMT_BF_PRIV : 0);
    static_assert(!PRIV || S::BF16 != 0, "private rows: the bf16-exchange instances only (priv_kernel picks)");
    static_assert(EX || (NTF == NTD && NTR == NTD && NTE == NTD), "the generic shape keeps one array width");
    const int tid = pmt_tid(), lane = tid & 63, g = lane >> 4, wave = uniform((int)(tid >> 6));
    const GroupGeom gg = group_geometry(bt, grp);
    const int side = gg.side;
    const int D = S::DIM_D ? S::DIM_D : uniform(M->d_model), E = S::DIM_E ? S::DIM_E : uniform(M->feature_dim), K = uniform(M->num_clusters);
    const int Er = S::DIM_R ? S::DIM_R : uniform(M->read_embed_dim), Ev = uniform(M->variant_embed_dim);
    const int h = S::DIM_H ? S::DIM_H : (uniform(M->d_ffn) >> 1), L = uniform(M->num_blocks), F = S::DIM_F ? S::DIM_F : uniform(M->num_read_features);
    const int nte = (E + 15) >> 4;

    // ---- setup ------------------------------------------------------------------------------------------------------
    for (int i = tid; i <= gg.nsets; i += PMT_THREADS) {
        sh.off[0][i] = bt.ref_offsets[gg.v0 + i] - gg.ref_base;
        sh.off[1][i] = bt.alt_offsets[gg.v0 + i] - gg.alt_base;
    }
    for (int i = tid; i < PMT_GROUP_MAX_SETS * PMT_ZW; i += PMT_THREADS) (&sh.gsum[0][0][0])[i] = 0.f;
    lds_barrier();
    // per-set upstream gradients (reference feature_clustering.py:121-135 differentiated)
    for (int i = tid; i < gg.nsets; i += PMT_THREADS) {
        const int b = gg.v0 + i;
        const float* lk = out.logits_bk + (size_t)b * (K + 2);
        const float dlogit = dout.d_logits_b ? dout.d_logits_b[b] : 0.f;
        const float capped = out.logits_b[b];
        const float th = capped / PMT_MAX_LOGIT_F;
        const float draw = dlogit * (1.f - th * th);
        float mx = -INFINITY;
        for (int k = 0; k < K; ++k) mx = fmaxf(mx, lk[2 + k]);
        float se = 0.f;
        for (int k = 0; k < K; ++k) se += expf(lk[2 + k] - mx);
        const float* dk = dout.d_logits_bk ? dout.d_logits_bk + (size_t)b * (K + 2) : nullptr;
        sh.dl[i][0] = (dk ? dk[0] : 0.f) - draw;
        sh.dl[i][1] = dk ? dk[1] : 0.f;
        for (int k = 0; k < K; ++k) sh.dl[i][2 + k] = (dk ? dk[2 + k] : 0.f) + draw * expf(lk[2 + k] - mx) / se;
    }
    for (int i = tid; i < gg.nsets * 2 * PMT_MAX_WIDTH; i += PMT_THREADS) {
        const int set = i / (2 * PMT_MAX_WIDTH), rem = i - set * 2 * PMT_MAX_WIDTH, s = rem / PMT_MAX_WIDTH, p = rem - s * PMT_MAX_WIDTH;
        const int f = pos_to_feat(p);
        const float* src = s == 1 ? dout.d_features_be : dout.d_ref_features_be;
        const float n = (float)(sh.off[s][set + 1] - sh.off[s][set]);
        sh.dfeat()[set][s][p] = (src && f < E) ? src[(size_t)(gg.v0 + set) * E + f] / (n + 1e-4f) : 0.f;
    }
    lds_barrier();

    TileMeta tm[PMT_RT];
    unsigned mask_all = 0;
    const float* stash_tile[PMT_RT];
    int tile_of[PMT_RT];
    const int nslots = stash_num_slots(M);
#pragma unroll
    for (int rt = 0; rt < PMT_RT; ++rt) {
        tm[rt] = tile_meta(gg, rt, &sh.off[0][0]);
        if (tm[rt].present) mask_all |= 1u << rt;
        tile_of[rt] = bt.group_tile_base[grp] + (tm[rt].present ? gg.tile_begin + rt : 0);  // (an absent tile: the group's first, see load_slot_tiles)
        stash_tile[rt] = stash + (size_t)tile_of[rt] * (size_t)(nslots * PMT_SLOT_FLOATS);
        if (PMT_BWD_DEBUG && bt.debug_flags && (uniform(bt.debug_flags[1]) & 256)) stash_tile[rt] = stash + (size_t)rt * (size_t)(nslots * PMT_SLOT_FLOATS);  // timing experiment: every read hits L2
    }
    BwdCtx c{M, theta, phi, packed, gtheta, gphi, &sh.stage[0], &sh.aux[0][0], &sh.aux_dst[0], g, mask_all,
             gg.tile_begin, gg.ntiles, gg.tiles_ref, 0,
             (PMT_BWD_DEBUG && bt.debug_flags) ? uniform(bt.debug_flags[1]) : 0,
             (PMT_BWD_DEBUG && bt.debug_flags) ? reinterpret_cast<unsigned long long*>(bt.debug_flags + 8) : nullptr};
    c.wr = gg.wr;
    c.priv = priv;
    if (PMT_BWD_TRACE && bt.debug_flags && uniform(bt.debug_flags[2]) == grp + 1) c.trace = bt.debug_flags + 64 + wave * 512;
    trace_ev(c, 1);
    c.wbase = stage_wbase(lane) + (PMT_RT == 1 ? 2 * (wave & 1) : 0);  // (one tile per wave: the odd wave of a pair writes the upper halves)
    c.rbase = stage_rbase(lane);
    c.pf_sink = (c.dbg & 64) ? &sh.pf_sink[0] : nullptr;  // stash prefetch: OFF (measured slower, see DESIGN)
    const unsigned long long t_kernel0 = prof_now();
    // a read set split over several groups is OWNED by the group that holds its first alt read: per-set terms are added once
    auto owns = [&](int set) { return !LAYERED || (sh.off[1][set] >= 0 && sh.off[1][set] < gg.nalt); };
    const size_t tile_global = (size_t)(bt.group_tile_base[grp] + gg.tile_begin);
    if (wave == 0 && !(c.dbg & 16) && !(LAYERED && lay.slice > 0)) {  // d(log cluster weights): summed over the sets of the group
        const int k = lane & 15;
        float a = 0.f;
        if (k < K)
            for (int set = g; set < gg.nsets; set += 4)
                if (owns(set)) a += sh.dl[set][2 + k];
        a = group_sum(a);
        if (g == 0 && k < K) atomicAdd(&gphi[M->head.log_w_k_phi + k], a);
    }
    const int n_read_ops = uniform(M->read_mlp.n_ops), n_red_ops = uniform(M->reducer.n_ops);
    const int slot_x0 = n_read_ops - 1, slot_red = slot_x0 + L + 1;
    const int slot_z0 = slot_red + (n_red_ops - 1);  // the blocks' z (stash_num_slots)
    const int slot_last_in = n_red_ops > 1 ? slot_red + (n_red_ops - 2) : slot_x0 + L;  // input of the last reducer op
    const PmtOp& red_last = M->reducer.ops[n_red_ops - 1];

    PmtDrop drop;  // (the instances that carry dropout: the masks of the step's seed, regenerated -- pmt_dropout.hpp)
    if constexpr (S::DROP) {
        drop = drop_setup(M, bt.dropout_seed, uniform(M->reducer.dropout));
#pragma unroll
        for (int rt = 0; rt < PMT_RT; ++rt) drop.row[rt] = tm[rt].row;
        c.drop = &drop;
    }
    // ---- recompute the tail of the forward: last reducer op, translation, rotation -> a ----------------------------
    f4 dy[PMT_RT][NTD];  // running gradient (d_model wide)
    if (LAYERED && lay.slice > 0) {
#pragma unroll
        for (int rt = 0; rt < PMT_RT; ++rt) {
#pragma unroll
            for (int t = 0; t < NTD; ++t) dy[rt][t] = f4{0.f, 0.f, 0.f, 0.f};
            if (mask_all & (1u << rt)) stash_load<NTD>(lay.dy_scratch + (tile_global + rt) * PMT_SLOT_FLOATS, dy[rt]);
        }
    }
    if (!(LAYERED && lay.slice > 0)) {
        f4 e[PMT_RT][NTE];  // reducer output, then + translation (the rotation's input)
        {
            f4 r[PMT_RT][NTD];
            load_slot_tiles<NTD>(stash_tile, mask_all, slot_last_in, r, c.pf_sink);
            if constexpr (EX) {
                const PmtLinear& Lr = M->lin[uniform(red_last.lin[0])];
                init_bias<NTE>(e, uniform(Lr.b_pvec) >= 0 ? packed + uniform(Lr.b_pvec) : nullptr, E, g);
                if constexpr (S::BF16) linear_acc_bf16<NTD, NTE, false, BFP>(e, r, packed + uniform(Lr.wb_frag));
                else linear_acc<NTD, NTE, false, true, S::DIM_D>(e, r, packed + uniform(Lr.w_frag), D, E);
                if constexpr (S::DROP) {
                    if (drop.on != 0) drop_apply<NTE>(drop, uniform(red_last.lin[0]), e, g);
                }
                if (uniform(red_last.selu_after) != 0) {
#pragma unroll
                    for (int rt = 0; rt < PMT_RT; ++rt)
#pragma unroll
                        for (int t = 0; t < NTE; ++t) e[rt][t] = selu4(e[rt][t]);
                }
            } else {  // forward of the last reducer op only, whatever its kind (with this step's dropout masks, if any)
                float* const no_stash[PMT_RT] = {};
                int no_slot = 0;
                run_mlp<false, NTD, false>(M, M->reducer, r, theta, g, 0u, no_stash, no_slot, 0, packed, n_red_ops - 1, n_red_ops, &drop);
#pragma unroll
                for (int rt = 0; rt < PMT_RT; ++rt)
#pragma unroll
                    for (int t = 0; t < NTE; ++t) e[rt][t] = r[rt][t < NTD ? t : 0];
            }
        }
        const PmtLinear& R = M->lin[uniform(M->rotation_lin)];
        f4 a[PMT_RT][NTE];
#pragma unroll
        for (int t = 0; t < NTE; ++t) {
            const f4 tr = load_pvec(packed + uniform(M->translation_pvec), t, g);
#pragma unroll
            for (int rt = 0; rt < PMT_RT; ++rt) {
                e[rt][t] = e[rt][t] + tr;  // e + t, the rotation's input
                a[rt][t] = f4{0.f, 0.f, 0.f, 0.f};
            }
        }
        if constexpr (S::BF16) linear_acc_bf16<NTE, NTE, false, BFP>(a, e, packed + uniform(R.wb_frag));
        else linear_acc<NTE, NTE, false, EX, S::DIM_E>(a, e, packed + uniform(R.w_frag), E, E);

        // ---- head backward (alt reads) + set-mean gradients -> d(a) in da ------------------------------------------
        f4 da[PMT_RT][NTE];
        f4 sig[NTE], dsig[NTE];
#pragma unroll
        for (int t = 0; t < NTE; ++t) {
            sig[t] = f4{1.f, 1.f, 1.f, 1.f};
            dsig[t] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (t < nte && feat_of(t, j, g) < E) sig[t][j] = phi[uniform(M->head.stdev_e_phi) + feat_of(t, j, g)];
        }
#pragma unroll
        for (int rt = 0; rt < PMT_RT; ++rt) {
            const int set = tm[rt].set;
#pragma unroll
            for (int t = 0; t < NTE; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    da[rt][t][j] = (tm[rt].valid && t < nte) ? sh.dfeat()[set][side][16 * t + 4 * g + j] : 0.f;
        }
        for (int k = -1; k < K; ++k) {  // k = -1: the two diagonal Gaussians; k >= 0: artifact cluster k
            f4 v[NTE], dvk[NTE];
            float d_tau = 0.f, d_mu = 0.f, d_lam = 0.f, d_sg = 0.f;
            float tau = 1.f, mu = 0.f, sg = 1.f, lam = 1.f;
            if (k >= 0) {
                tau = uniform(phi[uniform(M->head.art_stdev_k_phi) + k]);
                mu = uniform(theta[uniform(M->head.mu_k_src) + k]);
                sg = uniform(phi[uniform(M->head.sigma_k_phi) + k]);
                lam = uniform(phi[uniform(M->head.lambda_k_phi) + k]);
            }
#pragma unroll
            for (int t = 0; t < NTE; ++t) {
                v[t] = f4{0.f, 0.f, 0.f, 0.f};
                dvk[t] = f4{0.f, 0.f, 0.f, 0.f};
                if (k >= 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (t < nte && feat_of(t, j, g) < E) v[t][j] = phi[uniform(M->head.dirs_ke_phi) + k * E + feat_of(t, j, g)];
                }
            }
            if (side == 1) {
                // (wave-uniform reciprocals, once per cluster: every division below was an IEEE sequence of ~10 instructions)
                const float inv_s2 = fast_rcp(1.4142135623730951f * sg), inv_tau = fast_rcp(tau), inv_tau2 = inv_tau * inv_tau;
                const float var = sg * sg, inv_lam = fast_rcp(lam), inv_2var = fast_rcp(1.4142135623730951f * var);
#pragma unroll
                for (int rt = 0; rt < PMT_RT; ++rt) {
                    const int set = tm[rt].set;
                    const bool ok = tm[rt].valid;
                    if (k < 0) {
                        const float g0 = ok ? sh.dl[set][0] : 0.f, g1 = ok ? sh.dl[set][1] : 0.f;
#pragma unroll
                        for (int t = 0; t < NTE; ++t)
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                if (t < nte && feat_of(t, j, g) < E) {
                                    const float av = a[rt][t][j], inv_sv = fast_rcp(sig[t][j]), inv2 = inv_sv * inv_sv;
                                    da[rt][t][j] += -(g0 + 0.25f * g1) * av * inv2;
                                    dsig[t][j] += g0 * (-inv_sv + av * av * inv2 * inv_sv) + g1 * (-inv_sv + 0.25f * av * av * inv2 * inv_sv);
                                }
                        continue;
                    }
                    const float G = ok ? sh.dl[set][2 + k] : 0.f;
                    float p = 0.f;
#pragma unroll
                    for (int t = 0; t < NTE; ++t) p += (a[rt][t][0] * v[t][0] + a[rt][t][1] * v[t][1]) + (a[rt][t][2] * v[t][2] + a[rt][t][3] * v[t][3]);
                    p = group_sum(p);
                    float o2 = 0.f, eu = 0.f;
#pragma unroll
                    for (int t = 0; t < NTE; ++t)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float ee = a[rt][t][j] - p * v[t][j];
                            o2 += ee * ee;
                            eu += ee * v[t][j];
                        }
                    o2 = group_sum(o2);
                    eu = group_sum(eu);
                    const float zz = (mu + lam * var - p) * inv_s2;
                    const float Lp = dlogerfc_dev(zz);
                    const float demg_dp = -Lp * inv_s2 - lam;
                    const float c_o = -0.5f * G * inv_tau2;  // d(loss)/d(o2)
#pragma unroll
                    for (int t = 0; t < NTE; ++t)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (t < nte && feat_of(t, j, g) < E) {
                                const float ee = a[rt][t][j] - p * v[t][j];
                                da[rt][t][j] += c_o * (2.f * ee - 2.f * eu * v[t][j]) + G * demg_dp * v[t][j];
                                dvk[t][j] += c_o * (-2.f * eu * a[rt][t][j] - 2.f * p * ee) + G * demg_dp * a[rt][t][j];
                            }
                    if (g == 0) {  // scalar parameter gradients: once per read
                        d_tau += G * (-(float)(E - 1) * inv_tau + o2 * inv_tau2 * inv_tau);
                        d_mu += G * (Lp * inv_s2 + lam);
                        d_lam += G * (inv_lam + Lp * sg * 0.7071067811865476f + mu + lam * var - p);
                        d_sg += G * (Lp * (-(mu - p) * inv_2var + lam * 0.7071067811865476f) + lam * lam * sg);
                    }
                }
            }
            if (k >= 0) {  // every wave pushes (zeros from a ref wave): the slab layout is workgroup-uniform
                aux_push_vec_x<NTE, EX>(c, enc_phi(uniform(M->head.dirs_ke_phi) + k * E), dvk, E);
                const int enc4[4] = {enc_phi(uniform(M->head.art_stdev_k_phi) + k), uniform(M->head.mu_k_src) + k,
                                     enc_phi(uniform(M->head.lambda_k_phi) + k), enc_phi(uniform(M->head.sigma_k_phi) + k)};
                const float val4[4] = {d_tau, d_mu, d_lam, d_sg};
                aux_push_scalars<4>(c, enc4, val4);
            }
        }
        aux_push_vec_x<NTE, EX>(c, enc_phi(uniform(M->head.stdev_e_phi)), dsig, E);
        prof_add(c, 4, t_kernel0);
        trace_ev(c, 14);
        unsigned long long t_rot = prof_now();
        // ---- rotation + translation backward: a = Q (e + t) ------------------------------------------------------------
        linear_wgrad<NTE, NTE, BFB>(c, R, da, e);
        f4 de[PMT_RT][NTE];
        init_bias<NTE>(de, nullptr, E, g);
        if constexpr (S::BF16) linear_acc_bf16<NTE, NTE, false, PMT_DG(BFB)>(de, da, packed + uniform(R.wtb_frag));
        else linear_acc<NTE, NTE, false, EX, S::DIM_E>(de, da, packed + uniform(R.wt_frag), E, E);
        f4 dt[NTE];
#pragma unroll
        for (int t = 0; t < NTE; ++t) {
            dt[t] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int rt = 0; rt < PMT_RT; ++rt) dt[t] = dt[t] + de[rt][t];
        }
        aux_push_vec_x<NTE, EX>(c, uniform(M->translation_src), dt, E);
        prof_add(c, 5, t_rot);
        trace_ev(c, 15);
        // ---- last reducer op ------------------------------------------------------------------------------------------
        if constexpr (EX) {
            f4 r[PMT_RT][NTD];
            load_slot_tiles<NTD>(stash_tile, mask_all, slot_last_in, r);
            linear_op_backward<NTD, NTE, true, S::DIM_D, S::DIM_E, BFB, S::DROP>(c, red_last, de, r, dy, true);
        } else {
#pragma unroll
            for (int rt = 0; rt < PMT_RT; ++rt)
#pragma unroll
                for (int t = 0; t < NTD; ++t) dy[rt][t] = de[rt][t < NTE ? t : 0];
        }
    }
    unsigned long long t_ph = prof_now();

    // ---- reducer backward ---------------------------------------------------------------------------------------------
    if (!(LAYERED && lay.slice > 0))
    mlp_backward<NTD, EX, S::DIM_D, BFB, S::DROP>(c, M->reducer, dy, true,
                          [&](int op, f4 (&x)[PMT_RT][NTD]) { load_slot_tiles<NTD>(stash_tile, mask_all, op == 0 ? slot_x0 + L : slot_red + op - 1, x, c.pf_sink); },
                          0, EX ? n_red_ops - 1 : n_red_ops);

    prof_add(c, 6, t_ph);
    trace_ev(c, 16);
    // ---- gated blocks backward -----------------------------------------------------------------------------------------
    // xhat_l (the normalised block input, stashed by the forward with one rstd per read) is needed three times per block: by
    // phase 1 and twice by phase 4, which keeps its copy for both uses.  (Requesting either read a phase early -- under the
    // LDS-only exchange before it -- was measured SLOWER, 3.48 -> 3.55 / 3.96 ms: the 32 registers held across the phases
    // in between spill, and every small parameter load behind the request waits for it, the memory counter being in order.)
    auto load_xhat = [&](f4 (&xh)[PMT_RT][NTD], int l) {
#pragma unroll
        for (int rt = 0; rt < PMT_RT; ++rt) {
            stash_load<NTD>(stash_tile[rt] + (slot_x0 + l) * PMT_SLOT_FLOATS, xh[rt]);
        }
    };
    const bool joined = LAYERED && lay.join.on != 0;
    const int l_first = (c.dbg & 4) ? -1 : ((LAYERED && lay.slice > 0) ? L - lay.slice : L - 1);
    for (int l = l_first; l >= 0; --l) {
        const bool first_half = !LAYERED || joined || l == L - 1 - lay.slice;  // phases 1-2 (up to the per-set sums of d(gate))
        // Register discipline (this loop body used to spill thousands of VGPRs): nothing of width D except the running
        // gradient dy stays live across phases.  xhat_l = the normalised x_l (stashed by the forward together with one
        // rstd per read) is re-read from the stash (L2/HBM, 4 KB per tile) each of the three times it is needed; z2 / gate
        // are recomputed from z2hat.
        const PmtBlock& B = M->blocks[l];
        const PmtLinear& P1 = M->lin[uniform(B.proj1[side])];
        const PmtLinear& P2 = M->lin[uniform(B.proj2[side])];
        const float* lw_p = packed + uniform(B.norm_w_pvec);
        const float* lb_p = packed + uniform(B.norm_b_pvec);
        // n[rt] = LayerNorm_D(x_l[rt]) = xhat * w + b for every tile of this wave (absent tiles: xhat = 0)
        auto affine_n = [&](f4 (&n)[PMT_RT][NTD], const f4 (&xh)[PMT_RT][NTD]) {
#pragma unroll
            for (int t = 0; t < NTD; ++t) {
                const f4 lw = load_pvec(lw_p, t, g), lb = load_pvec(lb_p, t, g);
#pragma unroll
                for (int rt = 0; rt < PMT_RT; ++rt) n[rt][t] = xh[rt][t] * lw + lb;
            }
        };
        f4 xh4[PMT_RT][NTD];  // phase 4's xhat_l and rstd
        float rs4[PMT_RT];
        bool xh4_requested = false;
        auto load_xh4 = [&]() {
            load_xhat(xh4, l);
#pragma unroll
            for (int rt = 0; rt < PMT_RT; ++rt)
                rs4[rt] = rstd_stash[((size_t)tile_of[rt] * L + l) * 16 + (lane & 15)];
        };
        t_ph = prof_now();
        // ---- phase 1: z = selu(W1 n + b1).  The exact-width instances take it from the stash, where the forward left both halves
        // after their SELU (2 KiB per tile): recomputing it from xhat_l -- a stash read twice the size, a LayerNorm affine and the
        // 60 -> 2 x 10 projection with its splits -- was 6 % of the kernel.  The generic instance still recomputes.
        constexpr int HT = PMT_HT;  // tiles per half of the hidden layer: z[..][0 .. HT) = z1, z[..][HT .. 2 HT) = z2 (after its SELU)
        constexpr int PARK = 5 * HT + 1;  // tiles a layered launch parks per read tile (PMT_BWD_PARK_TILES)
        f4 z[PMT_RT][2 * HT];
        if (first_half) {
            if constexpr (EX && PMT_STASH_Z) {
#pragma unroll
                for (int rt = 0; rt < PMT_RT; ++rt) {
                    stash_load<2 * HT>(stash_tile[rt] + (size_t)(slot_z0 + l) * PMT_SLOT_FLOATS, z[rt]);
                }
                // (3: phase 4's xhat_l requested TOGETHER with z -- the two HBM round trips of a block overlap instead of following each
                //  other; the 32 registers it holds through phases 2 - 3 are paid for with a shallower fragment prefetch)
                if (PMT_BWD_XH4_AT_P3 == 3) {
                    __builtin_amdgcn_sched_barrier(0);  // (the scheduler otherwise sinks the request down to its use)
                    load_xh4();
                    __builtin_amdgcn_sched_barrier(0);
                    xh4_requested = true;
                }
            } else {
                f4 n[PMT_RT][NTD], xq[PMT_RT][NTD];
                load_xhat(xq, l);
                affine_n(n, xq);
#pragma unroll
                for (int t = 0; t < 2 * HT; ++t) {
                    const f4 b_t = load_pvec(packed + uniform(P1.b_pvec), t, g);
#pragma unroll
                    for (int rt = 0; rt < PMT_RT; ++rt) z[rt][t] = b_t;
                }
                // (the gate multiplies by this: fp32-equivalent products -- three f16 MFMAs on two-piece splits like the forward's, or
                //  with PMT_BWD_Z_F16 = 0 the six bf16 MFMAs on three-piece splits of round 3)
                if constexpr (S::BF16 && PMT_BWD_Z_F16) linear_acc_f16<NTD, 2 * HT, false>(z, n, packed + uniform(P1.wh_frag));
                else if constexpr (S::BF16) linear_acc_bf16<NTD, 2 * HT, false, BFP>(z, n, packed + uniform(P1.wb_frag));
                else linear_acc<NTD, 2 * HT, false, EX, S::DIM_D>(z, n, packed + uniform(P1.w_frag), D, PMT_SPLIT0 + h);
#pragma unroll
                for (int rt = 0; rt < PMT_RT; ++rt)
#pragma unroll
                    for (int t = 0; t < 2 * HT; ++t) z[rt][t] = selu4(z[rt][t]);
            }
        }
        f4 sw[HT], sb[HT], rho[HT];
#pragma unroll
        for (int t = 0; t < HT; ++t) {
            sw[t] = load_pvec(packed + uniform(B.sgu_norm_w_pvec), t, g);
            sb[t] = load_pvec(packed + uniform(B.sgu_norm_b_pvec), t, g);
        }
        const float w = uniform(phi[uniform(B.reg_weight_phi)]) + 0.25f;
#pragma unroll
        for (int t = 0; t < HT; ++t) rho[t] = load_pvec(packed + uniform(B.ref_reg_pvec), t, g);
        const float alpha = uniform(theta[uniform(B.alpha_src[side])]), beta = uniform(theta[uniform(B.beta_src[side])]);
        const float beta_ref = uniform(theta[uniform(B.beta_src[0])]), beta_alt = uniform(theta[uniform(B.beta_src[1])]);
        const float gamma = uniform(theta[uniform(B.gamma_src)]);
        // gate of one tile from z2hat (recomputed wherever it is needed)
        auto gate_of = [&](int rt, int t, f4 z2hat_rt, f4& z2_out, f4& m_ref, f4& m_alt) -> f4 {  // tile t of the half
            const int set = tm[rt].set;
            const float n_ref = (float)(sh.off[0][set + 1] - sh.off[0][set]);
            const float n_alt = (float)(sh.off[1][set + 1] - sh.off[1][set]);
            const float* zs = zsum_stash + ((size_t)(gg.v0 + set) * L + l) * PMT_ZW + 16 * t;
            m_ref = (*reinterpret_cast<const f4*>(zs + 4 * g) + w * rho[t]) * fast_rcp(n_ref + w);
            m_alt = *reinterpret_cast<const f4*>(zs + 16 * HT + 4 * g) * fast_rcp(n_alt + 1e-4f);
            z2_out = z2hat_rt * sw[t] + sb[t];
            const f4 gt = z2_out * alpha + 1.0f;
            return side == 0 ? gt + beta * m_ref : (gt + beta * m_alt) + gamma * m_ref;
        };
        prof_add(c, 8, t_ph);
        trace_ev(c, 18);
        t_ph = prof_now();
        // ---- phase 2: d(u) = W2^T dy, d(gate), per-set sums of d(gate), proj2 weight gradient ------------------------------
        f4 z2hat[PMT_RT][HT], dgate[PMT_RT][HT], du[PMT_RT][HT];
        float rstd2[PMT_RT];
        float d_alpha = 0.f, d_beta = 0.f, d_gamma = 0.f, d_reg_w = 0.f;
        // the block's scalar gradients in ONE push (zeros from the waves of the other side: the slab layout is workgroup-uniform)
        auto push_gate_scalars = [&](float extra, int extra_enc) {
            const int enc8[8] = {uniform(B.alpha_src[0]), uniform(B.alpha_src[1]), uniform(B.beta_src[0]), uniform(B.beta_src[1]),
                                 uniform(B.gamma_src), extra_enc, -1, -1};
            const float val8[8] = {side == 0 ? d_alpha : 0.f, side == 1 ? d_alpha : 0.f, side == 0 ? d_beta : 0.f, side == 1 ? d_beta : 0.f,
                                   d_gamma, extra, 0.f, 0.f};
            aux_push_scalars<8, AUXCHK>(c, enc8, val8);
        };
#pragma unroll
        for (int rt = 0; rt < PMT_RT; ++rt)
#pragma unroll
            for (int t = 0; t < HT; ++t) du[rt][t] = f4{0.f, 0.f, 0.f, 0.f};
        if (first_half) {
            if constexpr (S::BF16) linear_acc_bf16<NTD, HT, false, PMT_DG(BFB)>(du, dy, packed + uniform(P2.wtb_frag));
            else linear_acc<NTD, HT, false, EX, S::DIM_D>(du, dy, packed + uniform(P2.wt_frag), D, h);
        }
        if (first_half) {
            f4 u[PMT_RT][HT];
#pragma unroll
            for (int rt = 0; rt < PMT_RT; ++rt) {
                f4 zin[HT], zo[HT];
#pragma unroll
                for (int t = 0; t < HT; ++t) zin[t] = z[rt][HT + t];
                layernorm_tile<HT>(zo, z2hat[rt], rstd2[rt], zin, h, sw, sb, g);
                const int set = tm[rt].set;
                const bool ok = tm[rt].valid;
#pragma unroll
                for (int t = 0; t < HT; ++t) {
                    f4 z2v, m_ref, m_alt;
                    const f4 gt = gate_of(rt, t, z2hat[rt][t], z2v, m_ref, m_alt);
                    u[rt][t] = z[rt][t] * gt;
                    const SegPlan sp = seg_plan(ok ? set : -1);  // per-set sums of d(gate): segmented reduce over the tile's reads
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float dg = (ok && feat_of(t, j, g) < h) ? du[rt][t][j] * z[rt][t][j] : 0.f;
                        dgate[rt][t][j] = dg;
                        d_alpha += dg * z2v[j];
                        d_beta += dg * (side == 0 ? m_ref[j] : m_alt[j]);
                        if (side == 1) d_gamma += dg * m_ref[j];
                    }
                    const f4 sg4 = seg_sum4<!S::BF16>(dgate[rt][t], sp);  // (guarded in the fp32 instances only: pmt_device.hpp, seg_sum)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (sp.last && feat_of(t, j, g) < h) atomicAdd(&sh.gsum[set][side][16 * t + 4 * g + j], sg4[j]);
                }
            }
            prof_add(c, 9, t_ph);
            trace_ev(c, 19);
            t_ph = prof_now();
            // proj2 weight gradients of both sides in one exchange round
            if constexpr (S::BF16 != 0) wgrad_exchange_bf<NTD, HT, 2, BFB>(c, M->lin[uniform(B.proj2[0])], M->lin[uniform(B.proj2[1])], dy, u, 1.0f);
            else wgrad_exchange<NTD, HT, 2>(c, M->lin[uniform(B.proj2[0])], M->lin[uniform(B.proj2[1])], dy, u, 1.0f);
        }
        if (c.dbg & 1) __syncthreads();  // (the exchange's barriers, skipped by that switch, complete gsum)
        if constexpr (LAYERED) {
            float* pk[PMT_RT];
#pragma unroll
            for (int rt = 0; rt < PMT_RT; ++rt) pk[rt] = lay.park + (tile_global + rt) * (PARK * 256);
            if (joined) {  // publish this group's part of the block's d(gate) sums, wait for the other groups of its split sets
                lds_barrier();
                pmt_join_sets(lay.join, &sh.gsum[0][0][0], lay.gsum_g + ((size_t)gg.v0 * L + l) * PMT_ZW, L * PMT_ZW,
                              lay.join.arrivals + (size_t)gg.v0 * L + l, L, bt.set_groups + gg.v0, gg.nsets);
                lds_barrier();
                push_gate_scalars(0.f, -1);
            } else if (first_half) {  // end of this launch: join the global sums, park the per-read state and the running gradient
                for (int i = tid; i < gg.nsets * PMT_ZW; i += PMT_THREADS) {
                    const float v = (&sh.gsum[0][0][0])[i];
                    if (v != 0.f) atomicAdd(&lay.gsum_g[((size_t)(gg.v0 + (i / PMT_ZW)) * L + l) * PMT_ZW + (i % PMT_ZW)], v);
                }
                push_gate_scalars(0.f, -1);
#pragma unroll
                for (int rt = 0; rt < PMT_RT; ++rt)
                    if (mask_all & (1u << rt)) {
                        f4 st[PARK];  // [z1 | z2 | z2hat | d(gate) | d(u): HT tiles each][rstd]
#pragma unroll
                        for (int t = 0; t < HT; ++t) {
                            st[t] = z[rt][t]; st[HT + t] = z[rt][HT + t]; st[2 * HT + t] = z2hat[rt][t]; st[3 * HT + t] = dgate[rt][t]; st[4 * HT + t] = du[rt][t];
                        }
                        st[5 * HT] = f4{rstd2[rt], rstd2[rt], rstd2[rt], rstd2[rt]};
                        stash_store<PARK>(pk[rt], st);
                        stash_store<NTD>(lay.dy_scratch + (tile_global + rt) * PMT_SLOT_FLOATS, dy[rt]);
                    }
                aux_flush(c);
                return;
            } else {
#pragma unroll
            for (int rt = 0; rt < PMT_RT; ++rt) {
                f4 st[PARK];
#pragma unroll
                for (int q = 0; q < PARK; ++q) st[q] = f4{0.f, 0.f, 0.f, 0.f};
                if (mask_all & (1u << rt)) stash_load<PARK>(pk[rt], st);
#pragma unroll
                for (int t = 0; t < HT; ++t) {
                    z[rt][t] = st[t]; z[rt][HT + t] = st[HT + t]; z2hat[rt][t] = st[2 * HT + t]; dgate[rt][t] = st[3 * HT + t]; du[rt][t] = st[4 * HT + t];
                }
                rstd2[rt] = st[5 * HT][0];
            }
            for (int i = tid; i < gg.nsets * PMT_ZW; i += PMT_THREADS)
                (&sh.gsum[0][0][0])[i] = lay.gsum_g[((size_t)(gg.v0 + (i / PMT_ZW)) * L + l) * PMT_ZW + (i % PMT_ZW)];
            lds_barrier();
            }
        }
        prof_add(c, 10, t_ph);
        trace_ev(c, 20);
        t_ph = prof_now();
        // per-set coupling: d(m_ref), d(m_alt), d(ref_regularizer), d(reg_weight)
        // One thread per (set, position); a wave covers 4 sets x 16 positions per pass.  d(ref_regularizer) and
        // d(reg_weight) are sums over the sets: summed over the passes per lane, over the 4 sets of a pass across the
        // lane groups, over the waves and into global memory through the small-parameter slab.
        {
            float a_w = 0.f;
#pragma unroll
            for (int t = 0; t < HT; ++t) {  // (one pass per tile of the half: position 16 t + p)
                const int p = 16 * t + (lane & 15), f = pos_to_feat(p);
                const float rho_f = f < h ? theta[B.ref_reg_src + f] : 0.f;
                float a_rho = 0.f;
                for (int i = tid; i < gg.nsets * 16; i += PMT_THREADS) {
                    const int set = i >> 4;
                    const float n_ref = (float)(sh.off[0][set + 1] - sh.off[0][set]);
                    const float n_alt = (float)(sh.off[1][set + 1] - sh.off[1][set]);
                    const float gr = sh.gsum[set][0][p], ga = sh.gsum[set][1][p];
                    const float dm_ref = beta_ref * gr + gamma * ga, dm_alt = beta_alt * ga;
                    const float inv_ref = fast_rcp(n_ref + w);
                    sh.dmean[set][0][p] = dm_ref * inv_ref;
                    sh.dmean[set][1][p] = dm_alt * fast_rcp(n_alt + 1e-4f);
                    if (f < h && owns(set)) {
                        const float zs = zsum_stash[((size_t)(gg.v0 + set) * L + l) * PMT_ZW + p];
                        const float m_ref = (zs + w * rho_f) * inv_ref;
                        a_rho += dm_ref * w * inv_ref;
                        a_w += dm_ref * (rho_f - m_ref) * inv_ref;
                    }
                }
                aux_push_row16<AUXCHK>(c, uniform(B.ref_reg_src) + 16 * t, group_sum(a_rho), h - 16 * t);
            }
            if constexpr (LAYERED) aux_push_scalar(c, enc_phi(uniform(B.reg_weight_phi)), a_w);
            else d_reg_w = a_w;  // (pushed with the block's other scalars, phase 3)
        }
        lds_barrier();
        for (int i = tid; i < gg.nsets * PMT_ZW; i += PMT_THREADS) (&sh.gsum[0][0][0])[i] = 0.f;
        prof_add(c, 11, t_ph);
        trace_ev(c, 21);
        t_ph = prof_now();
        // ---- phase 3: finish d(z2), LayerNorm(h) backward, SELU backward -> d(zpre) ---------------------------------------
        if (PMT_BWD_XH4_AT_P3 == 1) { load_xh4(); xh4_requested = true; }  // phase 4's stash read, requested here: phase 3 is ~3.6 k cycles of arithmetic to hide it under
        f4 dz[PMT_RT][2 * HT];
        {
            f4 dsw[HT], dsb[HT];
#pragma unroll
            for (int t = 0; t < HT; ++t) dsw[t] = dsb[t] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int rt = 0; rt < PMT_RT; ++rt) {
                const int set = tm[rt].set;
                const bool ok = tm[rt].valid;
                f4 dz2v[HT], dxr[HT];
#pragma unroll
                for (int t = 0; t < HT; ++t) {
                    f4 z2v, m_ref, m_alt;
                    const f4 gt = gate_of(rt, t, z2hat[rt][t], z2v, m_ref, m_alt);
                    f4 dz2 = dgate[rt][t] * alpha;
                    const f4 dm = *reinterpret_cast<const f4*>(&sh.dmean[set][side][16 * t + 4 * g]);
                    if (ok) dz2 = dz2 + dm;
                    dz2v[t] = dz2;
                    const f4 dz1 = ok ? du[rt][t] * gt : f4{0.f, 0.f, 0.f, 0.f};
                    dz[rt][t] = selu_bwd4(dz1, z[rt][t]);
                }
                layernorm_bwd_tile<HT>(dxr, dz2v, z2hat[rt], rstd2[rt], h, sw, dsw, dsb, g);
#pragma unroll
                for (int t = 0; t < HT; ++t) dz[rt][HT + t] = selu_bwd4(dxr[t], z[rt][HT + t]);
            }
            if (PMT_BWD_XH4_AT_P3 == 2) {  // behind the phase's last global load, ahead of its cross-lane sums (LDS only)
                __builtin_amdgcn_sched_barrier(0);
                load_xh4();
                __builtin_amdgcn_sched_barrier(0);
                xh4_requested = true;
            }
            aux_push_vec_x<HT, EX, AUXCHK>(c, uniform(B.sgu_norm_w_src), dsw, h);
            aux_push_vec_x<HT, EX, AUXCHK>(c, uniform(B.sgu_norm_b_src), dsb, h);
            if constexpr (!LAYERED) push_gate_scalars(d_reg_w, enc_phi(uniform(B.reg_weight_phi)));  // (layered: pushed at the end of the launch that computed them)
        }
        prof_add(c, 12, t_ph);
        trace_ev(c, 22);
        t_ph = prof_now();
        // ---- phase 4: proj1 weight gradient (needs n again), d(n) = W1^T d(zpre), LayerNorm(D) backward ---------------------
        {
            f4 n[PMT_RT][NTD];
            if (!xh4_requested) load_xh4();  // (the generic instances and the later slices of a layered backward: no early request)
            affine_n(n, xh4);
            if constexpr (S::BF16 != 0) wgrad_exchange_bf<2 * HT, NTD, 2, BFB>(c, M->lin[uniform(B.proj1[0])], M->lin[uniform(B.proj1[1])], dz, n, 1.0f);
            else wgrad_exchange<2 * HT, NTD, 2>(c, M->lin[uniform(B.proj1[0])], M->lin[uniform(B.proj1[1])], dz, n, 1.0f);
        }
        prof_add(c, 13, t_ph);
        trace_ev(c, 23);
        t_ph = prof_now();
        {
            f4 dn[PMT_RT][NTD];
            init_bias<NTD>(dn, nullptr, D, g);
            if constexpr (S::BF16) linear_acc_bf16<2 * HT, NTD, false, PMT_DG(BFB)>(dn, dz, packed + uniform(P1.wtb_frag));
            else linear_acc<2 * HT, NTD, false, EX, 0, S::DIM_H>(dn, dz, packed + uniform(P1.wt_frag), PMT_SPLIT0 + h, D);
            f4 lw[NTD], dlw[NTD], dlb[NTD];
#pragma unroll
            for (int t = 0; t < NTD; ++t) {
                lw[t] = load_pvec(lw_p, t, g);
                dlw[t] = f4{0.f, 0.f, 0.f, 0.f};
                dlb[t] = f4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int rt = 0; rt < PMT_RT; ++rt) {
                // one tile at a time (the scheduling barrier keeps the compiler from interleaving the tiles' temporaries)
                __builtin_amdgcn_sched_barrier(0);
                layernorm_bwd_inplace_tile<NTD>(dy[rt], dn[rt], xh4[rt], rs4[rt], D, lw, dlw, dlb, g);
            }
            __builtin_amdgcn_sched_barrier(0);
            aux_push_vec_x<NTD, EX, AUXCHK>(c, uniform(B.norm_w_src), dlw, D);
            aux_push_vec_x<NTD, EX, AUXCHK>(c, uniform(B.norm_b_src), dlb, D);
        }
        prof_add(c, 14, t_ph);
        trace_ev(c, 24);
    }
    t_ph = prof_now();

    // ---- split d(x_0): variant-embedding part -> per-set sums; read-embedding part -> read MLP backward --------------
    lds_barrier();  // the last exchange has been consumed by every wave: the end of the stage becomes dv
    for (int i = tid; i < PMT_GROUP_MAX_SETS * PMT_MAX_WIDTH; i += PMT_THREADS) (&sh.dv()[0][0])[i] = 0.f;
    lds_barrier();
#pragma unroll
    for (int rt = 0; rt < PMT_RT; ++rt) {
        const int set = tm[rt].set;
        const SegPlan sp = seg_plan(tm[rt].valid ? set : -1);
#pragma unroll
        for (int t = 0; t < NTD; ++t) {
            if (16 * t + 15 < Er) continue;  // (no lane group holds a variant-embedding feature in this tile of registers)
            f4 part;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int f = feat_of(t, j, g);
                part[j] = (tm[rt].valid && f >= Er && f < D) ? dy[rt][t][j] : 0.f;
            }
            const f4 s4 = seg_sum4<!S::BF16>(part, sp);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int f = feat_of(t, j, g);
                if (f >= Er) {
                    if (sp.last && f < D) atomicAdd(&sh.dv()[set][f - Er], s4[j]);
                    dy[rt][t][j] = 0.f;
                }
            }
        }
    }
    lds_barrier();
    for (int i = tid; i < gg.nsets * Ev; i += PMT_THREADS) {  // (written out now: the read MLP's exchanges reuse the LDS)
        const int set = i / Ev, f = i - set * Ev;
        if (LAYERED) atomicAdd(&gvar[(size_t)(gg.v0 + set) * Ev + f], sh.dv()[set][f]);  // several groups per read set
        else gvar[(size_t)(gg.v0 + set) * Ev + f] = sh.dv()[set][f];
    }
    const int fmt = bt.read_format;
    auto decode_reads = [&](f4 (&x)[PMT_RT][NTF]) {
#pragma unroll
        for (int rt = 0; rt < PMT_RT; ++rt) {
            const unsigned char* rowp = nullptr;
            if (tm[rt].valid) {
                const long long src = bt.read_index ? bt.read_index[tm[rt].row] : (long long)tm[rt].row;
                rowp = reinterpret_cast<const unsigned char*>(bt.reads) + (size_t)src * (size_t)bt.read_row_bytes;
            }
            if (NTF == 4 && S::DIM_F == 61 && fmt == PMT_READS_PACKED_U8 && bt.read_row_bytes == 12) {
#pragma unroll
                for (int t = 0; t < NTF; ++t) x[rt][t] = f4{0.f, 0.f, 0.f, 0.f};
                if constexpr (NTF == 4) {
                    if (rowp) decode_packed12(x[rt], rowp, g);
                }
                continue;
            }
#pragma unroll
            for (int t = 0; t < NTF; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) x[rt][t][j] = rowp ? read_feature_b(rowp, fmt, feat_of(t, j, g), F) : 0.f;
        }
    };
    if constexpr (EX) {
        f4 dr[PMT_RT][NTR];
#pragma unroll
        for (int rt = 0; rt < PMT_RT; ++rt)
#pragma unroll
            for (int t = 0; t < NTR; ++t) dr[rt][t] = dy[rt][t < NTD ? t : 0];
        if constexpr (S::DROP) drop.on = drop_setup(M, bt.dropout_seed, uniform(M->read_mlp.dropout)).on;
        mlp_backward<NTR, true, S::DIM_R, BFB, S::DROP>(c, M->read_mlp, dr, true,
                                [&](int op, f4 (&x)[PMT_RT][NTR]) { load_slot_tiles<NTR>(stash_tile, mask_all, op - 1, x, c.pf_sink); }, 1, n_read_ops);
        f4 xf[PMT_RT][NTF], dxf[PMT_RT][NTF];
        decode_reads(xf);
        linear_op_backward<NTF, NTR, true, S::DIM_F, S::DIM_R, BFB, S::DROP>(c, M->read_mlp.ops[0], dr, xf, dxf, false);
    } else {
        drop.on = drop_setup(M, bt.dropout_seed, uniform(M->read_mlp.dropout)).on;
        mlp_backward<NTD, false>(c, M->read_mlp, dy, false,
                                 [&](int op, f4 (&x)[PMT_RT][NTD]) {
                                     if (op > 0) load_slot_tiles<NTD>(stash_tile, mask_all, op - 1, x); else decode_reads(x);
                                 }, 0, n_read_ops);
    }
    prof_add(c, 16, t_ph);
    trace_ev(c, 26);
    aux_flush(c);
    prof_add(c, 7, t_kernel0);
    trace_ev(c, 17);  // whole kernel, per wave
}

// Launched either with one workgroup per group or (partials != nullptr) as a fixed number of PERSISTENT workgroups that
// take the groups round-robin: a workgroup then owns row blockIdx.x of `partials` ([gridDim.x][emit_len], a mirror of the
// emit-table region of `packed`) and adds its weight-gradient blocks there with plain 16-byte loads and stores -- the ~250 M
// float atomics per step this replaces are throughput-bound in L2 and cost a tenth of the kernel.
// pmt_grad_fold_kernel sums the rows into the gradient buffers afterwards.
template <typename S, bool LAYERED = false, bool PRIV = false>
__global__ __launch_bounds__(PMT_THREADS, PMT_BWD_WAVES_PER_SIMD) void pmt_backward_kernel(
    const PmtModel* __restrict__ M, const float* __restrict__ theta, const float* __restrict__ phi,
    const float* __restrict__ packed, PmtBatch bt, PmtOutputs out, PmtOutputGrads dout, const float* __restrict__ stash,
    const float* __restrict__ zsum_stash, const float* __restrict__ rstd_stash, float* __restrict__ gtheta, float* __restrict__ gphi,
    float* __restrict__ gvar, PmtBwdLayered lay, float* __restrict__ partials, int emit_base, int emit_len) {
    __shared__ __attribute__((aligned(16))) BwdShared sh;
    const int ngroups = bt.num_groups_dev != nullptr ? uniform(bt.num_groups_dev[0]) : bt.num_groups;  // (device count: graph replay)
    float* priv = PRIV ? partials + (size_t)blockIdx.x * (size_t)emit_len - emit_base : nullptr;  // (the host picks PRIV = (partials != nullptr))
#if PMT_BWD_PRIO
    // the second-dispatched half of the workgroup loses every arbitration for its SIMD's issue slots to the older half
    // (MI355X_MICROARCH.md, two waves per SIMD, item 4): one static priority for it, set once
    if (threadIdx.x >= PMT_THREADS / 2) __builtin_amdgcn_s_setprio(1);
#endif
    if (LAYERED && lay.join.on) {
        // joined execution: groups go out by ticket, so the groups that have started always form a prefix and a group never waits
        // for one that nobody will run (pmt_device.hpp: PmtJoin)
        for (;;) {
            const int grp = pmt_join_ticket(lay.join, &sh.ticket);
            if (grp >= ngroups) break;
            backward_group<S, LAYERED, PRIV>(M, theta, phi, packed, bt, out, dout, stash, zsum_stash, rstd_stash, gtheta, gphi, gvar, lay, grp, sh, priv);
            lds_barrier();
        }
        return;
    }
#ifdef PMT_X_NOLOOP
    const int grp = blockIdx.x;
    if (grp < ngroups) {
#else
    for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
#endif
        backward_group<S, LAYERED, PRIV>(M, theta, phi, packed, bt, out, dout, stash, zsum_stash, rstd_stash, gtheta, gphi, gvar, lay, grp, sh, priv);
        lds_barrier();  // the next group reuses the LDS
    }
}

// rows of partial sums -> gradient buffers (and the rows back to zero).  grid (blocks over a linear's entries, linear); a
// block = 32 float4 entries x 8 slices of the rows, joined in LDS: ~100 MB of rows stream through at HBM / MALL speed
__global__ __launch_bounds__(256) void pmt_grad_fold_kernel(const PmtModel* __restrict__ M, const float* __restrict__ packed,
                                                            float* __restrict__ partials, int rows, float* __restrict__ gtheta,
                                                            float* __restrict__ gphi) {
    __shared__ f4 part[8][32];
    const PmtLinear& L = M->lin[blockIdx.y];
    const int tab = L.emit_tab;
    if (tab < 0) return;
    const int out_v = L.out_split > 0 ? PMT_SPLIT0 + L.out_split : L.out_dim;
    const int nmt = (out_v + 15) >> 4, nkt = (L.in_dim + 15) >> 4, nw = nmt * nkt * 256, n = nw + nmt * 16;  // (multiples of 4)
    const int q = threadIdx.x & 31, slice = threadIdx.x >> 5, i = (blockIdx.x * 32 + q) * 4;
    if (blockIdx.x * 128 >= n) return;
    const size_t emit_len = (size_t)M->emit_len;
    f4 s = f4{0.f, 0.f, 0.f, 0.f};
    if (i < n) {
        float* p = partials + (tab - M->emit_base) + i;
        for (int r = slice; r < rows; r += 8) {
            f4* pr = reinterpret_cast<f4*>(p + (size_t)r * emit_len);
            s = s + *pr;
            *pr = f4{0.f, 0.f, 0.f, 0.f};
        }
    }
    part[slice][q] = s;
    __syncthreads();
    if (slice == 0 && i < n) {
#pragma unroll
        for (int k = 1; k < 8; ++k) s = s + part[k][q];
        const int* dst = reinterpret_cast<const int*>(packed) + tab + i;
        float* g = (i < nw && L.w_src < 0) ? gphi : gtheta;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (dst[j] >= 0 && s[j] != 0.f) atomicAdd(g + dst[j], s[j]);
    }
}

// the instance that adds into private rows, where the shape has one (a generic-only build aliases every shape to the fp32 instances)
template <typename S, bool LAYERED>
static auto priv_kernel() {
    if constexpr (S::BF16 != 0) return pmt_backward_kernel<S, LAYERED, true>;
    else return pmt_backward_kernel<S, LAYERED, false>;
}
// persistent launch with private partial sums: only the bf16-exchange instances know them
static bool use_partials(const PmtModel* m, int shape, const float* partials, int rows) {
    return partials != nullptr && rows > 0 && m->emit_len > 0 && shape >= 2;  // (2, 3, 4, 6: the instances with the bf16 exchange)
}
static int fold_partials(const PmtModel* model_host, const PmtModel* model_dev, const float* packed, float* partials, int rows,
                         float* grad_theta, float* grad_phi, hipStream_t s) {
    hipLaunchKernelGGL(pmt_grad_fold_kernel, dim3((PMT_MAX_WIDTH * PMT_MAX_WIDTH + PMT_MAX_WIDTH + 127) / 128, model_host->n_linear), dim3(256), 0, s,
                       model_dev, packed, partials, rows, grad_theta, grad_phi);
    return hipGetLastError() == hipSuccess ? PMT_OK : PMT_E_LAUNCH;
}

extern "C" int pmt_backward(const PmtModel* model_host, const PmtModel* model_dev, const float* theta, const float* phi,
                            const float* packed, const PmtBatch* batch, const PmtOutputs* out, const PmtOutputGrads* dout,
                            const float* stash, float* grad_theta, float* grad_phi, float* grad_variant_embed,
                            float* grad_partials, int32_t num_partials, void* stream) {
    if (!model_host || !model_dev || !batch || !out || !dout || !stash || !grad_theta || !grad_phi || !grad_variant_embed)
        return PMT_E_INVALID;
    const int rc = pmt_model_check(model_host);
    if (rc != PMT_OK) return rc;
    if (batch->num_groups <= 0) return batch->num_groups == 0 ? PMT_OK : PMT_E_INVALID;
    if (batch->group_span) return PMT_E_UNSUPPORTED;  // read sets split over workgroups: pmt_backward_layered
    if (!batch->reads || !batch->ref_offsets || !batch->alt_offsets || !batch->group_start || !batch->group_tile_base ||
        batch->total_tiles <= 0 || !out->logits_b || !out->logits_bk)
        return PMT_E_INVALID;
    const float* zsum_stash = stash + (size_t)batch->total_tiles * (size_t)pmt_stash_slots(model_host) * PMT_SLOT_FLOATS;
    const float* rstd_stash = zsum_stash + (size_t)batch->num_variants * (size_t)(model_host->num_blocks > 0 ? model_host->num_blocks : 1) * PMT_ZW;
    const int shape = pmt_shape_for(model_host, batch);
    const bool part = use_partials(model_host, shape, grad_partials, num_partials);
    const int grid = part && num_partials < batch->num_groups ? num_partials : batch->num_groups;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    // (private rows are a compile-time property of the bf16-exchange instances: use_partials is false for the others)
    auto kernel = part ? (shape == 4 ? priv_kernel<ShapeP0XD, false>() : shape == 3 ? priv_kernel<ShapeP0XB, false>()
                          : shape == 2 ? priv_kernel<ShapeP0X, false>() : priv_kernel<ShapeP0T, false>())
                  : shape == 4 ? pmt_backward_kernel<ShapeP0XD> : shape == 3 ? pmt_backward_kernel<ShapeP0XB> : shape == 2 ? pmt_backward_kernel<ShapeP0X>
                  : shape == 6 ? pmt_backward_kernel<ShapeP0T> : shape == 1 ? pmt_backward_kernel<ShapeP0> : pmt_backward_kernel<ShapeAny>;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(PMT_THREADS), 0, s, model_dev,
                       theta, phi, packed, *batch, *out, *dout, stash, zsum_stash, rstd_stash, grad_theta, grad_phi, grad_variant_embed,
                       PmtBwdLayered{}, part ? grad_partials : nullptr, model_host->emit_base, model_host->emit_len);
    if (hipGetLastError() != hipSuccess) return PMT_E_LAUNCH;
    return part ? fold_partials(model_host, model_dev, packed, grad_partials, grid, grad_theta, grad_phi, s) : PMT_OK;
}

extern "C" size_t pmt_layered_backward_scratch_floats(const PmtModel* m, int64_t total_tiles, int32_t num_variants) {
    if (!m) return 0;
    const size_t nb = (size_t)(m->num_blocks > 0 ? m->num_blocks : 1);
    // parked state (layered launches only) | per-set d(gate) sums | joined execution: arrival counters [B][L], ticket, fault word
    return (size_t)total_tiles * (PMT_SLOT_FLOATS + PMT_BWD_PARK_TILES * 256) + (size_t)num_variants * nb * PMT_ZW + (size_t)num_variants * nb + 8;
}

extern "C" int pmt_backward_layered(const PmtModel* model_host, const PmtModel* model_dev, const float* theta, const float* phi,
                                    const float* packed, const PmtBatch* batch, const PmtOutputs* out, const PmtOutputGrads* dout,
                                    const float* stash, float* scratch, float* grad_theta, float* grad_phi,
                                    float* grad_variant_embed, float* grad_partials, int32_t num_partials, void* stream) {
    if (!model_host || !model_dev || !batch || !out || !dout || !stash || !scratch || !grad_theta || !grad_phi || !grad_variant_embed)
        return PMT_E_INVALID;
    const int rc = pmt_model_check(model_host);
    if (rc != PMT_OK) return rc;
    if (batch->num_groups <= 0) return batch->num_groups == 0 ? PMT_OK : PMT_E_INVALID;
    if (!batch->reads || !batch->ref_offsets || !batch->alt_offsets || !batch->group_span || !batch->group_tile_base ||
        batch->total_tiles <= 0 || !out->logits_b || !out->logits_bk)
        return PMT_E_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int L = model_host->num_blocks;
    const size_t nb = (size_t)(L > 0 ? L : 1), B = (size_t)batch->num_variants;
    const float* zsum_stash = stash + (size_t)batch->total_tiles * (size_t)pmt_stash_slots(model_host) * PMT_SLOT_FLOATS;
    const float* rstd_stash = zsum_stash + B * nb * PMT_ZW;
    PmtBwdLayered lay;
    lay.dy_scratch = scratch;
    lay.park = lay.dy_scratch + (size_t)batch->total_tiles * PMT_SLOT_FLOATS;
    lay.gsum_g = lay.park + (size_t)batch->total_tiles * PMT_BWD_PARK_TILES * 256;
    if (hipMemsetAsync(lay.gsum_g, 0, B * nb * PMT_ZW * sizeof(float), s) != hipSuccess) return PMT_E_LAUNCH;
    const int shape = pmt_shape_for(model_host, batch, true);
    const bool part = use_partials(model_host, shape, grad_partials, num_partials);
    const int grid = part && num_partials < batch->num_groups ? num_partials : batch->num_groups;
    auto kernel = (shape >= 2 && shape != 6) ? (part ? priv_kernel<ShapeP0X, true>() : pmt_backward_kernel<ShapeP0X, true>)
                  : (shape == 1 || shape == 6) ? pmt_backward_kernel<ShapeP0, true> : pmt_backward_kernel<ShapeAny, true>;
    int* join_words = reinterpret_cast<int*>(lay.gsum_g + B * nb * PMT_ZW);
    lay.join = PmtJoin{0, join_words + B * nb, join_words, batch->join_fault ? batch->join_fault : join_words + B * nb + 1};
    if (batch->set_groups != nullptr && L > 0) {  // ONE launch: the groups of a split read set join their sums through HBM
        lay.join.on = 1;
        if (hipMemsetAsync(join_words, 0, (B * nb + 8) * sizeof(int), s) != hipSuccess) return PMT_E_LAUNCH;
    }
    for (int slice = 0; slice <= (lay.join.on ? 0 : L); ++slice) {
        lay.slice = slice;
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(PMT_THREADS), 0, s, model_dev, theta, phi, packed, *batch, *out, *dout,
                           stash, zsum_stash, rstd_stash, grad_theta, grad_phi, grad_variant_embed, lay,
                           part ? grad_partials : nullptr, model_host->emit_base, model_host->emit_len);
    }
    if (hipGetLastError() != hipSuccess) return PMT_E_LAUNCH;
    return part ? fold_partials(model_host, model_dev, packed, grad_partials, grid, grad_theta, grad_phi, s) : PMT_OK;
}
