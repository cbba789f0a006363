// Fused read-set forward for gfx950: one workgroup (8 waves) owns a group of whole read sets; activations stay in
// registers from the packed input bytes to the per-set outputs.  See pmt_device.hpp for the register layout.
//
// Replaces (reference paths): data/batch.py:51-56 (decode), architecture/mlp.py:75-76, artifact_model.py:243-263,
// gated_mlp.py:177-251, sets/ragged_sets.py:144-158, euclidean_transformation.py:19-20,
// feature_clustering.py:82-135, exponentially_modified_gaussian.py:30-89, artifact_model.py:291-292.
#include "pmt_device.hpp"
#include "pmt_mlp_device.hpp"

#ifndef PMT_LAYERED_WIDE
#define PMT_LAYERED_WIDE 0  // 1: the instances for split read sets at 256 registers / two waves per SIMD (A/B switch)
#endif
#ifndef PMT_F16_X1
#define PMT_F16_X1 1  // the first read-MLP linear on one-piece inputs (0: two pieces like every other layer; A/B switch)
#endif
struct FwdShared {  // (the float tables first: their rows are read and cleared 16 bytes at a time)
    float zsum[3][PMT_GROUP_MAX_SETS][2][16 * PMT_HT];
    float fsum[PMT_GROUP_MAX_SETS][2][PMT_MAX_WIDTH];
    float hsum[PMT_GROUP_MAX_SETS][PMT_MAX_CLUSTERS + 2];
    int off[2][PMT_GROUP_MAX_SETS + 1];
    int ticket;  // joined execution: the group this workgroup drew (pmt_join_ticket)
};

// ---- input decode ------------------------------------------------------------------------------------------------
// value of read feature f for the row at `row` (reference data/batch.py:51-56; uint8 wrap quirk of
// data/plain_text_data.py:510-511: (u - 128) / 32 evaluated in uint8).
DEV float read_feature(const unsigned char* __restrict__ row, int fmt, int f, int F) {
    if (f >= F) return 0.f;
    if (fmt == PMT_READS_PACKED_U8) {
        if (f < 56) return (float)((row[f >> 3] >> (7 - (f & 7))) & 1);
        const unsigned u = row[7 + (f - 56)];
        return (float)((u + 128u) & 0xFFu) * (1.0f / 32.0f);
    } else if (fmt == PMT_READS_F16) {
        return (float)reinterpret_cast<const _Float16*>(row)[f];
    } else {
        return reinterpret_cast<const float*>(row)[f];
    }
}

// an activation array about to become a pair of f16 operand pieces: any value beyond their range goes into the caller's fault word
// (v_max3 over the registers, one compare, one ballot; NaN compares false: a non-finite stream shows in the outputs themselves)
#ifndef PMT_F16_RANGE_CHECK
#define PMT_F16_RANGE_CHECK 6  // bits: 2 the residual stream at the reducer, 4 the reducer's output (both free).  (Read rows need no check:
                               // the f16 instances run packed rows only, whose values are below 8 -- wide_range_operands.)
#endif
template <int NT, int WHICH>
DEV void f16_range_check(const f4 (&v)[PMT_RT][NT], int* __restrict__ fault) {
    if (!(PMT_F16_RANGE_CHECK & WHICH)) return;
    float m = 0.f;
#pragma unroll
    for (int rt = 0; rt < PMT_RT; ++rt)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            m = __builtin_fmaxf(__builtin_fmaxf(m, __builtin_fabsf(v[rt][t][0])), __builtin_fabsf(v[rt][t][1]));
            m = __builtin_fmaxf(__builtin_fmaxf(m, __builtin_fabsf(v[rt][t][2])), __builtin_fabsf(v[rt][t][3]));
        }
    if (fault != nullptr && wave_any(m > PMT_F16_OPERAND_MAX) && (pmt_tid() & 63) == 0) atomicOr(fault, PMT_FAULT_F16_RANGE);
}

// log erfc with the asymptotic branch for z > 5 (reference exponentially_modified_gaussian.py:30-55)
DEV float logerfc_dev(float z) {
    const float zc = fmaxf(z, 2.f);
    const float z2 = zc * zc, z4 = z2 * z2, z6 = z2 * z4;
    const float asym = -z2 - logf(zc * 1.7724538509055159f) + log1pf(-1.f / (2.f * z2) + 3.f / (4.f * z4) - 15.f / (8.f * z6));
    const float builtin = logf(fmaxf(erfcf(z), 1.0e-12f));
    return z > 5.f ? asym : builtin;
}

// Layered execution (pmt_forward_layered): launch `slice` finishes block slice - 1 and starts block slice; per-set sums
// live in HBM (zsum_g / fsum_g / hsum_g, float atomics), activations rest in x_scratch / z_scratch between launches.
// JOINED execution (join.on, when the batch carries PmtBatch.set_groups): the same per-set sums in HBM, but ONE launch: a group
// publishes its part of a block's z2 sums, waits for the other groups of its split sets and carries on -- nothing is parked.
struct PmtLayeredArgs {
    int slice;
    float* x_scratch;   // [total_tiles][PMT_SLOT_FLOATS]
    float* z_scratch;   // [total_tiles][512 PMT_HT]: z1 (after SELU) and z2 (after SELU + LayerNorm)
    float* zsum_g;      // [B][L][PMT_ZW] (the training stash's per-set z2 sums have the same layout and ARE this buffer)
    float* fsum_g;      // [B][2][PMT_MAX_WIDTH]
    float* hsum_g;      // [B][PMT_MAX_CLUSTERS + 2]
    PmtJoin join;       // join.on: ONE launch, all blocks, the groups of a split read set join their sums through HBM (pmt_device.hpp)
};

// development: per-wave event log of ONE workgroup (PmtBatch.debug_flags[2] = workgroup + 1; scripts/fwd_trace.py); compiled in
// with -DPMT_FWD_TRACE=1 only
#ifndef PMT_FWD_TRACE
#define PMT_FWD_TRACE 0
#endif
struct FwdTrace {
    int* buf = nullptr;
    int n = 0;
    DEV void ev(int id) {
        if (!PMT_FWD_TRACE || buf == nullptr) return;
        if (n < 250 && (threadIdx.x & 63) == 0) {
            buf[2 * n] = id;
            buf[2 * n + 1] = (int)__builtin_readcyclecounter();
        }
        ++n;
    }
};

template <bool TRAIN, typename S, bool LAYERED = false>
__global__ __launch_bounds__(PMT_THREADS, (S::EXACT && PMT_NT <= 4 && !(LAYERED && PMT_LAYERED_WIDE)) ? 4 : 2) void pmt_forward_kernel(const PmtModel* __restrict__ M,
                                                                      const float* __restrict__ theta,
                                                                      const float* __restrict__ phi,
                                                                      const float* __restrict__ packed, PmtBatch bt,
                                                                      PmtOutputs out, float* __restrict__ stash,
                                                                      float* __restrict__ zsum_stash,
                                                                      float* __restrict__ rstd_stash, PmtLayeredArgs lay) {
    constexpr int NTF = S::NTF, NTR = S::NTR, NTD = S::NTD, NTE = S::NTE;
    constexpr bool EX = S::EXACT;
    constexpr bool SEG_GUARD = S::BF16 != PMT_F16X2;  // (pmt_device.hpp, seg_sum: the f16 instances cannot meet a non-finite activation)
    static_assert(EX || (NTF == NTD && NTR == NTD && NTE == NTD), "the generic shape keeps one array width");
    __shared__ __attribute__((aligned(16))) FwdShared sh;
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4;
    if constexpr (S::BF16 == PMT_F16X2) fp16_saturate_on();  // f16 operand pieces: beyond +-65504 saturate, never inf (linear_acc_f16)
    if (bt.num_groups_dev != nullptr && (int)blockIdx.x >= uniform(bt.num_groups_dev[0])) return;  // grid sized for a capacity (graph replay)
    const bool joined = LAYERED && lay.join.on != 0;
    // joined: groups go out by ticket, in the order the workgroups actually start (dispatch order is not promised): the started
    // groups are always a prefix, which is what makes waiting for a neighbouring group safe
    const int grp = joined ? pmt_join_ticket(lay.join, &sh.ticket) : (int)blockIdx.x;
    const GroupGeom gg = group_geometry(bt, grp);
    const int side = gg.side;

    const int D = S::DIM_D ? S::DIM_D : uniform(M->d_model), E = S::DIM_E ? S::DIM_E : uniform(M->feature_dim), K = uniform(M->num_clusters);
    const int Er = S::DIM_R ? S::DIM_R : uniform(M->read_embed_dim), Ev = uniform(M->variant_embed_dim);
    const int h = S::DIM_H ? S::DIM_H : (uniform(M->d_ffn) >> 1), L = uniform(M->num_blocks), F = S::DIM_F ? S::DIM_F : uniform(M->num_read_features);
    FwdTrace tr;
    if (PMT_FWD_TRACE && bt.debug_flags && uniform(bt.debug_flags[2]) == grp + 1) tr.buf = bt.debug_flags + 64 + (tid >> 6) * 512;
    tr.ev(1);

    // ---- group setup: local offsets, zero the per-set accumulators -------------------------------------------
    for (int i = tid; i <= gg.nsets; i += PMT_THREADS) {
        sh.off[0][i] = bt.ref_offsets[gg.v0 + i] - gg.ref_base;
        sh.off[1][i] = bt.alt_offsets[gg.v0 + i] - gg.alt_base;
    }
    {   // only the rows of the group's sets (typically a third of the capacity), 16 bytes per store
        const f4 zero = f4{0.f, 0.f, 0.f, 0.f};
        for (int i = tid; i < 3 * gg.nsets * (PMT_ZW / 4); i += PMT_THREADS) {
            const int b = i / (gg.nsets * (PMT_ZW / 4)), r = i - b * (gg.nsets * (PMT_ZW / 4));
            reinterpret_cast<f4*>(&sh.zsum[b][0][0][0])[r] = zero;
        }
        for (int i = tid; i < gg.nsets * (2 * PMT_MAX_WIDTH / 4); i += PMT_THREADS) reinterpret_cast<f4*>(&sh.fsum[0][0][0])[i] = zero;
        for (int i = tid; i < gg.nsets * (PMT_MAX_CLUSTERS + 2); i += PMT_THREADS) (&sh.hsum[0][0])[i] = 0.f;
    }
    lds_barrier();
    tr.ev(2);

    TileMeta tm[PMT_RT];
    unsigned mask_all = 0;  // tiles that exist: gates memory traffic only
    float* stash_tile[PMT_RT];
#pragma unroll
    for (int rt = 0; rt < PMT_RT; ++rt) {
        tm[rt] = tile_meta(gg, rt, &sh.off[0][0]);
        if (tm[rt].present) mask_all |= 1u << rt;
        stash_tile[rt] = nullptr;
        if (TRAIN)
            stash_tile[rt] = stash + (size_t)(bt.group_tile_base[grp] + gg.tile_begin + rt) * (size_t)(stash_num_slots(M) * PMT_SLOT_FLOATS);
    }

    int slot = 0;
    const int n_read_ops = uniform(M->read_mlp.n_ops), n_red_ops = uniform(M->reducer.n_ops);
    const int slot_z0 = (n_read_ops - 1) + (L + 1) + (n_red_ops - 1);  // the blocks' z (stash_num_slots)

    // ---- decode the packed read rows straight into the B-operand layout, then the read MLP -------------------------
    f4 x[PMT_RT][NTD];
    const size_t tile_global = (size_t)(bt.group_tile_base[grp] + gg.tile_begin);
    if (LAYERED && lay.slice > 0) {  // resume: activations of the previous launch
#pragma unroll
        for (int rt = 0; rt < PMT_RT; ++rt) {
#pragma unroll
            for (int t = 0; t < NTD; ++t) x[rt][t] = f4{0.f, 0.f, 0.f, 0.f};
            if (mask_all & (1u << rt)) stash_load<NTD>(lay.x_scratch + (tile_global + rt) * PMT_SLOT_FLOATS, x[rt]);
        }
        slot = (n_read_ops - 1) + lay.slice;
    } else {
        f4 xf[PMT_RT][NTF];
        const int fmt = bt.read_format;
#pragma unroll
        for (int rt = 0; rt < PMT_RT; ++rt) {
            const unsigned char* rowp = nullptr;
            if (tm[rt].valid) {
                const long long src = bt.read_index ? bt.read_index[tm[rt].row] : (long long)tm[rt].row;
                rowp = reinterpret_cast<const unsigned char*>(bt.reads) + (size_t)src * (size_t)bt.read_row_bytes;
            }
            if (NTF == 4 && (S::DIM_F == 61 || (S::DIM_F == 0 && EX && F == 61)) && fmt == PMT_READS_PACKED_U8 && bt.read_row_bytes == 12) {
#pragma unroll
                for (int t = 0; t < NTF; ++t) xf[rt][t] = f4{0.f, 0.f, 0.f, 0.f};
                if constexpr (NTF == 4) {
                    if (rowp) decode_packed12(xf[rt], rowp, g);
                }
                continue;
            }
#pragma unroll
            for (int t = 0; t < NTF; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) xf[rt][t][j] = rowp ? read_feature(rowp, fmt, feat_of(t, j, g), F) : 0.f;
        }
        tr.ev(5);
        // (the f16 instances only ever see PACKED rows -- bits and k / 32 quantiles: pmt_forward picks the wide-range instances for float16 /
        //  float32 read rows, wide_range_operands below)
        if constexpr (EX) {
            f4 xr[PMT_RT][NTR];
            PmtDrop drop;  // (only the dropout instance, ShapeP0XD, touches it)
            if constexpr (S::DROP) {
                drop = drop_setup(M, bt.dropout_seed, uniform(M->read_mlp.dropout));
#pragma unroll
                for (int rt = 0; rt < PMT_RT; ++rt) drop.row[rt] = tm[rt].row;
            }
            // packed rows (bits, k / 32 quantiles) and float16 rows are exact in ONE f16 piece: no low piece, two MFMAs per product
            if (PMT_F16_X1 && S::BF16 == PMT_F16X2 && fmt != PMT_READS_F32)
                run_linear_op<NTF, NTR, true, S::DIM_F, S::DIM_R, S::BF16, S::DROP, true>(M, M->read_mlp.ops[0], xr, xf, g, packed, &drop);
            else
                run_linear_op<NTF, NTR, true, S::DIM_F, S::DIM_R, S::BF16, S::DROP>(M, M->read_mlp.ops[0], xr, xf, g, packed, &drop);
            tr.ev(6);
            run_mlp<TRAIN, NTR, true, S::DIM_R, S::BF16, S::DROP>(M, M->read_mlp, xr, theta, g, mask_all, stash_tile, slot, 1, packed, 1, n_read_ops, &drop);
#pragma unroll
            for (int rt = 0; rt < PMT_RT; ++rt)
#pragma unroll
                for (int t = 0; t < NTD; ++t) x[rt][t] = t < NTR ? xr[rt][t < NTR ? t : 0] : f4{0.f, 0.f, 0.f, 0.f};
        } else {
            PmtDrop drop = drop_setup(M, bt.dropout_seed, uniform(M->read_mlp.dropout));
#pragma unroll
            for (int rt = 0; rt < PMT_RT; ++rt) drop.row[rt] = tm[rt].row;
            run_mlp<TRAIN, NTD, false>(M, M->read_mlp, xf, theta, g, mask_all, stash_tile, slot, 1, packed, 0, n_read_ops, &drop);
#pragma unroll
            for (int rt = 0; rt < PMT_RT; ++rt)
#pragma unroll
                for (int t = 0; t < NTD; ++t) x[rt][t] = xf[rt][t < NTF ? t : 0];
        }
    }

    tr.ev(3);
    // ---- broadcast-concat of the per-variant embedding (reference artifact_model.py:246-251) --------------------
    if (!(LAYERED && lay.slice > 0)) {
#pragma unroll
    for (int rt = 0; rt < PMT_RT; ++rt) {
        const float* vrow = bt.variant_embed + (size_t)(gg.v0 + tm[rt].set) * (size_t)Ev;
#pragma unroll
        for (int t = 0; t < NTD; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int f = feat_of(t, j, g);
                if (f >= Er && f < D) x[rt][t][j] = vrow[f - Er];
            }
    }

    }
    tr.ev(4);
    // ---- L gated ref/alt blocks; this wave's tiles all use the weights of its side ------------------------------
    for (int l = (LAYERED && lay.slice > 0) ? lay.slice - 1 : 0; l < L; ++l) {
        // the lane coordinates are re-derived from an opaque thread id in every block: per-lane addresses computed from them
        // would otherwise be hoisted out of the loop and, at 128 registers, spilled -- and a spill reload is a vector memory
        // load that queues behind the stash stores (vmcnt retires in order)
        // (pmt_forward_train.hip only, where pmt_tid() is opaque: the filter instance has the registers, and hoisting serves it better)
        const int tid = pmt_tid(), lane = tid & 63, g = lane >> 4;
        TileMeta tmb[PMT_RT];  // and the tiles' set ids pass through an opaque copy: what is derived from them (LDS offsets, segment keys) is not hoisted either
#pragma unroll
        for (int rt = 0; rt < PMT_RT; ++rt) {
            tmb[rt] = tm[rt];
            if (PMT_OPAQUE_TID) asm volatile("" : "+v"(tmb[rt].set));
        }
        const PmtBlock& B = M->blocks[l];
        const bool first_half = !LAYERED || joined || l == lay.slice;  // LayerNorm, proj1, SELU, per-set sums of z2
        constexpr int HT = PMT_HT;  // tiles per half of the hidden layer: z[..][0 .. HT) = z1, z[..][HT .. 2 HT) = z2
        f4 z[PMT_RT][2 * HT];
        // packed region A of this block: [W1_ref | W1_alt | b1_ref | b1_alt | LN(D) w,b | LN(h) w,b | rho]
        const PmtLinear& P1r = M->lin[uniform(B.proj1[0])];
        const int baseA = uniform(P1r.w_frag);
        const float* stA = packed + baseA;
        const int buf = l % 3;
        if (first_half) {
            f4 lw[NTD], lb[NTD];
#pragma unroll
            for (int t = 0; t < NTD; ++t) {
                lw[t] = load_pvec(stA + (uniform(B.norm_w_pvec) - baseA), t, g);
                lb[t] = load_pvec(stA + (uniform(B.norm_b_pvec) - baseA), t, g);
            }
            f4 n[PMT_RT][NTD];
#pragma unroll
            for (int rt = 0; rt < PMT_RT; ++rt) {
                f4 xhat[NTD];
                float rstd;
                layernorm_tile<NTD>(n[rt], xhat, rstd, x[rt], D, lw, lb, g);
                // the backward needs x_l only through its LayerNorm: stash xhat (same bytes as x_l) + one rstd per read
                if (TRAIN && (mask_all & (1u << rt))) {
                    stash_store<NTD>(stash_tile[rt] + slot * PMT_SLOT_FLOATS, xhat);
                    if (g == 0)
                        rstd_stash[((size_t)(bt.group_tile_base[grp] + gg.tile_begin + rt) * L + l) * 16 + (lane & 15)] = rstd;
                }
            }
            if (TRAIN) ++slot;
            const float* bp = stA + (uniform(M->lin[uniform(B.proj1[side])].b_pvec) - baseA);
#pragma unroll
            for (int t = 0; t < 2 * HT; ++t) {
                const f4 bt_ = load_pvec(bp, t, g);
#pragma unroll
                for (int rt = 0; rt < PMT_RT; ++rt) z[rt][t] = bt_;
            }
            if constexpr (S::BF16) linear_acc_mx<NTD, 2 * HT, false, S::BF16>(z, n, packed, M->lin[uniform(B.proj1[side])]);
            else linear_acc<NTD, 2 * HT, false, EX, S::DIM_D>(z, n, stA + side * frag_floats_dev(P1r), D, PMT_SPLIT0 + h);
        // SELU, LayerNorm(h) on z2, per-set sums (reference gated_mlp.py:228-239)
        f4 sw[HT], sb[HT];  // (loaded here, behind the first projection: eight registers less across it)
#pragma unroll
        for (int t = 0; t < HT; ++t) {
            sw[t] = load_pvec(stA + (uniform(B.sgu_norm_w_pvec) - baseA), t, g);
            sb[t] = load_pvec(stA + (uniform(B.sgu_norm_b_pvec) - baseA), t, g);
        }
#pragma unroll
        for (int rt = 0; rt < PMT_RT; ++rt) {
            f4 zin[HT], zo[HT], zh[HT];
#pragma unroll
            for (int t = 0; t < HT; ++t) {
                z[rt][t] = selu4(z[rt][t]);
                zin[t] = selu4(z[rt][HT + t]);
            }
            if (TRAIN && EX && PMT_STASH_Z && (mask_all & (1u << rt))) {  // the (exact-width) backward starts the block from these instead of recomputing them
                f4 zs[2 * HT];
#pragma unroll
                for (int t = 0; t < HT; ++t) { zs[t] = z[rt][t]; zs[HT + t] = zin[t]; }
                stash_store<2 * HT>(stash_tile[rt] + (size_t)(slot_z0 + l) * PMT_SLOT_FLOATS, zs);
            }
            float rstd;
            layernorm_tile<HT>(zo, zh, rstd, zin, h, sw, sb, g);
            {   // per-set sums of z2: segmented reduce over the tile's reads, one LDS add per set and value
                const SegPlan sp = seg_plan(tmb[rt].valid ? tmb[rt].set : -1);
#pragma unroll
                for (int t = 0; t < HT; ++t) {
                    z[rt][HT + t] = zo[t];
                    float* dst = &sh.zsum[buf][tmb[rt].set][side][16 * t + 4 * g];
                    const f4 s4 = seg_sum4<SEG_GUARD>(tmb[rt].valid ? zo[t] : f4{0.f, 0.f, 0.f, 0.f}, sp);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (sp.last && feat_of(t, j, g) < h) atomicAdd(dst + j, s4[j]);
                }
            }
        }
        }
        if constexpr (LAYERED) {
            if (joined) {  // publish this group's part of the block's sums, wait for the other groups of its split sets, read the totals
                lds_barrier();
                pmt_join_sets(lay.join, &sh.zsum[buf][0][0][0], lay.zsum_g + ((size_t)gg.v0 * L + l) * PMT_ZW, L * PMT_ZW,
                              lay.join.arrivals + (size_t)gg.v0 * L + l, L, bt.set_groups + gg.v0, gg.nsets);
            } else if (first_half) {  // end of this launch: the group's partial sums join the global ones; park x and z
                lds_barrier();
                for (int i = tid; i < gg.nsets * PMT_ZW; i += PMT_THREADS) {
                    const float v = (&sh.zsum[buf][0][0][0])[i];
                    if (v != 0.f) atomicAdd(&lay.zsum_g[((size_t)(gg.v0 + (i / PMT_ZW)) * L + l) * PMT_ZW + (i % PMT_ZW)], v);
                }
#pragma unroll
                for (int rt = 0; rt < PMT_RT; ++rt)
                    if (mask_all & (1u << rt)) {
                        stash_store<NTD>(lay.x_scratch + (tile_global + rt) * PMT_SLOT_FLOATS, x[rt]);
                        stash_store<2 * HT>(lay.z_scratch + (tile_global + rt) * (512 * HT), z[rt]);
                    }
                return;
            } else {
            // second half of block l = slice - 1: z from the previous launch, the COMPLETE per-set sums from HBM
#pragma unroll
            for (int rt = 0; rt < PMT_RT; ++rt) {
#pragma unroll
                for (int t = 0; t < 2 * HT; ++t) z[rt][t] = f4{0.f, 0.f, 0.f, 0.f};
                if (mask_all & (1u << rt)) stash_load<2 * HT>(lay.z_scratch + (tile_global + rt) * (512 * HT), z[rt]);
            }
            for (int i = tid; i < gg.nsets * PMT_ZW; i += PMT_THREADS)
                (&sh.zsum[buf][0][0][0])[i] = lay.zsum_g[((size_t)(gg.v0 + (i / PMT_ZW)) * L + l) * PMT_ZW + (i % PMT_ZW)];
            }
        }
        tr.ev(10);
        lds_barrier();
        tr.ev(11);
        if (TRAIN && !LAYERED) {
            for (int i = tid; i < gg.nsets * PMT_ZW; i += PMT_THREADS)
                zsum_stash[((size_t)(gg.v0 + (i / PMT_ZW)) * L + l) * PMT_ZW + (i % PMT_ZW)] = (&sh.zsum[buf][0][0][0])[i];
        }
        {
            const int zb = (l + 2) % 3;
            for (int i = tid; i < gg.nsets * PMT_ZW; i += PMT_THREADS) (&sh.zsum[zb][0][0][0])[i] = 0.f;
        }
        // gate and second projection with the residual as the accumulator input
        {
            const float w = uniform(phi[uniform(B.reg_weight_phi)]) + 0.25f;
            const float alpha = uniform(theta[uniform(B.alpha_src[side])]), beta = uniform(theta[uniform(B.beta_src[side])]);
            const float gamma = uniform(theta[uniform(B.gamma_src)]);
            f4 u[PMT_RT][HT];
#pragma unroll
            for (int t = 0; t < HT; ++t) {
                const f4 rho = load_pvec(stA + (uniform(B.ref_reg_pvec) - baseA), t, g);  // (loaded here, not at the top of the block: four registers less across its first half)
#pragma unroll
                for (int rt = 0; rt < PMT_RT; ++rt) {
                    const int set = tmb[rt].set;
                    const float n_ref = (float)(sh.off[0][set + 1] - sh.off[0][set]);
                    const float n_alt = (float)(sh.off[1][set + 1] - sh.off[1][set]);
                    const f4 s_ref = *reinterpret_cast<const f4*>(&sh.zsum[buf][set][0][16 * t + 4 * g]);
                    const f4 m_ref = (s_ref + w * rho) * fast_rcp(n_ref + w);
                    f4 gate = z[rt][HT + t] * alpha + 1.0f;
                    if (side == 0) {
                        gate = gate + beta * m_ref;
                    } else {
                        const f4 s_alt = *reinterpret_cast<const f4*>(&sh.zsum[buf][set][1][16 * t + 4 * g]);
                        const f4 m_alt = s_alt * fast_rcp(n_alt + 1e-4f);
                        gate = (gate + beta * m_alt) + gamma * m_ref;
                    }
                    u[rt][t] = z[rt][t] * gate;
                }
            }
            const PmtLinear& P2r = M->lin[uniform(B.proj2[0])];
            const int baseB = uniform(P2r.w_frag);
            const float* p2_frags = packed + baseB;  // [W2_ref | W2_alt | b2_ref | b2_alt]
            const float* bp = p2_frags + (uniform(M->lin[uniform(B.proj2[side])].b_pvec) - baseB);
#pragma unroll
            for (int t = 0; t < NTD; ++t) {
                const f4 b = load_pvec(bp, t, g);
#pragma unroll
                for (int rt = 0; rt < PMT_RT; ++rt) x[rt][t] = x[rt][t] + b;
            }
            if constexpr (S::BF16) linear_acc_mx<HT, NTD, false, S::BF16>(x, u, packed, M->lin[uniform(B.proj2[side])]);
            else linear_acc<HT, NTD, false, EX, S::DIM_H>(x, u, p2_frags + side * frag_floats_dev(P2r), h, D);
        }
        tr.ev(12);
    }
    if (TRAIN) {  // x_L, the reducer's input
#pragma unroll
        for (int rt = 0; rt < PMT_RT; ++rt)
            if (mask_all & (1u << rt)) stash_store<NTD>(stash_tile[rt] + slot * PMT_SLOT_FLOATS, x[rt]);
        ++slot;
    }
    // The f16 operand pieces saturate at +-65504 (linear_acc_f16).  Inside the gated blocks every matrix operand is LayerNorm'ed or a
    // SELU of one, and the read MLP's inputs are bytes; the ONE place an unnormalised activation of any size becomes a matrix
    // operand is here, where the residual stream x_L enters the reducer (and, below, where the reducer's output enters the
    // rotation).  A value beyond the range is reported in the caller's fault word (bit 1), never silently clipped.
    if constexpr (S::BF16 == PMT_F16X2) f16_range_check<NTD, 2>(x, bt.join_fault);

    tr.ev(19);
    // ---- reducer MLP, then translation + rotation ----------------------------------------------------------------
    f4 e[PMT_RT][NTE];
    if constexpr (EX) {
        PmtDrop drop;
        if constexpr (S::DROP) {
            drop = drop_setup(M, bt.dropout_seed, uniform(M->reducer.dropout));
#pragma unroll
            for (int rt = 0; rt < PMT_RT; ++rt) drop.row[rt] = tm[rt].row;
        }
        run_mlp<TRAIN, NTD, true, S::DIM_D, S::BF16, S::DROP>(M, M->reducer, x, theta, g, mask_all, stash_tile, slot, 1, packed, 0, n_red_ops - 1, &drop);
        if (TRAIN && n_red_ops > 1) {
#pragma unroll
            for (int rt = 0; rt < PMT_RT; ++rt)
                if (mask_all & (1u << rt)) stash_store<NTD>(stash_tile[rt] + slot * PMT_SLOT_FLOATS, x[rt]);
            ++slot;
        }
        run_linear_op<NTD, NTE, true, S::DIM_D, S::DIM_E, S::BF16, S::DROP>(M, M->reducer.ops[n_red_ops - 1], e, x, g, packed, &drop);
    } else {
        PmtDrop drop = drop_setup(M, bt.dropout_seed, uniform(M->reducer.dropout));
#pragma unroll
        for (int rt = 0; rt < PMT_RT; ++rt) drop.row[rt] = tm[rt].row;
        run_mlp<TRAIN, NTD, false>(M, M->reducer, x, theta, g, mask_all, stash_tile, slot, 1, packed, 0, n_red_ops, &drop);
#pragma unroll
        for (int rt = 0; rt < PMT_RT; ++rt)
#pragma unroll
            for (int t = 0; t < NTE; ++t) e[rt][t] = x[rt][t < NTD ? t : 0];
    }
    f4 a[PMT_RT][NTE];
    {
        const PmtLinear& R = M->lin[uniform(M->rotation_lin)];
        const float* stR = packed + uniform(R.w_frag);  // [Q fragments | translation]
#pragma unroll
        for (int t = 0; t < NTE; ++t) {
            const f4 tr = load_pvec(stR + (uniform(M->translation_pvec) - uniform(R.w_frag)), t, g);
#pragma unroll
            for (int rt = 0; rt < PMT_RT; ++rt) {
                e[rt][t] = e[rt][t] + tr;
                a[rt][t] = f4{0.f, 0.f, 0.f, 0.f};
            }
        }
        if constexpr (S::BF16 == PMT_F16X2) f16_range_check<NTE, 4>(e, bt.join_fault);
        if constexpr (S::BF16) linear_acc_mx<NTE, NTE, false, S::BF16>(a, e, packed, R);
        else linear_acc<NTE, NTE, false, EX, S::DIM_E>(a, e, stR, E, E);
    }

    tr.ev(21);
    // ---- per-set feature sums (both sides) and the clustering head (alt reads) -------------------------------------
    const int nte = (E + 15) >> 4;
    {
        const float* hp = phi;  // head parameters live in phi (materialised parametrizations), mu in theta
        f4 sig[NTE], isig[NTE];
        float sum_log_sig = 0.f, sum_log_2sig = 0.f;
#pragma unroll
        for (int t = 0; t < NTE; ++t) {
            sig[t] = f4{1.f, 1.f, 1.f, 1.f};
            isig[t] = f4{1.f, 1.f, 1.f, 1.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int f = feat_of(t, j, g);
                if (t < nte && f < E) {
                    sig[t][j] = hp[uniform(M->head.stdev_e_phi) + f];
                    isig[t][j] = 1.0f / sig[t][j];
                    sum_log_sig += logf(sig[t][j]);
                    sum_log_2sig += logf(2.f * sig[t][j]);
                }
            }
        }
        sum_log_sig = group_sum(sum_log_sig);
        sum_log_2sig = group_sum(sum_log_2sig);
        const float c0 = -(0.5f * (float)E) * PMT_LOG2PI - sum_log_sig;
        const float c1 = -(0.5f * (float)E) * PMT_LOG2PI - sum_log_2sig;

#pragma unroll
        for (int rt = 0; rt < PMT_RT; ++rt) {
            if (!(mask_all & (1u << rt))) continue;
            const int set = tm[rt].set;
            const SegPlan sp = seg_plan(tm[rt].valid ? set : -1);
#pragma unroll
            for (int t = 0; t < NTE; ++t)
                if (t < nte) {
                    const f4 s4 = seg_sum4<SEG_GUARD>(tm[rt].valid ? a[rt][t] : f4{0.f, 0.f, 0.f, 0.f}, sp);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (16 * t + 4 * j < E && sp.last && feat_of(t, j, g) < E) atomicAdd(&sh.fsum[set][side][16 * t + 4 * g + j], s4[j]);
                }
            if (side != 1) continue;
            // nonartifact / outlier diagonal Gaussians
            float q0 = 0.f, q1 = 0.f;
#pragma unroll
            for (int t = 0; t < NTE; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (t < nte && feat_of(t, j, g) < E) {
                        const float r0 = a[rt][t][j] * isig[t][j], r1 = 0.5f * r0;
                        q0 += r0 * r0;
                        q1 += r1 * r1;
                    }
            q0 = group_sum(q0);
            q1 = group_sum(q1);
            {
                const float s0 = seg_sum<SEG_GUARD>(tm[rt].valid ? c0 - 0.5f * q0 : 0.f, sp), s1 = seg_sum<SEG_GUARD>(tm[rt].valid ? c1 - 0.5f * q1 : 0.f, sp);
                if (sp.last && g == 0) {
                    atomicAdd(&sh.hsum[set][0], s0);
                    atomicAdd(&sh.hsum[set][1], s1);
                }
            }
            // artifact clusters: projection on the unit direction, orthogonal distance, EMG along the direction.  The cheap
            // part (p, o2: a few FMAs and two cross-group sums) runs for every cluster in all lanes; the expensive scalar part
            // (erfc, logs) runs ONCE per four clusters, cluster k0 + g in lane group g, instead of once per cluster with a
            // quarter of the lanes enabled.
            for (int k0 = 0; k0 < K; k0 += 4) {
              float p_sel = 0.f, o2_sel = 0.f;
              for (int k = k0; k < min(K, k0 + 4); ++k) {
                const float* vk = hp + uniform(M->head.dirs_ke_phi) + k * E;
                f4 v[NTE];
                float p = 0.f;
#pragma unroll
                for (int t = 0; t < NTE; ++t) {
                    v[t] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int f = feat_of(t, j, g);
                        if (t < nte && f < E) {
                            v[t][j] = vk[f];
                            p += a[rt][t][j] * v[t][j];
                        }
                    }
                }
                p = group_sum(p);
                float o2 = 0.f;
#pragma unroll
                for (int t = 0; t < NTE; ++t)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (t < nte && feat_of(t, j, g) < E) {
                            const float d = a[rt][t][j] - p * v[t][j];
                            o2 += d * d;
                        }
                o2 = group_sum(o2);
                if (g == (k & 3)) { p_sel = p; o2_sel = o2; }
              }
              {
                const int k = k0 + g;
                const float p = p_sel, o2 = o2_sel;
                float lk = 0.f;
                if (tm[rt].valid && k < K) {
                    const float tau = hp[uniform(M->head.art_stdev_k_phi) + k];
                    const float mu = theta[uniform(M->head.mu_k_src) + k];
                    const float sg = hp[uniform(M->head.sigma_k_phi) + k];
                    const float lam = hp[uniform(M->head.lambda_k_phi) + k];
                    const float od = sqrtf(o2);
                    const float orth = -(0.5f * (float)(E - 1)) * PMT_LOG2PI - (float)(E - 1) * logf(tau) - (od * od) / (2.f * (tau * tau));
                    const float var = sg * sg;
                    const float zz = (mu + lam * var - p) / (1.4142135623730951f * sg);
                    const float par = logf(lam * 0.5f) + logerfc_dev(zz) + (lam * 0.5f) * (2.f * mu + lam * var - 2.f * p);
                    lk = orth + par;
                }
                const float s = seg_sum<SEG_GUARD>(lk, sp);  // (every lane of the row takes part in the scan)
                if (sp.last && k < K) atomicAdd(&sh.hsum[set][2 + k], s);
              }
            }
        }
    }
    tr.ev(22);
    lds_barrier();
    if constexpr (LAYERED) {  // the group's partial head sums join the global ones; pmt_finalize_kernel writes the outputs
        for (int i = tid; i < gg.nsets * (K + 2); i += PMT_THREADS) {
            const int set = i / (K + 2), k = i - set * (K + 2);
            const float v = sh.hsum[set][k];
            if (v != 0.f) atomicAdd(&lay.hsum_g[(size_t)(gg.v0 + set) * (PMT_MAX_CLUSTERS + 2) + k], v);
        }
        for (int i = tid; i < gg.nsets * 2 * PMT_MAX_WIDTH; i += PMT_THREADS) {
            const float v = (&sh.fsum[0][0][0])[i];
            if (v != 0.f) atomicAdd(&lay.fsum_g[(size_t)gg.v0 * 2 * PMT_MAX_WIDTH + i], v);
        }
        return;
    }

    tr.ev(23);
    // ---- per-set finalisation (reference feature_clustering.py:121-135, ragged_sets.py:144-155) -----------------
    for (int i = tid; i < gg.nsets; i += PMT_THREADS) {
        const int b = gg.v0 + i;
        float* lk = out.logits_bk + (size_t)b * (K + 2);
        const float l0 = sh.hsum[i][0];
        lk[0] = l0;
        lk[1] = sh.hsum[i][1];
        float mx = -INFINITY;
        for (int k = 0; k < K; ++k) {
            const float v = sh.hsum[i][2 + k] + phi[M->head.log_w_k_phi + k];
            lk[2 + k] = v;
            mx = fmaxf(mx, v);
        }
        float se = 0.f;
        for (int k = 0; k < K; ++k) se += expf(lk[2 + k] - mx);
        const float logit = (mx + logf(se)) - l0;
        out.logits_b[b] = PMT_MAX_LOGIT_F * tanhf(logit / PMT_MAX_LOGIT_F);
    }
    for (int i = tid; i < gg.nsets * 2 * E; i += PMT_THREADS) {
        const int set = i / (2 * E), rem = i - set * 2 * E, s = rem / E, f = rem - s * E;
        const int pos = 16 * (f >> 4) + 4 * (f & 3) + ((f & 15) >> 2);
        const float n = (float)(sh.off[s][set + 1] - sh.off[s][set]);
        const float v = sh.fsum[set][s][pos] / (n + 1e-4f);
        (s == 1 ? out.features_be : out.ref_features_be)[(size_t)(gg.v0 + set) * E + f] = v;
    }
}

// ---- translation units ---------------------------------------------------------------------------------------------------
// pmt_forward_train.hip includes this file with PMT_FORWARD_TRAIN_TU (and PMT_OPAQUE_TID 1) defined and compiles ONE instance,
// the training forward of the production shape; everything else -- the other instances, the layered path, the C entry points --
// is compiled here.
#ifdef PMT_FORWARD_TRAIN_TU
extern "C" int pmt_forward_launch_train_p0x(int groups, void* stream, const PmtModel* model_dev, const float* theta, const float* phi,
                                            const float* packed, const PmtBatch* batch, const PmtOutputs* out, float* stash,
                                            float* zsum_stash, float* rstd_stash, const PmtLayeredArgs* lay, int bf16x3) {
#if PMT_GENERIC_ONLY
    return PMT_E_UNSUPPORTED;  // (never asked for: pmt_shape_id is 0 in this build)
#else
    const hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const PmtLayeredArgs none{};
    // bf16x3 (PmtModel.force_shape = 5): the round-3 instances, six bf16 MFMAs per product; otherwise three f16 MFMAs
#define PMT_LAUNCH_FWD(TRAIN, SHAPE, LAYERED, LAY) \
    hipLaunchKernelGGL((pmt_forward_kernel<TRAIN, SHAPE, LAYERED>), dim3(groups), dim3(PMT_THREADS), 0, s, model_dev, theta, phi, packed, \
                       *batch, *out, stash, zsum_stash, rstd_stash, LAY)
    if (lay != nullptr && stash == nullptr) {  // one launch of the layered filter forward (it parks activations between launches: stores, too)
        if (bf16x3) PMT_LAUNCH_FWD(false, ShapeP0X, true, *lay); else PMT_LAUNCH_FWD(false, ShapeP0XH, true, *lay);
    } else if (lay != nullptr) {  // one launch of the layered training forward
        if (bf16x3) PMT_LAUNCH_FWD(true, ShapeP0X, true, *lay); else PMT_LAUNCH_FWD(true, ShapeP0XH, true, *lay);
    } else if (batch->dropout_seed != 0) {  // a training step with dropout (the caller checked the model's dropout_p): the instance with the masks
        if (bf16x3) PMT_LAUNCH_FWD(true, ShapeP0XD, false, none); else PMT_LAUNCH_FWD(true, ShapeP0XHD, false, none);
    } else {
        if (bf16x3) PMT_LAUNCH_FWD(true, ShapeP0X, false, none); else PMT_LAUNCH_FWD(true, ShapeP0XH, false, none);
    }
#undef PMT_LAUNCH_FWD
    return hipGetLastError() == hipSuccess ? PMT_OK : PMT_E_LAUNCH;
#endif
}
#else
extern "C" int pmt_forward_launch_train_p0x(int groups, void* stream, const PmtModel* model_dev, const float* theta, const float* phi,
                                            const float* packed, const PmtBatch* batch, const PmtOutputs* out, float* stash,
                                            float* zsum_stash, float* rstd_stash, const PmtLayeredArgs* lay, int bf16x3);

// Which instances of an exact-width shape run a batch.  The f16 instances (two f16 pieces per operand, MODE.FP16_OVFL, segmented
// scans without the non-finite guard) are built on what PACKED read rows guarantee: bits and k / 32 quantiles, nothing beyond 8,
// nothing non-finite.  float16 / float32 read rows (PMT_READS_F16 / _F32: a caller's own tensors) can hold anything -- an inf
// would be clamped to finite garbage without a trace, a NaN would pass v_cvt and leak through the unguarded scans into the sets
// that share its row of 16 lanes -- so they take the instances whose operands carry fp32's range and whose scans are
// guarded: three bf16 pieces (round 3's form, 1.3 x the time) or, for a tile-exact model, fp32 MFMAs.  The reference's semantics then
// hold: a non-finite read makes ITS variant's outputs non-finite and no other's (sets/ragged_sets.py:144-158).
static bool wide_range_operands(const PmtModel* m, const PmtBatch* b) { return m->force_shape == 5 || b->read_format != PMT_READS_PACKED_U8; }

// per-set outputs from the global sums of a layered forward (same arithmetic as the finalisation above)
__global__ __launch_bounds__(256) void pmt_finalize_kernel(const PmtModel* __restrict__ M, const float* __restrict__ phi, PmtBatch bt,
                                                           PmtOutputs out, const float* __restrict__ fsum_g, const float* __restrict__ hsum_g) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= bt.num_variants) return;
    const int K = M->num_clusters, E = M->feature_dim;
    const float* hs = hsum_g + (size_t)b * (PMT_MAX_CLUSTERS + 2);
    float* lk = out.logits_bk + (size_t)b * (K + 2);
    const float l0 = hs[0];
    lk[0] = l0;
    lk[1] = hs[1];
    float mx = -INFINITY;
    for (int k = 0; k < K; ++k) {
        const float v = hs[2 + k] + phi[M->head.log_w_k_phi + k];
        lk[2 + k] = v;
        mx = fmaxf(mx, v);
    }
    float se = 0.f;
    for (int k = 0; k < K; ++k) se += expf(lk[2 + k] - mx);
    const float logit = (mx + logf(se)) - l0;
    out.logits_b[b] = PMT_MAX_LOGIT_F * tanhf(logit / PMT_MAX_LOGIT_F);
    const float n_ref = (float)(bt.ref_offsets[b + 1] - bt.ref_offsets[b]), n_alt = (float)(bt.alt_offsets[b + 1] - bt.alt_offsets[b]);
    for (int f = 0; f < E; ++f) {
        const int pos = 16 * (f >> 4) + 4 * (f & 3) + ((f & 15) >> 2);
        out.ref_features_be[(size_t)b * E + f] = fsum_g[((size_t)b * 2 + 0) * PMT_MAX_WIDTH + pos] / (n_ref + 1e-4f);
        out.features_be[(size_t)b * E + f] = fsum_g[((size_t)b * 2 + 1) * PMT_MAX_WIDTH + pos] / (n_alt + 1e-4f);
    }
}

extern "C" size_t pmt_layered_scratch_floats(const PmtModel* m, int64_t total_tiles, int32_t num_variants) {
    if (!m) return 0;
    const size_t nb = (size_t)(m->num_blocks > 0 ? m->num_blocks : 1);
    // parked activations (layered launches only) | per-set sums | joined execution: arrival counters [B][L], ticket, fault word
    return (size_t)total_tiles * (PMT_SLOT_FLOATS + 512 * PMT_HT) + (size_t)num_variants * (nb * PMT_ZW + 2 * PMT_MAX_WIDTH + PMT_MAX_CLUSTERS + 2) +
           (size_t)num_variants * nb + 8;
}

extern "C" int pmt_forward_layered(const PmtModel* model_host, const PmtModel* model_dev, const float* theta, const float* phi,
                                   const float* packed, const PmtBatch* batch, const PmtOutputs* out, float* stash, float* scratch,
                                   void* stream) {
    if (!model_host || !model_dev || !batch || !out || !scratch) return PMT_E_INVALID;
    const int rc = pmt_model_check(model_host);
    if (rc != PMT_OK) return rc;
    if (batch->num_groups <= 0) return batch->num_groups == 0 ? PMT_OK : PMT_E_INVALID;
    if (!batch->reads || !batch->ref_offsets || !batch->alt_offsets || !batch->variant_embed || !batch->group_span ||
        !batch->group_tile_base || batch->total_tiles <= 0 || !out->logits_b || !out->logits_bk || !out->features_be || !out->ref_features_be)
        return PMT_E_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int shape = pmt_shape_for(model_host, batch, true);
    const bool p0 = shape == 1;
    const int L = model_host->num_blocks;
    const size_t nb = (size_t)(L > 0 ? L : 1), B = (size_t)batch->num_variants;
    PmtLayeredArgs lay;
    lay.x_scratch = scratch;
    lay.z_scratch = lay.x_scratch + (size_t)batch->total_tiles * PMT_SLOT_FLOATS;
    float* sums = lay.z_scratch + (size_t)batch->total_tiles * (512 * PMT_HT);
    float* zsum_stash = nullptr;
    float* rstd_stash = nullptr;
    if (stash) {
        zsum_stash = stash + (size_t)batch->total_tiles * (size_t)pmt_stash_slots(model_host) * PMT_SLOT_FLOATS;
        rstd_stash = zsum_stash + B * nb * PMT_ZW;
    }
    lay.zsum_g = stash ? zsum_stash : sums;  // training: the stash's per-set sums ARE the global sums
    lay.fsum_g = sums + B * nb * PMT_ZW;
    lay.hsum_g = lay.fsum_g + B * 2 * PMT_MAX_WIDTH;
    if (hipMemsetAsync(lay.zsum_g, 0, B * nb * PMT_ZW * sizeof(float), s) != hipSuccess) return PMT_E_LAUNCH;
    if (hipMemsetAsync(lay.fsum_g, 0, B * (2 * PMT_MAX_WIDTH + PMT_MAX_CLUSTERS + 2) * sizeof(float), s) != hipSuccess) return PMT_E_LAUNCH;
    auto kernel = stash ? (p0 ? pmt_forward_kernel<true, ShapeP0, true> : pmt_forward_kernel<true, ShapeAny, true>)
                        : (p0 ? pmt_forward_kernel<false, ShapeP0, true> : pmt_forward_kernel<false, ShapeAny, true>);
    // joined execution: one launch in which the groups of a split read set exchange their per-set sums through HBM
    int* join_words = reinterpret_cast<int*>(lay.hsum_g + B * (PMT_MAX_CLUSTERS + 2));
    lay.join = PmtJoin{0, join_words + B * nb, join_words, batch->join_fault ? batch->join_fault : join_words + B * nb + 1};
    if (batch->set_groups != nullptr && L > 0) {
        lay.join.on = 1;
        if (hipMemsetAsync(join_words, 0, (B * nb + 8) * sizeof(int), s) != hipSuccess) return PMT_E_LAUNCH;
    }
    for (int slice = 0; slice <= (lay.join.on ? 0 : L); ++slice) {
        lay.slice = slice;
        if (shape >= 2) {  // the layered instances of the production shape: pmt_forward_train.hip  (layered: no plain-bf16 instance)
            const int rct = pmt_forward_launch_train_p0x(batch->num_groups, stream, model_dev, theta, phi, packed, batch, out, stash, zsum_stash,
                                                         rstd_stash, &lay, wide_range_operands(model_host, batch));
            if (rct != PMT_OK) return rct;
            continue;
        }
        hipLaunchKernelGGL(kernel, dim3(batch->num_groups), dim3(PMT_THREADS), 0, s, model_dev, theta, phi, packed, *batch, *out, stash,
                           zsum_stash, rstd_stash, lay);
    }
    hipLaunchKernelGGL(pmt_finalize_kernel, dim3((batch->num_variants + 255) / 256), dim3(256), 0, s, model_dev, phi, *batch, *out,
                       lay.fsum_g, lay.hsum_g);
    return hipGetLastError() == hipSuccess ? PMT_OK : PMT_E_LAUNCH;
}

// ---- host launcher ---------------------------------------------------------------------------------------------------
extern "C" int pmt_forward(const PmtModel* model_host, const PmtModel* model_dev, const float* theta, const float* phi,
                           const float* packed, const PmtBatch* batch, const PmtOutputs* out, float* stash,
                           void* stream) {
    if (!model_host || !model_dev || !batch || !out) return PMT_E_INVALID;
    const int rc = pmt_model_check(model_host);
    if (rc != PMT_OK) return rc;
    if (batch->num_groups <= 0) return batch->num_groups == 0 ? PMT_OK : PMT_E_INVALID;
    if (!batch->reads || !batch->ref_offsets || !batch->alt_offsets || !batch->variant_embed || !batch->group_start ||
        !out->logits_b || !out->logits_bk || !out->features_be || !out->ref_features_be)
        return PMT_E_INVALID;
    if (batch->group_span) return PMT_E_UNSUPPORTED;  // split read sets: pmt_forward_layered
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    int shape = pmt_shape_for(model_host, batch);  // pmt_device.hpp: 2 ShapeP0X, 1 ShapeP0, 0 ShapeAny, 4 ShapeP0XD (dropout step)
    if (shape == 4 && !stash) shape = 0;            // (train mode without a backward to follow: the generic instance)
    const bool p0 = shape == 1;
    float* zsum_stash = nullptr;
    float* rstd_stash = nullptr;
    if (stash) {
        if (!batch->group_tile_base || batch->total_tiles <= 0) return PMT_E_INVALID;
        // per-set z2 sums follow the per-tile activation slots (layout: pmt_stash_bytes)
        zsum_stash = stash + (size_t)batch->total_tiles * (size_t)pmt_stash_slots(model_host) * PMT_SLOT_FLOATS;
        rstd_stash = zsum_stash + (size_t)batch->num_variants * (size_t)(model_host->num_blocks > 0 ? model_host->num_blocks : 1) * PMT_ZW;
    }
    auto kernel = stash ? (p0 ? pmt_forward_kernel<true, ShapeP0> : pmt_forward_kernel<true, ShapeAny>)
                        : (p0 ? pmt_forward_kernel<false, ShapeP0> : pmt_forward_kernel<false, ShapeAny>);
    const bool wide_range = wide_range_operands(model_host, batch);
    if ((shape == 2 || shape == 4) && stash)  // its own translation unit (pmt_forward_train.hip); 4: with the step's dropout masks
        return pmt_forward_launch_train_p0x(batch->num_groups, stream, model_dev, theta, phi, packed, batch, out, stash, zsum_stash, rstd_stash, nullptr,
                                            wide_range);
    if (shape == 2) kernel = wide_range ? pmt_forward_kernel<false, ShapeP0X> : pmt_forward_kernel<false, ShapeP0XH>;
    // the shape's tiles, widths at run time: f16 pieces -- or, for read rows that can hold anything, the fp32 tile-exact instance (`kernel` above)
    if (shape == 6 && !wide_range) kernel = stash ? pmt_forward_kernel<true, ShapeP0TH> : pmt_forward_kernel<false, ShapeP0TH>;
    if (shape == 6 && wide_range) kernel = stash ? pmt_forward_kernel<true, ShapeP0> : pmt_forward_kernel<false, ShapeP0>;
    if (shape == 3) kernel = stash ? pmt_forward_kernel<true, ShapeP0XB> : pmt_forward_kernel<false, ShapeP0XB>;  // plain bf16 products
    hipLaunchKernelGGL(kernel, dim3(batch->num_groups), dim3(PMT_THREADS), 0, s, model_dev, theta, phi, packed, *batch, *out,
                       stash, zsum_stash, rstd_stash, PmtLayeredArgs{});
    return hipGetLastError() == hipSuccess ? PMT_OK : PMT_E_LAUNCH;
}
#endif  // PMT_FORWARD_TRAIN_TU
