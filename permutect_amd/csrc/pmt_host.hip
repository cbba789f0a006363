// Host helpers of the C ABI plus the small utility kernels: parameter re-pack into MFMA fragment order, exclusive
// scans of the per-variant read counts, fused global-norm clip + AdamW.
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>

#include <thread>
#include <vector>

#include "pmt_device.hpp"

extern "C" int pmt_abi_version(void) { return PMT_ABI_VERSION; }
#ifndef PMT_BUILD_ID
#define PMT_BUILD_ID "0000000000000000"  // (csrc/Makefile passes the hash of the sources)
#endif
extern "C" int pmt_build_id(char* out, int32_t capacity) {
    static const char tagged[] = "PMT_BUILD_ID=" PMT_BUILD_ID;  // (the tag: engine/instances.py reads the id out of the FILE of a library it has not mapped)
    const char* id = tagged + 13;
    const int n = (int)sizeof(tagged) - 13;
    if (!out || capacity < n) return PMT_E_INVALID;
    memcpy(out, id, n);
    return n - 1;
}

extern "C" int pmt_struct_bytes(int which) {
    switch (which) {
        case 0: return (int)sizeof(PmtModel);
        case 1: return (int)sizeof(PmtBatch);
        case 2: return (int)sizeof(PmtOutputs);
        case 3: return (int)sizeof(PmtOutputGrads);
        case 4: return (int)sizeof(PmtAdamW);
        case 5: return (int)sizeof(PmtLinear);
        case 6: return (int)sizeof(PmtOp);
        case 7: return (int)sizeof(PmtMlp);
        case 8: return (int)sizeof(PmtBlock);
        case 9: return (int)sizeof(PmtHead);
        case 10: return (int)sizeof(PmtPhiProgram);
        case 11: return (int)sizeof(PmtLossArgs);
        case 12: return (int)sizeof(PmtDownsample);
        case 13: return (int)sizeof(PmtRecordArgs);
        case 14: return (int)sizeof(PmtBalanceArgs);
        case 15: return (int)sizeof(PmtEvalArgs);
        default: return PMT_E_INVALID;
    }
}

extern "C" int pmt_stash_slots(const PmtModel* m) { return (m->read_mlp.n_ops - 1) + (m->num_blocks + 1) + (m->reducer.n_ops - 1) + m->num_blocks; }

// Which register-array shape the read-set kernels run with (pmt_device.hpp: Shape; the tile counts and widths of the exact
// instances are the BUILD's shape, PMT_SH_*: the production hyperparameters by default).
//   0 = ShapeAny: any supported model (fp32 MFMAs, every tile guarded);
//   1 = ShapeP0: every layer fills the shape's tile arrays exactly -- read features and the first read linear NTF -> NTR tiles, the
//       rest of the read MLP NTR wide, d_model and the reducer NTD wide up to a last LINEAR NTD -> NTE, feature_dim NTE -- fp32
//       MFMAs, widths at run time (PmtModel.force_shape = 1 only);
//   6 = ShapeP0T / ShapeP0TH: the same tile-exact models on the 16-bit matrix pipes, widths at run time;
//   2 = ShapeP0X / ShapeP0XH: additionally every width equals the build's (compiled in); 3 = the same with plain bf16 products.
// PmtModel.force_shape = 2 / 1 forces the generic / the fp32 tile-exact instance (the parity tests cover all).
extern "C" int pmt_shape_info(int32_t* nine) {  // the build's shape: tile counts NTF, NTR, NTD, NTE, then widths F, R, D, H, E
    const int32_t v[9] = {PMT_SH_NTF, PMT_SH_NTR, PMT_SH_NTD, PMT_SH_NTE, PMT_SH_F, PMT_SH_R, PMT_SH_D, PMT_SH_H, PMT_SH_E};
    if (nine) memcpy(nine, v, sizeof(v));
    return PMT_OK;
}
static int tiles_of(int dim) { return (dim + 15) / 16; }
static bool mlp_ops_have_width(const PmtModel* m, const PmtMlp* mlp, int first, int last, int width) {
    for (int i = first; i < last; ++i) {
        const PmtOp* o = &mlp->ops[i];
        const int nl = o->kind == PMT_OP_SKIP ? o->n_layers : 1;
        for (int k = 0; k < nl; ++k) {
            const PmtLinear* L = &m->lin[o->lin[k]];
            if (L->in_dim != width || L->out_dim != width) return false;
        }
    }
    return true;
}
static bool mlp_ops_have_tiles(const PmtModel* m, const PmtMlp* mlp, int first, int last, int nt) {
    for (int i = first; i < last; ++i) {
        const PmtOp* o = &mlp->ops[i];
        const int nl = o->kind == PMT_OP_SKIP ? o->n_layers : 1;
        if (nl > 2) return false;  // (skip blocks of three and four layers: the generic instances' interpreter only)
        for (int k = 0; k < nl; ++k) {
            const PmtLinear* L = &m->lin[o->lin[k]];
            if (tiles_of(L->in_dim) != nt || tiles_of(L->out_dim) != nt) return false;
        }
    }
    return true;
}
extern "C" int pmt_limits(int32_t* four) {  // the build's compile-time limits: widest activation, d_ffn / 2, floats of a stash slot, waves per group
    const int32_t v[4] = {PMT_MAX_WIDTH, PMT_MAX_HALF_FFN, PMT_MAX_WIDTH * 16, PMT_GROUP_WAVES};
    if (four) memcpy(four, v, sizeof(v));
    return PMT_OK;
}
extern "C" int pmt_shape_id(const PmtModel* m) {
    if (m->force_shape == 2 || PMT_GENERIC_ONLY) return 0;  // (5 = auto for every kernel but the forward, whose launchers read it themselves)
    const PmtMlp* rm = &m->read_mlp;
    const PmtMlp* red = &m->reducer;
    if (rm->n_ops < 1 || red->n_ops < 1 || m->num_blocks < 0) return 0;
    const PmtOp* first = &rm->ops[0];
    const PmtOp* last = &red->ops[red->n_ops - 1];
    if (first->kind != PMT_OP_LINEAR || last->kind != PMT_OP_LINEAR) return 0;
    const PmtLinear* Lf = &m->lin[first->lin[0]];
    const PmtLinear* Ll = &m->lin[last->lin[0]];
    const bool ok = tiles_of(m->num_read_features) == PMT_SH_NTF && tiles_of(Lf->in_dim) == PMT_SH_NTF && tiles_of(Lf->out_dim) == PMT_SH_NTR &&
                    mlp_ops_have_tiles(m, rm, 1, rm->n_ops, PMT_SH_NTR) && tiles_of(m->read_embed_dim) == PMT_SH_NTR &&
                    tiles_of(m->d_model) == PMT_SH_NTD && m->d_ffn >= 2 && mlp_ops_have_tiles(m, red, 0, red->n_ops - 1, PMT_SH_NTD) &&
                    tiles_of(Ll->in_dim) == PMT_SH_NTD && tiles_of(Ll->out_dim) == PMT_SH_NTE && tiles_of(m->feature_dim) == PMT_SH_NTE;
    if (!ok) return 0;
    if (m->force_shape == 1) return 1;
    const bool exact = m->num_read_features == PMT_SH_F && Lf->in_dim == PMT_SH_F && Lf->out_dim == PMT_SH_R &&
                       mlp_ops_have_width(m, rm, 1, rm->n_ops, PMT_SH_R) && m->read_embed_dim == PMT_SH_R && m->d_model == PMT_SH_D &&
                       m->d_ffn == 2 * PMT_SH_H && mlp_ops_have_width(m, red, 0, red->n_ops - 1, PMT_SH_D) && Ll->in_dim == PMT_SH_D &&
                       Ll->out_dim == PMT_SH_E && m->feature_dim == PMT_SH_E;
    if (!exact) return 6;  // the shape's tiles, other widths: the 16-bit pipes with the widths read at run time
    return m->force_shape == 3 ? 3 : 2;  // 3: the exact widths with plain bf16 products (asked for explicitly only)
}

// ---------------------------------------------------------------------------------------------------------------------
// descriptor validation
// ---------------------------------------------------------------------------------------------------------------------
static int check_linear(const PmtModel* m, int id, int in_dim, int out_dim, int max_in = PMT_MAX_WIDTH) {
    if (id < 0 || id >= m->n_linear) return PMT_E_INVALID;
    const PmtLinear* l = &m->lin[id];
    if (l->in_dim != in_dim || l->out_dim != out_dim) return PMT_E_INVALID;
    if (in_dim < 1 || out_dim < 1 || in_dim > max_in || out_dim > PMT_MAX_WIDTH) return PMT_E_UNSUPPORTED;
    if (l->w_frag < 0 || l->wt_frag < 0 || (l->w_frag & 3) || (l->wt_frag & 3)) return PMT_E_INVALID;
    if (l->b_pvec >= 0 && (l->b_pvec & 3)) return PMT_E_INVALID;
    return PMT_OK;
}

static int check_mlp(const PmtModel* m, const PmtMlp* mlp, int max_in = PMT_MAX_WIDTH) {
    if (mlp->n_ops < 1 || mlp->n_ops > PMT_MAX_OPS) return PMT_E_UNSUPPORTED;
    int width = mlp->in_dim;
    if (width > PMT_MAX_WIDTH && mlp->ops[0].kind != PMT_OP_LINEAR) return PMT_E_UNSUPPORTED;
    for (int i = 0; i < mlp->n_ops; ++i) {
        const PmtOp* o = &mlp->ops[i];
        if (o->kind == PMT_OP_LINEAR) {
            if (o->lin[0] < 0 || o->lin[0] >= m->n_linear) return PMT_E_INVALID;
            const int out = m->lin[o->lin[0]].out_dim;
            const int rc = check_linear(m, o->lin[0], width, out, i == 0 ? max_in : PMT_MAX_WIDTH);
            if (rc) return rc;
            width = out;
        } else if (o->kind == PMT_OP_SKIP) {
            if (o->n_layers < 1 || o->n_layers > PMT_MAX_SKIP_LAYERS) return PMT_E_UNSUPPORTED;  /* (three and four: the generic instances, pmt_shape_id) */
            if (o->alpha_src < 0) return PMT_E_INVALID;
            for (int k = 0; k < o->n_layers; ++k) {
                const int rc = check_linear(m, o->lin[k], width, width);
                if (rc) return rc;
                if (m->lin[o->lin[k]].b_pvec < 0) return PMT_E_INVALID;
            }
        } else {
            return PMT_E_INVALID;
        }
    }
    return width == mlp->out_dim ? PMT_OK : PMT_E_INVALID;
}

extern "C" int pmt_model_check(const PmtModel* m) {
    if (!m || m->abi_version != PMT_ABI_VERSION) return PMT_E_INVALID;
    if (m->n_linear < 1 || m->n_linear > PMT_MAX_LINEAR) return PMT_E_UNSUPPORTED;
    if (m->num_read_features < 1 || m->num_read_features > PMT_MAX_WIDTH) return PMT_E_UNSUPPORTED;
    if (m->d_model != m->read_embed_dim + m->variant_embed_dim || m->d_model > PMT_MAX_WIDTH) return PMT_E_UNSUPPORTED;
    if (m->d_ffn < 2 || (m->d_ffn & 1) || m->d_ffn / 2 > PMT_MAX_HALF_FFN) return PMT_E_UNSUPPORTED;
    if (m->num_blocks > 0 && (m->d_ffn / 2 + 15) / 16 != PMT_HT) return PMT_E_UNSUPPORTED;  // (the two-tile layout of a PMT_MAX_HALF_FFN = 32 build is not the one-tile model's)
    if (m->num_blocks < 0 || m->num_blocks > PMT_MAX_BLOCKS) return PMT_E_UNSUPPORTED;
    if (m->feature_dim < 2 || m->feature_dim > PMT_MAX_WIDTH) return PMT_E_UNSUPPORTED;
    if (m->num_clusters < 1 || m->num_clusters > PMT_MAX_CLUSTERS) return PMT_E_UNSUPPORTED;
    if (!(m->dropout_p >= 0.f && m->dropout_p < 1.f)) return PMT_E_INVALID;
    if (m->read_mlp.in_dim != m->num_read_features || m->read_mlp.out_dim != m->read_embed_dim) return PMT_E_INVALID;
    if (m->reducer.in_dim != m->d_model || m->reducer.out_dim != m->feature_dim) return PMT_E_INVALID;
    int rc = check_mlp(m, &m->read_mlp);
    if (rc) return rc;
    rc = check_mlp(m, &m->reducer);
    if (rc) return rc;
    for (int k = 0; k < 3; ++k) {
        if (m->row_mlp[k].n_ops == 0 && k == PMT_ROWS_SOURCE) continue;
        if ((rc = check_mlp(m, &m->row_mlp[k], PMT_MAX_ROW_INPUT))) return rc;
    }
    const int h = m->d_ffn / 2;
    for (int l = 0; l < m->num_blocks; ++l) {
        const PmtBlock* b = &m->blocks[l];
        for (int s = 0; s < 2; ++s) {
            if ((rc = check_linear(m, b->proj1[s], m->d_model, m->d_ffn))) return rc;
            if (m->lin[b->proj1[s]].out_split != h || m->lin[b->proj1[s]].b_pvec < 0) return PMT_E_INVALID;
            if ((rc = check_linear(m, b->proj2[s], h, m->d_model))) return rc;
            if (m->lin[b->proj2[s]].b_pvec < 0) return PMT_E_INVALID;
        }
    }
    if ((rc = check_linear(m, m->rotation_lin, m->feature_dim, m->feature_dim))) return rc;
    return PMT_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// group planning (host)
// ---------------------------------------------------------------------------------------------------------------------
// ref tiles and alt tiles go to disjoint waves, PMT_GROUP_TILES / PMT_GROUP_WAVES tiles per wave (group_geometry)
static bool group_fits(long long ref_reads, long long alt_reads) {
    const long long per = PMT_GROUP_TILES / PMT_GROUP_WAVES;
    const long long tr = (ref_reads + 15) / 16, ta = (alt_reads + 15) / 16;
    return (tr + per - 1) / per + (ta + per - 1) / per <= PMT_GROUP_WAVES;
}

extern "C" int pmt_plan_groups(const int32_t* ref_counts, const int32_t* alt_counts, int32_t num_variants,
                               int32_t* group_start, int32_t* group_tile_base, int32_t* bad_variant) {
    if (!ref_counts || !alt_counts || !group_start || !group_tile_base || num_variants < 0) return PMT_E_INVALID;
    int groups = 0;
    long long ref = 0, alt = 0;
    int sets = 0;
    long long tile_base = 0;
    group_start[0] = 0;
    group_tile_base[0] = 0;
    for (int b = 0; b < num_variants; ++b) {
        const long long r = ref_counts[b], a = alt_counts[b];
        if (r < 0 || a < 0) return PMT_E_INVALID;
        if (!group_fits(r, a)) {
            if (bad_variant) *bad_variant = b;
            return PMT_E_CAPACITY;
        }
        if (sets > 0 && (!group_fits(ref + r, alt + a) || sets + 1 > PMT_GROUP_MAX_SETS)) {
            tile_base += (ref + 15) / 16 + (alt + 15) / 16;
            ++groups;
            group_start[groups] = b;
            group_tile_base[groups] = (int32_t)tile_base;
            ref = alt = 0;
            sets = 0;
        }
        ref += r;
        alt += a;
        ++sets;
    }
    if (sets > 0) {
        tile_base += (ref + 15) / 16 + (alt + 15) / 16;
        ++groups;
        group_start[groups] = num_variants;
        group_tile_base[groups] = (int32_t)tile_base;
    }
    return groups;
}

// ---- pmt_plan_groups ON THE DEVICE ---------------------------------------------------------------------------------------
// A DownsampledBatch keeps about half of its parent's reads (reference training/downsampler.py: a mixture of Beta fractions with mean
// one half), and how many is decided on the device (pmt_downsample_counts).  Until round 5 it ran on its PARENT's plan -- the parent's
// counts bound its own -- so every training step of the real loop launched the parent's ~3 400 workgroups with half-empty tiles and
// took the parent's time (a workgroup costs the same full or not).  Planning from the downsampled counts needs them where the
// planner runs; bringing them to the host would be a synchronisation per step, so the planner goes to the device instead:
// the batch is cut into chunks of consecutive variants, one thread packs a chunk exactly like pmt_plan_groups (next-fit over
// consecutive variants, the same group_fits and set limit; a chunk boundary closes a group: ~1 / 13 of a group per 256 variants
// lost), a prefix over the chunks' group and tile counts places their groups, and a second pass writes them.  The kernels take the
// group count from the device (PmtBatch.num_groups_dev) under a grid sized for a capacity.
// Next-fit over the same order never makes MORE groups when the items shrink (by induction the k-th boundary does not move left),
// so capacity = the parent's groups + the number of chunks; an overflow (a caller's wrong capacity) raises the fault word.
// One WAVE per chunk of 256 consecutive variants: its lanes load the chunk's counts coalesced (four registers per side), and the
// packing itself -- sequential by nature -- runs on the scalar unit, every lane in step, reading one variant's counts at a time out
// of the registers with v_readlane (~15 scalar instructions per variant: ~2 us per chunk).  Two launches: the chunks' group and
// tile counts, then (after a prefix over them, which every wave takes for itself) the groups written out.  The first version gave a
// chunk to one THREAD of a single workgroup, whose 256 dependent, uncoalesced loads took 174 us per 65 536-variant batch.
#define PLAN_CHUNK 256
#define PLAN_WAVES 4
__device__ __forceinline__ bool group_fits_reads(int ref_reads, int alt_reads) {
    const int per = PMT_GROUP_TILES / PMT_GROUP_WAVES;
    const int tr = (ref_reads + 15) >> 4, ta = (alt_reads + 15) >> 4;
    return (tr + per - 1) / per + (ta + per - 1) / per <= PMT_GROUP_WAVES;
}
// next-fit over the chunk [v0, v1); WRITE: groups g0 + 1 .. go to group_start / group_tile_base (lane 0 stores).  Wave-uniform throughout.
template <bool WRITE>
__device__ void plan_chunk_wave(const int* __restrict__ ro, const int* __restrict__ ao, int v0, int v1, int g0, int t0, int* __restrict__ group_start,
                                int* __restrict__ group_tile_base, int capacity, int& groups, int& tiles, int& bad) {
    const int lane = threadIdx.x & 63;
    int rc[PLAN_CHUNK / 64], ac[PLAN_CHUNK / 64];
#pragma unroll
    for (int k = 0; k < PLAN_CHUNK / 64; ++k) {
        const int b = v0 + 64 * k + lane;
        rc[k] = b < v1 ? ro[b + 1] - ro[b] : 0;
        ac[k] = b < v1 ? ao[b + 1] - ao[b] : 0;
    }
    int ref = 0, alt = 0, sets = 0, g = g0, t = t0;
    const int n = v1 - v0;
#pragma unroll
    for (int k = 0; k < PLAN_CHUNK / 64; ++k) {
        const int m = min(64, n - 64 * k);
        for (int l = 0; l < m; ++l) {
            const int r = __builtin_amdgcn_readlane(rc[k], l), a = __builtin_amdgcn_readlane(ac[k], l);
            if (!group_fits_reads(r, a)) bad = 1;  // (a read set beyond one workgroup: the caller's batch is not one for this planner)
            if (sets > 0 && (!group_fits_reads(ref + r, alt + a) || sets + 1 > PMT_GROUP_MAX_SETS)) {
                t += ((ref + 15) >> 4) + ((alt + 15) >> 4);
                ++g;
                if (WRITE && g <= capacity && lane == 0) { group_start[g] = v0 + 64 * k + l; group_tile_base[g] = t; }
                ref = alt = 0;
                sets = 0;
            }
            ref += r; alt += a; ++sets;
        }
    }
    if (sets > 0) {
        t += ((ref + 15) >> 4) + ((alt + 15) >> 4);
        ++g;
        if (WRITE && g <= capacity && lane == 0) { group_start[g] = v1; group_tile_base[g] = t; }
    }
    groups = g - g0;
    tiles = t - t0;
}
// launch 1: per chunk (groups, tiles) -> counts[2 c], counts[2 c + 1]
__global__ __launch_bounds__(64 * PLAN_WAVES) void pmt_plan_count_kernel(const int* __restrict__ ro, const int* __restrict__ ao, int n, int nchunks,
                                                                          int* __restrict__ counts) {
    const int c = blockIdx.x * PLAN_WAVES + (threadIdx.x >> 6);
    if (c >= nchunks) return;
    int groups = 0, tiles = 0, bad = 0;
    plan_chunk_wave<false>(ro, ao, c * PLAN_CHUNK, min(n, (c + 1) * PLAN_CHUNK), 0, 0, nullptr, nullptr, 0, groups, tiles, bad);
    if ((threadIdx.x & 63) == 0) {
        counts[2 * c] = groups;
        counts[2 * c + 1] = tiles | (bad ? (1 << 30) : 0);
    }
}
// launch 2: every wave sums the chunks in front of its own (coalesced, a wave reduction), then packs its chunk again, writing
__global__ __launch_bounds__(64 * PLAN_WAVES) void pmt_plan_write_kernel(const int* __restrict__ ro, const int* __restrict__ ao, int n, int nchunks,
                                                                          const int* __restrict__ counts, int* __restrict__ group_start,
                                                                          int* __restrict__ group_tile_base, int capacity,
                                                                          int* __restrict__ num_groups_dev, int* __restrict__ fault) {
    const int c = blockIdx.x * PLAN_WAVES + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= nchunks) return;
    int g0 = 0, t0 = 0, total = 0, bad = 0;
    for (int i = lane; i < nchunks; i += 64) {
        const int gi = counts[2 * i], ti = counts[2 * i + 1];
        bad |= ti >> 30;
        total += gi;
        if (i < c) { g0 += gi; t0 += ti & ((1 << 30) - 1); }
    }
    for (int d = 32; d > 0; d >>= 1) {
        g0 += __shfl_xor(g0, d); t0 += __shfl_xor(t0, d); total += __shfl_xor(total, d); bad |= __shfl_xor(bad, d);
    }
    if (c == 0 && lane == 0) {
        group_start[0] = 0;
        group_tile_base[0] = 0;
        num_groups_dev[0] = min(total, capacity);
        if ((total > capacity || bad) && fault != nullptr) atomicOr(fault, PMT_FAULT_PLAN);
    }
    int groups = 0, tiles = 0, bad2 = 0;
    plan_chunk_wave<true>(ro, ao, c * PLAN_CHUNK, min(n, (c + 1) * PLAN_CHUNK), g0, t0, group_start, group_tile_base, capacity, groups, tiles, bad2);
}

extern "C" int pmt_plan_device_chunks(int32_t num_variants) {  // how many chunks pmt_plan_groups_device cuts a batch into (capacity = the parent's groups + this)
    return (num_variants + PLAN_CHUNK - 1) / PLAN_CHUNK;
}
extern "C" int pmt_plan_groups_device(const int32_t* ref_offsets, const int32_t* alt_offsets, int32_t num_variants, int32_t* group_start,
                                      int32_t* group_tile_base, int32_t capacity, int32_t* num_groups_dev, int32_t* fault, int32_t* scratch,
                                      void* stream) {
    if (!ref_offsets || !alt_offsets || !group_start || !group_tile_base || !num_groups_dev || !scratch || num_variants < 1 || capacity < 1) return PMT_E_INVALID;
    const int nchunks = pmt_plan_device_chunks(num_variants);
    const hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid((nchunks + PLAN_WAVES - 1) / PLAN_WAVES), block(64 * PLAN_WAVES);
    hipLaunchKernelGGL(pmt_plan_count_kernel, grid, block, 0, s, ref_offsets, alt_offsets, num_variants, nchunks, scratch);
    hipLaunchKernelGGL(pmt_plan_write_kernel, grid, block, 0, s, ref_offsets, alt_offsets, num_variants, nchunks, scratch, group_start, group_tile_base,
                       capacity, num_groups_dev, fault);
    return hipGetLastError() == hipSuccess ? PMT_OK : PMT_E_LAUNCH;
}

// An ORDER of the batch's variants in which pmt_plan_groups packs fuller groups.  A workgroup costs the same whether its
// 16 tile slots are full or not, and the kernels run in rounds of (CUs x workgroups per CU) workgroups, so the number of
// groups is what the batch costs.  Taking the variants as they come fills a group to ~91 % (the next variant often does not
// fit the last wave of one side); here the group under construction may take, out of the next `window` unplaced variants,
// the first one that still fits: ~97 % (65 536 WGS-shaped variants: 3 643 -> 3 414 groups).  The caller composes the
// batch in the returned order (order[i] = index of the variant to put at position i); oversized variants are emitted when they
// become the oldest unplaced one and are left to pmt_plan_groups_split.
extern "C" int pmt_pack_order(const int32_t* ref_counts, const int32_t* alt_counts, int32_t num_variants, int32_t window,
                              int32_t* order) {
    if (!ref_counts || !alt_counts || !order || num_variants < 0 || window < 1) return PMT_E_INVALID;
    for (int i = 0; i < num_variants; ++i)
        if (ref_counts[i] < 0 || alt_counts[i] < 0) return PMT_E_INVALID;
    // `cand`: the (at most `window`) oldest unplaced variants, oldest first; `next`: the first variant not yet in it
    std::vector<int> cand;
    cand.reserve((size_t)window);
    int next = 0, placed = 0;
    auto refill = [&] {
        while ((int)cand.size() < window && next < num_variants) cand.push_back(next++);
    };
    refill();
    while (!cand.empty()) {
        long long R = 0, A = 0;
        int sets = 0;
        for (;;) {
            size_t found = cand.size();
            for (size_t c = 0; c < cand.size(); ++c) {
                const int i = cand[c];
                const long long r = ref_counts[i], a = alt_counts[i];
                // an oversized variant is a group (or several) of its own: it goes when it is the oldest one and the group is empty
                const bool oversized = !group_fits(r, a);
                if (oversized ? (sets == 0 && c == 0) : (sets < PMT_GROUP_MAX_SETS && group_fits(R + r, A + a))) {
                    found = c;
                    break;
                }
            }
            if (found == cand.size()) break;
            const int v = cand[found];
            cand.erase(cand.begin() + (long)found);
            order[placed++] = v;
            R += ref_counts[v];
            A += alt_counts[v];
            ++sets;
            refill();
            if (!group_fits(ref_counts[v], alt_counts[v])) break;  // (the oversized one closes its group)
        }
        if (sets == 0 && !cand.empty()) {  // the oldest is oversized but an ordinary group was being asked for: cannot happen
            order[placed++] = cand[0];     // with sets == 0 (it would have been taken); kept as a guard against a stall
            cand.erase(cand.begin());
            refill();
        }
    }
    return placed == num_variants ? PMT_OK : PMT_E_INVALID;
}

// The same for consecutive batches of `batch` variants at once (a chunk of the dataset), on `threads` host threads and
// outside the Python GIL: order[k * batch + i] = position, within the whole array, of the variant to put at place i of batch k.
extern "C" int pmt_pack_order_batches(const int32_t* ref_counts, const int32_t* alt_counts, int32_t num_variants, int32_t batch,
                                      int32_t window, int32_t threads, int32_t* order) {
    if (!ref_counts || !alt_counts || !order || num_variants < 0 || batch < 1 || window < 1) return PMT_E_INVALID;
    const int nb = (num_variants + batch - 1) / batch;
    if (threads < 1) threads = 1;
    if (threads > nb) threads = nb;
    std::vector<int> rc((size_t)(nb > 0 ? nb : 1), PMT_OK);
    auto work = [&](int t) {
        for (int k = t; k < nb; k += threads) {
            const int lo = k * batch, n = (lo + batch <= num_variants ? batch : num_variants - lo);
            rc[k] = pmt_pack_order(ref_counts + lo, alt_counts + lo, n, window, order + lo);
            for (int i = 0; i < n; ++i) order[lo + i] += lo;
        }
    };
    if (threads <= 1) {
        work(0);
    } else {
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; ++t) pool.emplace_back(work, t);
        for (auto& th : pool) th.join();
    }
    for (int k = 0; k < nb; ++k)
        if (rc[k] != PMT_OK) return rc[k];
    return PMT_OK;
}

// Everything the device chunk loader needs from the HOST for one chunk of the dataset, in one call outside the Python GIL
// (reference data/reads_dataset.py:141-196 does the same bookkeeping one Datum at a time in Python): the chunk's read counts,
// the order in which its variants are consumed (shuffled or not, then ordered inside every batch for the group packer), and
// every batch's group plan.  Returns the number of batches, PMT_E_CAPACITY when some read set needs the split plan (the caller
// then plans that chunk batch by batch with pmt_plan_groups_split), or another error.
//   ints / row_stride / ref_col / alt_col: the dataset's int16 table (row stride in elements) and its two count columns
//   ref_host, alt_host [n]: out, counts in chunk order
//   ids [n]: out, variant ids (0 .. n-1) in consumption order: batch k is ids[k * batch, (k + 1) * batch)
//   plans: out, per batch [group_start (g + 1 ints) | group_tile_base (g + 1 ints)], packed back to back; capacity in ints
//   batch_info [nb][4]: out, {offset of the batch's plan in `plans`, groups g, total tiles, total reads}
static inline uint64_t splitmix64(uint64_t& x) {
    uint64_t z = (x += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
extern "C" int pmt_prepare_chunk(const int16_t* ints, int64_t row_stride, int32_t ref_col, int32_t alt_col, int32_t n, int32_t shuffle,
                                 uint64_t seed, int32_t batch, int32_t window, int32_t threads, int32_t* ref_host, int32_t* alt_host,
                                 int64_t* ids, int32_t* plans, int64_t plans_capacity, int32_t* batch_info, int32_t* offsets) {
    if (!ints || !ref_host || !alt_host || !ids || !plans || !batch_info || n < 0 || batch < 1 || window < 1 || row_stride < 1) return PMT_E_INVALID;
    const int nb = (n + batch - 1) / batch;
    if (threads < 1) threads = 1;
    auto run = [&](int jobs, auto&& body) {  // body(job) for job in [0, jobs) on up to `threads` host threads
        const int nt = threads < jobs ? threads : jobs;
        if (nt <= 1) {
            for (int j = 0; j < jobs; ++j) body(j);
            return;
        }
        std::vector<std::thread> pool;
        for (int t = 0; t < nt; ++t)
            pool.emplace_back([&, t] { for (int j = t; j < jobs; j += nt) body(j); });
        for (auto& th : pool) th.join();
    };
    // the counts: two int16 per 100-byte row (a cache line per variant), gathered by all threads; the shuffle meanwhile
    std::thread shuffler([&] {
        for (int i = 0; i < n; ++i) ids[i] = i;
        if (shuffle) {  // Fisher-Yates on a splitmix64 stream; j = floor(u * (i + 1) / 2^64): unbiased to 2^-45 for i < 2^19
            uint64_t state = seed;
            for (int i = n - 1; i > 0; --i) {
                const int j = (int)(((unsigned __int128)splitmix64(state) * (uint64_t)(i + 1)) >> 64);
                const int64_t t = ids[i]; ids[i] = ids[j]; ids[j] = t;
            }
        }
    });
    const int slices = threads * 4;
    run(slices, [&](int sl) {
        const int lo = (int)((long long)n * sl / slices), hi = (int)((long long)n * (sl + 1) / slices);
        for (int i = lo; i < hi; ++i) {
            ref_host[i] = ints[(size_t)i * row_stride + ref_col];
            alt_host[i] = ints[(size_t)i * row_stride + alt_col];
        }
    });
    shuffler.join();
    for (int i = 0; i < n; ++i)
        if (ref_host[i] < 0 || alt_host[i] < 0) return PMT_E_INVALID;
    // Packing order inside every batch (pmt_pack_order is sequential: ~60 ns per variant).  A large batch is ordered in two
    // halves on two threads -- the only cost is one more partly filled group where the halves meet.
    const int parts = batch >= 16384 ? 2 : 1;
    std::vector<int> rc((size_t)(nb > 0 ? nb : 1) * parts, PMT_OK);
    run(nb * parts, [&](int job) {
        const int k = job / parts, part = job % parts;
        const int blo = k * batch, m = (blo + batch <= n ? batch : n - blo);
        const int lo = blo + (int)((long long)m * part / parts), cnt = blo + (int)((long long)m * (part + 1) / parts) - lo;
        std::vector<int32_t> r((size_t)cnt), a((size_t)cnt), order((size_t)cnt);
        std::vector<int64_t> tmp((size_t)cnt);
        for (int i = 0; i < cnt; ++i) { r[i] = ref_host[ids[lo + i]]; a[i] = alt_host[ids[lo + i]]; }
        const int e = pmt_pack_order(r.data(), a.data(), cnt, window, order.data());
        if (e != PMT_OK) { rc[job] = e; return; }
        for (int i = 0; i < cnt; ++i) tmp[i] = ids[lo + order[i]];
        for (int i = 0; i < cnt; ++i) ids[lo + i] = tmp[i];
    });
    for (size_t j = 0; j < rc.size(); ++j)
        if (rc[j] != PMT_OK) return rc[j];
    // every batch plans into its own slot of a scratch area (its size is not known beforehand), packed together afterwards
    const size_t slot = 2 * ((size_t)batch + 1);
    std::vector<int32_t> scratch((size_t)(nb > 0 ? nb : 1) * slot);
    std::vector<int> groups((size_t)(nb > 0 ? nb : 1), 0);
    run(nb, [&](int k) {
        const int lo = k * batch, m = (lo + batch <= n ? batch : n - lo);
        std::vector<int32_t> r((size_t)m), a((size_t)m);
        long long reads = 0;
        for (int i = 0; i < m; ++i) { r[i] = ref_host[ids[lo + i]]; a[i] = alt_host[ids[lo + i]]; reads += (long long)r[i] + a[i]; }
        int32_t* gs = scratch.data() + (size_t)k * slot;
        int32_t bad = -1;
        groups[k] = pmt_plan_groups(r.data(), a.data(), m, gs, gs + batch + 1, &bad);  // >= 0: the number of groups
        batch_info[4 * k + 3] = (int32_t)reads;
        if (offsets != nullptr) {  // the exclusive scans of the batch's counts, in its consumption order (pmt_compose_batch_planned)
            int32_t* ro = offsets + (size_t)k * 2 * ((size_t)batch + 1);
            int32_t* ao = ro + batch + 1;
            int32_t rs = 0, as = 0;
            for (int i = 0; i < m; ++i) { ro[i] = rs; ao[i] = as; rs += r[i]; as += a[i]; }
            ro[m] = rs; ao[m] = as;
        }
    });
    int64_t at = 0;
    for (int k = 0; k < nb; ++k) {
        if (groups[k] < 0) return groups[k];
        const int g = groups[k];
        if (at + 2 * ((int64_t)g + 1) > plans_capacity) return PMT_E_WORKSPACE;
        const int32_t* gs = scratch.data() + (size_t)k * slot;
        const int32_t* gt = gs + batch + 1;
        memcpy(plans + at, gs, sizeof(int32_t) * ((size_t)g + 1));
        memcpy(plans + at + g + 1, gt, sizeof(int32_t) * ((size_t)g + 1));
        batch_info[4 * k] = (int32_t)at; batch_info[4 * k + 1] = g; batch_info[4 * k + 2] = gt[g];
        at += 2 * ((int64_t)g + 1);
    }
    return nb;
}

extern "C" int pmt_plan_groups_split(const int32_t* ref_counts, const int32_t* alt_counts, int32_t num_variants, int32_t* span,
                                     int32_t* tile_base, int32_t max_groups, int32_t* needs_layered) {
    if (!ref_counts || !alt_counts || !span || !tile_base || !needs_layered || num_variants < 0 || max_groups < 1) return PMT_E_INVALID;
    const long long per = PMT_GROUP_TILES / PMT_GROUP_WAVES;
    int g = 0;
    long long tiles = 0;
    *needs_layered = 0;
    auto emit = [&](long long v0, long long v1, long long rb, long long re, long long ab, long long ae) {
        if (g >= max_groups) return false;
        int32_t* sp = span + 6 * (size_t)g;
        sp[0] = (int32_t)v0; sp[1] = (int32_t)v1; sp[2] = (int32_t)rb; sp[3] = (int32_t)re; sp[4] = (int32_t)ab; sp[5] = (int32_t)ae;
        tile_base[g] = (int32_t)tiles;
        tiles += (re - rb + 15) / 16 + (ae - ab + 15) / 16;
        ++g;
        return true;
    };
    bool oversized = false;
    for (int b = 0; b < num_variants; ++b) {
        if (ref_counts[b] < 0 || alt_counts[b] < 0) return PMT_E_INVALID;
        oversized = oversized || !group_fits(ref_counts[b], alt_counts[b]);
    }
    long long ref_row = 0, alt_row = 0;         // exclusive scans so far
    if (!oversized) {  // every set fits a workgroup: whole sets per group, exactly pmt_plan_groups
        long long gs = 0, gref0 = 0, galt0 = 0, gref = 0, galt = 0;
        int sets = 0;
        for (int b = 0; b < num_variants; ++b) {
            const long long r = ref_counts[b], a = alt_counts[b];
            if (sets > 0 && (!group_fits(gref + r, galt + a) || sets + 1 > PMT_GROUP_MAX_SETS)) {
                if (!emit(gs, b, gref0, ref_row, galt0, alt_row)) return PMT_E_WORKSPACE;
                gs = b; gref0 = ref_row; galt0 = alt_row; gref = galt = 0; sets = 0;
            }
            gref += r; galt += a; ++sets;
            ref_row += r; alt_row += a;
        }
        if (sets > 0 && !emit(gs, num_variants, gref0, ref_row, galt0, alt_row)) return PMT_E_WORKSPACE;
        tile_base[g] = (int32_t)tiles;
        return g;
    }
    // Some set exceeds a workgroup: the batch runs layered, where a group may hold ANY contiguous run of ref rows and of alt
    // rows (per-set sums are joined in HBM).  So the rows are STREAMED into groups -- a set's ref rows, then its alt rows, the
    // next set right behind -- and a group is closed only when nothing more fits: ~ceil(tiles / 16) groups instead of one
    // partly empty last group per oversized set (600-read sets: 3 groups of 16 tile slots for 38 tiles each before).
    *needs_layered = 1;
    auto waves_of = [&](long long rows) { return ((rows + 15) / 16 + per - 1) / per; };
    long long g_v0 = 0, g_rb = 0, g_ab = 0, gref = 0, galt = 0;
    int sets = 0;
    bool open = false;
    for (int b = 0; b < num_variants; ++b) {
        const long long r = ref_counts[b], a = alt_counts[b];
        long long rdone = 0, adone = 0;
        bool counted = false;  // set b has rows in the open group
        while (rdone < r || adone < a) {
            if (!open) {
                g_v0 = b; g_rb = ref_row + rdone; g_ab = alt_row + adone; gref = galt = 0; sets = 0;
                open = true; counted = false;
            }
            bool placed = false;
            if (counted || sets < PMT_GROUP_MAX_SETS) {
                if (rdone < r) {
                    const long long room = (PMT_GROUP_WAVES - waves_of(galt)) * per * 16 - gref;
                    const long long x = (r - rdone) < room ? (r - rdone) : room;
                    if (x > 0) { gref += x; rdone += x; placed = true; }
                }
                if (rdone == r && adone < a) {
                    const long long room = (PMT_GROUP_WAVES - waves_of(gref)) * per * 16 - galt;
                    const long long y = (a - adone) < room ? (a - adone) : room;
                    if (y > 0) { galt += y; adone += y; placed = true; }
                }
            }
            if (placed && !counted) { counted = true; ++sets; }
            if (rdone < r || adone < a) {  // the group is full (or has its sets): close it; set b continues in the next one
                if (!emit(g_v0, counted ? b + 1 : b, g_rb, g_rb + gref, g_ab, g_ab + galt)) return PMT_E_WORKSPACE;
                open = false;
            }
        }
        ref_row += r; alt_row += a;
    }
    if (open && gref + galt > 0 && !emit(g_v0, num_variants, g_rb, g_rb + gref, g_ab, g_ab + galt)) return PMT_E_WORKSPACE;
    tile_base[g] = (int32_t)tiles;
    return g;
}

extern "C" size_t pmt_stash_bytes(const PmtModel* m, int64_t total_tiles, int32_t num_variants) {
    if (!m) return 0;
    const size_t tile_part = (size_t)total_tiles * (size_t)pmt_stash_slots(m) * PMT_SLOT_FLOATS;
    const size_t nb = (size_t)(m->num_blocks > 0 ? m->num_blocks : 1);
    const size_t zsum_part = (size_t)num_variants * nb * PMT_ZW;   // per-set z2 sums of every block
    const size_t rstd_part = (size_t)total_tiles * nb * 16;    // LayerNorm(D) rstd of every read of every block
    return (tile_part + zsum_part + rstd_part) * sizeof(float);
}

// ---------------------------------------------------------------------------------------------------------------------
// parameter re-pack: natural layouts (theta / phi) -> MFMA A-fragment order and tile-position vectors
// ---------------------------------------------------------------------------------------------------------------------
// virtual row -> source row for a linear whose output rows are split into two 16-row tiles (out_split = h)
DEV int split_row(int v, int h) {
    if (h <= 0) return v;
    if (v < PMT_SPLIT0) return v < h ? v : -1;
    return (v - PMT_SPLIT0) < h ? h + (v - PMT_SPLIT0) : -1;
}

// one job per workgroup: (lin, 0) forward frags, (lin, 1) transposed frags, (lin, 2) bias, (lin, 3 / 4) the same two matrices
// as bf16 pieces, (lin, 5) the weight-gradient emit table, (lin, 6) the forward matrix as two f16 pieces, then per-block vectors
#define PMT_PACK_KINDS 7
__global__ void pmt_pack_kernel(const PmtModel* __restrict__ M, const float* __restrict__ theta,
                                const float* __restrict__ phi, float* __restrict__ packed) {
    const int job = blockIdx.x;
    const int n_lin_jobs = M->n_linear * PMT_PACK_KINDS;
    if (job < n_lin_jobs) {
        const PmtLinear& L = M->lin[job / PMT_PACK_KINDS];
        const int kind = job % PMT_PACK_KINDS;
        const int h = L.out_split;
        const int out_v = h > 0 ? PMT_SPLIT0 + h : L.out_dim;  // virtual output rows
        if (kind == 5) {
            // PmtLinear.emit_tab: destination of every element of every 16 x 16 block of dW as the matrix core leaves it
            // (C layout: lane (g, c) register j = row position 4 g + j, column position c; position p of a tile = feature
            // 4 (p & 3) + (p >> 2), pmt_device.hpp), then the bias gradient's destinations per out tile in position order
            if (L.emit_tab < 0) return;
            int* tab = reinterpret_cast<int*>(packed + L.emit_tab);
            const int nmt = (out_v + 15) >> 4, nkt = (L.in_dim + 15) >> 4;
            const int w_off = L.w_src >= 0 ? L.w_src : -(L.w_src + 2);
            for (int i = threadIdx.x; i < nmt * nkt * 256; i += blockDim.x) {
                const int j = i & 3, lane = (i >> 2) & 63, blk = i >> 8;
                const int ot = blk / nkt, it = blk - ot * nkt;
                const int pf = 16 * ot + 4 * j + (lane >> 4), o = pf < out_v ? split_row(pf, h) : -1;
                const int cp = lane & 15, col = 16 * it + 4 * (cp & 3) + (cp >> 2);
                tab[i] = (o >= 0 && o < L.out_dim && col < L.in_dim) ? w_off + o * L.in_dim + col : -1;
            }
            int* btab = tab + nmt * nkt * 256;
            for (int i = threadIdx.x; i < nmt * 16; i += blockDim.x) {
                const int p = i & 15, pf = 16 * (i >> 4) + 4 * (p & 3) + (p >> 2), o = pf < out_v ? split_row(pf, h) : -1;
                btab[i] = (L.b_src >= 0 && o >= 0 && o < L.out_dim) ? L.b_src + o : -1;
            }
            return;
        }
        if (kind == 6) {
            // PmtLinear.wh_frag: two f16 pieces per weight, (out tile, k block) order, the element order of the bf16 pieces
            // below.  hi = f16(w), lo = f16(2^12 (w - hi)): w - hi is exact in fp32 and at most 2^-11 |w|, so the scaled low
            // piece is a NORMAL f16 with all 11 bits wherever hi is normal (unscaled it would be a denormal for |w| < 1/4 --
            // every weight of a 60-wide layer); linear_acc_f16 keeps the products of the low pieces in an accumulator of
            // their own and scales it back.  |w| beyond the f16 range saturates at 65504 (no trained weight is near it).
            if (L.wh_frag < 0) return;
            const float* W = src_ptr(L.w_src, theta, phi);
            const int nmt = (out_v + 15) >> 4, nkt = (L.in_dim + 15) >> 4, nkb = (nkt + 1) >> 1;
            _Float16* dst = reinterpret_cast<_Float16*>(packed + L.wh_frag);
            const int total = nkb * nmt * 512;
            for (int i = threadIdx.x; i < total; i += blockDim.x) {
                const int e = i & 7, lane = (i >> 3) & 63, blk = i >> 9;
                const int kb = blk % nkb, mt = blk / nkb;
                const int m = lane & 15, kg = lane >> 4;
                const int mv = 16 * mt + 4 * (m & 3) + (m >> 2);
                const int kv = 16 * (2 * kb + (e >> 2)) + 4 * (e & 3) + kg;
                float val = 0.f;
                if (mv < out_v && kv < L.in_dim && (2 * kb + (e >> 2)) < nkt) {
                    const int orow = split_row(mv, h);
                    if (orow >= 0 && orow < L.out_dim) val = W[(size_t)orow * L.in_dim + kv];
                }
                val = fminf(fmaxf(val, -65504.f), 65504.f);
                const _Float16 hi = (_Float16)val;
                const _Float16 lo = (_Float16)((val - (float)hi) * 4096.f);
                const size_t base = (size_t)blk * 2 * 512 + lane * 8 + e;
                dst[base] = hi;
                dst[base + 512] = lo;
            }
            return;
        }
        if (kind >= 3) {
            // k block kb of 32 = activation tiles 2 kb and 2 kb + 1; lane (m, kg) element e: tile 2 kb + (e >> 2), register
            // e & 3 of lane group kg, i.e. the SAME feature order as the fp32 activation registers, so a layer's C-layout
            // output still feeds the next layer without any shuffle
            const int off = kind == 3 ? L.wb_frag : L.wtb_frag;
            if (off < 0) return;
            const float* W = src_ptr(L.w_src, theta, phi);
            const int Mv = kind == 3 ? out_v : L.in_dim, Kv = kind == 3 ? L.in_dim : out_v;
            const int nmt = (Mv + 15) >> 4, nkt = (Kv + 15) >> 4, nkb = (nkt + 1) >> 1;
            __bf16* dst = reinterpret_cast<__bf16*>(packed + off);
            const int total = nkb * nmt * 512;
            for (int i = threadIdx.x; i < total; i += blockDim.x) {
                const int e = i & 7, lane = (i >> 3) & 63, blk = i >> 9;
                const int mt = blk % nmt, kb = blk / nmt;
                const int m = lane & 15, kg = lane >> 4;
                const int mv = 16 * mt + 4 * (m & 3) + (m >> 2);
                const int kv = 16 * (2 * kb + (e >> 2)) + 4 * (e & 3) + kg;
                float val = 0.f;
                if (mv < Mv && kv < Kv && (2 * kb + (e >> 2)) < nkt) {
                    const int orow = kind == 3 ? split_row(mv, h) : split_row(kv, h);
                    const int icol = kind == 3 ? kv : mv;
                    if (orow >= 0 && orow < L.out_dim && icol < L.in_dim) val = W[(size_t)orow * L.in_dim + icol];
                }
                __bf16 hi, mid, lo;
                split_bf16x3(val, hi, mid, lo);
                const size_t base = (size_t)blk * 3 * 512 + lane * 8 + e;
                dst[base] = hi;
                dst[base + 512] = mid;
                dst[base + 1024] = lo;
            }
            return;
        }
        if (kind == 2) {
            if (L.b_pvec < 0) return;
            const float* b = src_ptr(L.b_src, theta, phi);
            const int n = ((out_v + 15) >> 4) * 16;
            for (int i = threadIdx.x; i < n; i += blockDim.x) {
                const int t = i >> 4, g = (i >> 2) & 3, j = i & 3;
                const int v = 16 * t + 4 * j + g;
                const int r = v < out_v ? split_row(v, h) : -1;
                packed[L.b_pvec + i] = (r >= 0 && r < L.out_dim) ? b[r] : 0.f;
            }
            return;
        }
        const float* W = src_ptr(L.w_src, theta, phi);
        // forward: A_v[m][k] = W[row(m)][k], Mv = out_v, Kv = in_dim; transposed: A_v[m][k] = W[row(k)][m]
        const int Mv = kind == 0 ? out_v : L.in_dim, Kv = kind == 0 ? L.in_dim : out_v;
        const int nmt = (Mv + 15) >> 4, nkt = (Kv + 15) >> 4;
        float* dst = packed + (kind == 0 ? L.w_frag : L.wt_frag);
        const int total = nmt * nkt * 256;
        for (int i = threadIdx.x; i < total; i += blockDim.x) {
            const int j = i & 3, lane = (i >> 2) & 63, tile = i >> 8;
            const int mt = tile % nmt, kt = tile / nmt;  // kt-major: the order linear_acc walks the fragments
            const int m = lane & 15, g = lane >> 4;
            const int mv = 16 * mt + 4 * (m & 3) + (m >> 2);
            const int kv = 16 * kt + 4 * j + g;
            float val = 0.f;
            if (mv < Mv && kv < Kv) {
                const int orow = kind == 0 ? split_row(mv, h) : split_row(kv, h);
                const int icol = kind == 0 ? kv : mv;
                if (orow >= 0 && orow < L.out_dim && icol < L.in_dim) val = W[(size_t)orow * L.in_dim + icol];
            }
            dst[i] = val;
        }
        return;
    }
    int v = job - n_lin_jobs;  // vector jobs
    int src = -1, dst = -1, n = 0;
    if (v < M->num_blocks * 5) {
        const PmtBlock& B = M->blocks[v / 5];
        const int hh = M->d_ffn / 2;
        switch (v % 5) {
            case 0: src = B.norm_w_src; dst = B.norm_w_pvec; n = M->d_model; break;
            case 1: src = B.norm_b_src; dst = B.norm_b_pvec; n = M->d_model; break;
            case 2: src = B.sgu_norm_w_src; dst = B.sgu_norm_w_pvec; n = hh; break;
            case 3: src = B.sgu_norm_b_src; dst = B.sgu_norm_b_pvec; n = hh; break;
            default: src = B.ref_reg_src; dst = B.ref_reg_pvec; n = hh; break;
        }
    } else {
        src = M->translation_src; dst = M->translation_pvec; n = M->feature_dim;
    }
    const int padded = ((n + 15) >> 4) * 16;
    const float* s = src_ptr(src, theta, phi);
    for (int i = threadIdx.x; i < padded; i += blockDim.x) {
        const int t = i >> 4, g = (i >> 2) & 3, j = i & 3;
        const int f = 16 * t + 4 * j + g;
        packed[dst + i] = f < n ? s[f] : 0.f;
    }
}

extern "C" int pmt_pack_params(const PmtModel* model_host, const PmtModel* model_dev, const float* theta,
                               const float* phi, float* packed, void* stream) {
    if (!model_host || !model_dev || !theta || !packed) return PMT_E_INVALID;
    const int rc = pmt_model_check(model_host);
    if (rc) return rc;
    const int jobs = model_host->n_linear * PMT_PACK_KINDS + model_host->num_blocks * 5 + 1;
    hipLaunchKernelGGL(pmt_pack_kernel, dim3(jobs), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), model_dev, theta,
                       phi, packed);
    return hipGetLastError() == hipSuccess ? PMT_OK : PMT_E_LAUNCH;
}

// ---------------------------------------------------------------------------------------------------------------------
// exclusive scans of ref / alt counts.  The counts sit in a strided column of the batch's integer table: every element in
// its own cache line (and, at 464 bytes per row, a new page every 9 rows), so the scan is bound by how many of those loads
// are in flight, and every element should be read exactly once.  Three small launches, no scratch memory:
//   1. every 1024-thread workgroup scans its own segment (local exclusive prefixes) and leaves the segment's TOTAL in the
//      first slot of the next segment (whose own local prefix is known to be 0); the last total goes to o[n];
//   2. one wave turns the totals in those slots into segment offsets (and completes o[n]);
//   3. every workgroup adds its offset to the rest of its segment.
// ---------------------------------------------------------------------------------------------------------------------
#define PMT_SCAN_SEG 4096
template <typename T>
__global__ __launch_bounds__(1024) void pmt_scan_local_kernel(const T* __restrict__ c0, const T* __restrict__ c1,
                                                              long long stride, int n, int* __restrict__ o0, int* __restrict__ o1) {
    const T* c = blockIdx.y == 0 ? c0 : c1;
    int* o = blockIdx.y == 0 ? o0 : o1;
    __shared__ int wave_tot[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int begin = blockIdx.x * PMT_SCAN_SEG, end = min(n, begin + PMT_SCAN_SEG);
    int v[4], local = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = begin + tid * 4 + k;
        v[k] = i < end ? (int)c[(size_t)i * stride] : 0;
        local += v[k];
    }
    int incl = local;  // inclusive scan of `local` across the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int up = __shfl_up(incl, d);
        if (lane >= d) incl += up;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int wave_prefix = 0, total = 0;
    for (int w = 0; w < 16; ++w) {
        if (w < wave) wave_prefix += wave_tot[w];
        total += wave_tot[w];
    }
    int run = wave_prefix + incl - local;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = begin + tid * 4 + k;
        if (i < end && i != begin) o[i] = run;  // (slot `begin` belongs to the previous segment's total until step 2)
        run += v[k];
    }
    if (tid == 0) {
        if (begin == 0) o[0] = 0;
        o[end == n ? n : begin + PMT_SCAN_SEG] = total;  // n > 0 here, so end == n only in the last segment
    }
}
__global__ __launch_bounds__(64) void pmt_scan_offsets_kernel(int n, int* __restrict__ o0, int* __restrict__ o1) {
    int* o = blockIdx.x == 0 ? o0 : o1;
    if (threadIdx.x != 0) return;  // at most n / 4096 totals: a serial walk of a few hundred L2 hits at the very most
    int run = 0;
    for (int s = PMT_SCAN_SEG; s < n; s += PMT_SCAN_SEG) {
        run += o[s];
        o[s] = run;
    }
    o[n] += run;  // (step 1 left the last segment's total there)
}
__global__ __launch_bounds__(1024) void pmt_scan_add_kernel(int n, int* __restrict__ o0, int* __restrict__ o1) {
    int* o = blockIdx.y == 0 ? o0 : o1;
    const int begin = (blockIdx.x + 1) * PMT_SCAN_SEG, end = min(n, begin + PMT_SCAN_SEG);
    const int off = o[begin];
    for (int i = begin + 1 + threadIdx.x; i < end; i += 1024) o[i] += off;
}

extern "C" int pmt_scan_counts(const void* ref_counts, const void* alt_counts, int32_t count_elem_bytes,
                               int64_t count_stride, int32_t num_variants, int32_t* ref_offsets, int32_t* alt_offsets,
                               void* stream) {
    if (!ref_counts || !alt_counts || !ref_offsets || !alt_offsets || num_variants < 0 || count_stride < 1) return PMT_E_INVALID;
    if (count_elem_bytes != 4 && count_elem_bytes != 8) return PMT_E_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (num_variants == 0) {
        if (hipMemsetAsync(ref_offsets, 0, sizeof(int32_t), s) != hipSuccess || hipMemsetAsync(alt_offsets, 0, sizeof(int32_t), s) != hipSuccess)
            return PMT_E_LAUNCH;
        return PMT_OK;
    }
    const int blocks = (num_variants + PMT_SCAN_SEG - 1) / PMT_SCAN_SEG;
    if (count_elem_bytes == 4)
        hipLaunchKernelGGL(pmt_scan_local_kernel<int32_t>, dim3(blocks, 2), dim3(1024), 0, s, (const int32_t*)ref_counts,
                           (const int32_t*)alt_counts, (long long)count_stride, num_variants, ref_offsets, alt_offsets);
    else
        hipLaunchKernelGGL(pmt_scan_local_kernel<int64_t>, dim3(blocks, 2), dim3(1024), 0, s, (const int64_t*)ref_counts,
                           (const int64_t*)alt_counts, (long long)count_stride, num_variants, ref_offsets, alt_offsets);
    if (blocks > 1) {
        hipLaunchKernelGGL(pmt_scan_offsets_kernel, dim3(2), dim3(64), 0, s, num_variants, ref_offsets, alt_offsets);
        hipLaunchKernelGGL(pmt_scan_add_kernel, dim3(blocks - 1, 2), dim3(1024), 0, s, num_variants, ref_offsets, alt_offsets);
    }
    return hipGetLastError() == hipSuccess ? PMT_OK : PMT_E_LAUNCH;
}

// ---------------------------------------------------------------------------------------------------------------------
// batch composition on the device: gather index of a batch drawn from a dataset chunk that is resident in HBM
// ---------------------------------------------------------------------------------------------------------------------
// On disk (reference data/memory_mapped_data.py:39-40) a datum's rows are its ref reads then its alt reads; a batch wants
// all ref rows of all its variants, then all alt rows (reference data/batch.py:45-47).  One workgroup per variant.
__global__ __launch_bounds__(64) void pmt_read_index_kernel(const long long* __restrict__ row_start, const int* __restrict__ ref_off,
                                                            const int* __restrict__ alt_off, int nb, long long* __restrict__ index) {
    const int b = blockIdx.x;
    const int r0 = ref_off[b], nr = ref_off[b + 1] - r0, a0 = alt_off[b], na = alt_off[b + 1] - a0;
    const long long total_ref = ref_off[nb], start = row_start[b];
    for (int i = threadIdx.x; i < nr; i += 64) index[r0 + i] = start + i;
    for (int i = threadIdx.x; i < na; i += 64) index[total_ref + a0 + i] = start + nr + i;
}

extern "C" int pmt_build_read_index(const int64_t* row_start, const int32_t* ref_offsets, const int32_t* alt_offsets,
                                    int32_t num_variants, int64_t* read_index, void* stream) {
    if (!row_start || !ref_offsets || !alt_offsets || !read_index || num_variants < 0) return PMT_E_INVALID;
    if (num_variants == 0) return PMT_OK;
    hipLaunchKernelGGL(pmt_read_index_kernel, dim3(num_variants), dim3(64), 0, reinterpret_cast<hipStream_t>(stream),
                       (const long long*)row_start, ref_offsets, alt_offsets, num_variants, (long long*)read_index);
    return hipGetLastError() == hipSuccess ? PMT_OK : PMT_E_LAUNCH;
}

// One batch composed from a chunk of the dataset that is resident in HBM as it lies on disk (reference data/batch.py:41-62 does
// this collate on the host, Datum by Datum): the batch's variants are rows `ids` of the chunk.  ONE call = the per-variant
// tables in the Batch's dtypes (int16 -> int64, float16 -> float32: reference data/batch.py:47-49), the rows' read offsets, the
// exclusive scans of the counts and the gather index of the reads (ref rows of all variants, then alt rows).
__global__ __launch_bounds__(256) void pmt_gather_rows_kernel(const short* __restrict__ ints, int int_cols, const _Float16* __restrict__ floats,
                                                              int float_cols, const long long* __restrict__ row_start,
                                                              const long long* __restrict__ ids, int n, long long* __restrict__ ints_out,
                                                              float* __restrict__ floats_out, long long* __restrict__ row_start_out) {
    // a 64-lane slice per variant row: coalesced along the columns
    const int rows_per_block = 256 / 64, lane = threadIdx.x & 63;
    const int b = blockIdx.x * rows_per_block + (threadIdx.x >> 6);
    if (b >= n) return;
    const long long src = ids[b];
    const short* ip = ints + (size_t)src * int_cols;
    for (int c = lane; c < int_cols; c += 64) ints_out[(size_t)b * int_cols + c] = (long long)ip[c];
    const _Float16* fp = floats + (size_t)src * float_cols;
    for (int c = lane; c < float_cols; c += 64) floats_out[(size_t)b * float_cols + c] = (float)fp[c];
    if (lane == 0) row_start_out[b] = row_start[src];
}

// The same with the exclusive scans of the counts already known (pmt_prepare_chunk computes them on the host, beside the group
// plan): ONE launch per batch -- a 64-lane slice per variant gathers its two table rows AND writes its entries of the read index --
// where pmt_compose_batch needs five (gather, three for the scans, one workgroup per variant for the index).  Composition runs
// beside the consumer's read-set kernels, which leave it only the slots their retiring workgroups free: every launch less counts.
__global__ __launch_bounds__(256) void pmt_compose_planned_kernel(const short* __restrict__ ints, int int_cols, const _Float16* __restrict__ floats,
                                                                  int float_cols, const long long* __restrict__ row_start,
                                                                  const long long* __restrict__ ids, int n, const int* __restrict__ ref_off,
                                                                  const int* __restrict__ alt_off, long long* __restrict__ ints_out,
                                                                  float* __restrict__ floats_out, long long* __restrict__ row_start_out,
                                                                  long long* __restrict__ index) {
    const int rows_per_block = 256 / 64, lane = threadIdx.x & 63;
    const int b = blockIdx.x * rows_per_block + (threadIdx.x >> 6);
    if (b >= n) return;
    const long long src = ids[b];
    const long long start = row_start[src];
    const int r0 = ref_off[b], nr = ref_off[b + 1] - r0, a0 = alt_off[b], na = alt_off[b + 1] - a0;
    const long long total_ref = ref_off[n];
    const short* ip = ints + (size_t)src * int_cols;
    for (int c = lane; c < int_cols; c += 64) ints_out[(size_t)b * int_cols + c] = (long long)ip[c];
    const _Float16* fp = floats + (size_t)src * float_cols;
    for (int c = lane; c < float_cols; c += 64) floats_out[(size_t)b * float_cols + c] = (float)fp[c];
    for (int i = lane; i < nr; i += 64) index[r0 + i] = start + i;
    for (int i = lane; i < na; i += 64) index[total_ref + a0 + i] = start + nr + i;
    if (lane == 0) row_start_out[b] = start;
}

extern "C" int pmt_compose_batch_planned(const int16_t* chunk_ints, int32_t int_cols, const void* chunk_floats_f16, int32_t float_cols,
                                         const int64_t* chunk_row_start, const int64_t* ids, int32_t num_variants,
                                         const int32_t* ref_offsets, const int32_t* alt_offsets, int64_t* int_tensor, float* float_tensor,
                                         int64_t* row_start, int64_t* read_index, void* stream) {
    if (!chunk_ints || !chunk_floats_f16 || !chunk_row_start || !ids || !int_tensor || !float_tensor || !row_start || !ref_offsets ||
        !alt_offsets || !read_index || num_variants < 0 || int_cols < 2 || float_cols < 1)
        return PMT_E_INVALID;
    if (num_variants == 0) return PMT_OK;
    hipLaunchKernelGGL(pmt_compose_planned_kernel, dim3((num_variants + 3) / 4), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       (const short*)chunk_ints, int_cols, (const _Float16*)chunk_floats_f16, float_cols, (const long long*)chunk_row_start,
                       (const long long*)ids, num_variants, ref_offsets, alt_offsets, (long long*)int_tensor, float_tensor, (long long*)row_start,
                       (long long*)read_index);
    return hipGetLastError() == hipSuccess ? PMT_OK : PMT_E_LAUNCH;
}

extern "C" int pmt_compose_batch(const int16_t* chunk_ints, int32_t int_cols, const void* chunk_floats_f16, int32_t float_cols,
                                 const int64_t* chunk_row_start, const int64_t* ids, int32_t num_variants, int32_t ref_col, int32_t alt_col,
                                 int64_t* int_tensor, float* float_tensor, int64_t* row_start, int32_t* ref_offsets, int32_t* alt_offsets,
                                 int64_t* read_index, void* stream) {
    if (!chunk_ints || !chunk_floats_f16 || !chunk_row_start || !ids || !int_tensor || !float_tensor || !row_start || !ref_offsets ||
        !alt_offsets || !read_index || num_variants < 0 || int_cols < 2 || float_cols < 1 || ref_col < 0 || alt_col < 0 || ref_col >= int_cols ||
        alt_col >= int_cols)
        return PMT_E_INVALID;
    if (num_variants == 0) return pmt_scan_counts(int_tensor, int_tensor, 8, int_cols, 0, ref_offsets, alt_offsets, stream);
    hipLaunchKernelGGL(pmt_gather_rows_kernel, dim3((num_variants + 3) / 4), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       (const short*)chunk_ints, int_cols, (const _Float16*)chunk_floats_f16, float_cols, (const long long*)chunk_row_start,
                       (const long long*)ids, num_variants, (long long*)int_tensor, float_tensor, (long long*)row_start);
    if (hipGetLastError() != hipSuccess) return PMT_E_LAUNCH;
    int rc = pmt_scan_counts(int_tensor + ref_col, int_tensor + alt_col, 8, int_cols, num_variants, ref_offsets, alt_offsets, stream);
    if (rc != PMT_OK) return rc;
    return pmt_build_read_index(row_start, ref_offsets, alt_offsets, num_variants, read_index, stream);
}

// ---------------------------------------------------------------------------------------------------------------------
// fused clip_grad_norm_(max_norm) + AdamW over one flat buffer (reference misc_utils.py:128-129)
// ---------------------------------------------------------------------------------------------------------------------
#define PMT_OPT_BLOCKS 256
__global__ __launch_bounds__(256) void pmt_sumsq_kernel(const float* __restrict__ g, long long n, float* __restrict__ partial,
                                                        int* __restrict__ step_counter) {
    if (step_counter != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *step_counter += 1;  // this update's 1-based step number
    double acc = 0.0;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += 256ll * gridDim.x) acc += (double)g[i] * (double)g[i];
    __shared__ double sh[256];
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if (threadIdx.x < d) sh[threadIdx.x] += sh[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = (float)sh[0];
}

__global__ __launch_bounds__(256) void pmt_adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v, long long n,
                                                        PmtAdamW hp, const float* __restrict__ partial, int n_partial,
                                                        float* __restrict__ norm_out, const int* __restrict__ step_counter) {
    if (step_counter != nullptr) hp.step = *step_counter;  // (incremented by pmt_sumsq_kernel of this launch pair)
    __shared__ double sh[256];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n_partial; i += 256) acc += (double)partial[i];
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if (threadIdx.x < d) sh[threadIdx.x] += sh[threadIdx.x + d];
        __syncthreads();
    }
    const float total = (float)sqrt(sh[0]);
    if (norm_out && blockIdx.x == 0 && threadIdx.x == 0) *norm_out = total;
    const float coef = hp.max_grad_norm > 0.f ? fminf(hp.max_grad_norm / (total + 1e-6f), 1.0f) : 1.0f;
    const float bc1 = 1.f - powf(hp.beta1, (float)hp.step);
    const float bc2_sqrt = sqrtf(1.f - powf(hp.beta2, (float)hp.step));
    const float step_size = hp.lr / bc1;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += 256ll * gridDim.x) {
        const float gi = g[i] * coef;
        float pi = p[i] * (1.f - hp.lr * hp.weight_decay);
        const float mi = hp.beta1 * m[i] + (1.f - hp.beta1) * gi;
        const float vi = hp.beta2 * v[i] + (1.f - hp.beta2) * gi * gi;
        const float denom = sqrtf(vi) / bc2_sqrt + hp.eps;
        pi -= step_size * (mi / denom);
        p[i] = pi;
        m[i] = mi;
        v[i] = vi;
    }
}

extern "C" int pmt_clip_adamw(float* theta, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                              const PmtAdamW* hyper, float* scratch, float* grad_norm_out, void* stream) {
    if (!theta || !grad || !exp_avg || !exp_avg_sq || !hyper || !scratch || n < 0 || (hyper->step < 1 && hyper->step != PMT_STEP_ON_DEVICE))
        return PMT_E_INVALID;
    int* step_counter = hyper->step == PMT_STEP_ON_DEVICE ? reinterpret_cast<int*>(scratch) + PMT_STEP_SLOT : nullptr;
    if (n == 0) return PMT_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    int blocks = (int)((n + 1023) / 1024);
    if (blocks > PMT_OPT_BLOCKS) blocks = PMT_OPT_BLOCKS;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(pmt_sumsq_kernel, dim3(blocks), dim3(256), 0, s, grad, (long long)n, scratch, step_counter);
    hipLaunchKernelGGL(pmt_adamw_kernel, dim3(blocks), dim3(256), 0, s, theta, grad, exp_avg, exp_avg_sq, (long long)n,
                       *hyper, (const float*)scratch, blocks, grad_norm_out, (const int*)step_counter);
    return hipGetLastError() == hipSuccess ? PMT_OK : PMT_E_LAUNCH;
}

// ---------------------------------------------------------------------------------------------------------------------
// host staging copy (dataset chunk -> pinned buffer), multi-threaded and GIL-free
// ---------------------------------------------------------------------------------------------------------------------
extern "C" int pmt_host_copy(void* dst, const void* src, size_t bytes, int32_t threads) {
    if ((!dst || !src) && bytes > 0) return PMT_E_INVALID;
    if (threads < 1) threads = 1;
    if (threads > 64) threads = 64;
    const size_t min_part = (size_t)1 << 20;
    size_t parts = bytes / min_part;
    if (parts > (size_t)threads) parts = (size_t)threads;
    if (parts <= 1) {
        memcpy(dst, src, bytes);
        return PMT_OK;
    }
    const size_t per = ((bytes / parts) + 4095) & ~(size_t)4095;
    std::vector<std::thread> pool;
    pool.reserve(parts);
    for (size_t i = 0; i < parts; ++i) {
        const size_t lo = i * per, hi = (i + 1 == parts || (i + 1) * per > bytes) ? bytes : (i + 1) * per;
        if (lo >= hi) break;
        pool.emplace_back([=] { memcpy((char*)dst + lo, (const char*)src + lo, hi - lo); });
        if (hi == bytes) break;
    }
    for (auto& t : pool) t.join();
    return PMT_OK;
}

// The dropout masks as the kernels generate them (pmt_dropout.hpp), for the parity tests.
extern "C" int pmt_dropout_mask(uint64_t seed, float p, int32_t lin, int64_t row0, int64_t rows, int32_t width, float* out) {
    if (!out || rows < 0 || width < 0 || !(p >= 0.f && p < 1.f) || lin < 0) return PMT_E_INVALID;
    const unsigned thresh = pmt_drop_threshold(p);
    const float scale = 1.0f / (1.0f - p);
    for (int64_t r = 0; r < rows; ++r) {
        const unsigned key = pmt_drop_row_key((unsigned)seed, (unsigned)(seed >> 32), lin, (int)(row0 + r));
        for (int f = 0; f < width; ++f) out[r * width + f] = (seed == 0 || pmt_drop_keep(key, f, thresh)) ? (seed == 0 ? 1.0f : scale) : 0.f;
    }
    return PMT_OK;
}

// The same for a table of fixed-size rows, with one byte range of every row cleared on the way (the posterior hand-off's integer
// rows are the dataset's own with the two read counts zeroed, reference tools/filter_variants.py:305-308): one pass over the
// memory instead of a copy and a strided clearing pass.
extern "C" int pmt_host_copy_rows(void* dst, const void* src, int64_t rows, int64_t row_bytes, int64_t zero_offset, int64_t zero_bytes,
                                  int32_t threads) {
    if (rows < 0 || row_bytes < 1 || zero_offset < 0 || zero_bytes < 0 || zero_offset + zero_bytes > row_bytes) return PMT_E_INVALID;
    if ((!dst || !src) && rows > 0) return PMT_E_INVALID;
    if (threads < 1) threads = 1;
    if (threads > 64) threads = 64;
    int64_t parts = rows * row_bytes / ((int64_t)1 << 20);
    if (parts > threads) parts = threads;
    if (parts < 1) parts = 1;
    auto work = [=](int64_t lo, int64_t hi) {
        char* d = (char*)dst + lo * row_bytes;
        memcpy(d, (const char*)src + lo * row_bytes, (size_t)((hi - lo) * row_bytes));
        if (zero_bytes > 0)
            for (int64_t r = 0; r < hi - lo; ++r) memset(d + r * row_bytes + zero_offset, 0, (size_t)zero_bytes);
    };
    // (blocks of 4096 rows: the clearing pass then finds the rows it just copied in the cache)
    auto run = [=](int64_t lo, int64_t hi) {
        for (int64_t b = lo; b < hi; b += 4096) work(b, b + 4096 < hi ? b + 4096 : hi);
    };
    if (parts == 1) {
        run(0, rows);
        return PMT_OK;
    }
    std::vector<std::thread> pool;
    for (int64_t i = 0; i < parts; ++i) pool.emplace_back(run, rows * i / parts, rows * (i + 1) / parts);
    for (auto& t : pool) t.join();
    return PMT_OK;
}

