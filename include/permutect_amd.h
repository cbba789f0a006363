/*
 * permutect_amd.h -- C ABI of the MI355X (gfx950) engine for the Permutect artifact-model hot path.
 *
 * The reference (broadinstitute/permutect) is pure Python on PyTorch and has no FFI or operator registry; the seam
 * this library plugs into is the Python object boundary of `ArtifactModel`
 * (reference permutect/architecture/artifact_model.py:239-325) plus `backpropagate`
 * (reference permutect/misc_utils.py:125-129).  Each entry point below names the reference code it replaces.
 * INTEGRATION.md shows the ctypes binding a maintainer would add on the reference side.
 *
 * Conventions
 *   - plain C: pointers + sizes only.  Device pointers are raw HBM addresses; `stream` is a hipStream_t passed as
 *     void* (NULL = default stream).  The caller owns every buffer; the library allocates nothing on the device
 *     and keeps no global state, so it is thread-safe per stream and graph-capturable.
 *   - every function returns 0 on success or a negative PMT_E_* code; it never throws and never falls back to a
 *     CPU path.
 *   - offsets inside descriptors are in units of floats into the buffer named in the field comment.
 */
#ifndef PERMUTECT_AMD_H
#define PERMUTECT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PMT_ABI_VERSION 10

/* bits of the fault word (PmtBatch.join_fault) */
#define PMT_FAULT_JOIN 1
#define PMT_FAULT_F16_RANGE 2
#define PMT_FAULT_PLAN 4        /* pmt_plan_groups_device: more groups than the capacity given, or a read set beyond one workgroup */

/* error codes */
#define PMT_OK 0
#define PMT_E_INVALID (-1)      /* bad argument / descriptor out of supported range */
#define PMT_E_UNSUPPORTED (-2)  /* configuration the gfx950 kernels do not cover (stated in DESIGN.md) */
#define PMT_E_CAPACITY (-3)     /* a read set does not fit the register-resident group capacity */
#define PMT_E_LAUNCH (-4)       /* HIP launch failure (hipGetLastError is left set) */
#define PMT_E_WORKSPACE (-5)    /* workspace too small */

/* compile-time limits of the kernels */
#ifndef PMT_MAX_WIDTH
#define PMT_MAX_WIDTH 64        /* widest activation (features) kept register-resident: 4 tiles of 16.  The WIDE build of the
                                   library (csrc/Makefile: `make wide`, -DPMT_MAX_WIDTH=128) keeps 8: pmt_limits() reports it */
#endif
#ifndef PMT_MAX_HALF_FFN
#define PMT_MAX_HALF_FFN 16     /* d_ffn / 2: one 16-feature tile per half of a gated block's hidden layer.  A build with 32 (csrc/Makefile:
                                   `make wide32`) gives every half TWO tiles -- it runs models with d_ffn / 2 in 17 .. 32 only */
#endif
#define PMT_MAX_CLUSTERS 16
#define PMT_MAX_OPS 8           /* top-level ops per MLP program */
#define PMT_MAX_ROW_INPUT 128    /* widest input of a per-variant row MLP (info vector) */
#define PMT_MAX_CNN_TAPS 192     /* widest im2col column of a haplotype-CNN convolution: in_channels * kernel_size (64 x 3; 32 x 5) */
#define PMT_ROWS_INFO 0
#define PMT_ROWS_ALT_COUNT 1
#define PMT_ROWS_SOURCE 2
#define PMT_MAX_SKIP_LAYERS 4
#define PMT_MAX_BLOCKS 16
#define PMT_MAX_LINEAR 96
#ifndef PMT_GROUP_WAVES
#define PMT_GROUP_WAVES 8       /* waves per workgroup */
#endif
#define PMT_GROUP_TILES (2 * PMT_GROUP_WAVES)      /* 16-read tiles per group (two per wave) */
#define PMT_GROUP_MAX_SETS (8 * PMT_GROUP_WAVES)   /* read sets (variants) per group */
#define PMT_TILE 16

/* read-row formats accepted at the boundary (reference data/batch.py:41-62, data/datum.py:35) */
#define PMT_READS_PACKED_U8 0   /* [R][7 + nf] uint8: 7 MSB-first bit-packed bytes, then nf quantile bytes */
#define PMT_READS_F16 1         /* [R][F] float16 (Batch.reads_re as the reference collates it) */
#define PMT_READS_F32 2         /* [R][F] float32 (Batch.copy_to(device, float32)); any values: float16 / float32 rows run the wide-range, guarded instances (pmt_forward.hip) */

/* One nn.Linear.  Weights are consumed in MFMA fragment order from the packed buffer. */
typedef struct PmtLinear {
    int32_t in_dim, out_dim;
    int32_t w_frag;      /* packed: A fragments of W   [out][in]  (forward, y = W x + b)            */
    int32_t wt_frag;     /* packed: A fragments of W^T [in][out]  (backward, dx = W^T dy)           */
    int32_t b_pvec;      /* packed: bias in tile-position order, -1 = no bias                       */
    int32_t w_src;       /* natural-layout source of W: offset into theta (>=0) or phi (<= -2: -(off+2)) */
    int32_t b_src;       /* same for the bias, -1 = none                                            */
    int32_t out_split;   /* 0, or h: the 2h output rows are laid out as two 16-row tiles (rows 0..h-1 -> tile 0,
                            rows h..2h-1 -> tile 1) so that z1 / z2 of the gating unit are tile aligned       */
    int32_t wb_frag;     /* packed: W as THREE bf16 pieces (hi + mid + lo = the fp32 value) in the operand order of
                            v_mfma_f32_16x16x32_bf16: [k block of 32][out tile][piece][lane][8 bf16]; -1 = none */
    int32_t wtb_frag;    /* the same for W^T                                                                   */
    int32_t wh_frag;     /* packed: W as TWO f16 pieces in the operand order of v_mfma_f32_16x16x32_f16:
                            [out tile][k block of 32][piece][lane][8 f16], piece 0 = f16(w), piece 1 = f16(2^12 (w - piece 0)):
                            the low piece is stored scaled so that it keeps its 11 bits whatever the weight's size
                            (pmt_device.hpp: linear_acc_f16); -1 = none                                        */
    int32_t emit_tab;    /* packed (read as int32): where the weight-gradient blocks of this linear go.  For every 16 x 16
                            block (out tile ot, in tile it) of dW in the matrix core's C layout, [(ot * nkt + it)][lane][4]
                            element offsets into the buffer w_src names (theta, or phi when w_src <= -2), -1 = padding;
                            then per out tile 16 offsets of the bias gradient in position order (theta), -1 = none.
                            Written by pmt_pack_params; lets pmt_backward add a block with four atomics and no index
                            arithmetic.  -1 = none                                                                */
} PmtLinear;

#define PMT_OP_LINEAR 0         /* y = W x + b, optional SELU after   (reference mlp.py:55-61)          */
#define PMT_OP_SKIP 1           /* y = x + alpha * f(x), f = (SELU, Linear) x n   (reference mlp.py:15-22) */
typedef struct PmtOp {
    int32_t kind;
    int32_t n_layers;                       /* SKIP: number of (SELU, Linear) pairs; LINEAR: 1 */
    int32_t selu_after;                     /* LINEAR only */
    int32_t alpha_src;                      /* SKIP: theta offset of the scalar alpha */
    int32_t lin[PMT_MAX_SKIP_LAYERS];       /* indices into PmtModel.lin */
} PmtOp;

typedef struct PmtMlp {
    int32_t n_ops, in_dim, out_dim;
    int32_t dropout;            /* 1: the reference built this MLP with dropout_p (an nn.Dropout behind every Linear) */
    PmtOp ops[PMT_MAX_OPS];
} PmtMlp;

/* One GatedRefAltMLPBlock (reference gated_mlp.py:148-251). */
typedef struct PmtBlock {
    int32_t norm_w_pvec, norm_b_pvec;       /* packed LayerNorm(D) weight / bias */
    int32_t norm_w_src, norm_b_src;         /* theta offsets */
    int32_t proj1[2], proj2[2];             /* linear ids, [0] = ref, [1] = alt; proj1 rows are permuted so
                                               that z1 and z2 each start on a 16-row tile */
    int32_t sgu_norm_w_pvec, sgu_norm_b_pvec;
    int32_t sgu_norm_w_src, sgu_norm_b_src;
    int32_t alpha_src[2], beta_src[2];      /* theta offsets of alpha_ref/alt, beta_ref/alt */
    int32_t gamma_src;
    int32_t ref_reg_pvec, ref_reg_src;      /* ref_regularizer [h] */
    int32_t reg_weight_phi;                 /* phi offset of exp(reg_weight.original) (the +0.25 is applied in-kernel) */
} PmtBlock;

/* FeatureClustering + EMG head (reference feature_clustering.py:82-135, exponentially_modified_gaussian.py:30-89).
 * All entries are offsets into phi (materialised values) except mu (theta). */
typedef struct PmtHead {
    int32_t stdev_e_phi;        /* [E] nonartifact stdev                                   */
    int32_t dirs_ke_phi;        /* [K][E] unit direction vectors (already normalised)      */
    int32_t art_stdev_k_phi;    /* [K]                                                     */
    int32_t log_w_k_phi;        /* [K] log_softmax cluster weights                         */
    int32_t mu_k_src;           /* [K] theta offset                                        */
    int32_t sigma_k_phi;        /* [K]                                                     */
    int32_t lambda_k_phi;       /* [K]                                                     */
    int32_t reserved;
} PmtHead;

/* Haplotype CNN (reference architecture/dna_sequence_convolution.py:31-111): a short list of 1-D layers applied to the
 * [10][S] one-hot of the ref/alt haplotypes; evaluated per variant entirely in LDS by pmt_cnn_forward / _backward. */
#define PMT_MAX_CNN_LAYERS 12
#define PMT_CNN_CONV 0
#define PMT_CNN_POOL 1
#define PMT_CNN_LEAKY_RELU 2
#define PMT_CNN_SELU 3
#define PMT_CNN_FLATTEN 4
#define PMT_CNN_LINEAR 5
typedef struct PmtCnnLayer {
    int32_t kind;
    int32_t in_ch, in_len, out_ch, out_len;     /* LINEAR (after FLATTEN): in_ch = features, in_len = out_len = 1 */
    int32_t kernel, stride, padding, dilation;  /* CONV / POOL */
    int32_t w_src, b_src;                       /* theta offsets of weight ([out][in][k] or [out][in]) and bias */
    int32_t in_off, out_off;                    /* per-variant offsets of this layer's input / output in the backward's
                                                   activation region (the one-hot input sits at offset 0)         */
    int32_t lin;                                /* CONV: PmtLinear id of the weight viewed as [out_ch][in_ch * kernel]
                                                   (packed fragments for the implicit-GEMM kernels), else -1      */
    int32_t reserved[2];
} PmtCnnLayer;
typedef struct PmtCnn {
    int32_t n_layers;
    int32_t seq_len;    /* S = haplotypes_length / 2 */
    int32_t out_dim;    /* width of the haplotype embedding */
    int32_t max_act;    /* max floats of any single activation (including the 10*S input) */
    int32_t sum_act;    /* floats of input + every layer output (backward keeps them all in LDS) */
    int32_t reserved[3];
    PmtCnnLayer layers[PMT_MAX_CNN_LAYERS];
} PmtCnn;

typedef struct PmtModel {
    int32_t abi_version;
    int32_t num_read_features;  /* F */
    int32_t read_embed_dim;     /* E_r */
    int32_t variant_embed_dim;  /* E_v = info embedding + haplotype embedding widths */
    int32_t d_model;            /* D = E_r + E_v */
    int32_t d_ffn;              /* 2 h */
    int32_t num_blocks;         /* L */
    int32_t feature_dim;        /* E */
    int32_t num_clusters;       /* K */
    int32_t n_linear;
    int32_t theta_size, phi_size, packed_size; /* floats */
    int32_t translation_src;    /* theta offset of pre_clustering_transform.translation_e [E] */
    int32_t translation_pvec;
    int32_t rotation_lin;       /* linear id, weight source = phi (materialised Q), no bias */
    PmtMlp read_mlp;            /* read_embedding  */
    PmtMlp reducer;             /* reducer         */
    PmtMlp row_mlp[3];          /* per-variant row MLPs (pmt_rows_*): [0] info_embedding, [1] alt_count_predictor,
                                   [2] source_predictor (n_ops = 0 when there is a single source).  Their first
                                   linear may read up to PMT_MAX_ROW_INPUT features                               */
    PmtBlock blocks[PMT_MAX_BLOCKS];
    PmtHead head;
    PmtCnn cnn;                 /* haplotypes_cnn */
    PmtLinear lin[PMT_MAX_LINEAR];
    /* Kernel-instance selection, part of the descriptor (the library reads no environment variables and keeps no state).
     * 0 everywhere = the library's own choice; the other values exist so that the parity tests can run every instance. */
    int32_t force_shape;        /* read-set kernels: 0 auto, 1 at most the tile-exact instance, 2 the generic instance,
                                   3 plain bf16 products where the exact-width instance applies (NOT a parity mode:
                                   one bf16 MFMA per product instead of the fp32-equivalent ones; bench.py --dtype bf16),
                                   5 the exact-width FORWARD with its products as six bf16 MFMAs on three-piece splits (the
                                   round-3 form; auto = three f16 MFMAs on two-piece splits, pmt_device.hpp: linear_acc_f16) */
    int32_t force_cnn;          /* haplotype CNN: 0 auto, 1 general (workgroup-per-chunk) kernels, 2 wave-per-variant
                                   kernels (pmt_cnn2), 3 batched-column kernels (pmt_cnn3)                               */
    int32_t cnn_debug;          /* development switches of pmt_cnn2_backward (0 in production)                           */
    int32_t emit_base;          /* packed: the linears' emit tables lie back to back in [emit_base, emit_base + emit_len) */
    int32_t emit_len;           /*   (floats); a row of pmt_backward's `grad_partials` mirrors exactly this region        */
    float dropout_p;            /* reference parameters.py:26 / mlp.py:57-58: nn.Dropout(p) behind every Linear of the MLPs flagged
                                   PmtMlp.dropout.  Applied only to a batch that brings a dropout_seed (train mode); such a batch
                                   runs the generic kernel instances */
} PmtModel;

/* Inputs of one forward / backward pass.  Reads are ordered as the reference's Batch orders them: all ref reads
 * of all variants, then all alt reads (reference data/batch.py:45-47). */
typedef struct PmtBatch {
    int32_t num_variants;           /* B */
    int32_t num_groups;             /* G */
    int32_t read_format;            /* PMT_READS_* */
    int32_t read_row_bytes;         /* stride of one read row */
    const void* reads;              /* device */
    const int64_t* read_index;      /* device, optional gather (DownsampledBatch.read_indices), NULL = identity */
    const int32_t* ref_offsets;     /* device [B+1] exclusive scan of ref counts; [B] = total ref reads */
    const int32_t* alt_offsets;     /* device [B+1] exclusive scan of alt counts */
    const float* variant_embed;     /* device [B][E_v]: info embedding | haplotype embedding */
    const int32_t* group_start;     /* device [G+1] first variant of each group (pmt_plan_groups) */
    const int32_t* group_tile_base; /* device [G+1] first stash tile of each group (pmt_plan_groups) */
    int64_t total_tiles;            /* host value of group_tile_base[G] (sizes the stash)             */
    int32_t* debug_flags;           /* device, optional [64 + 4096]: [0] unused, [1] development
                                       switches of the backward kernel (0 in production), [8:56] 24 x u64 cycle counters */
    const int32_t* group_span;      /* device [G][6] or NULL.  With it a group may cover only PART of a read set (read sets
                                       beyond one workgroup, pmt_plan_groups_split): v0, v1 (variants [v0, v1)), ref_begin,
                                       ref_end, alt_begin, alt_end (rows of the batch's ref / alt regions); group_start is
                                       then ignored.  Only pmt_forward_layered / pmt_backward_layered accept it. */
    const int32_t* num_groups_dev;  /* device, optional [1]: the number of groups of THIS batch when num_groups is only the
                                       capacity the launch is sized for: workgroups beyond it return at once.  Lets one
                                       captured HIP graph (fixed grids) serve batches with different group plans;
                                       pmt_forward / pmt_backward only */
    const int32_t* set_groups;      /* device, optional [B] (with group_span): how many groups cover each variant, i.e. the
                                       number of g with span[g].v0 <= b < span[g].v1.  With it pmt_forward_layered /
                                       pmt_backward_layered run ONE launch each way in which the groups of a split read set
                                       join their per-set sums through HBM (arrival counters) while the activations stay in
                                       registers; NULL = num_blocks + 1 launches with the activations parked in between */
    uint64_t dropout_seed;          /* 0 = no dropout (eval mode, or dropout_p = 0).  Otherwise the seed of THIS step's masks
                                       (pmt_dropout_mask): the forward and the backward of one step get the same value */
    int32_t* join_fault;            /* device, optional [1], caller-owned and never cleared by the library: the launches' fault
                                       word.  Bit 0 (PMT_FAULT_JOIN): a joined launch (set_groups) whose bounded wait for another
                                       workgroup gave up -- its numbers are wrong.  Bit 1 (PMT_FAULT_F16_RANGE): a forward whose
                                       residual stream or reducer output left the range of its f16 operand pieces (+-65504: they
                                       saturate) -- the logits of that launch are wrong.  One persistent word serves every launch
                                       of a run; the host reads it where it synchronises anyway (end of an epoch / of a filtering
                                       pass) and raises.  NULL = join faults go to the launch's own scratch, range faults nowhere */
} PmtBatch;

typedef struct PmtOutputs {
    float* logits_b;                /* [B]            capped artifact logit                    */
    float* logits_bk;               /* [B][K+2]       nonartifact, outlier, K clusters         */
    float* features_be;             /* [B][E]         alt-set mean embedding                   */
    float* ref_features_be;         /* [B][E]         ref-set mean embedding                   */
} PmtOutputs;

/* dL/d(outputs), same shapes as PmtOutputs; any pointer may be NULL (= zero). */
typedef struct PmtOutputGrads {
    const float* d_logits_b;
    const float* d_logits_bk;
    const float* d_features_be;
    const float* d_ref_features_be;
} PmtOutputGrads;

typedef struct PmtAdamW {
    float lr, beta1, beta2, eps, weight_decay, max_grad_norm;
    int32_t step;                   /* 1-based step number of this update, or PMT_STEP_ON_DEVICE: the step count lives in
                                       `scratch` (int32 at float index PMT_STEP_SLOT, zero before the first step) and the
                                       launch increments it itself, so that a captured graph can be replayed */
    int32_t reserved;
} PmtAdamW;
#define PMT_STEP_ON_DEVICE (-1)
#define PMT_STEP_SLOT 1000

/* ---- host-side helpers (no GPU needed) ------------------------------------------------------------------- */

int pmt_abi_version(void);

/* The shape the exact-width kernel instances of THIS build of the library are compiled for: tile counts (16 features each) of the
 * read features, the read-MLP widths, d_model / the reducer's widths, feature_dim, then the widths themselves: read features,
 * read-MLP width, d_model, d_ffn / 2, feature_dim.  The default build carries the production hyperparameters
 * (4, 2, 4, 1; 61, 30, 60, 10, 10); `make -C permutect_amd/csrc instance SHAPE="..."` builds the library around another shape,
 * and permutect_amd/engine/instances.py picks (or builds) the library whose TILE counts a model fills.  pmt_shape_id: which
 * instance a model runs in this build -- 0 generic (fp32 MFMAs, any supported model), 1 the build's tiles with fp32 MFMAs (asked for
 * through force_shape only), 6 the build's tiles on the 16-bit matrix pipes with the widths read at run time, 2 the build's widths
 * compiled in, 3 the same with plain bf16 products.  Replaces nothing in the reference: its ATen kernels take any width
 * (architecture/mlp.py:32-67, parameters.py:70-156). */
int pmt_shape_info(int32_t* nine);
int pmt_shape_id(const struct PmtModel* m);
/* The compile-time limits of THIS build of the library: four[0] widest register-resident activation (PMT_MAX_WIDTH: 64; the wide
 * build `libpermutect_amd_wide.so` -- csrc/Makefile `make wide`, generic instances only -- 128), [1] largest d_ffn / 2, [2] floats of
 * one stash slot, [3] waves per workgroup of the read-set kernels.  permutect_amd/engine/instances.py loads the wide build for a
 * model with a layer wider than the default build's limit (the reference takes any width: architecture/mlp.py:32-67). */
int pmt_limits(int32_t* four);

/* A hash of the sources THIS build of the library was compiled from (16 hex digits and a terminating 0 into `out`; returns the
 * number of characters written, PMT_E_INVALID if `capacity` < 17).  The default library and every per-shape / wide build of one tree
 * carry the same id; permutect_amd/engine/instances.py refuses (and rebuilds) a per-shape library left over from other sources. */
int pmt_build_id(char* out, int32_t capacity);

/* sizeof() of the ABI structs as compiled into the library, for binding self-checks:
 * 0 PmtModel, 1 PmtBatch, 2 PmtOutputs, 3 PmtOutputGrads, 4 PmtAdamW, 5 PmtLinear, 6 PmtOp, 7 PmtMlp, 8 PmtBlock, 9 PmtHead,
 * 10 PmtPhiProgram, 11 PmtLossArgs, 12 PmtDownsample, 13 PmtRecordArgs, 14 PmtBalanceArgs, 15 PmtEvalArgs */
int pmt_struct_bytes(int which);

/* Validates a descriptor against the kernels' limits. */
int pmt_model_check(const PmtModel* model);

/* Materialised parametrizations ("phi"): the reference registers torch parametrizations on a few small tensors
 * (reference architecture/feature_clustering.py:41-75, exponentially_modified_gaussian.py:66-75, gated_mlp.py:196-201,
 * euclidean_transformation.py:14-16); one segment per tensor maps theta[theta_off ...] (the `.original` leaf) to
 * phi[phi_off ...] (the value the kernels consume). */
#define PMT_PHI_EXP 0           /* phi = exp(theta)                                        (PositiveNumber)          */
#define PMT_PHI_BOUNDED 1       /* phi = p1 * sigmoid(theta) + p0                          (BoundedNumber)           */
#define PMT_PHI_UNIT_ROWS 2     /* every row x of [rows][cols]: x / |x|                    (UnitVector)              */
#define PMT_PHI_LOG_SOFTMAX 3   /* log_softmax of every row                                (LogWeights)              */
#define PMT_PHI_ORTHOGONAL 4    /* base @ expm(tril(X) - tril(X)^T), X = theta [n][n]      (torch orthogonal, matrix_exp map,
                                   use_trivialization; base == NULL means identity)                              */
#define PMT_MAX_PHI_SEGS 48
#define PMT_MAX_ORTHO_DIM 32
typedef struct PmtPhiSeg {
    int32_t kind, theta_off, phi_off, rows, cols, reserved;
    float p0, p1;
    int32_t base_rs, base_cs;   /* ORTHOGONAL: element strides of `base` (torch keeps it as a transposed view) */
    const float* base;          /* ORTHOGONAL: device base matrix, base[i][k] = base[i * base_rs + k * base_cs] */
} PmtPhiSeg;
typedef struct PmtPhiProgram {
    int32_t n_segs, reserved;
    PmtPhiSeg seg[PMT_MAX_PHI_SEGS];
} PmtPhiProgram;

/* phi <- parametrizations(theta): one launch, one workgroup per segment.  Replaces the forward of
 * torch.nn.utils.parametrize for the tensors above (~60 tiny launches per step through torch). */
int pmt_phi_forward(const PmtPhiProgram* prog, const float* theta, float* phi, void* stream);
/* grad_theta[segment's leaf] += J^T grad_phi: one launch.  Replaces autograd through the parametrizations (the adjoint of
 * matrix_exp alone is ~100 launches through torch). */
int pmt_phi_backward(const PmtPhiProgram* prog, const float* theta, const float* phi, const float* grad_phi,
                     float* grad_theta, void* stream);

/* One integer column of a batch as the caller holds it: element i at ptr + i * stride elements of elem_bytes (4: int32, 8: int64)
 * -- a column of Batch.int_tensor (int64, row stride), a DownsampledBatch's own counts (int32, dense).  ptr NULL = all zeros. */
typedef struct PmtIntColumn {
    const void* ptr;
    int64_t stride;
    int32_t elem_bytes, reserved;
} PmtIntColumn;

/* The (source, label, variant type, ref-count bin, alt-count bin) cell of a variant (reference data/batch.py:228-230,
 * data/count_binning.py:9-26): what the balancer's and the downsampler's tables are indexed by. */
typedef struct PmtBinning {
    int32_t num_sources, num_variant_types, num_ref_bins, num_alt_bins, count_bin_skip, max_ref_count, max_alt_count, reserved;
} PmtBinning;

/* Read downsampling of a training batch (reference data/batch.py:389-439, training/downsampler.py:105-123). */
typedef struct PmtDownsample {
    int32_t num_variants;
    int32_t reference_alt_gather;   /* 1: reproduce the reference's un-offset alt gather; 0: the intended rows */
    uint64_t seed;                  /* counter-based generator: decisions are a function of (seed, row) */
    int64_t force_random;           /* the reference's randint(0, 100): alt read  end - (force_random % count) - 1  is always kept */
    const int32_t* ref_offsets;     /* [B+1] exclusive scans of the PARENT batch's counts (pmt_scan_counts) */
    const int32_t* alt_offsets;
    const float* ref_weights_b4;    /* [B][4] mixture weights over the Beta shapes (1,1) (1,5) (5,1) (5,5), or NULL = uniform */
    const float* alt_weights_b4;
    const float* ref_fracs_in;      /* [B] keep fractions given by the caller (then no fractions are drawn), or NULL */
    const float* alt_fracs_in;
    /* The mixture weights looked up by the kernel itself (then *_weights_b4 are NULL): the Downsampler's tables [S][3][V][R][A][4]
     * (exp of its log-weights, reference training/downsampler.py:105-112) indexed by the variant's cell -- its label, variant type and
     * source columns and the bins of its PARENT counts (the offsets above).  Replaces ~25 torch launches per training step. */
    const float* ref_weight_table;
    const float* alt_weight_table;
    PmtIntColumn labels, variant_types, sources;
    PmtBinning bins;
} PmtDownsample;
/* Launch 1: draw the keep fractions (unless given) and count the kept reads per variant. */
int pmt_downsample_counts(const PmtDownsample* args, float* ref_fracs, float* alt_fracs, int32_t* new_ref_counts,
                          int32_t* new_alt_counts, void* stream);
/* Launch 2 (after the exclusive scans of the new counts): the gather index PmtBatch.read_index consumes; kept rows in
 * ascending order, all kept ref rows of all variants then all kept alt rows, like the reference's read_indices. */
int pmt_downsample_index(const PmtDownsample* args, const float* ref_fracs, const float* alt_fracs,
                         const int32_t* new_ref_offsets, const int32_t* new_alt_offsets, int64_t* read_index, void* stream);

/* Per-variant losses (reference architecture/artifact_model.py:267-325). */
typedef struct PmtLossArgs {
    int32_t num_variants, num_clusters, num_sources, reserved;
    float max_outlier_logit;        /* constants.MAX_OUTLIER_LOGIT: the outlier logit is clipped from above */
    float max_alt_count;            /* constants.MAX_ALT_COUNT: alt-count target = alt_count / max_alt_count */
    const float* logits_b;          /* [B]      capped artifact logit (PmtOutputs.logits_b) */
    const float* logits_bk;         /* [B][K+2] (PmtOutputs.logits_bk) */
    const float* alt_count_raw;     /* [B]      output of the alt-count adversary's MLP (before the sigmoid) */
    const float* source_logits;     /* [B][S]   output of the source adversary's MLP, or NULL (num_sources == 1) */
    const int64_t* labels;          /* Label per variant (0 artifact, 1 variant, 2 unlabeled), element stride label_stride */
    int64_t label_stride;
    const int64_t* alt_counts;      /* element stride alt_count_stride */
    int64_t alt_count_stride;
    const int64_t* sources;         /* source index per variant (NULL if num_sources == 1), element stride source_stride */
    int64_t source_stride;
    const float* weights;           /* [B] BatchOutput.weights */
    const float* source_weights;    /* [B] BatchOutput.source_weights */
} PmtLossArgs;
typedef struct PmtLossOutputs {     /* forward: the five per-variant loss vectors; backward: their upstream gradients (NULL = 0) */
    float* supervised_b;
    float* unsupervised_b;
    float* alt_count_b;
    float* source_b;
    float* total_b;                 /* weights * (supervised + unsupervised + alt_count) + source_weights * source */
} PmtLossOutputs;
typedef struct PmtLossInputGrads {
    float* d_logits_b;              /* [B]      */
    float* d_logits_bk;             /* [B][K+2] */
    float* d_alt_count_raw;         /* [B]      */
    float* d_source_logits;         /* [B][S] or NULL */
} PmtLossInputGrads;
/* One launch each; replace ~85 elementwise torch launches per training step (BCEWithLogits, clip, logsumexp, sigmoid, MSE,
 * softmax, products and their autograd). */
int pmt_losses_forward(const PmtLossArgs* args, const PmtLossOutputs* out, void* stream);
int pmt_losses_backward(const PmtLossArgs* args, const PmtLossOutputs* grad_out, const PmtLossInputGrads* grad_in, void* stream);

/* Loss bookkeeping (reference training/loss_recorder.py:15-24, metrics/loss_metrics.py:50-54): adds one step's losses into
 * six histograms [6][num_bins] = (primary totals, primary counts, alt-count totals, alt-count counts, source totals,
 * source counts), each a flattened [S][3][V][R][A] tensor indexed by source, label, variant type, ref-count bin,
 * alt-count bin (reference data/count_binning.py:61-66, data/batch.py:228-230).  One launch instead of 8 index_add_. */
typedef struct PmtRecordArgs {
    int32_t num_variants, num_bins;                 /* num_bins = S * 3 * V * R * A */
    int32_t num_variant_types, num_ref_bins, num_alt_bins, count_bin_skip, max_ref_count, max_alt_count;
    PmtIntColumn labels, variant_types, sources, ref_counts, alt_counts;   /* (sources.ptr NULL = source 0; int32 or int64 columns as the batch holds them) */
    const float* weights;         /* [B] BatchOutput.weights */
    const float* source_weights;  /* [B] */
    const float* supervised_b;    /* [B] the four loss vectors of pmt_losses_forward */
    const float* unsupervised_b;
    const float* alt_count_b;
    const float* source_b;
} PmtRecordArgs;
int pmt_record_losses(const PmtRecordArgs* args, float* histograms, void* stream);

/* The tallies of an evaluation step (reference metrics/evaluation_metrics.py:49-66 -> AccuracyMetrics, metrics/loss_metrics.py:226-243;
 * training/model_training.py:204-228 runs it three times per parent batch after every validation epoch): the weights of the LABELED
 * variants over (source, label, variant type, ref-count bin, alt-count bin, LOGIT bin) and, per label, the weight called artifact / not
 * and the weighted logit sum -- one launch for ~30 tensor ops.  `flat`: [2 epoch types][nhist] histograms, then [2][3 labels][3] statistics
 * (training/loss_recorder.py: EvaluationCounts.flat). */
typedef struct PmtEvalArgs {
    int32_t num_variants, epoch_index;              /* 0: training data, 1: validation data */
    int32_t num_logit_bins, min_logit, max_logit, logit_bin_skip;   /* reference data/count_binning.py:14-26 */
    PmtBinning bins;
    PmtIntColumn labels, variant_types, sources, ref_counts, alt_counts;
    const float* logits_b;
    const float* weights_b;
    int64_t nhist;                                  /* S * 3 * V * R * A * num_logit_bins */
} PmtEvalArgs;
int pmt_record_evaluation(const PmtEvalArgs* args, float* flat, void* stream);

/* The balancer's step (reference training/balancer.py:55-119 `process_batch_and_compute_weights`): running counts per cell, pseudo-counts
 * of the unlabeled data from the model's artifact probability, the weight tables re-derived from them (when `recompute`: the reference
 * does it every DATA_BEFORE_RECOMPUTE variants) and this batch's weights looked up -- ~60 torch launches per training step in two:
 *   launch 1: counts[cell] += 1; unlabeled variants: pseudo_counts[cell as artifact] += p, [cell as variant] += 1 - p
 *             (p = sigmoid(logits_b); a workgroup's additions are joined in LDS first);
 *   launch 2: tables_out = recompute ? att * tables_in + (1 - att) * clip((1 + ratio^-+1) / 2, 0.01, 100) : tables_in with
 *             ratio = (counts[artifact] + 0.01) / (counts[variant] + 0.01) per cell (every workgroup derives the tables it reads from,
 *             workgroup 0 stores them), then weights_b = labeled ? weights[cell] : p * unlabeled_weights[cell as artifact] +
 *             (1 - p) * unlabeled_weights[cell as variant], and source_weights_b = weights_b * source_weights[source]
 *             (BatchOutput.weights / .source_weights of reference artifact_model.py:285-288).
 * tables_in and tables_out must be different buffers when recompute != 0 (the caller swaps them). */
typedef struct PmtBalanceArgs {
    int32_t num_variants, recompute;
    float attenuation;                  /* ATTENUATION_PER_DATUM ^ (variants since the last recomputation) */
    int32_t reserved;
    PmtBinning bins;
    PmtIntColumn labels, variant_types, sources, ref_counts, alt_counts;
    const float* logits_b;              /* [B] capped artifact logits of this batch */
    float* counts;                      /* [S][3][V][R][A] running counts (added to) */
    float* pseudo_counts;
    const float* weights_in;            /* the three tables before this step: [S][3][V][R][A], the same, [S] */
    const float* unlabeled_weights_in;
    const float* source_weights_in;
    float* weights_out;                 /* ... after it */
    float* unlabeled_weights_out;
    float* source_weights_out;
    float* weights_b;                   /* [B] out */
    float* source_weights_b;            /* [B] out: weights_b * the source's weight */
} PmtBalanceArgs;
int pmt_balance_step(const PmtBalanceArgs* args, void* stream);

/* The float rows of the posterior hand-off, one launch per batch (reference tools/filter_variants.py:302-320 builds a Datum per
 * variant in Python: `set(CACHED_ARTIFACT_LOGIT, logit)` stores the logit through the float16 scalar array, `set_info_1d(embedding)`
 * replaces the info columns by the float32 embedding, data/datum.py:199-211).  Row i of the batch becomes row dest_ids[i] (NULL: i)
 * of `block`: columns [0, n_scalars) = the batch's float columns rounded through float16, column logit_col = float16(logits_b[i]);
 * columns [n_scalars, n_scalars + e) = features_be[i].  Replaces five torch launches per batch in tools/posterior_data.py. */
int pmt_posterior_rows(const float* float_rows, int64_t float_stride, int32_t n_scalars, int32_t logit_col, const float* logits_b,
                       const float* features_be, int32_t e, const int64_t* dest_ids, int32_t n, float* block, int64_t block_stride,
                       void* stream);

/* Partition variants into register-resident groups: greedy over consecutive variants so that each group has
 * <= PMT_GROUP_MAX_SETS sets and its tiles fit the workgroup: ref tiles and alt tiles go to disjoint waves, two per wave,
 * i.e. ceil(ceil(ref/16) / 2) + ceil(ceil(alt/16) / 2) <= PMT_GROUP_WAVES (so never more than PMT_GROUP_TILES tiles).  Counts are HOST arrays
 * (upper bounds are fine: a DownsampledBatch reuses its parent's plan).  group_start / group_tile_base must hold
 * num_variants + 1 ints.  Returns the number of groups, or PMT_E_CAPACITY if one variant alone exceeds a group
 * (*bad_variant receives its index). */
int pmt_plan_groups(const int32_t* ref_counts, const int32_t* alt_counts, int32_t num_variants,
                    int32_t* group_start, int32_t* group_tile_base, int32_t* bad_variant);

/* pmt_plan_groups on the DEVICE, from exclusive scans of the counts that live there (a DownsampledBatch's: how many reads it keeps is
 * decided by pmt_downsample_counts and never comes to the host): two small launches, stream-ordered.  The batch is cut into
 * pmt_plan_device_chunks(num_variants) chunks of consecutive variants, each packed next-fit like pmt_plan_groups (a chunk boundary closes
 * a group).  group_start / group_tile_base: device, capacity + 1 ints; num_groups_dev: device, [1] -- PmtBatch.num_groups_dev of the
 * launches that use the plan, whose PmtBatch.num_groups is then `capacity` (the grid).  capacity >= the groups of ANY plan of larger
 * counts in the same order + the number of chunks is always enough; an overflow, or a read set beyond one workgroup, sets
 * PMT_FAULT_PLAN in *fault (may be NULL).  The reference has no counterpart (its ATen kernels need no plan). */
int pmt_plan_groups_device(const int32_t* ref_offsets, const int32_t* alt_offsets, int32_t num_variants, int32_t* group_start,
                           int32_t* group_tile_base, int32_t capacity, int32_t* num_groups_dev, int32_t* fault, int32_t* scratch,
                           void* stream);   /* scratch: device, 2 * pmt_plan_device_chunks(num_variants) ints (contents irrelevant) */
int pmt_plan_device_chunks(int32_t num_variants);

/* An order of the batch's variants in which pmt_plan_groups packs fuller groups (a workgroup costs the same full or not, so
 * the number of groups is what a batch costs): the group under construction takes, out of the next `window` unplaced
 * variants, the first that still fits.  order[i] (host, num_variants ints) = the variant to put at position i of the batch.
 * The reference composes batches in random order (data/reads_dataset.py:141-196), so any order is as good to it. */
int pmt_pack_order(const int32_t* ref_counts, const int32_t* alt_counts, int32_t num_variants, int32_t window, int32_t* order);
/* The same for consecutive batches of `batch` variants (the batches of a dataset chunk) in one call, on `threads` host
 * threads: order[k * batch + i] = position in the whole array of the variant to put at place i of batch k. */
int pmt_pack_order_batches(const int32_t* ref_counts, const int32_t* alt_counts, int32_t num_variants, int32_t batch,
                           int32_t window, int32_t threads, int32_t* order);

/* Like pmt_plan_groups, but a variant whose reads exceed one workgroup is split over several groups (each a run of whole
 * tiles of its ref rows and / or alt rows).  span: host [max_groups][6] (PmtBatch.group_span), tile_base: host
 * [max_groups + 1].  *needs_layered is set when some group covers only part of a read set (then only pmt_forward_layered
 * can run the batch).  Returns the number of groups or PMT_E_WORKSPACE if max_groups is too small. */
int pmt_plan_groups_split(const int32_t* ref_counts, const int32_t* alt_counts, int32_t num_variants, int32_t* span,
                          int32_t* tile_base, int32_t max_groups, int32_t* needs_layered);

/* Floats of scratch pmt_forward_layered needs. */
size_t pmt_layered_scratch_floats(const PmtModel* model, int64_t total_tiles, int32_t num_variants);

/* The forward for batches with read sets of ANY size (BASELINE config: 600 reads per variant): the same kernel, run as
 * num_blocks + 1 launches.  A read set couples its reads only through per-set sums (the gating mean fields of every block,
 * reference gated_mlp.py:236-239, and the head's sums, feature_clustering.py:111-113); launch s finishes block s - 1 from
 * the complete sums and starts block s, the sums accumulate in HBM with float atomics, activations rest in `scratch` between
 * launches, and a last launch finalises the per-set outputs.  Same results as pmt_forward on batches it accepts.
 * `stash` (optional) as in pmt_forward; pmt_backward_layered is its backward. */
int pmt_forward_layered(const PmtModel* model_host, const PmtModel* model_dev, const float* theta, const float* phi,
                        const float* packed, const PmtBatch* batch, const PmtOutputs* out, float* stash, float* scratch,
                        void* stream);

/* Floats of scratch pmt_backward_layered needs. */
size_t pmt_layered_backward_scratch_floats(const PmtModel* model, int64_t total_tiles, int32_t num_variants);

/* Backward of pmt_forward_layered (same arguments as pmt_backward plus `scratch`): num_blocks + 1 launches of the backward
 * kernel.  Going down the stack, a block's backward needs the per-set sums of d(gate) over ALL reads of the set before the
 * gradient can pass its first projection, so launch s finishes block L - s from the complete sums (accumulated in HBM by
 * launch s - 1) and starts block L - s - 1; the running gradient and the half-finished block's per-read state rest in
 * `scratch` between launches.  Per-set parameter terms are added by the group that holds the set's first alt read.
 * grad_variant_embed must be zeroed by the caller (several groups add to a row).  grad_partials / num_partials: as in
 * pmt_backward (the rows are folded once, after the last launch). */
int pmt_backward_layered(const PmtModel* model_host, const PmtModel* model_dev, const float* theta, const float* phi,
                         const float* packed, const PmtBatch* batch, const PmtOutputs* out, const PmtOutputGrads* dout,
                         const float* stash, float* scratch, float* grad_theta, float* grad_phi, float* grad_variant_embed,
                         float* grad_partials, int32_t num_partials, void* stream);

/* Bytes of activation stash a training forward needs for `total_tiles` tiles (group_tile_base[G]) and B variants. */
size_t pmt_stash_bytes(const PmtModel* model, int64_t total_tiles, int32_t num_variants);

/* ---- device entry points ---------------------------------------------------------------------------------- */

/* Re-pack natural-layout parameters (theta: the flat leaf buffer the optimizer updates; phi: materialised
 * parametrizations) into MFMA fragment order.  Run after every optimizer step.  `model_dev` is a device copy of
 * the descriptor. */
int pmt_pack_params(const PmtModel* model_host, const PmtModel* model_dev, const float* theta, const float* phi,
                    float* packed, void* stream);

/* Exclusive scans of per-variant counts (int32 or int64, device) into ref_offsets/alt_offsets [B+1].
 * Replaces the host-synchronising torch.sum(...).item() of reference artifact_model.py:241. */
int pmt_scan_counts(const void* ref_counts, const void* alt_counts, int32_t count_elem_bytes, int64_t count_stride,
                    int32_t num_variants, int32_t* ref_offsets, int32_t* alt_offsets, void* stream);

/* Gather index of a batch whose variants are rows of a dataset chunk resident in HBM: row_start[b] = first read row of
 * variant b inside the chunk (its ref rows, then its alt rows, as on disk: reference data/memory_mapped_data.py:39-40);
 * read_index[R] lists the chunk rows in batch order (all ref rows of all variants, then all alt rows: reference
 * data/batch.py:45-47) and is what PmtBatch.read_index consumes.  Replaces the per-Datum Python collate
 * (reference data/batch.py:41-62) and the re-upload of every read for every batch. */
int pmt_build_read_index(const int64_t* row_start, const int32_t* ref_offsets, const int32_t* alt_offsets,
                         int32_t num_variants, int64_t* read_index, void* stream);

/* Host-side staging copy for the dataset loader (reference data/reads_dataset.py:141-196 reads the memory map Datum by
 * Datum): copies `bytes` from src (a memory-mapped file or host array) to dst (a pinned staging buffer) with `threads`
 * worker threads, outside the Python GIL.  Pure host code, no HIP call. */
/* One batch composed on the device from a chunk of the dataset resident in HBM as it lies on disk (int16 / float16 per-variant
 * tables, CSR row starts): the batch's variants are rows ids[0 .. num_variants) of the chunk.  Writes the Batch's tables in the
 * reference's dtypes (data/batch.py:47-49: int64 [B][int_cols], float32 [B][float_cols]), the variants' first read rows, the
 * exclusive scans of the two count columns ([B + 1] each) and the gather index of the reads (all ref rows, then all alt rows;
 * sized by the caller: the batch's total reads).  Everything device memory; asynchronous on `stream`. */
int pmt_compose_batch(const int16_t* chunk_ints, int32_t int_cols, const void* chunk_floats_f16, int32_t float_cols,
                      const int64_t* chunk_row_start, const int64_t* ids, int32_t num_variants, int32_t ref_col, int32_t alt_col,
                      int64_t* int_tensor, float* float_tensor, int64_t* row_start, int32_t* ref_offsets, int32_t* alt_offsets,
                      int64_t* read_index, void* stream);
/* pmt_compose_batch with the scans of the counts supplied (device copies of pmt_prepare_chunk's `offsets` rows): one launch. */
int pmt_compose_batch_planned(const int16_t* chunk_ints, int32_t int_cols, const void* chunk_floats_f16, int32_t float_cols,
                              const int64_t* chunk_row_start, const int64_t* ids, int32_t num_variants,
                              const int32_t* ref_offsets, const int32_t* alt_offsets, int64_t* int_tensor, float* float_tensor,
                              int64_t* row_start, int64_t* read_index, void* stream);

int pmt_host_copy(void* dst, const void* src, size_t bytes, int32_t threads);
/* The same for `rows` rows of `row_bytes` bytes with bytes [zero_offset, zero_offset + zero_bytes) of every row cleared in the
 * same pass: the integer rows of the posterior hand-off (reference tools/filter_variants.py:305-308: the datum's own integer
 * array with REF_COUNT and ALT_COUNT set to zero). */
int pmt_host_copy_rows(void* dst, const void* src, int64_t rows, int64_t row_bytes, int64_t zero_offset, int64_t zero_bytes,
                       int32_t threads);
/* The host side of one chunk of the device chunk loader in ONE call (no Python, no GIL): the chunk's read counts, the order in
 * which its variants are consumed (optionally shuffled: Fisher-Yates on a splitmix64 stream seeded by `seed`; then ordered
 * inside every batch of `batch` variants by pmt_pack_order with `window`), and every batch's group plan (pmt_plan_groups).
 * Replaces the per-Datum bookkeeping of reference data/reads_dataset.py:141-196 + data/batch.py:41-62 for a whole chunk.
 *   ints: the dataset's int16 table (row_stride in elements), ref_col / alt_col: its REF_COUNT / ALT_COUNT columns
 *   ref_host, alt_host [n]: out; ids [n]: out, ids in consumption order (batch k = ids[k * batch, (k + 1) * batch))
 *   plans [plans_capacity ints]: out, per batch group_start (g + 1) then group_tile_base (g + 1), back to back
 *   batch_info [batches][4]: out, {offset in plans, groups, total tiles, total reads}
 * Returns the number of batches; PMT_E_CAPACITY if a read set exceeds one workgroup (plan such a chunk with
 * pmt_plan_groups_split); PMT_E_WORKSPACE if `plans` is too small (2 * (n + batches) ints always suffice). */
int pmt_prepare_chunk(const int16_t* ints, int64_t row_stride, int32_t ref_col, int32_t alt_col, int32_t n, int32_t shuffle,
                      uint64_t seed, int32_t batch, int32_t window, int32_t threads, int32_t* ref_host, int32_t* alt_host,
                      int64_t* ids, int32_t* plans, int64_t plans_capacity, int32_t* batch_info, int32_t* offsets);
/* (offsets: optional, host, [batches][2][batch + 1]: per batch the exclusive scans of its ref counts and of its alt counts in its
 *  consumption order -- what pmt_compose_batch_planned takes, so that composing a batch on the device is one launch) */

/* Fused read-set forward: decode -> read MLP -> concat -> L gated ref/alt blocks -> reducer -> rotation ->
 * clustering head + per-set sums.  Replaces ArtifactModel.calculate_features + FeatureClustering.calculate_logits
 * + RaggedSets.means_over_sets (reference artifact_model.py:239-297).  `stash` = NULL for inference; otherwise the
 * activations the backward pass re-reads are written there (pmt_stash_bytes). */
int pmt_forward(const PmtModel* model_host, const PmtModel* model_dev, const float* theta, const float* phi,
                const float* packed, const PmtBatch* batch, const PmtOutputs* out, float* stash, void* stream);

/* Fused backward of pmt_forward.  Accumulates into grad_theta / grad_phi (same layouts as theta / phi; the caller zeroes
 * them) and writes grad_variant_embed [B][E_v].  Replaces autograd over the same graph (reference misc_utils.py:127).
 * grad_partials (optional, device): [num_partials][PmtModel.emit_len] floats, ALL ZERO on entry and all zero again on return.
 * With it the kernel runs as min(groups, num_partials) persistent workgroups, each adding its weight-gradient blocks into its
 * OWN row with plain loads and stores, and a second small launch folds the rows into grad_theta / grad_phi; without it
 * (NULL / 0) every block goes out as global float atomics, whose throughput then costs a tenth of the kernel's time.  One
 * row per compute unit (256) is the intended size. */
int pmt_backward(const PmtModel* model_host, const PmtModel* model_dev, const float* theta, const float* phi,
                 const float* packed, const PmtBatch* batch, const PmtOutputs* out, const PmtOutputGrads* dout,
                 const float* stash, float* grad_theta, float* grad_phi, float* grad_variant_embed, float* grad_partials,
                 int32_t num_partials, void* stream);

/* The dropout mask of one linear's output as the kernels generate it (host computation, no GPU; pmt_dropout.hpp): out[r][f] =
 * 1 / (1 - p) where element (row0 + r, f) of the output of linear `lin` (index into PmtModel.lin) is kept under `seed`, else 0.
 * For the parity tests: the oracle applies exactly these masks (reference mlp.py:57-58 draws its own from torch's generator). */
int pmt_dropout_mask(uint64_t seed, float p, int32_t lin, int64_t row0, int64_t rows, int32_t width, float* out);

/* Row-wise MLP over N independent rows -- the per-variant branches: `which` = PMT_ROWS_INFO (info_embedding,
 * reference artifact_model.py:244), PMT_ROWS_ALT_COUNT (alt_count_predictor, :180-183, :276-279) or PMT_ROWS_SOURCE
 * (source_predictor, :267-274).  in: [n_rows] rows of in_dim floats with the given row stride (floats); out likewise, so a
 * result can be written straight into a column block of a wider matrix.  stash = NULL for inference, else
 * pmt_rows_stash_bytes() bytes that the backward pass re-reads. */
size_t pmt_rows_stash_bytes(const PmtModel* model, int which, int32_t n_rows);
int pmt_rows_forward(const PmtModel* model_host, const PmtModel* model_dev, int which, const float* theta,
                     const float* packed, const float* in, int64_t in_stride, int32_t n_rows, float* out,
                     int64_t out_stride, float* stash, uint64_t dropout_seed, void* stream);
/* (dropout_seed: as PmtBatch.dropout_seed; row r of the masks is row r of `in`.)
 * Backward of pmt_rows_forward: accumulates parameter gradients into grad_theta and, if d_in != NULL, writes
 * d_in = d_in_scale * dL/d(in)  (d_in_scale = -alpha implements the reference's gradient reversal,
 * gradient_reversal/functional.py:18-22).
 * `workspace` (optional, device, >= pmt_rows_workspace_floats(model, which) floats, ALL ZERO at the first use; every call
 * leaves it zero again; may be shared by the three MLPs if sized for the largest): the workgroups add their weight-gradient
 * blocks into replicas of the MLP's parameter range and a second launch sums the replicas into grad_theta.  With NULL every
 * workgroup adds to grad_theta itself: hundreds of float atomics per address at the same moment, which the L2 serialises. */
int pmt_rows_backward(const PmtModel* model_host, const PmtModel* model_dev, int which, const float* theta,
                      const float* packed, const float* in, int64_t in_stride, int32_t n_rows, const float* d_out,
                      int64_t d_out_stride, const float* stash, float* grad_theta, float* d_in, int64_t d_in_stride,
                      float d_in_scale, float* workspace, size_t workspace_floats, uint64_t dropout_seed, void* stream);
size_t pmt_rows_workspace_floats(const PmtModel* model_host, int which);

/* Haplotype CNN: haplotypes = device int64 [n][H] rows (values 0..4: A, C, G, T, indel; ref half then alt half,
 * reference data/batch.py:110-130) with the given row stride in elements; out = [n][cnn.out_dim] with row stride.
 * Replaces Batch.get_one_hot_haplotypes_bcs + DNASequenceConvolution.forward (artifact_model.py:245). */
int pmt_cnn_forward(const PmtModel* model_host, const PmtModel* model_dev, const float* theta, const float* packed,
                    const int64_t* haplotypes, int64_t hap_stride, int32_t n, float* out, int64_t out_stride, float* stash,
                    void* stream);
/* Floats per variant of the optional activation stash of a training forward (every layer output, as the backward keeps
 * them in LDS), or 0 when this model runs on kernels that recompute instead (then pass stash = NULL). */
size_t pmt_cnn_stash_floats(const PmtModel* model_host);
/* Backward: accumulates parameter gradients into grad_theta.  With `stash` (written by pmt_cnn_forward for the
 * same haplotypes and weights) the layer outputs are loaded; with NULL the forward is recomputed in LDS (a third of the
 * kernel's time).
 * `workspace` (optional, device, >= pmt_cnn_workspace_floats(model) floats, contents irrelevant before and after): the
 * workgroups store their weight-gradient sums as private rows and a second launch folds the rows into grad_theta.  With
 * NULL (or too small a workspace) every wave adds its sums with global float atomics: thousands of adds to each address
 * within a few microseconds, which the L2 serialises (half of the kernel's time at 65 536 variants). */
int pmt_cnn_backward(const PmtModel* model_host, const PmtModel* model_dev, const float* theta, const float* packed,
                     const int64_t* haplotypes, int64_t hap_stride, int32_t n, const float* d_out, int64_t d_out_stride,
                     const float* stash, float* grad_theta, float* workspace, size_t workspace_floats, void* stream);
/* Floats of workspace pmt_cnn_backward can use for this model on the current device (0: its kernels for this model use atomics). */
size_t pmt_cnn_workspace_floats(const PmtModel* model_host);

/* Global-norm clip + AdamW over the flat parameter buffer, one launch sequence, no host sync.
 * Replaces nn.utils.clip_grad_norm_(max_norm=1.0) + torch.optim.AdamW.step (reference misc_utils.py:128-129).
 * `scratch` holds >= 1024 floats.  grad_norm_out (device, optional) receives the pre-clip global L2 norm. */
int pmt_clip_adamw(float* theta, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                   const PmtAdamW* hyper, float* scratch, float* grad_norm_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PERMUTECT_AMD_H */
