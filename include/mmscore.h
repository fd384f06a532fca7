/*
 * libmmscore -- C ABI of the MI355X-native (gfx950) cross-modal pair scorer.
 *
 * The reference (zuokai/KDDCUP_2020_MultimodalitiesRecall_2nd_Place) has no FFI of its own: its
 * hot path is entered through three Python/TF1/PyTorch call surfaces.  Each entry point below
 * replaces one of them (paths relative to the reference root):
 *
 *   mms_score_zk      <- model_triple.model_attention_channel_e(...)  code/imagebert_zk/model_triple.py:162-214,
 *                        driven by sess.run([probs, ...])              code/imagebert_zk/evaluate_normal.py:227-238
 *   mms_score_lds     <- bertmodel(... features ...)                   code/imagebert_lds/src/run_pretraining_predict_score.py:288-336,
 *                        get_next_sentence_output                      :479-501, sess.run :566-570
 *   mms_score_lxmert  <- KDDModel.forward(...)                         code/lxmert/src/tasks/kdd_model.py:183-214,
 *                        called from KDD.predict                       :97-100
 *   mms_load_weight   <- tf.train.Saver.restore / load_state_dict      evaluate_normal.py:204-212,
 *                        run_pretraining_predict_score.py:558-563, kdd_model.py:131-152
 *                        (tensor names are the reference's own checkpoint variable names)
 *
 * Conventions
 *   - plain C, no torch / STL types; every function returns an int status (0 = MMS_OK) and never
 *     throws; mms_last_error(h) gives the message of the last failing call on that handle.
 *   - all batch pointers are DEVICE pointers (HBM-resident inputs) unless the field says host;
 *     the caller owns every input/output buffer, the handle owns weights and workspace.
 *   - one handle per GPU/stream; calls on one handle must be serialised by the caller; kernels are
 *     enqueued asynchronously on the hipStream_t passed as `stream` (NULL = default stream).
 *     Host synchronisation points inside a scoring call (each a 4-byte device->host read + hipStreamSynchronize of `stream`):
 *       * dense label ids (uniq_label_ids == NULL): the count of distinct label tuples, once per call;
 *       * lxmert / ensemble with pack_tokens = 1 and >= 2 pairs: the count of distinct (input_ids, input_mask) rows, once per call
 *         -- ALSO when the caller passes uniq_label_ids (the language layers run once per distinct query, api.hip lx_query_stage);
 *       * ensemble: the per-wave counts of pairs whose query the sen2forest rewrite changed, once per call;
 *       * a workspace that has to grow (first call, or a larger batch than any before) allocates, which synchronises the device.
 *     A zk / lds call with de-duplicated labels on a warmed-up handle is fully asynchronous.
 *     Memory: the workspace is sized for the largest launch wave seen (<= chunk_pairs pairs, ~1 MB per pair); lxmert additionally
 *     keeps the language rows of the distinct queries of the largest BATCH seen (Q x text_len x 768 x 4 bytes, Q <= B / 2) outside
 *     the per-wave workspace; nothing shrinks before mms_destroy.
 *   - integer dtypes are the reference feed dtypes (zk: int32 ids, int64 labels; lds/lxmert: int64).
 *   - the library reads no environment variable (A/B knobs and timing-only diagnostics live in the separate lab build,
 *     csrc/Makefile `make lab`, which the package never loads).
 */
#ifndef MMSCORE_H
#define MMSCORE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMS_OK 0
#define MMS_ERR_ARG 1
#define MMS_ERR_WEIGHT 2
#define MMS_ERR_HIP 3
#define MMS_ERR_STATE 4

#define MMS_MODEL_ZK 0
#define MMS_MODEL_LDS 1
#define MMS_MODEL_LXMERT 2

typedef struct mms_handle mms_handle;

typedef struct mms_config {
    int32_t model;            /* MMS_MODEL_* */
    int32_t layers;           /* zk/lds: encoder layers (12); lxmert: language layers (9) */
    int32_t r_layers;         /* lxmert: relational (vision) layers (5) */
    int32_t x_layers;         /* lxmert: cross-modality layers (5) */
    int32_t vocab;            /* 21128 */
    int32_t inter;            /* 3072, multiple of 128 */
    int32_t max_pos;          /* 512 */
    int32_t type_vocab;       /* 2 */
    int32_t text_len;         /* zk/lds 20, lxmert 23 */
    int32_t precision;        /* 1: bf16 activations, one MFMA pass; 2: split-bf16 activations (hi+lo), two passes, GEMM
                                 weights stored as bf16; 3: activations AND weights split (three passes): follows an
                                 arbitrary fp32 checkpoint to ~2e-5; 4: fp8 (OCP e4m3) weights with one fp32 scale per output
                                 channel and fp8 activations on the four big encoder GEMM classes (QKV, attention output,
                                 FFN up / down), fp8 MFMA -- BASELINE.json config 5; OUTSIDE the 1e-3 logit contract, its measured
                                 deviation is reported by the tests and bench.py */
    int32_t chunk_pairs;      /* max pairs per internal launch wave (0 = default 32768); a batch is cut into equal chunks */
    int32_t stop_after;       /* debug: run only the first n encoder layers (-1 = all) and skip nothing else */
    int32_t device;           /* HIP device ordinal */
    int32_t pack_tokens;      /* 1 = skip token rows that cannot change a logit; 0 = the reference's padded rows.
                                 zk / lxmert: padded tokens whose keys are masked are dropped (a masked key's softmax weight is
                                 exactly 0 in fp32).  lds has no mask (pixelmodel.py:189-190), but its feature and label tokens carry
                                 no position embedding, so identical rows of a pair (the zero-padded boxes, boxes of one class) stay
                                 identical through every layer: one representative is kept and its key carries log(multiplicity) --
                                 the same softmax and the same P V up to fp32 round-off */
    int32_t fuse_layernorm;   /* mask -- bit 0: the attention-output projections, bit 1: the FFN-down projections of launches with
                                 >= 16384 rows add bias + residual and apply the LayerNorm in their GEMM epilogue (gemm_pp_ln.h: the
                                 residual rides in as eight extra K stages against an identity built in registers, the three column
                                 tiles of a row panel exchange row statistics across workgroups; precision mode 2 only), instead of
                                 writing the fp32 sum for a separate LayerNorm kernel.  Same results to fp32 round-off (one-pass
                                 variance); 12 -> 6 KB of HBM traffic per row and LayerNorm, no LayerNorm launches; +1.5 % on the
                                 bench batch.  scorers.py passes 3; a launch whose workgroups are not all
                                 resident (GPU shared with another process) falls back to the two-kernel route by itself */
    int32_t fuse_attention;   /* 1: the self-attention sub-layers of launches with >= 16384 rows run the Q / K / V projection and the
                                 attention in ONE kernel (qkv_attn.hip: a workgroup projects one head of a 256-row tile, keeps the 192
                                 result columns in LDS and attends from there), so the fp32 [rows][2304] Q | K | V tensor never goes
                                 through HBM.  Precision modes 2 and 3 (mode 3: a 192-row tile variant with the weights' lo plane in the ring); context
                                 rows bit-identical to the two-kernel route.
                                 2: the attention's Q K^T and P V on split-bf16 MFMAs (hi + lo operands, three products, v_exp_f32 / v_rcp_f32 softmax)
                                 instead of exact-fp32 MFMAs: ~2^-16 relative on the scores.  Precision mode 2 runs the 8 x 1-wave kernel (Q operands
                                 straight from the accumulators, K / V staged as bf16 planes in LDS, 16-query tiles attended block-diagonally over the
                                 pairs they touch with an online softmax): any pair length that fits a 128-row sub-tile, lxmert's 10-token box stream
                                 included, and BOTH directions of lxmert's cross-attention (lxrt/modeling.py:460-464) in one launch per X layer.
                                 scorers.py's "auto": 2 in precision mode 2, 1 in mode 3 */
} mms_config;

/* zk feed, code/imagebert_zk/evaluate_normal.py:141-152.  np_idx_class_labels [B,10,8] is either passed as the reference
 * feeds it (`label_ids`, dense: the library finds the distinct 8-id tuples itself -- the label-text encoder depends on the
 * tuple only -- which costs one 4-byte device->host read on `stream` per call), or already de-duplicated by the caller:
 * uniq_label_ids [U,8] + label_index [B,10] (U = B*10 and index = arange reproduces the dense graph).  uniq_label_ids ==
 * NULL selects the dense form. */
typedef struct mms_zk_batch {
    int64_t n_pairs;
    const int32_t* num_boxes;      /* [B] */
    const float* boxes_5;          /* [B,10,5] */
    const float* feats;            /* [B,10,2048] */
    const int32_t* uniq_label_ids; /* [U,8] or NULL */
    int64_t n_uniq_labels;
    const int32_t* label_index;    /* [B,10] -> row of uniq_label_ids (values outside [0,U) are clamped) */
    const int32_t* query_ids;      /* [B,text_len] */
    const int32_t* len_query;      /* [B] */
    const int64_t* labels;         /* [B] (AM-softmax head is label dependent, model_triple.py:68-81) */
    const int32_t* segment_ids;    /* [B,text_len+10]; ids are clamped to [0, type_vocab) */
    const int32_t* label_ids;      /* [B,10,8] dense np_idx_class_labels, read when uniq_label_ids == NULL */
} mms_zk_batch;

/* lds feed, code/imagebert_lds/src/run_pretraining_predict_score.py:526-548 (boxes is unused there) */
typedef struct mms_lds_batch {
    int64_t n_pairs;
    const int64_t* input_ids;      /* [B,text_len] */
    const int64_t* segment_ids;    /* [B,text_len] */
    const float* features;         /* [B,10,2048] */
    const int64_t* labelfeat;      /* [B,10,8] */
} mms_lds_batch;

/* lxmert feed, code/lxmert/src/tasks/kdd_model.py:183-186 (label text dense or de-duplicated, as for zk) */
typedef struct mms_lxmert_batch {
    int64_t n_pairs;
    const int64_t* input_ids;      /* [B,text_len] */
    const int64_t* input_mask;     /* [B,text_len] */
    const int64_t* uniq_label_ids; /* [U,8] or NULL */
    int64_t n_uniq_labels;
    const int32_t* label_index;    /* [B,10] */
    const float* feats;            /* [B,10,2048] */
    const float* boxes;            /* [B,10,4] */
    const float* visual_attention_mask; /* [B,10] */
    const int64_t* label_ids;      /* [B,10,8] dense boxes_label_input_ids, read when uniq_label_ids == NULL */
    float* x_norm;                 /* optional OUTPUT, device fp32 [B,768]: pooled / max(||pooled||, 1e-12), the first element
                                      of KDDModel.forward's return tuple (kdd_model.py:204-205,214); NULL = not wanted */
} mms_lxmert_batch;

/* The three models on the SAME pairs in one call -- BASELINE.json config 5, what code/main.py:41-59 merges from four score files:
 * imagebert_zk on the query, imagebert_zk on its sen2forest rewrite (evaluate_normal_sen2fs.py, load_data_v4.py:153-154),
 * imagebert_lds and lxmert.  One TSV record feeds all of them (load_data_v4.py:133-163, load_data_pred.py:94-121,
 * lxmert/src/utils.py:23-59), so the image side is passed ONCE: the 2048-d box features are split into operand planes once per
 * launch wave, the class-name tuples are de-duplicated once, zk's image-token stage (label conv, kdd_conv2, kdd_dense1,
 * kdd_featureemb: model_triple.py:189-195, pixelbert.py:449-452) runs once for both query variants.  All ids are int32 here. */
typedef struct mms_ensemble_batch {
    int64_t n_pairs;
    const float* feats;            /* [B,10,2048] */
    const float* boxes_5;          /* [B,10,5] = corners/[h,w,h,w] + area (load_data_v4.py:141-147); lxmert reads the 4 corners */
    const int32_t* num_boxes;      /* [B]; lxmert's visual mask = box index < min(num_boxes, 10) */
    const int32_t* label_ids;      /* [B,10,8] */
    const int32_t* query_ids;      /* [B,zk.text_len]  zk + lds query ([CLS] .. [SEP], zero padded) */
    const int32_t* len_query;      /* [B] */
    const int32_t* s2f_query_ids;  /* [B,zk.text_len]  the sen2forest rewrite of the same query */
    const int32_t* s2f_len_query;  /* [B] */
    const int64_t* labels;         /* [B] zk AM-softmax label (1 on testB, ground truth on valid: load_data_v4.py:259-265) */
    const int32_t* lx_input_ids;   /* [B,lxmert.text_len]  lxmert tokenizer flavour */
    const int32_t* lx_input_mask;  /* [B,lxmert.text_len] */
} mms_ensemble_batch;

/* ABI revision of the structs and entry points declared in this header.  It changes whenever a struct gains a field or an entry
 * point changes its signature (r1: 1; r2 added mms_config.fuse_layernorm, mms_zk_batch.label_ids, mms_lxmert_batch.label_ids / x_norm
 * without bumping it; r3: 3, then 4 with mms_config.fuse_attention; r4: 5 -- mms_dbg_gemm takes the engine per call,
 * the process-global mms_set_gemm_variant and the lab engines' hooks left the product library, fuse_layernorm became a mask; r5: 6 -- mms_dbg_qkv_attn and mms_side_lane_flops added).  A caller built against another revision would make the library read past its structs, so compare
 * BEFORE the first mms_create:  if (mms_version() != MMS_ABI_VERSION) abort();   (lib.py's load() does) */
#define MMS_ABI_VERSION 6
int mms_version(void);
const char* mms_global_error(void);              /* message of the last failing mms_create */

int mms_create(const mms_config* cfg, mms_handle** out);
void mms_destroy(mms_handle* h);
const char* mms_last_error(const mms_handle* h);

/* host fp32 tensor, copied; names = reference checkpoint variable names (see weights.py) */
int mms_load_weight(mms_handle* h, const char* name, const float* host_data, const int64_t* shape, int32_t rank);
/* builds the device-resident model: bf16 [N][K] GEMM matrices (fused QKV), fp32 tables/vectors */
int mms_finalize(mms_handle* h);

/* logits / probs: device fp32 [B,2] (probs may be NULL).
 * Pairs are independent: how a caller groups them into calls (or mms_config.chunk_pairs into launch waves) changes a logit by fp32 summation
 * order at most (~1e-5 relative), and not at all between launches of the same SIZE REGIME.  A launch's regime is decided by its PADDED row bound
 * (pairs x sequence length of the stream; the live count stays on the device), per projection class -- every boundary that changes a summation order:
 * GENERATED REGIMES BEGIN (tools/gen_regime_doc.py from csrc/regimes.h -- do not edit)
 *   rows <= 128     SKINNY_ROWS      [speed choice only: bit-identical across it]  every projection on the skinny kernel (gemm_skinny.hip), K sliced
 *                                    exactly as the tile route of the same projection slices it
 *   rows <  1024    TINY_ROWS        wide projections (N >= 1536, K = 768: QKV, K | V, FFN-up): K in 4 slices, summed in fixed order
 *                                    (k_splitk_reduce)
 *   rows >= 1024    FUSED_ATTN_ROWS  mms_config.fuse_attention: a stream's QKV projection + attention in one kernel (qkv_attn.hip); 1 = the
 *                                    two-kernel route's arithmetic (bit-identical to it), 2 = split-bf16 attention over 16-query tiles of a packed
 *                                    sub-tile: a pair's logits depend on its place in the launch by fp32 round-off (<= 1e-4 relative)
 *   rows <  4096    TALL_ROWS        the long-K projections in front of the encoder (K >= 2048, N = 768: zk kdd_conv1 as im2col, kdd_conv2 / visn_fc
 *                                    / featureemb over the box rows): K in 8 slices
 *   rows >= 4096    SPLITK_HALF_ROWS the LayerNorm-followed K >= 2048 projections (FFN-down) of the split-K regime: 4 K slices instead of 8
 *   rows >= 5120    PP_WIDE_ROWS     [speed choice only: bit-identical across it]  wide projections (N >= 1536) on the persistent ping-pong engine
 *                                    instead of the 128 x 256 tiles (same contraction order per element)
 *   rows <  8192    SPLITK_ROWS      the N = 768 projections that a LayerNorm follows (attention output, FFN-down): K in 4 (K = 768) / 8 or 4 (K >=
 *                                    2048) slices, summed by the LayerNorm kernel
 *   rows >= 11264   SPLITK2_LO_ROWS  FFN-down (K = 3072) of launches below PP_ROWS: 2 K slices (one pass between SPLITK_ROWS and here)
 *   rows >= 16384   PP_ROWS          [speed choice only: bit-identical across it]  persistent ping-pong engines for every projection (same
 *                                    contraction order per element as the tiles: bit-identical, tested)
 *   rows >= 98304   LNF_ROWS         mms_config.fuse_layernorm: bias + residual + LayerNorm in the GEMM epilogue (one-pass variance) -- not in a call
 *                                    that runs on launch lanes, see LANE_ROWS / ENS_LANE_ROWS
 *   rows <  200000  ENS_LANE_ROWS    mms_score_ensemble: the three members of a launch wave (rows of its longest member; ~5000 pairs) side by side on
 *                                    three streams, every LayerNorm by its own kernel
 *   rows <  400000  LANE_ROWS        lxmert calls (pairs x (text_len + 10) rows; ~12 500 pairs): the two streams' launch chains on two lanes, every
 *                                    LayerNorm by its own kernel (with or without per-launch timing)
 * GENERATED REGIMES END
 * (below TINY_ROWS / SPLITK_ROWS / PP_ROWS a projection runs on the register-staged / LDS-DMA 128 x 256 tiles, one pass over K unless a line above says otherwise.)
 * Batch composition enters in two more places: lxmert with pack_tokens runs its language layers once per DISTINCT query when at least half of the pairs
 * share theirs (the rows of that stage = distinct queries x text_len), and label texts are encoded once per distinct 8-id tuple (rows = tuples).
 * mms_score_ensemble runs the three members of a launch wave of fewer than 5000 pairs side by side on three streams (same rule: every LayerNorm by its own kernel; a member's
 * scores equal the single-model call's bit for bit wherever that call would not have used the fused epilogue either, i.e. below 98304 token rows).
 * lxmert's two launch lanes (a side stream between fork / join events: the vision stream's sub-layers beside the language stream's between two cross attentions, the
 * distinct-query stage beside the box stream's layers) run the same kernels on the same operands: no numerical effect beyond the LayerNorm route named above.
 * Not numerical boundaries (same arithmetic, tested bit-identical): the ping-pong engine taken for the wide projections (N >= 1536) from 5120 rows on and for every
 * projection from 16384, the LDS-DMA tile variant taken when a launch has no more workgroups than CUs, the
 * read-back-free label de-duplication of calls of <= 51 pairs, packed vs. dense token layout.
 * A call is ~110 (zk, lds) to ~220 (lxmert) dependent launches: 0.6 - 1.1 ms at 1 pair, so batch thousands of pairs per call when
 * throughput matters ("Call sizes"). */
int mms_score_zk(mms_handle* h, const mms_zk_batch* b, float* logits, float* probs, void* stream);
int mms_score_lds(mms_handle* h, const mms_lds_batch* b, float* logits, float* probs, void* stream);
int mms_score_lxmert(mms_handle* h, const mms_lxmert_batch* b, float* logits, float* probs, void* stream);
/* merged[i] = w[0]*zk(query) + w[1]*zk(sen2forest query) + w[2]*lds + w[3]*lxmert, each member = softmax(logit)[1]
 * (main.py:59: w = 0.2, 0.2, 0.3, 0.3).  merged: device fp32 [B]; member_scores: optional device fp32 [4,B] (the four columns
 * main.py reads from the score files).  The three handles must live on the same device; zk and lds share text_len. */
int mms_score_ensemble(mms_handle* zk, mms_handle* lds, mms_handle* lxmert, const mms_ensemble_batch* b, const float* weights4,
                       float* merged, float* member_scores, void* stream);

/* accumulated hipEvent time (ms) and launch count of the GEMM kernels since the last reset;
 * enable = 1 brackets every GEMM launch with events on its stream (bench / roofline only) */
int mms_gemm_timing(mms_handle* h, int32_t enable, int32_t reset, double* ms_out, int64_t* launches_out, double* flops_out);
/* the same, for one class of the timed GEMM launches: cls 0 = plain epilogue, 1 = fused bias + residual + LayerNorm epilogue
 * (mms_config.fuse_layernorm; their duration includes the residual stages and the LayerNorm work, their FLOP count the projection alone) */
int mms_gemm_timing_class(mms_handle* h, int32_t cls, double* ms_out, int64_t* launches_out, double* flops_out);
/* The fused QKV + attention launches (mms_config.fuse_attention) of the calls timed through mms_gemm_timing (enabled / reset there),
 * reported apart from the GEMM launches because their duration includes the attention of their pairs: total ms, launches, executed
 * projection FLOPs. */
int mms_fused_timing(mms_handle* h, double* ms_out, int64_t* launches_out, double* flops_out);
/* ... and the executed FLOPs of the launches that ran on a side lane (lxmert's distinct-query stage): counted, not timed -- a launch that shares the chip has no duration of its own */
int mms_side_lane_flops(mms_handle* h, double* flops_out);

/* ---- debug / test hooks: NOT part of the product ABI --------------------------------------------------------------------------------
 * The kernel-level parity tests and tools/ call the same kernels the scorers launch through these; they are built into libmmscore.so (the
 * tests must exercise the shipped binary) but may change with any revision and take no part in MMS_ABI_VERSION.  A host that scores pairs
 * does not define MMS_TEST_HOOKS and does not see them. */
#ifdef MMS_TEST_HOOKS
int mms_debug_read_x(mms_handle* h, float* dst_dev, int64_t rows, void* stream); /* current hidden state -> fp32 [rows,768] */
int mms_dbg_gemm(const float* a_f32, int64_t M, int64_t K, int64_t lda, const float* w_f32_nk, int64_t N,
                 const float* bias, const float* resid_f32, int32_t act, int32_t nsplit, int32_t out_planes,
                 int32_t engine /* enum Engine of csrc/regimes.h, per CALL, no global state:
 * GENERATED ENGINES BEGIN (tools/gen_regime_doc.py from csrc/regimes.h -- do not edit)
 *     0 ENG_AUTO          the per-shape choice of a forward (gemm_dispatch.hip pick_engine)
 *     1 ENG_TILE_128      gemm_tile.hip: register-staged 128 x 128 tile (the only tile for N % 256 != 0, and the three-pass tile of precision mode 3)
 *     3 ENG_TILE_DMA      gemm_tile.hip: 128 x 256 tile, LDS-DMA double buffered, one workgroup per CU (launches of no more workgroups than CUs)
 *     4 ENG_TILE          gemm_tile.hip: register-staged 128 x 256 tile, the default below PP_ROWS
 *     5 ENG_SKINNY        gemm_skinny.hip: <= 128 rows, one workgroup per 16 output columns, K sliced over its waves (GemmParams::wave_k_slices)
 *    16 ENG_TILE_256      gemm_tile.hip: 256 x 256 / 16 waves (kept for the kernel tests; no forward selects it since round 4)
 *    20 ENG_PP            gemm_pp.hip: 256 x 256 ping-pong phases, one tile per workgroup
 *    26 ENG_PP_PERSIST    gemm_pp.hip: the same, persistent workgroups (XCD-aware tile walk): launches of >= PP_ROWS rows
 *    27 ENG_PPW           gemm_ppw.hip: 256 x 128 three-pass ping-pong (precision mode 3, >= PP_ROWS rows)
 *    28 ENG_DW            lab build: gemm_dw.hip (128 x 256, two 4-wave workgroups per CU) -- measured and shelved
 *    54 ENG_SKINNY_K4     kernel tests: the skinny kernel with 4 wave-level K slices
 *    55 ENG_SKINNY_PARTS  gemm_skinny.hip with the K slices dealt to single-wave WORKGROUPS: GemmParams::k_splits fp32 partials (the split-K contract
 *                         of the tiles)
 *    58 ENG_SKINNY_K8     kernel tests: 8 wave-level K slices
 * GENERATED ENGINES END
 */,
                 float* c_f32, void* stream);
/* fp8 GEMM (precision 4, MX-scaled fp8 MFMA) on fp32 operands: A and W are quantised exactly as the forward does it (A: e4m3 RNE of the
 * value; W: per output channel the smallest POWER OF TWO scale with max|w| / scale <= 448, e4m3 RNE of w / scale; the scale is applied
 * by the MFMA instruction itself as the weight operand's e8m0 hardware scale).  N % 256 == 0, K % 128 == 0 */
int mms_dbg_gemm_f8(const float* a_f32, int64_t M, int64_t K, const float* w_f32_nk, int64_t N, const float* bias, int32_t act,
                    int32_t out_f8, float* c_f32, void* stream);
/* out = LayerNorm(A W^T + bias + resid) over N = 768 through the GEMM with the fused LayerNorm epilogue (gemm_pp_ln.h) and the
 * LayerNorm kernel queued behind it; *mode_out = 1 when the launch ran fused, 2 when it took the plain two-kernel route */
int mms_dbg_gemm_ln(const float* a_f32, int64_t M, int64_t K, const float* w_f32_nk, const float* bias, const float* resid_f32,
                    const float* gamma, const float* beta, int32_t f8, float* c_f32, int32_t* mode_out, void* stream);
/* the small-launch route of the same operation (calls of a few hundred pairs): register-staged tiles with the contraction split `splits`
 * ways into fp32 partials (K % (64 * splits) == 0; 1 = unsplit) + the LayerNorm kernel that sums them and adds bias + residual; splits < 0: |splits|
 * partials from the skinny kernel instead (K slices dealt to workgroups, M <= 512): bit-identical to the tile engine's */
int mms_dbg_proj_ln_splitk(const float* a_f32, int64_t M, int64_t K, const float* w_f32_nk, const float* bias, const float* resid_f32,
                           const float* gamma, const float* beta, int32_t splits, float* c_f32, void* stream);
/* time one GEMM shape on random data (`what`: an engine as in mms_dbg_gemm, or 52: the MX-fp8 engine; 60 / 61: the LayerNorm kernel with / without
 * residual; 62 / 63: N = 768 projection + residual + LayerNorm as two kernels / as one launch with the fused epilogue -- api.hip enum BenchMode) */
int mms_dbg_gemm_bench(int64_t M, int64_t N, int64_t K, int32_t nsplit, int32_t act, int32_t out_planes, int32_t resid,
                       int32_t what, int32_t iters, float* ms_out);
int mms_dbg_attention(const float* q, const float* k, const float* v, int64_t B, int32_t Sq, int32_t Sk,
                      const float* key_add, float* out_f32, void* stream);
/* ONE fused QKV-projection + attention launch (qkv_attn.hip) on fp32 operands (all device pointers but n_sub_out): x [rows1 + rows2][768], the rows of stream 1
 * first; a pair's tokens are consecutive rows of a stream -- off / cnt [n_pairs] per stream (first row relative to the stream, tokens), or NULL = dense, S tokens per
 * pair; rows2 == 0: self-attention of stream 1, else CROSS attention: the queries of either stream attend the keys of the OTHER stream of their pair (lxmert X layers);
 * w_qkv [2304][768] = [Wq; Wk; Wv] in torch Linear layout, bias_qkv [2304]; key_add: additive key mask by stream row or NULL; mode 1: exact-fp32 attention MFMAs
 * (self-attention, pairs of <= 48 tokens), 2: the split-bf16 route (qkv_attn2_kernel).  ctx_f32 [rows1 + rows2][768]; *n_sub_out = sub-tiles the plan made */
int mms_dbg_qkv_attn(const float* x, int64_t rows1, int64_t rows2, const int32_t* off1, const int32_t* cnt1, const int32_t* off2, const int32_t* cnt2,
                     int64_t n_pairs, int32_t S1, int32_t S2, const float* w_qkv, const float* bias_qkv, const float* key_add1, const float* key_add2,
                     int32_t mode, float* ctx_f32, int32_t* n_sub_out, void* stream);
/* launch counters since mms_create: which = 0 -> fused QKV + attention launches (mms_config.fuse_attention took effect), 1 -> GEMM launches
 * with the fused LayerNorm epilogue (mms_config.fuse_layernorm), 2 -> split-K launches of the small-call routes, 3 -> skinny-GEMM launches (launches of <= 128
 * padded rows, precision modes 2 and 3), 4 -> fork / join pairs of the second launch lane (lxmert handles; on a zk handle: the member lanes of mms_score_ensemble);
 * anything else: -1 */
int64_t mms_dbg_counter(mms_handle* h, int32_t which);
int mms_dbg_layernorm(const float* x, const float* gamma, const float* beta, int64_t M, float* out_f32, void* stream);
#endif /* MMS_TEST_HOOKS */

#ifdef __cplusplus
}
#endif
#endif
