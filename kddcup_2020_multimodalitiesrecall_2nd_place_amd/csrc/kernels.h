// Host-side launcher declarations for the libmmscore kernels (all launches are async on `st`).
#pragma once
#include "common.h"
#include "regimes.h"

// ---------------------------------------------------------------------------------------------
// GEMM  C[M,N] = act(A[M,K] * W[N,K]^T + bias (+ residual))      (gemm.hip)
// A is a split-plane activation (hi/lo bf16 in the hl32 layout of common.h: a_lo == a_hi + 32; lda, ldp, ldr are LOGICAL row lengths,
// multiples of 32), W is bf16 [N][K] stored as 16 x 32 tiles (wtile_off), N % 128 == 0, K % 64 == 0.  nsplit 1: hi plane only (1 MFMA pass), 2: A hi+lo (2 passes, parity mode for bf16-exact
// weights), 3: A hi+lo and W hi+lo (3 passes a_hi*w_hi + a_lo*w_hi + a_hi*w_lo: fp32-checkpoint-faithful).
// ---------------------------------------------------------------------------------------------
struct GemmParams {
    const bf16* a_hi; const bf16* a_lo; int lda;
    RowMap amap;                 // logical row -> A row
    const bf16* w;
    const bf16* w_lo;            // lo plane of the weights (precision mode 3) or nullptr
    const float* bias;           // [N] or nullptr
    int M, N, K;
    int out_kind, act;
    float* c_f32; int ldc;       // OUT_F32
    bf16* c_hi; bf16* c_lo; int ldp;  // OUT_PLANES
    RowMap cmap;                 // logical row -> output row
    const bf16* r_hi; const bf16* r_lo; int ldr;  // residual planes (logical rows) or nullptr
    const int* m_dev;            // optional device-side row count: M_eff = min(M, *m_dev) (packed mode)
    const int* a_index;          // optional row gather: logical row r reads A row a_index[r]
    RowMap rmap; const int* r_index;  // residual row of logical row r: r_index ? r_index[r] : rmap(r)
    unsigned long long* flop_counter;  // optional: block 0 adds 2*M_eff*N*K (executed algorithmic FLOPs)
    int reverse;                 // 1: walk the tiles from the last row panel to the first (cache-direction alternation, api.hip)
    // precision mode 4 (fp8 operands, gemm_pp.hip only): a_hi / w point at e4m3 bytes and lda / K count PAIRS of them (= the same
    // 2-byte units as the bf16 planes, so every address in the tile engine is unchanged); col_scale[n] = the weight row's
    // quantisation scale, applied to the accumulator before the bias; OUT_F8 stores e4m3 bytes at c_f8[row * ldf8 + col]
    // OUT_F32, head-major (hm_rows > 0): the 64-column group G = (hm_col0 + col) / 64 of row r goes to c_f32[(G * hm_rows + r) * 64 + ...]
    // -- the [Q | K | V] projection laid out as 36 contiguous [rows][64] blocks (12 heads each), which is how the attention kernel reads it
    int hm_rows, hm_col0;
    int f8;
    const float* col_scale;
    unsigned char* c_f8; int ldf8;
    // fused bias + residual + LayerNorm epilogue (gemm_pp_ln.h; N == 768, rows of the residual = rows of the output): ln_gamma != nullptr
    // selects it.  Outputs: planes c_hi / c_lo (+ c_f8) when the launch runs fused, plain fp32 c_f32 when it falls back (then the
    // LayerNorm kernel queued behind the GEMM does the work; it skips itself when ln_ctl[1] == 1).
    // precision mode 5 (gemm_mx.hip; h3 operands, common.h): a_hi = fp16 high plane [rows][lda] row-major (rows allocated up to a
    // multiple of 256), a8 = e4m3 low plane [rows][lda]; w = fp16 weights (w * 2^e16[n]) in 16 x 32 tiles, w8 = e4m3 weights
    // (w * 2^e8[n]) in 8-row x 128-byte tiles, w8_scale4 = per 64-channel group g and r < 16 one dword of e8m0 bytes
    // {127 + e16 - e8 of channels 64 g + r, + 16, + 32, + 48}, col_scale[n] = 2^-e16[n].  OUT_H3 stores c_h16 / c_l8 [rows][ldh].
    const unsigned char* a8; const unsigned char* w8; const unsigned* w8_scale4;
    f16* c_h16; unsigned char* c_l8; int ldh;
    const float* ln_gamma; const float* ln_beta;
    float* ln_stats;             // [rows][3 tiles][2] 8-byte {value, tag} granules
    unsigned ln_tag;             // unique per launch
    int* ln_ctl;                 // [0] workgroups checked in, [1] 0 undecided / 1 fused / 2 plain -- zeroed before the launch
    // split-K (gemm_tile.hip; small launches of the N = 768 projections): k_splits > 1 -> the grid holds k_splits copies of the tile grid, copy s
    // contracts K columns [s * K / k_splits, (s + 1) * K / k_splits) (a multiple of 64) and writes its fp32 partial to c_f32 + s * c_split_stride;
    // bias / residual / activation are NOT applied (the LayerNorm kernel behind it sums the partials: LnResid::nparts)
    int k_splits; long long c_split_stride;
    int wave_k_slices;           // ENG_SKINNY only: K slices dealt to the WAVES of a workgroup (1, 4 or 8; 0 -> 1), summed inside it -- bias / activation still in the epilogue.
                                 // NOT the split-K contract above: the two never share a field (ADVICE r4's fall-through bug lived in that overlap)
    int engine;                  // enum Engine (regimes.h).  ENG_AUTO: the per-shape choice (gemm_dispatch.hip pick_engine); else ONE named engine
    unsigned long long* ln_dbg;  // lab build only: per-tile phase stamps [virtual tile][6] of wall_clock64 (tools/ln_trace.py); nullptr otherwise
};
// CUs of the CURRENT device rounded down to a multiple of 8 (one per XCD-slot), cached per device ordinal; thread-safe (gemm_dispatch.hip)
int device_cu_count();
// padded row bound from which the persistent ping-pong engines (and the fused LayerNorm epilogue) take a launch (gemm_dispatch.hip)
int pp_rows();
bool launch_gemm(const GemmParams& p, int nsplit, hipStream_t st);      // false: no engine took the shape, nothing was launched
bool launch_gemm_tile(const GemmParams& p, int nsplit, Engine tile, hipStream_t st);  // gemm_tile.hip: ENG_TILE_128 / ENG_TILE_DMA / ENG_TILE / ENG_TILE_256
bool launch_gemm_skinny_parts(const GemmParams& p, int nsplit, hipStream_t st);        // ... with the K slices dealt to workgroups: k_splits fp32 partials, c_split_stride apart (ENG_SKINNY_PARTS; the LayerNorm kernel sums them)
bool launch_gemm_skinny(const GemmParams& p, int nsplit, hipStream_t st);              // gemm_skinny.hip: a handful of rows (api.hip: <= 128), precision modes 2 and 3, one workgroup per 16 output columns, K split over its waves (ENG_SKINNY, GemmParams::wave_k_slices); false: not taken
#ifdef MMS_LAB
bool launch_gemm_dw(const GemmParams& p, int nsplit, hipStream_t st);                                                       // lab: gemm_dw.hip (ENG_DW: 128x256 tiles, two 4-wave workgroups per CU)
#endif
bool launch_gemm_pp(const GemmParams& p, int nsplit, int diag, hipStream_t st, bool persist = false);                     // gemm_pp.hip (ENG_PP / ENG_PP_PERSIST: 256x256 ping-pong phases)
bool launch_gemm_pp_ln(const GemmParams& p, int nsplit, hipStream_t st);                 // gemm_pp.hip + gemm_pp_ln.h: N = 768 with the fused residual + LayerNorm epilogue (nsplit 2)
#ifdef MMS_LAB
bool launch_gemm_mx_hi_only(const GemmParams& p, hipStream_t st);                        // lab: timing reference, the high pass alone
#endif
bool launch_gemm_mx8(const GemmParams& p, hipStream_t st);                               // gemm_mx.hip: precision mode 4 on the MX-scaled fp8 instruction
#ifdef MMS_LAB
bool launch_gemm_mx(const GemmParams& p, hipStream_t st);                                // lab: gemm_mx.hip: fp16 high pass + MX-scaled e4m3 low pass ("precision 5", measured and shelved)
#endif
bool launch_gemm_ppw(const GemmParams& p, hipStream_t st);                               // gemm_ppw.hip: precision mode 3 (A and W split), 256x128 ping-pong phases

// ---------------------------------------------------------------------------------------------
// Attention for one (pair, head) per wavefront, S <= 48               (attn.hip)
// q/k/v: fp32 projections; row of (b, i) = base + b * S + i; o: split planes on the q rows.
// ---------------------------------------------------------------------------------------------
struct AttnParams {
    const float* q; int ldq;
    const float* k; const float* v; int ldkv;
    long long hs_q, hs_kv;       // element stride between heads: 64 inside a [rows][768+] row, rows * 64 for head-major blocks
    int q_base, Sq, kv_base, Sk;
    const float* key_add;        // additive mask indexed by kv row (relative to kv_base) or nullptr
    bf16* o_hi; bf16* o_lo; int ldo;
    unsigned char* o_f8;         // precision mode 4: the context goes out as e4m3 bytes (same row stride ldo) INSTEAD of the planes
    int B;
    // packed (ragged) mode: per-pair first row and live-token count; nullptr = dense (b*S, S).
    // Sq / Sk are then the MAXIMUM lengths (tile selection).
    const int* q_off; const int* q_cnt; const int* kv_off; const int* kv_cnt;
    int q_stride;                // dense q rows: q_base + b * q_stride (0 -> Sq)
    int o_compact;               // 1: Sq == 1 and the output row is b (CLS-only last layer)
    int reverse;                 // 1: last pair first
};
bool launch_attention(const AttnParams& p, hipStream_t st);   // false: (Sq, Sk) beyond the instantiated tiles (> 48 tokens)

// Fused QKV projection + self-attention of one token stream (qkv_attn.hip; precision mode 2).  The stream's pairs are packed into
// sub-tiles of <= 128 rows by launch_qkv_tile_plan (once per stream and call); the weights / bias are in head-major order
// ([12 heads][Q 64 | K 64 | V 64] rows).
struct QkvAttnParams {
    const bf16* a_hi; int lda;               // hidden-state planes (hl32), row stride in elements
    const bf16* w; const float* bias; int K; // head-major tiled [2304][K] weights, head-major bias
    const bf16* w_lo;                        // precision mode 3: the weights' lo plane (same tiling), else nullptr
    const int4* sub; const int* n_sub;       // sub-tile table {first row, rows, first pair, pairs} and its device-side length
    const int* pair_off; const int* pair_cnt; int S;   // packed: first row / live tokens per pair; dense (nullptr): S tokens per pair.  S = maximum tokens
    const float* key_add;                    // additive key mask by stream row, or nullptr
    bf16* o_hi; bf16* o_lo; int ldo;         // context planes
    int M; const int* m_dev;                 // rows of the stream (upper bound / device-side live count): grid size, FLOP count
    int reverse;
    int fast;                                // 1: attention on split-bf16 MFMAs (three products per operand pair) instead of exact-fp32 MFMAs
    // CROSS mode (fast only; lxmert X layers, lxrt/modeling.py:460-464: ONE attention module, both directions): a sub-tile holds the rows of its
    // pairs in BOTH streams -- sub[u] = {first row, rows, first pair, pairs} of stream 1, sub2[u] = {first row, rows, -, -} of stream 2 (rows
    // relative to their stream); stream 2 starts at row row0_b of the plane buffers (a_hi, o_hi).  Queries of one stream attend the keys of
    // the OTHER stream of their pair.  sub2 == nullptr: self-attention of one stream.
    const int4* sub2;
    const int* pair_rec;                     // fast: per pair, its rows inside its sub-tile (launch_qkv_tile_plan / launch_qkv_cross_plan write it next to the table)
    const int* pair_off2; const int* pair_cnt2; int S2;     // stream 2: first row / live tokens per pair (nullptr: dense, S2 tokens per pair)
    const float* key_add2;                   // additive key mask of stream 2 by stream row, or nullptr
    long long row0_b;
    int M2; const int* m_dev2;               // rows of stream 2 (upper bound / device-side live count)
    unsigned long long* flop_counter;
    unsigned long long* trace;               // lab builds: per-tile timeline (qkv_attn.hip), nullptr otherwise
    int lab_flags;                           // lab builds: timing-only knock-outs
};
// sub-tile table of a token stream (n pairs; off / cnt: first row / live tokens per pair or nullptr = dense, S tokens each); pair_rec: int[n] (QkvAttnParams::pair_rec)
// or nullptr; scratch: qkv_plan_scratch_ints(n) ints.  false: more pairs than one plan takes (40 K: the caller keeps the two-kernel attention route)
bool launch_qkv_tile_plan(const int* off, const int* cnt, int n, int S, int4* sub, int* n_sub, int passes, hipStream_t st, int* pair_rec, int* scratch);
// ... of a PAIR of streams (cross mode): a pair brings cnt[b] + cnt2[b] rows, sub / sub2 as QkvAttnParams describes them
bool launch_qkv_cross_plan(const int* off, const int* cnt, const int* off2, const int* cnt2, int n, int S, int S2,
                           int4* sub, int4* sub2, int* n_sub, int passes, hipStream_t st, int* pair_rec, int* scratch);
int qkv_plan_scratch_ints(int n);
bool launch_qkv_attn(const QkvAttnParams& p, hipStream_t st);   // false: shape not supported (S > 48, K % 64)

// ---------------------------------------------------------------------------------------------
// Row-wise kernels (one wavefront per 768-wide row)                   (rowops.hip)
// ---------------------------------------------------------------------------------------------
// optional residual of the LayerNorm input: row r adds planes row (r_index ? r_index[r] : rmap(r)) before normalising
struct LnResid { const bf16* hi = nullptr; const bf16* lo = nullptr; int ld = 0; RowMap rmap{0, 0, 0}; const int* r_index = nullptr; int reverse = 0;
                 unsigned char* o_f8 = nullptr;
                 const int* skip = nullptr;
                 int nparts = 1; long long part_stride = 0; const float* bias = nullptr; };   // nparts > 1: in = sum of nparts fp32 partials (split-K GEMM), + bias   // skip: the kernel returns at once when *skip == 1 (the producing GEMM already normalised: gemm_pp_ln.h)   // o_f8: additionally write the row as e4m3 bytes (row stride ldo; precision mode 4)
void launch_ln_to_planes(const float* in, int ld, const float* gamma, const float* beta,
                         bf16* o_hi, bf16* o_lo, int ldo, int M, hipStream_t st, const int* m_dev = nullptr, LnResid res = LnResid());
void launch_split_f32(const float* in, bf16* o_hi, bf16* o_lo, long long n, hipStream_t st);
// split-K reduce + epilogue: out = act(sum_s parts[s] + bias) as plain / head-major fp32 (c_f32) or split planes (c_hi / c_lo)
void launch_splitk_reduce(const float* parts, int S, long long stride, int M, int N, const int* m_dev, const float* bias, int act, float* c_f32, int ldc,
                          int hm_rows, int hm_col0, bf16* c_hi, bf16* c_lo, int ldp, hipStream_t st);
void launch_planes_to_f32(const bf16* hi, const bf16* lo, float* out, long long n, hipStream_t st);
void launch_tile_weights(const float* in, bf16* o_hi, bf16* o_lo, long long N, long long K, hipStream_t st);   // fp32 W[N][K] -> tiled bf16 (hi, optional lo)
void launch_mean8(const float* in, float* out, int U, hipStream_t st, const int* U_dev = nullptr);      // U_dev: device-side row count (U is then an upper bound)
void launch_scale_count(const int* in, int mul, int* out, hipStream_t st);                               // *out = *in * mul

// zk (code/imagebert_zk/model_triple.py:162-214, pixelbert.py:541-621)
void launch_zk_im2col(const float* E, const int* uniq_ids, int U, int vocab, bf16* o_hi, bf16* o_lo, hipStream_t st, const int* U_dev = nullptr);
void launch_zk_tokpre(const float* labfeat, const int* lab_index, int n_labels, const float* boxes5, const float* Wd,
                      const float* bd, const float* img, bf16* o_hi, bf16* o_lo, int rows, hipStream_t st,
                      const int* src = nullptr, const int* rows_dev = nullptr);      // src: compact rows, row r stands for box src[r]; r < *rows_dev
// live boxes of a packed zk wave: cnt / off per pair, box_idx[off[b] + j] = b * 10 + j, *rows_dev = their number (rowops.hip)
void launch_zk_box_plan(const int* len_query, const int* num_boxes, int T, int n, int* cnt, int* off, int* box_idx, int* rows_dev, hipStream_t st);
// n <= 1024 pairs: box plan + token plan (launch_zk_pack_plan) in ONE single-block launch; rows_dev[0] = live tokens, rows_dev[1] = live boxes; false: n too large
bool launch_zk_plans_small(const int* len_query, const int* num_boxes, int T, int n, int* b_cnt, int* b_off, int* box_idx, int* t_cnt, int* t_off,
                           int* tok_src, float* key_add, int* rows_dev, hipStream_t st);
void launch_split_f32_rows(const float* in, const int* idx, const int* rows_dev, int max_rows, int width, bf16* o_hi, bf16* o_lo, hipStream_t st);
void launch_zk_embed(const float* E, const float* type_tab, const float* pos_tab, const float* gamma,
                     const float* beta, const int* query_ids, const int* segment_ids, const float* tok,
                     int T, int vocab, bf16* o_hi, bf16* o_lo, int B, hipStream_t st);
// packed (ragged) execution: drop padded tokens whose keys are masked (results are identical: a
// masked key's softmax weight underflows to exactly 0 in fp32)
void launch_zk_pack_plan(const int* len_query, const int* num_boxes, int T, int n, int* off, int* cnt, int* tok_src,
                         float* key_add, int* rows_dev, hipStream_t st);
void launch_zk_embed_packed(const float* E, const float* type_tab, const float* pos_tab, const float* gamma,
                            const float* beta, const int* query_ids, const int* segment_ids, const float* tok,
                            int T, int vocab, const int* tok_src, const int* rows_dev, int max_rows,
                            bf16* o_hi, bf16* o_lo, hipStream_t st, const int* box_off = nullptr);      // box_off: tok holds live boxes only, pair b's at row box_off[b]
// lds: identical feature / label token rows of a pair merged, multiplicity as an additive log on the key (rowops.hip)
void launch_lds_pack_plan(const float* feats, const int64_t* labelfeat, int T, int n, int* nz_flags, int* off, int* cnt, int* tok_src,
                          float* key_add, int* rows_dev, hipStream_t st);
void launch_lx_pack_plan(const int64_t* input_mask, const float* visual_mask, int T, int n, int* l_off, int* l_cnt,
                         int* l_src, float* l_add, int* l_rows, int* v_off, int* v_cnt, int* v_src, float* v_add,
                         int* v_rows, hipStream_t st);
void launch_lx_embed_lang_packed(const float* E, const float* pos_tab, const float* type_tab, const float* gamma,
                                 const float* beta, const int64_t* input_ids, int T, int vocab, const int* src,
                                 const int* rows_dev, int max_rows, bf16* o_hi, bf16* o_lo, hipStream_t st);
void launch_zk_mask(const int* len_query, const int* num_boxes, int T, float* key_add, int B, hipStream_t st);
void launch_zk_head(const float* pooled, const float* am_kernel, const int64_t* labels, float scale,
                    float margin, float* logits, float* probs, int B, hipStream_t st);

// lds (code/imagebert_lds/src/pixelmodel.py:444-602, run_pretraining_predict_score.py:479-501)
void launch_lds_embed_text(const float* E, const float* type_tab, const float* pos_tab, const float* gamma,
                           const float* beta, const int64_t* input_ids, const int64_t* segment_ids,
                           int T, int S, int vocab, bf16* o_hi, bf16* o_lo, int B, hipStream_t st);
void launch_lds_label(const float* E, const float* wl, const int64_t* labelfeat, int vocab, int S,
                      int row_off, bf16* o_hi, bf16* o_lo, int B, hipStream_t st);
void launch_lds_head(const float* pooled, const float* W, const float* b, float* logits, float* probs,
                     int B, hipStream_t st);

// lxmert (code/lxmert/src/lxrt/modeling.py:269-297,496-533,872-927; tasks/kdd_model.py:167-214)
void launch_lx_embed_lang(const float* E, const float* pos_tab, const float* type_tab, const float* gamma,
                          const float* beta, const int64_t* input_ids, int T, int vocab,
                          bf16* o_hi, bf16* o_lo, int B, hipStream_t st);
void launch_lx_label_emb(const float* E, const float* pos_tab, const float* type_tab, const float* gamma,
                         const float* beta, const float* conv_w, const float* conv_b,
                         const int64_t* uniq_ids, int vocab, bf16* o_hi, bf16* o_lo, int U, hipStream_t st);
void launch_lx_visn(const float* xf, const float* g_x, const float* b_x, const float* boxes, int box_dim,
                    const float* Wb, const float* bb, const float* g_y, const float* b_y, const float* z,
                    const int* lab_index, int n_labels, bf16* o_hi, bf16* o_lo, int rows, hipStream_t st,
                    const int* src = nullptr, const int* rows_dev = nullptr, int xf_compact = 0);   // xf_compact: xf row = OUTPUT row (projection ran on the live boxes only)
void launch_ln_f32(const float* in, const float* gamma, const float* beta, float* out, int M, hipStream_t st);
void launch_lx_masks(const int64_t* input_mask, const float* visual_mask, int T, float* lang_add,
                     float* visn_add, int B, hipStream_t st);
void launch_lx_head(const float* h, const float* gamma, const float* beta, const float* W, const float* b,
                    float* logits, float* probs, int B, hipStream_t st);

// ---------------------------------------------------------------------------------------------
// Per-batch bookkeeping (batchops.hip): label-tuple de-duplication, feed conversions of the fused three-model entry point
// ---------------------------------------------------------------------------------------------
// slots: int[cap], cap a power of two >= 2 * rows; rep / uid / index: int[rows]; counter: int (number of distinct tuples on return);
// uniq32 / uniq64: [rows,8] worst case, either may be nullptr
void launch_label_dedup_i32(const int32_t* ids, int rows, int* slots, int cap, int* rep, int* uid, int* counter, int32_t* uniq32,
                            int64_t* uniq64, int* index, hipStream_t st);
void launch_label_dedup_i64(const int64_t* ids, int rows, int* slots, int cap, int* rep, int* uid, int* counter, int32_t* uniq32,
                            int64_t* uniq64, int* index, hipStream_t st);
// distinct (input_ids, input_mask) rows of a lxmert batch: index[b] = distinct-query number of pair b, rows_of[u] = the FIRST pair row holding query u; the queries
// are numbered in the order of their first occurrence (deterministic).  slots: int[2 * cap]
void launch_query_dedup(const int64_t* ids, const int64_t* mask, int T, int rows, int* slots, int cap, int* rep, int* uid, int* counter,
                        int* rows_of, int* index, hipStream_t st);
void launch_gather_i64_rows(const int64_t* in, const int* rows_of, int T, long long n_rows, int64_t* out, hipStream_t st);
// packed plane rows r < *rows_dev:  scatter: d[map[r]] = s[r];  gather: d[r] = s[idx[map[r] / T] * T + map[r] % T]
void launch_rows_scatter(const bf16* s_hi, const bf16* s_lo, const int* map, const int* rows_dev, int max_rows, bf16* d_hi, bf16* d_lo,
                         hipStream_t st);
void launch_rows_pick(const bf16* s_hi, const bf16* s_lo, const int* map, const int* rows_dev, int max_rows, bf16* d_hi, bf16* d_lo,
                      hipStream_t st);   // d[r] = s[map[r]]
void launch_rows_gather(const bf16* s_hi, const bf16* s_lo, const int* map, const int* idx, int T, const int* rows_dev, int max_rows,
                        bf16* d_hi, bf16* d_lo, hipStream_t st);
// ensemble, second zk member: diff[b] = the rewritten query of pair b differs from the original; compact list of those pairs; row gathers;
// out[b] = diff[b] ? changed[off[b]] : same[b] on [n,2] rows
void launch_query_differs(const int32_t* q1, const int32_t* l1, const int32_t* q2, const int32_t* l2, int T, int n, int* diff, hipStream_t st);
void launch_compact_list(const int* diff, const int* off, int n, int* list, hipStream_t st);
void launch_gather_rows_i32(const int32_t* in, const int* list, int width, long long n_rows, int32_t* out, hipStream_t st);
void launch_gather_rows_i64(const int64_t* in, const int* list, int width, long long n_rows, int64_t* out, hipStream_t st);
void launch_gather_rows_f32x4(const float* in, const int* list, int width, long long n_rows, float* out, hipStream_t st);
void launch_select_rows2(const int* diff, const int* off, const float* same, const float* changed, int n, float* out, hipStream_t st);
void launch_plan_scan(const int* cnt, int n, int* off, int* total_dev, hipStream_t st);   // exclusive scan of n counts (rowops.hip)
void launch_i32_to_i64(const int32_t* in, int64_t* out, long long n, hipStream_t st);
void launch_fill_i64(int64_t* out, long long n, int64_t v, hipStream_t st);
void launch_zk_segment_ids(int32_t* out, long long B, int T, hipStream_t st);
void launch_box_mask(const int32_t* num_boxes, float* mask, long long B, hipStream_t st);
void launch_corners(const float* boxes5, float* boxes4, long long B, hipStream_t st);
void launch_merge4(const float* const probs[4], const float w[4], float* merged, float* members, long long n, hipStream_t st);
void launch_xnorm(const bf16* hi, const bf16* lo, float* out, int rows, hipStream_t st);

// precision mode 5 helpers (rowops.hip): h3 operand planes, weight copies of gemm_mx.hip (N % 64 == 0, K % 128 == 0)
void launch_split_h3(const float* in, f16* o_h, unsigned char* o_l, long long n, hipStream_t st);
void launch_planes_to_h3(const bf16* hi, const bf16* lo, f16* o_h, unsigned char* o_l, long long n, hipStream_t st);
void launch_h3_to_f32(const f16* h, const unsigned char* l, float* out, long long n, hipStream_t st);
void launch_prep_w_mx(const float* w, f16* w16, unsigned char* w8, unsigned* w8_scale4, float* col_scale, int N, int K, hipStream_t st);
// precision mode 4 helpers (rowops.hip)
void launch_planes_to_f8(const bf16* hi, const bf16* lo, unsigned char* out, long long n, hipStream_t st);
void launch_f32_to_f8(const float* in, unsigned char* out, long long n, hipStream_t st);
void launch_f8_to_f32(const unsigned char* in, float* out, long long n, hipStream_t st);
// weight rows [N][K] fp32 (N % 64 == 0, K % 128 == 0) -> e4m3 bytes in 8 x 128-byte tiles + one power-of-two scale per row (the smallest
// 2^e with max|w| / 2^e <= 448) as e8m0 bytes packed for gemm_mx8_kernel (scale4) and, optionally, as fp32 (scale)
void launch_quant_rows_f8(const float* w, unsigned char* out, float* scale, unsigned* scale4, int N, int K, hipStream_t st);
