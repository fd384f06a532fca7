/*
 * libmmfeat -- native (C++17, multi-threaded) TSV record featurizer feeding the pair scorers.
 *
 * Replaces the per-line Python work of the reference's data loaders on the callers' side of the hot path
 * (paths relative to the reference root):
 *   read_line              code/imagebert_zk/load_data_v4.py:133-163, code/imagebert_lds/src/load_data_pred.py:94-121,
 *                          code/lxmert/src/utils.py:23-59   (tab split, base64 -> f32/i64 arrays, box normalisation + area,
 *                          label-text ids per box, "[CLS] query [SEP]" WordPiece ids, sen2forest rewrite)
 *   seq_padding(_2)        code/imagebert_zk/load_data_v4.py:78-102  (zero pad / truncate to 10 boxes, text_len tokens, 8 label ids)
 * and writes straight into caller-owned (ideally pinned) padded batch buffers, so one H2D copy per array follows.
 *
 * Tokenisation: ASCII queries are tokenised natively (lower-case, whitespace / ASCII-punctuation split, greedy
 * longest-match WordPiece) -- bit-identical to the BERT tokenizer on ASCII input.  A query containing any non-ASCII byte
 * is NOT tokenised here: its row in `needs_host_tokenizer` is set to 1 and the caller fills query ids with the full
 * Unicode tokenizer (featurizer.WordPieceTokenizer).  Label texts come pre-tokenised (there are only ~30 classes).
 *
 * Plain C ABI, no exceptions cross it; every function returns 0 on success or a negative error code, and
 * mmf_last_error() returns a thread-local message.
 */
#ifndef MMFEAT_H
#define MMFEAT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mmf_context mmf_context;

/* vocab: one token per line (id = line number).  max_chars: 200 (zk/lds tokenization.py:299) or 100 (lxmert).
 * never_split_specials: 1 = keep "[CLS]" "[SEP]" "[PAD]" "[UNK]" "[MASK]" whole (lxmert's HF tokenizer), 0 = stock BERT. */
/* base64 decoder tier: 0 scalar tables, 1 AVX2, 2 AVX-512 VBMI; the best one the CPU has is chosen at load.  set_to >= 0 selects one
 * (process-wide; tests run every tier against the scalar one), < 0 only reports.  Returns the tier in use, or -1. */
int mmf_b64_tier(int32_t set_to);

int mmf_create(const char* vocab_path, int32_t max_chars, int32_t never_split_specials, mmf_context** out);
void mmf_destroy(mmf_context* c);
const char* mmf_last_error(void);

/* class id -> pre-tokenised label text (<= 8 ids used, `len` = untruncated length, for the lxmert label mask) */
int mmf_set_label(mmf_context* c, int64_t class_id, const int32_t* ids, int32_t len);

/* ASCII tokeniser: returns the number of ids written (<= max_ids), or -2 if the text has non-ASCII bytes */
int mmf_tokenize_ascii(const mmf_context* c, const char* text, int64_t text_len, int32_t* ids, int32_t max_ids);

typedef struct mmf_batch_out {
    int64_t* product_id;        /* [B] */
    int64_t* query_id;          /* [B] */
    int32_t* num_boxes;         /* [B]  raw count (may exceed 10) */
    float* boxes;               /* [B,10,box_dim]  box_dim 5: 4 normalised corners + area (zk/lds); 4: corners (lxmert) */
    float* feats;               /* [B,10,2048], or NULL: the 2048-d features are not decoded (their base64 length is still checked) -- a second pass over
                                 * the same records that needs the query / label side only (pipeline.EnsembleScorer: the sen2forest and lxmert flavours) */
    int32_t* label_ids;         /* [B,10,8] */
    int32_t* label_len;         /* [B,10]  untruncated label-text length */
    int32_t* query_ids;         /* [B,text_len] zero padded / truncated */
    int32_t* query_len;         /* [B]  untruncated length incl. [CLS] [SEP] */
    uint8_t* needs_host_tokenizer; /* [B] 1 = query has non-ASCII bytes, query_ids/query_len not filled */
    int64_t* query_span;        /* [B,2] byte range of the query field inside `data` (for the host tokenizer) */
    int32_t* feat_rows_live;    /* [B] or NULL.  In: how many leading box rows of feats[i] may be non-zero (a fresh buffer: 10); out: the
                                 * record's box count.  With it the zero padding of a REUSED buffer is written only where the row's
                                 * previous record left data (60 % of the bytes of a row at 3.8 boxes per record); NULL: always. */
} mmf_batch_out;

/* lines: `n` records in one buffer, record i = data[offsets[i] .. offsets[i+1]) (no trailing newline needed).
 * Work is split over `threads` threads (<= 0: hardware concurrency): the caller plus helper threads the context keeps
 * between calls.  Every output row is fully written (padding zeroed; see feat_rows_live).  Calls on one context run one
 * at a time.  Returns 0, or -1000 - index of the first malformed record (message in mmf_last_error). */
int mmf_featurize(const mmf_context* c, const char* data, const int64_t* offsets, int64_t n, int32_t text_len,
                  int32_t box_dim, int32_t sen2forest, int32_t threads, const mmf_batch_out* out);

/* Same, with record i = data[starts[i] .. ends[i]) -- records need not be contiguous (skipped header / blank lines). */
int mmf_featurize_spans(const mmf_context* c, const char* data, const int64_t* starts, const int64_t* ends, int64_t n,
                        int32_t text_len, int32_t box_dim, int32_t sen2forest, int32_t threads, const mmf_batch_out* out);

/* Line splitter for a raw TSV buffer: fills starts/ends (newline excluded) for up to max_lines records, skipping blank
 * lines and lines containing "product_id" (the header test of code/lxmert/src/tasks/kdd_data.py:70-71).  *consumed = bytes of
 * `data` covered (resume the next call there).  Returns the number of records found, or -1. */
int64_t mmf_split_lines(const char* data, int64_t len, int64_t* starts, int64_t* ends, int64_t max_lines, int64_t* consumed);

/* query_id (the last tab-separated field) of `n` record spans, without decoding anything: what a rank needs to find ITS contiguous
 * query block of a shared TSV file (pipeline.stream_scores_tsv(shard=(rank, world))).  Returns 0, or
 * -1000 - index of the first record whose last field is not an integer. */
int mmf_query_ids(const char* data, const int64_t* starts, const int64_t* ends, int64_t n, int64_t* out);

/* Map the pages of [addr, addr + len) -- a read-only file mapping about to be split and decoded -- on `threads` threads
 * (MADV_POPULATE_READ per 2 MB piece).  A 50 KB record is ~12 pages; left to demand faulting, the ONE thread that splits
 * lines takes the faults of the whole file (fault-around maps the rest of each record with its head), which bounded the
 * 256-thread decode at ~450 k records/s (profiles/rd6_feat_sweep.txt).  Returns 0. */
int mmf_prefault(const mmf_context* c, const void* addr, int64_t len, int32_t threads);

/* [addr, addr + len) of a read-only FILE mapping has been decoded and will not be read again: the next mmf_featurize* call on
 * this context drops those pages from the page table (madvise MADV_DONTNEED, whole pages inside the range) on one of its threads
 * while the others decode.  Unmapping a 2.5 GB file at the end instead is 30 ms of serial kernel work -- a third of the pass
 * (profiles/rd6_feat_sweep.txt).  The data stays in the page cache; touching the range again simply faults it back in.
 * addr == NULL: forget every pending range (returns how many there were) -- REQUIRED before the mapping is unmapped (a range left pending would be dropped from
 * whatever is mapped at that address when the next decode runs: MADV_DONTNEED zeroes anonymous memory). */
int mmf_release_later(const mmf_context* c, const void* addr, int64_t len);

#ifdef __cplusplus
}
#endif
#endif
