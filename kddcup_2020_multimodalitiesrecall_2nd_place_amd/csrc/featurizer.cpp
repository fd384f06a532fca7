// libmmfeat: native multi-threaded TSV record featurizer (see include/mmfeat.h for the contract and the reference
// lines each step restates).  Host-only C++17, no HIP: built with g++ into csrc/libmmfeat.so.
#include <immintrin.h>
#include <linux/futex.h>
#include <sys/mman.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <atomic>
#include <climits>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/mmfeat.h"

namespace {

constexpr int N_BOX = 10, LABEL_LEN = 8, FEAT_DIM = 2048;
thread_local std::string g_err;

struct Label { int32_t ids[LABEL_LEN]; int32_t len; };

inline bool is_ascii_punct(unsigned char c) {
    return (c >= 33 && c <= 47) || (c >= 58 && c <= 64) || (c >= 91 && c <= 96) || (c >= 123 && c <= 126);
}
inline bool is_ascii_space(unsigned char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r'; }
inline bool is_ascii_ctrl(unsigned char c) { return (c < 32 && !is_ascii_space(c)) || c == 127; }

// base64 (standard alphabet).  Four pre-shifted tables turn each 4-character group into one OR + three byte stores;
// an invalid character sets bit 24+ of the OR.  Writes at most `cap` bytes; returns bytes written or -1 on a bad character.
int8_t B64[256];
alignas(64) int8_t B64V[128];            // AVX-512 tier: 6-bit value per ASCII character, 0x80 = outside the alphabet
alignas(64) int8_t B64PK[64];            // ... and the byte order that compacts sixteen 24-bit groups
uint32_t B64T[4][256];
constexpr uint32_t B64_BAD = 0x01000000u;
struct B64Init {
    B64Init() {
        for (int i = 0; i < 256; ++i) B64[i] = -1;
        const char* a = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
        for (int i = 0; i < 64; ++i) B64[(unsigned char)a[i]] = (int8_t)i;
        for (int i = 0; i < 256; ++i)
            for (int k = 0; k < 4; ++k) B64T[k][i] = B64[i] < 0 ? B64_BAD : (uint32_t)B64[i] << (18 - 6 * k);
        for (int c = 0; c < 128; ++c) B64V[c] = B64[c] < 0 ? (int8_t)0x80 : B64[c];
        for (int k = 0; k < 16; ++k) { B64PK[3 * k] = (int8_t)(4 * k + 2); B64PK[3 * k + 1] = (int8_t)(4 * k + 1); B64PK[3 * k + 2] = (int8_t)(4 * k); }
        for (int k = 48; k < 64; ++k) B64PK[k] = 0;
    }
} b64_init;

// Whole 4-character groups with the scalar tables; stops in front of '=' padding or a bad character (the caller's tail loop decides which).
inline void b64_groups_scalar(const unsigned char* u, int64_t n, unsigned char* out, int64_t cap, int64_t& i, int64_t& o) {
    for (; i + 4 <= n && o + 3 <= cap; i += 4, o += 3) {
        const uint32_t x = B64T[0][u[i]] | B64T[1][u[i + 1]] | B64T[2][u[i + 2]] | B64T[3][u[i + 3]];
        if (x & B64_BAD) break;
        out[o] = (unsigned char)(x >> 16); out[o + 1] = (unsigned char)(x >> 8); out[o + 2] = (unsigned char)x;
    }
}

// The bulk of a record is ~41 KB of base64 (2048 floats per box): vector decoders for it, chosen once per process by CPUID.  Both follow
// Mula & Lemire, "Faster Base64 Encoding and Decoding using AVX2 Instructions" (2018) / "Base64 encoding and decoding at almost the speed
// of a memory copy" (2019): translate characters to 6-bit values with byte shuffles, fold four of them into three bytes with two multiply-
// adds, compact.  A block that holds anything outside the alphabet ('=' included) is left to the scalar code, so results and error
// behaviour are those of the table decoder, bit for bit (tests/test_featurizer_native.py runs every tier the CPU has).
__attribute__((target("avx2")))
void b64_bulk_avx2(const unsigned char* u, int64_t n, unsigned char* out, int64_t cap, int64_t& i, int64_t& o) {
    const __m256i lut_lo = _mm256_setr_epi8(0x15, 0x11, 0x11, 0x11, 0x11, 0x11, 0x11, 0x11, 0x11, 0x11, 0x13, 0x1A, 0x1B, 0x1B, 0x1B, 0x1A,
                                            0x15, 0x11, 0x11, 0x11, 0x11, 0x11, 0x11, 0x11, 0x11, 0x11, 0x13, 0x1A, 0x1B, 0x1B, 0x1B, 0x1A);
    const __m256i lut_hi = _mm256_setr_epi8(0x10, 0x10, 0x01, 0x02, 0x04, 0x08, 0x04, 0x08, 0x10, 0x10, 0x10, 0x10, 0x10, 0x10, 0x10, 0x10,
                                            0x10, 0x10, 0x01, 0x02, 0x04, 0x08, 0x04, 0x08, 0x10, 0x10, 0x10, 0x10, 0x10, 0x10, 0x10, 0x10);
    const __m256i lut_roll = _mm256_setr_epi8(0, 16, 19, 4, -65, -65, -71, -71, 0, 0, 0, 0, 0, 0, 0, 0,
                                              0, 16, 19, 4, -65, -65, -71, -71, 0, 0, 0, 0, 0, 0, 0, 0);
    const __m256i m2f = _mm256_set1_epi8(0x2F);
    const __m256i pack = _mm256_setr_epi8(2, 1, 0, 6, 5, 4, 10, 9, 8, 14, 13, 12, -1, -1, -1, -1, 2, 1, 0, 6, 5, 4, 10, 9, 8, 14, 13, 12, -1, -1, -1, -1);
    const __m256i lanes = _mm256_setr_epi32(0, 1, 2, 4, 5, 6, -1, -1);
    while (i + 32 <= n && o + 32 <= cap) {          // 32 characters -> 24 bytes (the store is 32 wide: the next block overwrites the slack)
        __m256i v = _mm256_loadu_si256((const __m256i*)(u + i));
        const __m256i hi_n = _mm256_and_si256(_mm256_srli_epi32(v, 4), m2f);
        const __m256i lo_n = _mm256_and_si256(v, m2f);
        if (!_mm256_testz_si256(_mm256_shuffle_epi8(lut_lo, lo_n), _mm256_shuffle_epi8(lut_hi, hi_n))) break;
        const __m256i roll = _mm256_shuffle_epi8(lut_roll, _mm256_add_epi8(_mm256_cmpeq_epi8(v, m2f), hi_n));
        v = _mm256_add_epi8(v, roll);
        v = _mm256_madd_epi16(_mm256_maddubs_epi16(v, _mm256_set1_epi32(0x01400140)), _mm256_set1_epi32(0x00011000));
        v = _mm256_permutevar8x32_epi32(_mm256_shuffle_epi8(v, pack), lanes);
        _mm256_storeu_si256((__m256i*)(out + o), v);
        i += 32; o += 24;
    }
}

__attribute__((target("avx512f,avx512bw,avx512vbmi")))
void b64_bulk_avx512(const unsigned char* u, int64_t n, unsigned char* out, int64_t cap, int64_t& i, int64_t& o) {
    const __m512i lut0 = _mm512_load_si512(B64V), lut1 = _mm512_load_si512(B64V + 64);
    const __m512i pack = _mm512_load_si512(B64PK);
    while (i + 64 <= n && o + 64 <= cap) {          // 64 characters -> 48 bytes (store 64 wide, as above)
        const __m512i v = _mm512_loadu_si512(u + i);
        const __m512i t = _mm512_permutex2var_epi8(lut0, v, lut1);                  // by the low 7 bits of each character
        if (_mm512_movepi8_mask(_mm512_or_si512(v, t))) break;                       // non-ASCII byte, or a character outside the alphabet
        const __m512i m = _mm512_madd_epi16(_mm512_maddubs_epi16(t, _mm512_set1_epi32(0x01400140)), _mm512_set1_epi32(0x00011000));
        _mm512_storeu_si512(out + o, _mm512_permutexvar_epi8(pack, m));
        i += 64; o += 48;
    }
}

// threads <= 0: every hardware thread up to MMF_MAX_DEFAULT_THREADS.  On the 2 x 64-core / 256-thread host of the MI355X box 32 threads decode
// 580 .. 790 k records/s, 64 threads 470 .. 680 k, 128 threads 375 .. 610 k depending on the box and the run, 256 are always slower (a 8192-record
// call is ~3 ms of work per thread at that width, less than it takes to wake them): profiles/rd6_feat_sweep.txt, rd6_e2e_tsv.txt
constexpr int MMF_MAX_DEFAULT_THREADS = 32;
inline int default_threads() {
    const int hw = (int)std::thread::hardware_concurrency();
    return hw < 1 ? 1 : hw > MMF_MAX_DEFAULT_THREADS ? MMF_MAX_DEFAULT_THREADS : hw;
}

using BulkFn = void (*)(const unsigned char*, int64_t, unsigned char*, int64_t, int64_t&, int64_t&);
int g_b64_tier = -1;                     // 0 scalar, 1 AVX2, 2 AVX-512 VBMI
BulkFn g_b64_bulk = nullptr;
int b64_best_tier() {
    __builtin_cpu_init();
    if (__builtin_cpu_supports("avx512vbmi") && __builtin_cpu_supports("avx512bw")) return 2;
    if (__builtin_cpu_supports("avx2")) return 1;
    return 0;
}
bool b64_set_tier(int tier) {
    if (tier < 0 || tier > b64_best_tier()) return false;
    g_b64_tier = tier;
    g_b64_bulk = tier == 2 ? b64_bulk_avx512 : tier == 1 ? b64_bulk_avx2 : nullptr;
    return true;
}
struct B64Tier { B64Tier() { b64_set_tier(b64_best_tier()); } } b64_tier_init;

int64_t b64_decode(const char* s, int64_t n, unsigned char* out, int64_t cap) {
    const unsigned char* u = (const unsigned char*)s;
    int64_t o = 0, i = 0;
    if (g_b64_bulk) g_b64_bulk(u, n, out, cap, i, o);
    b64_groups_scalar(u, n, out, cap, i, o);
    uint32_t acc = 0;
    int bits = 0;
    for (; i < n && o < cap; ++i) {
        if (u[i] == '=') break;
        const int8_t v = B64[u[i]];
        if (v < 0) return -1;
        acc = (acc << 6) | (uint32_t)v;
        bits += 6;
        if (bits >= 8) {
            bits -= 8;
            out[o++] = (unsigned char)((acc >> bits) & 0xFF);
        }
    }
    return o;
}

inline int64_t b64_decoded_size(const char* s, int64_t n) {
    while (n > 0 && s[n - 1] == '=') --n;
    return n * 6 / 8;
}

bool parse_i64(const char* s, int64_t n, int64_t* out) {
    while (n > 0 && (*s == ' ')) { ++s; --n; }
    while (n > 0 && (s[n - 1] == ' ')) --n;
    if (n <= 0) return false;
    bool neg = false;
    int64_t i = 0, v = 0;
    if (s[0] == '-' || s[0] == '+') { neg = s[0] == '-'; i = 1; }
    if (i >= n) return false;
    for (; i < n; ++i) {
        if (s[i] < '0' || s[i] > '9') return false;
        v = v * 10 + (s[i] - '0');
    }
    *out = neg ? -v : v;
    return true;
}

// Persistent helper threads of one context: a call wakes them and works as worker 0 itself.  (Round 5 created and joined 255 std::threads
// per call.)  Helpers sleep on ONE futex word and are woken together -- a condition variable hands its mutex from thread to thread, and at
// 128 helpers that hand-over chain is as long as the 3 ms job of a 8192-record call (profiles/rd6_feat_sweep.txt).
class WorkerPool {
  public:
    ~WorkerPool() {
        stop_.store(true, std::memory_order_release);
        ctl_.fetch_add(1u << WANT_BITS, std::memory_order_release);
        futex(&ctl_, FUTEX_WAKE_PRIVATE, INT32_MAX);
        for (auto& t : threads_) t.join();
    }
    // fn(worker index 0 .. nt-1) on nt threads, the caller being worker 0; returns when all are done.  One job at a time per pool.
    template <class F> void run(int nt, F&& fn) {
        std::lock_guard<std::mutex> one(call_);
        if (nt > MAX_THREADS) nt = MAX_THREADS;
        std::function<void(int)> job = std::forward<F>(fn);
        if (nt > 1) {
            while ((int)threads_.size() < nt - 1) {
                const int id = (int)threads_.size() + 1;
                const uint32_t seen = ctl_.load(std::memory_order_relaxed);
                threads_.emplace_back([this, id, seen] { loop(id, seen); });
            }
            // job_ and remaining_ are published by the release store of ctl_ = (generation, thread count): a helper acts on them only after
            // it has read THAT value of ctl_, and only if the count in it includes the helper
            job_ = &job;
            remaining_.store((uint32_t)(nt - 1), std::memory_order_relaxed);
            const uint32_t gen = (ctl_.load(std::memory_order_relaxed) >> WANT_BITS) + 1;
            ctl_.store(gen << WANT_BITS | (uint32_t)nt, std::memory_order_release);
            futex(&ctl_, FUTEX_WAKE_PRIVATE, INT32_MAX);
        }
        job(0);
        if (nt > 1) {
            for (int spin = 0;; ++spin) {
                const uint32_t r = remaining_.load(std::memory_order_acquire);
                if (r == 0) break;
                if (spin < 4000) _mm_pause();
                else futex(&remaining_, FUTEX_WAIT_PRIVATE, r);
            }
        }
    }
    static constexpr int WANT_BITS = 12, MAX_THREADS = (1 << WANT_BITS) - 1;

  private:
    static long futex(std::atomic<uint32_t>* w, int op, uint32_t val) {
        return syscall(SYS_futex, reinterpret_cast<uint32_t*>(w), op, val, nullptr, nullptr, 0);
    }
    void loop(int id, uint32_t seen) {
        for (;;) {
            uint32_t c;
            while ((c = ctl_.load(std::memory_order_acquire)) == seen) futex(&ctl_, FUTEX_WAIT_PRIVATE, seen);
            seen = c;
            if (stop_.load(std::memory_order_acquire)) return;
            if (id >= (int)(c & (uint32_t)MAX_THREADS)) continue;
            (*job_)(id);
            if (remaining_.fetch_sub(1, std::memory_order_acq_rel) == 1) futex(&remaining_, FUTEX_WAKE_PRIVATE, 1);
        }
    }
    std::mutex call_;
    std::vector<std::thread> threads_;
    const std::function<void(int)>* job_ = nullptr;
    std::atomic<uint32_t> ctl_{0}, remaining_{0};
    std::atomic<bool> stop_{false};
};

}  // namespace

struct mmf_context {
    mutable WorkerPool pool;
    mutable std::mutex release_m;
    mutable std::vector<std::pair<void*, size_t>> release_q;      // mmf_release_later: ranges the next decode job unmaps beside its work
    std::unordered_map<std::string, int32_t> vocab;
    std::unordered_map<int64_t, Label> labels;
    int32_t max_chars = 200;
    bool never_split = false;
    int32_t unk = -1, cls = -1, sep = -1;

    // BERT Basic + WordPiece on pure-ASCII text (featurizer.WordPieceTokenizer restricted to ASCII).
    // Returns false when `text` has a non-ASCII byte.
    bool tokenize(const char* text, int64_t n, std::vector<int32_t>& ids) const {
        for (int64_t i = 0; i < n; ++i)
            if ((unsigned char)text[i] >= 128) return false;
        std::string word;
        auto flush_piece = [&](const std::string& piece) {   // greedy longest-match-first WordPiece
            if ((int32_t)piece.size() > max_chars) { ids.push_back(unk); return; }
            const size_t mark = ids.size();
            size_t start = 0;
            std::string sub;
            while (start < piece.size()) {
                size_t end = piece.size();
                int32_t found = -1;
                while (start < end) {
                    sub.assign(start > 0 ? "##" : "");
                    sub.append(piece, start, end - start);
                    auto it = vocab.find(sub);
                    if (it != vocab.end()) { found = it->second; break; }
                    --end;
                }
                if (found < 0) { ids.resize(mark); ids.push_back(unk); return; }
                ids.push_back(found);
                start = end;
            }
        };
        auto flush_word = [&]() {
            if (word.empty()) return;
            bool special = false;
            if (never_split)
                special = word == "[UNK]" || word == "[SEP]" || word == "[PAD]" || word == "[CLS]" || word == "[MASK]";
            if (special) {
                auto it = vocab.find(word);
                ids.push_back(it != vocab.end() ? it->second : unk);
            } else {
                std::string cur;
                for (char ch : word) {
                    unsigned char c = (unsigned char)ch;
                    if (c >= 'A' && c <= 'Z') c = (unsigned char)(c + 32);
                    if (is_ascii_punct(c)) {
                        if (!cur.empty()) { flush_piece(cur); cur.clear(); }
                        flush_piece(std::string(1, (char)c));
                    } else {
                        cur.push_back((char)c);
                    }
                }
                if (!cur.empty()) flush_piece(cur);
            }
            word.clear();
        };
        for (int64_t i = 0; i < n; ++i) {
            const unsigned char c = (unsigned char)text[i];
            if (c == 0 || is_ascii_ctrl(c)) continue;      // _clean_text drops NUL / control characters
            if (is_ascii_space(c)) flush_word();
            else word.push_back((char)c);
        }
        flush_word();
        return true;
    }
};

namespace {

struct Field { const char* p; int64_t n; };

// one record -> row `i` of the batch buffers.  Returns false (and sets `msg`) on a malformed record.
bool featurize_one(const mmf_context* c, const char* data, const char* line, int64_t len, int64_t i, int32_t text_len, int32_t box_dim, bool s2f,
                   const mmf_batch_out* o, std::string& msg) {
    // line.strip().split("\t")
    while (len > 0 && (is_ascii_space((unsigned char)line[0]))) { ++line; --len; }
    while (len > 0 && (is_ascii_space((unsigned char)line[len - 1]))) --len;
    Field f[9];
    int nf = 0;
    const char* start = line;
    for (int64_t k = 0; k <= len && nf < 9; ++k) {
        if (k == len || line[k] == '\t') {
            f[nf].p = start; f[nf].n = line + k - start; ++nf;
            start = line + k + 1;
        }
    }
    if (nf < 9) { msg = "record has fewer than 9 tab-separated fields"; return false; }
    // the 9th field runs to the end of the line in Python's split only if there are exactly 9 fields; extra tabs -> more
    // fields, of which the reference reads arr[8] only -- same here.
    int64_t pid, h, w, n, qid;
    if (!parse_i64(f[0].p, f[0].n, &pid) || !parse_i64(f[1].p, f[1].n, &h) || !parse_i64(f[2].p, f[2].n, &w) ||
        !parse_i64(f[3].p, f[3].n, &n) || !parse_i64(f[8].p, f[8].n, &qid)) { msg = "bad integer field"; return false; }
    if (n < 0 || h == 0 || w == 0) { msg = "bad num_boxes / image size"; return false; }
    if (b64_decoded_size(f[4].p, f[4].n) != n * 16 || b64_decoded_size(f[5].p, f[5].n) != n * FEAT_DIM * 4 ||
        b64_decoded_size(f[6].p, f[6].n) != n * 8) { msg = "base64 payload size does not match num_boxes"; return false; }
    o->product_id[i] = pid;
    o->query_id[i] = qid;
    o->num_boxes[i] = (int32_t)n;
    const int64_t nb = n < N_BOX ? n : N_BOX;

    float raw[N_BOX * 4];
    if (b64_decode(f[4].p, f[4].n, (unsigned char*)raw, nb * 16) != nb * 16) { msg = "bad base64 in boxes"; return false; }
    float* bx = o->boxes + i * N_BOX * box_dim;
    std::memset(bx, 0, sizeof(float) * N_BOX * box_dim);
    const double div[4] = {(double)h, (double)w, (double)h, (double)w};
    for (int64_t b = 0; b < nb; ++b) {
        for (int k = 0; k < 4; ++k) bx[b * box_dim + k] = (float)((double)raw[b * 4 + k] / div[k]);   // float64 division, stored fp32
        if (box_dim == 5) {
            const float d1 = raw[b * 4 + 2] - raw[b * 4 + 0], d2 = raw[b * 4 + 3] - raw[b * 4 + 1];
            const float prod = d1 * d2;
            bx[b * box_dim + 4] = prod / (float)(w * h);                                              // float32 throughout
        }
    }
    if (o->feats) {      // (NULL: the caller takes the features from another pass over the same records -- the fused three-model feed decodes them once, not three times)
    float* ft = o->feats + i * (int64_t)N_BOX * FEAT_DIM;
    // box rows [nb, 10) must read zero (seq_padding).  A caller that reuses its buffers tells us how many leading box rows of this batch
    // row may be non-zero (feat_rows_live, in / out): the rest is zero already and is not written again -- at 3.8 boxes per record the
    // padding is 60 % of the 80 KB row.
    const int64_t dirty = o->feat_rows_live ? (o->feat_rows_live[i] < 0 || o->feat_rows_live[i] > N_BOX ? N_BOX : o->feat_rows_live[i]) : N_BOX;
    if (o->feat_rows_live) o->feat_rows_live[i] = N_BOX;          // until the row is complete (an error return leaves it marked dirty)
    if (b64_decode(f[5].p, f[5].n, (unsigned char*)ft, nb * FEAT_DIM * 4) != nb * FEAT_DIM * 4) { msg = "bad base64 in features"; return false; }
    if (dirty > nb) std::memset(ft + nb * FEAT_DIM, 0, sizeof(float) * (dirty - nb) * FEAT_DIM);
    if (o->feat_rows_live) o->feat_rows_live[i] = (int32_t)nb;
    }

    int64_t cls_ids[N_BOX];
    if (b64_decode(f[6].p, f[6].n, (unsigned char*)cls_ids, nb * 8) != nb * 8) { msg = "bad base64 in class labels"; return false; }
    int32_t* lab = o->label_ids + i * N_BOX * LABEL_LEN;
    int32_t* lablen = o->label_len + i * N_BOX;
    std::memset(lab, 0, sizeof(int32_t) * N_BOX * LABEL_LEN);
    std::memset(lablen, 0, sizeof(int32_t) * N_BOX);
    for (int64_t b = 0; b < nb; ++b) {
        auto it = c->labels.find(cls_ids[b]);
        if (it == c->labels.end()) { msg = "class id " + std::to_string(cls_ids[b]) + " has no label text"; return false; }
        std::memcpy(lab + b * LABEL_LEN, it->second.ids, sizeof(int32_t) * LABEL_LEN);
        lablen[b] = it->second.len;
    }

    int32_t* q = o->query_ids + i * (int64_t)text_len;
    std::memset(q, 0, sizeof(int32_t) * text_len);
    o->query_span[2 * i] = f[7].p - data;
    o->query_span[2 * i + 1] = f[7].p - data + f[7].n;
    std::string query(f[7].p, (size_t)f[7].n);
    if (s2f) {                                                    // load_data_v4.py:153-154
        const std::string from = "sen department of", to = "forest style";
        size_t pos = 0;
        while ((pos = query.find(from, pos)) != std::string::npos) { query.replace(pos, from.size(), to); pos += to.size(); }
    }
    std::vector<int32_t> ids;
    ids.push_back(c->cls);
    if (!c->tokenize(query.data(), (int64_t)query.size(), ids)) {
        o->needs_host_tokenizer[i] = 1;
        o->query_len[i] = 0;
        return true;
    }
    ids.push_back(c->sep);
    o->needs_host_tokenizer[i] = 0;
    o->query_len[i] = (int32_t)ids.size();
    const int64_t k = (int64_t)ids.size() < text_len ? (int64_t)ids.size() : text_len;   // seq_padding truncation
    std::memcpy(q, ids.data(), sizeof(int32_t) * k);
    return true;
}

}  // namespace

extern "C" {

const char* mmf_last_error(void) { return g_err.c_str(); }

int mmf_b64_tier(int32_t set_to) {
    if (set_to >= 0 && !b64_set_tier(set_to)) { g_err = "mmf_b64_tier: this CPU has no such decoder"; return -1; }
    return g_b64_tier;
}

int mmf_create(const char* vocab_path, int32_t max_chars, int32_t never_split_specials, mmf_context** out) {
    if (!vocab_path || !out || max_chars <= 0) { g_err = "mmf_create: bad argument"; return -1; }
    std::ifstream f(vocab_path);
    if (!f) { g_err = std::string("cannot open vocab file ") + vocab_path; return -2; }
    mmf_context* c = new mmf_context();
    std::string line;
    int32_t id = 0;
    while (std::getline(f, line)) {
        size_t a = 0, b = line.size();
        while (a < b && is_ascii_space((unsigned char)line[a])) ++a;
        while (b > a && is_ascii_space((unsigned char)line[b - 1])) --b;
        c->vocab[line.substr(a, b - a)] = id++;          // later duplicates win, like the Python dict
    }
    auto get = [&](const char* t) { auto it = c->vocab.find(t); return it == c->vocab.end() ? -1 : it->second; };
    c->unk = get("[UNK]"); c->cls = get("[CLS]"); c->sep = get("[SEP]");
    if (c->unk < 0 || c->cls < 0 || c->sep < 0) { delete c; g_err = "vocab lacks [UNK] / [CLS] / [SEP]"; return -3; }
    c->max_chars = max_chars;
    c->never_split = never_split_specials != 0;
    *out = c;
    return 0;
}

void mmf_destroy(mmf_context* c) { delete c; }

int mmf_set_label(mmf_context* c, int64_t class_id, const int32_t* ids, int32_t len) {
    if (!c || len < 0 || (len > 0 && !ids)) { g_err = "mmf_set_label: bad argument"; return -1; }
    Label l{};
    l.len = len;
    for (int k = 0; k < LABEL_LEN && k < len; ++k) l.ids[k] = ids[k];
    c->labels[class_id] = l;
    return 0;
}

int mmf_tokenize_ascii(const mmf_context* c, const char* text, int64_t text_len, int32_t* ids, int32_t max_ids) {
    if (!c || !text || !ids || max_ids < 0) { g_err = "mmf_tokenize_ascii: bad argument"; return -1; }
    std::vector<int32_t> v;
    if (!c->tokenize(text, text_len, v)) return -2;
    const int n = (int)v.size() < max_ids ? (int)v.size() : max_ids;
    std::memcpy(ids, v.data(), sizeof(int32_t) * n);
    return (int)v.size() <= max_ids ? (int)v.size() : max_ids;
}

int64_t mmf_split_lines(const char* data, int64_t len, int64_t* starts, int64_t* ends, int64_t max_lines, int64_t* consumed) {
    if (!data || !starts || !ends || !consumed || len < 0 || max_lines < 0) { g_err = "mmf_split_lines: bad argument"; return -1; }
    static const char kHeader[] = "product_id";
    const size_t kh = sizeof(kHeader) - 1;
    int64_t pos = 0, n = 0;
    while (pos < len && n < max_lines) {
        // Fast path: a well-formed record's three base64 fields have lengths fixed by num_boxes (canonical padding), so they are
        // hopped over instead of scanned (49 KB of the ~50 KB line); base64 text can contain neither '\n' nor "product_id".
        int64_t scan_from = pos, head_end = -1;
        {
            int64_t t[4], nt = 0;
            for (int64_t k = pos; k < len && k < pos + 96 && nt < 4 && data[k] != '\n'; ++k)
                if (data[k] == '\t') t[nt++] = k;
            int64_t nb;
            if (nt == 4 && parse_i64(data + t[2] + 1, t[3] - t[2] - 1, &nb) && nb >= 0 && nb < (1 << 20)) {
                const int64_t l4 = (nb * 16 + 2) / 3 * 4, l5 = (nb * FEAT_DIM * 4 + 2) / 3 * 4, l6 = (nb * 8 + 2) / 3 * 4;
                const int64_t p4 = t[3] + 1, p5 = p4 + l4 + 1, p6 = p5 + l5 + 1, p7 = p6 + l6 + 1;
                if (p7 <= len && data[p5 - 1] == '\t' && data[p6 - 1] == '\t' && data[p7 - 1] == '\t') { head_end = t[3]; scan_from = p7; }
            }
        }
        const char* nl = (const char*)std::memchr(data + scan_from, '\n', (size_t)(len - scan_from));
        const int64_t end = nl ? nl - data : len;
        bool keep;
        if (head_end >= 0) {
            keep = !memmem(data + pos, (size_t)(head_end - pos), kHeader, kh) && !memmem(data + scan_from, (size_t)(end - scan_from), kHeader, kh);
        } else {
            bool blank = true;
            for (int64_t k = pos; k < end && blank; ++k) blank = is_ascii_space((unsigned char)data[k]) || data[k] == '\v' || data[k] == '\f';
            keep = !blank && !memmem(data + pos, (size_t)(end - pos), kHeader, kh);
        }
        if (keep) { starts[n] = pos; ends[n] = end; ++n; }
        pos = end + 1;
    }
    *consumed = pos < len ? pos : len;
    return n;
}

int mmf_query_ids(const char* data, const int64_t* starts, const int64_t* ends, int64_t n, int64_t* out) {
    if (!data || !starts || !ends || !out || n < 0) { g_err = "mmf_query_ids: bad argument"; return -1; }
    for (int64_t i = 0; i < n; ++i) {
        const char* line = data + starts[i];
        int64_t len = ends[i] - starts[i];
        while (len > 0 && is_ascii_space((unsigned char)line[len - 1])) --len;
        int64_t k = len;
        while (k > 0 && line[k - 1] != '\t') --k;              // the last field: query_id of the 9-field record (a few bytes from the end; the base64 fields are never scanned)
        if (k == 0 || !parse_i64(line + k, len - k, &out[i])) { g_err = "record " + std::to_string(i) + ": no integer query id in the last field"; return (int)(-1000 - (i < 2000000000 ? i : 2000000000)); }
    }
    return 0;
}

int mmf_prefault(const mmf_context* c, const void* addr, int64_t len, int32_t threads) {
    if (!c || !addr || len < 0) { g_err = "mmf_prefault: bad argument"; return -1; }
    if (len == 0) return 0;
#ifndef MADV_POPULATE_READ
#define MADV_POPULATE_READ 22            /* Linux 5.14 */
#endif
    const int64_t page = sysconf(_SC_PAGESIZE), piece = 2 << 20;
    const uintptr_t lo = (uintptr_t)addr & ~(uintptr_t)(page - 1), hi = (uintptr_t)addr + (uintptr_t)len;
    const int64_t pieces = (int64_t)((hi - lo + piece - 1) / piece);
    int nt = threads > 0 ? threads : default_threads();
    if (nt > 16) nt = 16;                      // 170 pieces per 8192-record batch: a wide wake-up costs more than it maps
    if ((int64_t)nt > pieces) nt = (int)pieces;
    std::atomic<int64_t> next(0);
    c->pool.run(nt, [&](int) {
        for (;;) {
            const int64_t k = next.fetch_add(1);
            if (k >= pieces) break;
            const uintptr_t a = lo + (uintptr_t)(k * piece), b = a + piece < hi ? a + piece : hi;
            if (madvise((void*)a, (size_t)(b - a), MADV_POPULATE_READ) != 0) {
                volatile unsigned char sink = 0;                     // older kernel: touch a byte per page
                for (uintptr_t q = a; q < b; q += (uintptr_t)page) sink = sink + *(const volatile unsigned char*)q;
            }
        }
    });
    return 0;
}

int mmf_release_later(const mmf_context* c, const void* addr, int64_t len) {
    if (c && !addr) {      // forget every pending range: the mapping they point into is about to go away
        std::lock_guard<std::mutex> g(c->release_m);
        const int dropped = (int)c->release_q.size();
        c->release_q.clear();
        return dropped;          // (how many ranges were pending)
    }
    if (!c || len < 0) { g_err = "mmf_release_later: bad argument"; return -1; }
    const uintptr_t page = (uintptr_t)sysconf(_SC_PAGESIZE);
    const uintptr_t lo = ((uintptr_t)addr + page - 1) & ~(page - 1), hi = ((uintptr_t)addr + (uintptr_t)len) & ~(page - 1);   // whole pages inside
    if (hi > lo) {
        std::lock_guard<std::mutex> g(c->release_m);
        c->release_q.emplace_back((void*)lo, (size_t)(hi - lo));
    }
    return 0;
}

int mmf_featurize(const mmf_context* c, const char* data, const int64_t* offsets, int64_t n, int32_t text_len, int32_t box_dim,
                  int32_t sen2forest, int32_t threads, const mmf_batch_out* out) {
    if (!offsets) { g_err = "mmf_featurize: bad argument"; return -1; }
    return mmf_featurize_spans(c, data, offsets, offsets + 1, n, text_len, box_dim, sen2forest, threads, out);
}

int mmf_featurize_spans(const mmf_context* c, const char* data, const int64_t* starts, const int64_t* ends, int64_t n, int32_t text_len,
                        int32_t box_dim, int32_t sen2forest, int32_t threads, const mmf_batch_out* out) {
    if (!c || !data || !starts || !ends || !out || n < 0 || text_len <= 0 || (box_dim != 4 && box_dim != 5)) { g_err = "mmf_featurize: bad argument"; return -1; }
    if (!out->product_id || !out->query_id || !out->num_boxes || !out->boxes || !out->label_ids || !out->label_len ||
        !out->query_ids || !out->query_len || !out->needs_host_tokenizer || !out->query_span) { g_err = "mmf_featurize: null output buffer"; return -1; }
    if (n == 0) return 0;
    int nt = threads > 0 ? threads : default_threads();
    if (nt < 1) nt = 1;
    if ((int64_t)nt > n) nt = (int)n;
    std::atomic<int64_t> next(0), first_bad(INT64_MAX);
    std::vector<std::string> msgs(nt);
    std::vector<std::pair<void*, size_t>> drop;
    { std::lock_guard<std::mutex> g(c->release_m); drop.swap(c->release_q); }
    auto worker = [&](int t) {
        if (t == nt - 1)                                      // one thread drops the consumed input pages while the others decode
            for (auto& r : drop) (void)madvise(r.first, r.second, MADV_DONTNEED);
        for (;;) {
            const int64_t i0 = next.fetch_add(16);            // small work items: records differ a lot in size
            if (i0 >= n) break;
            const int64_t i1 = i0 + 16 < n ? i0 + 16 : n;
            for (int64_t i = i0; i < i1; ++i) {
                std::string m;
                if (!featurize_one(c, data, data + starts[i], ends[i] - starts[i], i, text_len, box_dim, sen2forest != 0, out, m)) {
                    int64_t cur = first_bad.load();
                    while (i < cur && !first_bad.compare_exchange_weak(cur, i)) {}
                    if (first_bad.load() == i) msgs[t] = "record " + std::to_string(i) + ": " + m;
                }
            }
        }
    };
    c->pool.run(nt, worker);
    const int64_t bad = first_bad.load();
    if (bad != INT64_MAX) {
        const std::string prefix = "record " + std::to_string(bad) + ":";
        for (auto& m : msgs) if (m.compare(0, prefix.size(), prefix) == 0) g_err = m;
        return bad < 2000000000 ? (int)(-1000 - bad) : -2000001000;
    }
    return 0;
}

}  // extern "C"
