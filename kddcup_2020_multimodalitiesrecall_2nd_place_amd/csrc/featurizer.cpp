// libmmfeat: native multi-threaded TSV record featurizer (see include/mmfeat.h for the contract and the reference
// lines each step restates).  Host-only C++17, no HIP: built with g++ into csrc/libmmfeat.so.
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
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
uint32_t B64T[4][256];
constexpr uint32_t B64_BAD = 0x01000000u;
struct B64Init {
    B64Init() {
        for (int i = 0; i < 256; ++i) B64[i] = -1;
        const char* a = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
        for (int i = 0; i < 64; ++i) B64[(unsigned char)a[i]] = (int8_t)i;
        for (int i = 0; i < 256; ++i)
            for (int k = 0; k < 4; ++k) B64T[k][i] = B64[i] < 0 ? B64_BAD : (uint32_t)B64[i] << (18 - 6 * k);
    }
} b64_init;

int64_t b64_decode(const char* s, int64_t n, unsigned char* out, int64_t cap) {
    const unsigned char* u = (const unsigned char*)s;
    int64_t o = 0, i = 0;
    for (; i + 4 <= n && o + 3 <= cap; i += 4, o += 3) {                 // whole groups
        const uint32_t x = B64T[0][u[i]] | B64T[1][u[i + 1]] | B64T[2][u[i + 2]] | B64T[3][u[i + 3]];
        if (x & B64_BAD) break;                                          // '=' padding or a bad character: finish below
        out[o] = (unsigned char)(x >> 16); out[o + 1] = (unsigned char)(x >> 8); out[o + 2] = (unsigned char)x;
    }
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

}  // namespace

struct mmf_context {
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
    float* ft = o->feats + i * (int64_t)N_BOX * FEAT_DIM;
    if (b64_decode(f[5].p, f[5].n, (unsigned char*)ft, nb * FEAT_DIM * 4) != nb * FEAT_DIM * 4) { msg = "bad base64 in features"; return false; }
    std::memset(ft + nb * FEAT_DIM, 0, sizeof(float) * (N_BOX - nb) * FEAT_DIM);

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

int mmf_featurize(const mmf_context* c, const char* data, const int64_t* offsets, int64_t n, int32_t text_len, int32_t box_dim,
                  int32_t sen2forest, int32_t threads, const mmf_batch_out* out) {
    if (!offsets) { g_err = "mmf_featurize: bad argument"; return -1; }
    return mmf_featurize_spans(c, data, offsets, offsets + 1, n, text_len, box_dim, sen2forest, threads, out);
}

int mmf_featurize_spans(const mmf_context* c, const char* data, const int64_t* starts, const int64_t* ends, int64_t n, int32_t text_len,
                        int32_t box_dim, int32_t sen2forest, int32_t threads, const mmf_batch_out* out) {
    if (!c || !data || !starts || !ends || !out || n < 0 || text_len <= 0 || (box_dim != 4 && box_dim != 5)) { g_err = "mmf_featurize: bad argument"; return -1; }
    if (!out->product_id || !out->query_id || !out->num_boxes || !out->boxes || !out->feats || !out->label_ids || !out->label_len ||
        !out->query_ids || !out->query_len || !out->needs_host_tokenizer || !out->query_span) { g_err = "mmf_featurize: null output buffer"; return -1; }
    if (n == 0) return 0;
    int nt = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
    if (nt < 1) nt = 1;
    if ((int64_t)nt > n) nt = (int)n;
    std::atomic<int64_t> next(0), first_bad(INT64_MAX);
    std::vector<std::string> msgs(nt);
    auto worker = [&](int t) {
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
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(worker, t);
    worker(0);
    for (auto& th : pool) th.join();
    const int64_t bad = first_bad.load();
    if (bad != INT64_MAX) {
        const std::string prefix = "record " + std::to_string(bad) + ":";
        for (auto& m : msgs) if (m.compare(0, prefix.size(), prefix) == 0) g_err = m;
        return bad < 2000000000 ? (int)(-1000 - bad) : -2000001000;
    }
    return 0;
}

}  // extern "C"
