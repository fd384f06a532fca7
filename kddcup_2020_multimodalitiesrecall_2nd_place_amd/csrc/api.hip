// libmmscore C ABI: handle, weight container, workspace and the launch plans of the three forwards.
// See include/mmscore.h for the contract and the reference call sites each entry point replaces.
#include <cassert>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/mmscore.h"
#include "kernels.h"

namespace {

constexpr int H = MMS_HIDDEN;
thread_local std::string g_err;

struct HostTensor {
    std::vector<float> data;
    std::vector<int64_t> shape;
    int64_t numel() const { int64_t n = 1; for (auto s : shape) n *= s; return n; }
};

struct Planes {
    bf16* hi = nullptr;
    bf16* lo = nullptr;
    unsigned char* f8 = nullptr;   // precision mode 4: the same activation as e4m3 bytes (the fp8 GEMMs' A operand)
    // hl32 layout (common.h): hi and lo are ONE buffer of alternating 32-element blocks (lo == hi + 32); a LOGICAL element offset that is
    // a multiple of 32 (row offsets: every leading dimension is) is twice as far in the buffer
    Planes at(long long elem_off) const {
        assert((elem_off & 31) == 0 && "hl32 planes: a view must start on a 32-element block");     // every caller passes row * 768
        Planes p; p.hi = hi + 2 * elem_off; p.lo = lo + 2 * elem_off; p.f8 = f8 ? f8 + elem_off : nullptr; return p;
    }
};

// *8 / *s: e4m3 copy of the matrix (8 x 128-byte tiles) and its per-output-channel scales as packed e8m0 bytes (precision mode 4 only)
struct AttW { bf16* wqkv; float* bqkv; bf16* wo; float* bo; float* g; float* b; unsigned char* wqkv8 = nullptr; unsigned* wqkvs = nullptr;
              unsigned char* wo8 = nullptr; unsigned* wos = nullptr;
              bf16* wqkv_hm = nullptr; float* bqkv_hm = nullptr; };   // head-major [12][Q 64 | K 64 | V 64] rows: the fused QKV + attention kernel (qkv_attn.hip)
struct FfnW { bf16* wi; float* bi; bf16* wd; float* bd; float* g; float* b; unsigned char* wi8 = nullptr; unsigned* wis = nullptr;
              unsigned char* wd8 = nullptr; unsigned* wds = nullptr; };
struct LayerW { AttW att; FfnW ffn; };
struct XLayerW { AttW cross, lang_self, visn_self; FfnW lang_ffn, visn_ffn; };

inline uint16_t f2bf(float f) {  // round-to-nearest-even, same as v_cvt_pk_bf16_f32
    uint32_t u; std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

}  // namespace

struct mms_handle {
    mms_config cfg{};
    std::string err;
    std::unordered_map<std::string, HostTensor> host;
    std::vector<void*> w_allocs, ws_allocs, lab_allocs;
    bool finalized = false;
    int nsplit = 2;
    bool f8 = false;       // precision mode 4: the big encoder GEMMs run on e4m3 operands; everything else as mode 2
    int alternate = 1;     // successive big kernels walk their rows in opposite directions, so each starts on the rows its producer wrote last
                           // (still in the 256 MB Infinity Cache / L2); env MMS_ALTERNATE=0 turns it off (A/B)
    int flip = 0;
    int resid_in_ln = 1;   // residual add in the LayerNorm that follows (1, default) or in the GEMM epilogue (0; env MMS_RESID_IN_LN, A/B only)
    int x1_mask = 0;   // lab build only (env MMS_X1_MASK): GEMM classes forced to one pass: 1 qkv, 2 att-out, 4 ffn-up, 8 ffn-down
    struct WPlane { const bf16* base; long long elems; };
    std::vector<WPlane> w_planes;   // precision 3: hi plane [base, base+elems), lo plane right behind it

    // ---- weights (device) ----
    float *E = nullptr, *type_tab = nullptr, *pos_tab = nullptr, *emb_g = nullptr, *emb_b = nullptr;
    std::vector<LayerW> layers, r_layers;
    std::vector<XLayerW> x_layers;
    bf16* w_pool = nullptr; float* b_pool = nullptr;
    // zk
    bf16 *w_conv1 = nullptr, *w_conv2 = nullptr, *w_femb = nullptr;
    float *b_conv1 = nullptr, *b_conv2 = nullptr, *b_femb = nullptr, *w_dense1 = nullptr, *b_dense1 = nullptr, *am_kernel = nullptr;
    // lds
    bf16* w_feat = nullptr; float *b_feat = nullptr, *w_lab8 = nullptr, *w_cls = nullptr, *b_cls = nullptr;
    // lxmert
    bf16 *w_visn = nullptr, *w_labfc = nullptr, *w_fc0 = nullptr;
    float *b_visn = nullptr, *g_visn = nullptr, *be_visn = nullptr, *w_box = nullptr, *b_box = nullptr, *g_box = nullptr,
          *be_box = nullptr, *w_lconv = nullptr, *b_lconv = nullptr, *b_labfc = nullptr, *g_lab = nullptr, *be_lab = nullptr,
          *b_fc0 = nullptr, *g_fc2 = nullptr, *be_fc2 = nullptr, *w_fc3 = nullptr, *b_fc3 = nullptr;

    // ---- workspace (device), sized for ws_pairs ----
    int64_t ws_pairs = 0;
    Planes x, ctx, y, mid;
    float *qkv = nullptr, *t = nullptr, *key_add = nullptr, *key_add2 = nullptr, *pooled = nullptr, *hbuf = nullptr;
    int64_t x_rows = 0;  // rows of the hidden state per pair-chunk (for mms_debug_read_x)
    // packed (ragged) execution plan of the current chunk: per-pair first row / live count, per-row source
    // token, device-side live row totals (stream 0: zk tokens or lxmert language, stream 1: lxmert vision)
    int *pk_off[2] = {nullptr, nullptr}, *pk_cnt[2] = {nullptr, nullptr}, *pk_src[2] = {nullptr, nullptr}, *pk_rows = nullptr;
    unsigned long long* flop_counter = nullptr;
    // fused residual + LayerNorm epilogue (gemm_pp_ln.h): per-row statistics granules, per-launch control words (check-in count,
    // fused / plain decision), launch tag.  ln_slot counts the fused launches of the current call.
    float* ln_stats = nullptr; int* ln_ctl = nullptr; int ln_slot = 0; unsigned ln_tag = 0;

    static constexpr int LN_SLOTS = 2048;
    float* kparts = nullptr; int64_t kparts_floats = 0;    // fp32 partials of the split-K launches (small calls): slices x launch rows x N, sized with the workspace (ensure_kparts)
    // Second launch lane (lxmert calls below LANE_ROWS token rows): the language and the vision stream's sub-layers between two cross attentions -- and the
    // distinct-query stage beside the box stream's layers -- are independent chains of under-filled launches; the vision chain / the query stage is enqueued on `side`
    // between a fork and a join event.  Same kernels on the same operands: results do not change.  lane == 1 while the side chain is being enqueued: its split-K partials
    // and its FFN intermediate live in kparts_side / behind mid_side_off elements of `mid` (everything else a chain touches is addressed by its stream's row range).
    hipStream_t side = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool no_fused_ln = false;      // this call takes the two-kernel LayerNorm route throughout: it runs on lanes -- or WOULD without per-launch timing (the route must not depend on instrumentation)
    int lane = 0; bool lanes_on = false, lanes_q = false;      // lanes_on: the whole call (stream chains of the X layers, language beside box layers); lanes_q: the distinct-query stage only
    float* kparts_side = nullptr; int64_t kparts_side_floats = 0; int64_t mid_side_off = 0, mid_elems = 0;
    int fuse_ln = 0;       // mms_config.fuse_layernorm (lab build: env MMS_FUSE_LN overrides)
    int fuse_attn = 0;     // mms_config.fuse_attention: QKV projection + self-attention in one kernel (qkv_attn.hip; precision mode 2)
    int* qa_rec[3] = {nullptr, nullptr, nullptr};      // per-pair records of the tables (split-bf16 attention route)
    int* qa_jump = nullptr;                            // scratch of the plan kernel (jump tables)
    int4* qa_sub[4] = {nullptr, nullptr, nullptr, nullptr}; int* qa_nsub = nullptr;     // sub-tile tables of the (up to two) token streams of a launch wave; [2], [3]: the CROSS table of the stream pair (lxmert X layers)
    // label-text workspace, sized for lab_cap unique labels
    int64_t lab_cap = 0;
    Planes lab_planes; float *lab_f32 = nullptr, *lab_feat = nullptr; int64_t lab_feat_cap = 0;
    int64_t n_labels = 0;  // rows of lab_feat that are valid for the current call
    // label-tuple de-duplication workspace (dense label ids), sized for dd_rows = B*10 tuples
    std::vector<void*> dd_allocs;
    int64_t dd_rows = 0; int dd_cap = 0;
    int *dd_slots = nullptr, *dd_rep = nullptr, *dd_uid = nullptr, *dd_counter = nullptr, *dd_index = nullptr;
    int32_t* dd_uniq32 = nullptr; int64_t* dd_uniq64 = nullptr;
    // lxmert distinct-query stage: the language stream of the first l_layers runs once per distinct (input_ids, input_mask)
    // row; lq_store holds its output rows, dense by (query, token), for the pair chunks to copy from
    std::vector<void*> lq_allocs, lq_sub_allocs;      // lq_sub_allocs: the two sub-batch gather buffers (replaced, not piled up, when they grow)
    int64_t lq_pairs = 0, lq_store_q = 0, lq_sub = 0;
    int *lq_slots = nullptr, *lq_rep = nullptr, *lq_uid = nullptr, *lq_counter = nullptr, *lq_rows_of = nullptr, *lq_index = nullptr;
    int lq_cap = 0;
    int64_t *lq_ids = nullptr, *lq_mask = nullptr;
    int *lq_pk_off = nullptr, *lq_pk_cnt = nullptr, *lq_pk_src = nullptr; float* lq_key_add = nullptr;      // the stage's own packed plan (it may run beside a chunk's prologue: two lanes)
    int4* lq_qa_sub = nullptr; int *lq_qa_rec = nullptr, *lq_qa_jump = nullptr;      // ... and its own sub-tile table (plan_tiles)
    bool lq_join_pending = false;                      // the stage was enqueued on the side lane: lx_chunk joins before it reads lq_store
    Planes lq_store;
    bool lq_active = false;
    // fused three-model entry point: feed conversions and member outputs, sized for ens_pairs
    std::vector<void*> ens_allocs, ens_c_allocs;      // ens_c_allocs: the changed-query buffers of the second zk member (replaced when they grow)
    int64_t ens_pairs = 0;
    int32_t* ens_seg = nullptr; int64_t *ens_ids64 = nullptr, *ens_seg64 = nullptr, *ens_lab64 = nullptr, *ens_mask64 = nullptr;
    float *ens_vmask = nullptr, *ens_boxes4 = nullptr, *ens_logits = nullptr, *ens_probs = nullptr, *ens_tok = nullptr;
    // second zk member: pairs whose query the rewrite changed (flags, per-wave offsets / lists / counts, compacted feeds and outputs)
    int *ens_diff = nullptr, *ens_doff = nullptr, *ens_dlist = nullptr, *ens_dcount = nullptr;
    int32_t *ens_cq = nullptr, *ens_clen = nullptr, *ens_cnb = nullptr; int64_t* ens_clab = nullptr;
    float *ens_ctok = nullptr, *ens_clog = nullptr, *ens_cprob = nullptr; int64_t ens_cpairs = 0;
    Planes ens_featp; int64_t ens_featp_pairs = 0;      // the wave's split box features when the three members run side by side (they may not live in zk's FFN buffer then)

    // ---- gemm timing ----
    bool timing = false;
    std::vector<hipEvent_t> ev;
    std::vector<unsigned char> ev_cls;   // per timed launch (event pair): 0 plain GEMM epilogue, 1 fused bias + residual + LayerNorm epilogue
    size_t ev_used = 0;
    int64_t gemm_launches = 0;
    int64_t fused_attn_launches = 0;     // qkv_attn.hip launches since mms_create (mms_dbg_counter)
    bool zk_plans_merged = false;        // zk_image_tokens() already built this launch wave's token plan (small waves: launch_zk_plans_small)
    int64_t skinny_launches = 0;         // gemm_skinny.hip launches (launches of <= 128 padded rows) since mms_create
    int64_t lane_forks = 0;              // fork / join pairs of the second launch lane since mms_create (mms_dbg_counter 4)
    int64_t ln_fused_launches = 0, splitk_launches = 0;      // LayerNorm-fused GEMM launches / split-K launches (small-call routes) since mms_create
    std::vector<hipEvent_t> ev_fused; size_t ev_fused_used = 0; int64_t fused_timed = 0;     // timing of the fused launches, apart from the GEMMs' (mms_fused_timing)

    int fail(int code, const std::string& m) { err = m; return code; }
};

namespace {

#define HIP_TRY(h, expr)                                                                                  \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess)                                                                             \
            return (h)->fail(MMS_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));             \
    } while (0)


// two more events at the end of `v` (nothing is appended unless both were created: mms_destroy destroys every element)
int grow_event_pair(mms_handle* h, std::vector<hipEvent_t>& v) {
    hipEvent_t a, b;
    HIP_TRY(h, hipEventCreate(&a));
    hipError_t e = hipEventCreate(&b);
    if (e != hipSuccess) { (void)hipEventDestroy(a); return h->fail(MMS_ERR_HIP, std::string("hipEventCreate: ") + hipGetErrorString(e)); }
    v.push_back(a); v.push_back(b);
    return MMS_OK;
}

int dev_alloc(mms_handle* h, std::vector<void*>& pool, void** out, size_t bytes) {
    void* p = nullptr;
    HIP_TRY(h, hipMalloc(&p, bytes ? bytes : 16));
    pool.push_back(p);
    *out = p;
    return MMS_OK;
}
void free_pool(std::vector<void*>& pool) {
    for (void* p : pool) (void)hipFree(p);
    pool.clear();
}

const HostTensor* find(mms_handle* h, const std::string& name) {
    auto it = h->host.find(name);
    return it == h->host.end() ? nullptr : &it->second;
}

int need(mms_handle* h, const std::string& name, std::vector<int64_t> shape, const HostTensor** out) {
    const HostTensor* t = find(h, name);
    if (!t) return h->fail(MMS_ERR_WEIGHT, "missing weight: " + name);
    if (t->shape != shape) {
        std::string s = "weight " + name + " has shape [";
        for (auto d : t->shape) s += std::to_string(d) + ",";
        s += "] expected [";
        for (auto d : shape) s += std::to_string(d) + ",";
        return h->fail(MMS_ERR_WEIGHT, s + "]");
    }
    *out = t;
    return MMS_OK;
}

int upload_f32(mms_handle* h, const float* src, size_t n, float** out) {
    void* p;
    if (int rc = dev_alloc(h, h->w_allocs, &p, n * 4)) return rc;
    HIP_TRY(h, hipMemcpy(p, src, n * 4, hipMemcpyHostToDevice));
    *out = (float*)p;
    return MMS_OK;
}
int vec(mms_handle* h, const std::string& name, int64_t n, float** out) {
    const HostTensor* t;
    if (int rc = need(h, name, {n}, &t)) return rc;
    return upload_f32(h, t->data.data(), (size_t)n, out);
}
int tab(mms_handle* h, const std::string& name, std::vector<int64_t> shape, float** out) {
    const HostTensor* t;
    if (int rc = need(h, name, shape, &t)) return rc;
    return upload_f32(h, t->data.data(), (size_t)t->numel(), out);
}

// Build a bf16 [N][K] device matrix from a list of host sources.  Each source contributes n_i output
// rows; `in_out` sources are [K, n_i] (TF dense kernel), otherwise [n_i, K] (torch Linear weight).
struct MatSrc { const float* p; int64_t n; bool in_out; int64_t col0 = 0, ld = 0; };   // ld != 0: rows col0 .. col0 + n of a source with ld output rows
int upload_mat(mms_handle* h, const std::vector<MatSrc>& srcs, int64_t K, bf16** out) {
    int64_t N = 0;
    for (auto& s : srcs) N += s.n;
    // Stored as 16-row x 32-column tiles of 1 KiB (wtile_off, common.h): what one LDS-DMA piece of the GEMM engines fetches.  A row offset
    // that is a multiple of 16 is n * K elements into the buffer, as in a row-major matrix, so sub-matrix pointers (the K | V rows of a
    // fused [Q | K | V] matrix) are formed as before.
    // precision mode 3 keeps a second (tiled) plane lo = bf16(w - bf16(w)) right behind the hi plane
    if (N % 16 || K % 32) return h->fail(MMS_ERR_WEIGHT, "GEMM weight matrix must have N % 16 == 0 and K % 32 == 0");
    const bool with_lo = h->nsplit == 3;
    std::vector<uint16_t> buf((size_t)(N * K) * (with_lo ? 2 : 1));
    int64_t r0 = 0;
    for (auto& s : srcs) {
        for (int64_t n = 0; n < s.n; ++n)
            for (int64_t k = 0; k < K; ++k) {
                const float w = s.in_out ? s.p[k * (s.ld ? s.ld : s.n) + s.col0 + n] : s.p[(s.col0 + n) * K + k];
                const uint16_t hi = f2bf(w);
                const size_t at = (size_t)wtile_off(r0 + n, k, K);
                buf[at] = hi;
                if (with_lo) {
                    uint32_t u = (uint32_t)hi << 16; float hf; std::memcpy(&hf, &u, 4);
                    buf[(size_t)(N * K) + at] = f2bf(w - hf);
                }
            }
        r0 += s.n;
    }
    void* p;
    if (int rc = dev_alloc(h, h->w_allocs, &p, buf.size() * 2)) return rc;
    HIP_TRY(h, hipMemcpy(p, buf.data(), buf.size() * 2, hipMemcpyHostToDevice));
    *out = (bf16*)p;
    if (with_lo) h->w_planes.push_back({(const bf16*)p, N * K});
    return MMS_OK;
}
// precision mode 4: e4m3 copy [N][K] of the same sources + one power-of-two scale per output channel (quantised on the device:
// launch_quant_rows_f8, the routine mms_dbg_gemm_f8 exposes to the kernel tests)
int upload_mat_f8(mms_handle* h, const std::vector<MatSrc>& srcs, int64_t K, unsigned char** out, unsigned** scale) {
    int64_t N = 0;
    for (auto& s : srcs) N += s.n;
    std::vector<float> buf((size_t)(N * K));
    int64_t r0 = 0;
    for (auto& s : srcs) {
        for (int64_t n = 0; n < s.n; ++n)
            for (int64_t k = 0; k < K; ++k) buf[(size_t)((r0 + n) * K + k)] = s.in_out ? s.p[k * s.n + n] : s.p[n * K + k];
        r0 += s.n;
    }
    float* tmp = nullptr;
    HIP_TRY(h, hipMalloc((void**)&tmp, buf.size() * 4));
    hipError_t e = hipMemcpy(tmp, buf.data(), buf.size() * 4, hipMemcpyHostToDevice);
    void *q = nullptr, *sc = nullptr;
    int rc = e == hipSuccess ? MMS_OK : h->fail(MMS_ERR_HIP, std::string("hipMemcpy: ") + hipGetErrorString(e));
    if (!rc) rc = dev_alloc(h, h->w_allocs, &q, (size_t)(N * K));
    if (!rc && (N % 64 || K % 128)) rc = h->fail(MMS_ERR_WEIGHT, "fp8 GEMM weight matrix must have N % 64 == 0 and K % 128 == 0");
    if (!rc) rc = dev_alloc(h, h->w_allocs, &sc, (size_t)N);
    if (!rc) {
        launch_quant_rows_f8(tmp, (unsigned char*)q, nullptr, (unsigned*)sc, (int)N, (int)K, 0);
        e = hipDeviceSynchronize();
        if (e != hipSuccess) rc = h->fail(MMS_ERR_HIP, std::string("quantise weights: ") + hipGetErrorString(e));
    }
    (void)hipFree(tmp);
    *out = (unsigned char*)q; *scale = (unsigned*)sc;
    return rc;
}
int mat_f8(mms_handle* h, const std::string& name, int64_t N, int64_t K, bool in_out, unsigned char** out, unsigned** scale) {
    const HostTensor* t = find(h, name);
    if (!t) return h->fail(MMS_ERR_WEIGHT, "missing weight: " + name);
    return upload_mat_f8(h, {{t->data.data(), N, in_out}}, K, out, scale);
}
int mat(mms_handle* h, const std::string& name, int64_t N, int64_t K, bool in_out, bf16** out,
        std::vector<int64_t> shape_override = {}) {
    const HostTensor* t;
    std::vector<int64_t> shape = shape_override.empty() ? (in_out ? std::vector<int64_t>{K, N} : std::vector<int64_t>{N, K})
                                                        : shape_override;
    if (int rc = need(h, name, shape, &t)) return rc;
    return upload_mat(h, {{t->data.data(), N, in_out}}, K, out);
}
int cat_vec(mms_handle* h, const std::vector<std::string>& names, int64_t n_each, float** out) {
    std::vector<float> buf;
    for (auto& nm : names) {
        const HostTensor* t;
        if (int rc = need(h, nm, {n_each}, &t)) return rc;
        buf.insert(buf.end(), t->data.begin(), t->data.end());
    }
    return upload_f32(h, buf.data(), buf.size(), out);
}

// attention block weights; tf: scope names with /kernel [in,out]; pt: module names with .weight [out,in]
int load_att(mms_handle* h, bool tf, const std::string& self_scope, const std::string& out_scope, AttW* w) {
    const char* qkv[3] = {"query", "key", "value"};
    std::vector<MatSrc> srcs;
    std::vector<std::string> bnames;
    for (int i = 0; i < 3; ++i) {
        const HostTensor* t;
        const std::string nm = self_scope + (tf ? "/" : ".") + qkv[i] + (tf ? "/kernel" : ".weight");
        if (int rc = need(h, nm, {H, H}, &t)) return rc;
        srcs.push_back({t->data.data(), H, tf});
        bnames.push_back(self_scope + (tf ? "/" : ".") + qkv[i] + (tf ? "/bias" : ".bias"));
    }
    if (int rc = upload_mat(h, srcs, H, &w->wqkv)) return rc;
    if (h->f8) {
        if (int rc = upload_mat_f8(h, srcs, H, &w->wqkv8, &w->wqkvs)) return rc;
        if (int rc = mat_f8(h, out_scope + (tf ? "/dense/kernel" : ".dense.weight"), H, H, tf, &w->wo8, &w->wos)) return rc;
    }
    if (int rc = cat_vec(h, bnames, H, &w->bqkv)) return rc;
    if (h->fuse_attn && (h->nsplit == 2 || h->nsplit == 3) && !h->f8) {
        std::vector<MatSrc> hm;
        std::vector<float> bhm;
        for (int hd = 0; hd < MMS_HEADS; ++hd)
            for (int i = 0; i < 3; ++i) {
                hm.push_back({srcs[i].p, MMS_HEAD_DIM, tf, (int64_t)hd * MMS_HEAD_DIM, H});
                const HostTensor* bt;
                if (int rc = need(h, bnames[i], {H}, &bt)) return rc;
                bhm.insert(bhm.end(), bt->data.begin() + hd * MMS_HEAD_DIM, bt->data.begin() + (hd + 1) * MMS_HEAD_DIM);
            }
        if (int rc = upload_mat(h, hm, H, &w->wqkv_hm)) return rc;
        if (int rc = upload_f32(h, bhm.data(), bhm.size(), &w->bqkv_hm)) return rc;
    }
    if (int rc = mat(h, out_scope + (tf ? "/dense/kernel" : ".dense.weight"), H, H, tf, &w->wo)) return rc;
    if (int rc = vec(h, out_scope + (tf ? "/dense/bias" : ".dense.bias"), H, &w->bo)) return rc;
    if (int rc = vec(h, out_scope + (tf ? "/LayerNorm/gamma" : ".LayerNorm.weight"), H, &w->g)) return rc;
    if (int rc = vec(h, out_scope + (tf ? "/LayerNorm/beta" : ".LayerNorm.bias"), H, &w->b)) return rc;
    return MMS_OK;
}
int load_ffn(mms_handle* h, bool tf, const std::string& inter, const std::string& out, FfnW* w) {
    const int64_t I = h->cfg.inter;
    if (h->f8) {
        if (int rc = mat_f8(h, inter + (tf ? "/dense/kernel" : ".dense.weight"), I, H, tf, &w->wi8, &w->wis)) return rc;
        if (int rc = mat_f8(h, out + (tf ? "/dense/kernel" : ".dense.weight"), H, I, tf, &w->wd8, &w->wds)) return rc;
    }
    if (int rc = mat(h, inter + (tf ? "/dense/kernel" : ".dense.weight"), I, H, tf, &w->wi)) return rc;
    if (int rc = vec(h, inter + (tf ? "/dense/bias" : ".dense.bias"), I, &w->bi)) return rc;
    if (int rc = mat(h, out + (tf ? "/dense/kernel" : ".dense.weight"), H, I, tf, &w->wd)) return rc;
    if (int rc = vec(h, out + (tf ? "/dense/bias" : ".dense.bias"), H, &w->bd)) return rc;
    if (int rc = vec(h, out + (tf ? "/LayerNorm/gamma" : ".LayerNorm.weight"), H, &w->g)) return rc;
    if (int rc = vec(h, out + (tf ? "/LayerNorm/beta" : ".LayerNorm.bias"), H, &w->b)) return rc;
    return MMS_OK;
}

int finalize_tf_common(mms_handle* h) {
    const mms_config& c = h->cfg;
    if (int rc = tab(h, "bert/embeddings/word_embeddings", {c.vocab, H}, &h->E)) return rc;
    if (int rc = tab(h, "bert/embeddings/token_type_embeddings", {c.type_vocab, H}, &h->type_tab)) return rc;
    if (int rc = tab(h, "bert/embeddings/position_embeddings", {c.max_pos, H}, &h->pos_tab)) return rc;
    if (int rc = vec(h, "bert/embeddings/LayerNorm/gamma", H, &h->emb_g)) return rc;
    if (int rc = vec(h, "bert/embeddings/LayerNorm/beta", H, &h->emb_b)) return rc;
    h->layers.resize(c.layers);
    for (int i = 0; i < c.layers; ++i) {
        const std::string p = "bert/encoder/layer_" + std::to_string(i);
        if (int rc = load_att(h, true, p + "/attention/self", p + "/attention/output", &h->layers[i].att)) return rc;
        if (int rc = load_ffn(h, true, p + "/intermediate", p + "/output", &h->layers[i].ffn)) return rc;
    }
    if (int rc = mat(h, "bert/pooler/dense/kernel", H, H, true, &h->w_pool)) return rc;
    if (int rc = vec(h, "bert/pooler/dense/bias", H, &h->b_pool)) return rc;
    return MMS_OK;
}

int finalize_zk(mms_handle* h) {
    if (int rc = finalize_tf_common(h)) return rc;
    // kdd_conv1 [1,8,in,out] -> im2col weight [out][k*768 + in]
    const HostTensor* t;
    if (int rc = need(h, "kdd_conv1/weights", {1, MMS_LABEL_LEN, H, H}, &t)) return rc;
    if (int rc = upload_mat(h, {{t->data.data(), H, true}}, (int64_t)MMS_LABEL_LEN * H, &h->w_conv1)) return rc;
    if (int rc = vec(h, "kdd_conv1/biases", H, &h->b_conv1)) return rc;
    if (int rc = tab(h, "kdd_dense1/weights", {5, H}, &h->w_dense1)) return rc;
    if (int rc = vec(h, "kdd_dense1/biases", H, &h->b_dense1)) return rc;
    if (int rc = mat(h, "kdd_conv2/weights", H, MMS_FEAT, true, &h->w_conv2, {1, 1, MMS_FEAT, H})) return rc;
    if (int rc = vec(h, "kdd_conv2/biases", H, &h->b_conv2)) return rc;
    if (int rc = mat(h, "kdd_featureemb/fully_connected/weights", H, H, true, &h->w_femb)) return rc;
    if (int rc = vec(h, "kdd_featureemb/fully_connected/biases", H, &h->b_femb)) return rc;
    if (int rc = tab(h, "cls/seq_relationship/am_kernel", {H, 2}, &h->am_kernel)) return rc;
    return MMS_OK;
}

int finalize_lds(mms_handle* h) {
    if (int rc = finalize_tf_common(h)) return rc;
    if (int rc = mat(h, "featureemb/fully_connected/weights", H, MMS_FEAT, true, &h->w_feat)) return rc;
    if (int rc = vec(h, "featureemb/fully_connected/biases", H, &h->b_feat)) return rc;
    if (int rc = tab(h, "bert/embeddings/word_embeddings_labelembedding", {MMS_LABEL_LEN, 1}, &h->w_lab8)) return rc;
    if (int rc = tab(h, "cls/seq_relationship/output_weights", {2, H}, &h->w_cls)) return rc;
    if (int rc = vec(h, "cls/seq_relationship/output_bias", 2, &h->b_cls)) return rc;
    return MMS_OK;
}

int finalize_lxmert(mms_handle* h) {
    const mms_config& c = h->cfg;
    const std::string b = "lxrt_encoder.model.bert.";
    if (int rc = tab(h, b + "embeddings.word_embeddings.weight", {c.vocab, H}, &h->E)) return rc;
    if (int rc = tab(h, b + "embeddings.position_embeddings.weight", {c.max_pos, H}, &h->pos_tab)) return rc;
    if (int rc = tab(h, b + "embeddings.token_type_embeddings.weight", {c.type_vocab, H}, &h->type_tab)) return rc;
    if (int rc = vec(h, b + "embeddings.LayerNorm.weight", H, &h->emb_g)) return rc;
    if (int rc = vec(h, b + "embeddings.LayerNorm.bias", H, &h->emb_b)) return rc;
    const std::string v = b + "encoder.visn_fc.";
    if (int rc = mat(h, v + "visn_fc.weight", H, MMS_FEAT, false, &h->w_visn)) return rc;
    if (int rc = vec(h, v + "visn_fc.bias", H, &h->b_visn)) return rc;
    if (int rc = vec(h, v + "visn_layer_norm.weight", H, &h->g_visn)) return rc;
    if (int rc = vec(h, v + "visn_layer_norm.bias", H, &h->be_visn)) return rc;
    if (int rc = tab(h, v + "box_fc.weight", {H, 4}, &h->w_box)) return rc;
    if (int rc = vec(h, v + "box_fc.bias", H, &h->b_box)) return rc;
    if (int rc = vec(h, v + "box_layer_norm.weight", H, &h->g_box)) return rc;
    if (int rc = vec(h, v + "box_layer_norm.bias", H, &h->be_box)) return rc;
    if (int rc = tab(h, v + "label_conv.weight", {1, MMS_LABEL_LEN, 1, 1}, &h->w_lconv)) return rc;
    if (int rc = vec(h, v + "label_conv.bias", 1, &h->b_lconv)) return rc;
    if (int rc = mat(h, v + "label_fc.weight", H, H, false, &h->w_labfc)) return rc;
    if (int rc = vec(h, v + "label_fc.bias", H, &h->b_labfc)) return rc;
    if (int rc = vec(h, v + "label_layer_norm.weight", H, &h->g_lab)) return rc;
    if (int rc = vec(h, v + "label_layer_norm.bias", H, &h->be_lab)) return rc;
    h->layers.resize(c.layers);
    h->r_layers.resize(c.r_layers);
    h->x_layers.resize(c.x_layers);
    for (int i = 0; i < c.layers; ++i) {
        const std::string p = b + "encoder.layer." + std::to_string(i);
        if (int rc = load_att(h, false, p + ".attention.self", p + ".attention.output", &h->layers[i].att)) return rc;
        if (int rc = load_ffn(h, false, p + ".intermediate", p + ".output", &h->layers[i].ffn)) return rc;
    }
    for (int i = 0; i < c.r_layers; ++i) {
        const std::string p = b + "encoder.r_layers." + std::to_string(i);
        if (int rc = load_att(h, false, p + ".attention.self", p + ".attention.output", &h->r_layers[i].att)) return rc;
        if (int rc = load_ffn(h, false, p + ".intermediate", p + ".output", &h->r_layers[i].ffn)) return rc;
    }
    for (int i = 0; i < c.x_layers; ++i) {
        const std::string p = b + "encoder.x_layers." + std::to_string(i);
        XLayerW& x = h->x_layers[i];
        if (int rc = load_att(h, false, p + ".visual_attention.att", p + ".visual_attention.output", &x.cross)) return rc;
        if (int rc = load_att(h, false, p + ".lang_self_att.self", p + ".lang_self_att.output", &x.lang_self)) return rc;
        if (int rc = load_att(h, false, p + ".visn_self_att.self", p + ".visn_self_att.output", &x.visn_self)) return rc;
        if (int rc = load_ffn(h, false, p + ".lang_inter", p + ".lang_output", &x.lang_ffn)) return rc;
        if (int rc = load_ffn(h, false, p + ".visn_inter", p + ".visn_output", &x.visn_ffn)) return rc;
    }
    if (int rc = mat(h, b + "pooler.dense.weight", H, H, false, &h->w_pool)) return rc;
    if (int rc = vec(h, b + "pooler.dense.bias", H, &h->b_pool)) return rc;
    if (int rc = mat(h, "logit_fc.0.weight", 2 * H, H, false, &h->w_fc0)) return rc;
    if (int rc = vec(h, "logit_fc.0.bias", 2 * H, &h->b_fc0)) return rc;
    if (int rc = vec(h, "logit_fc.2.weight", 2 * H, &h->g_fc2)) return rc;
    if (int rc = vec(h, "logit_fc.2.bias", 2 * H, &h->be_fc2)) return rc;
    if (int rc = tab(h, "logit_fc.3.weight", {2, 2 * H}, &h->w_fc3)) return rc;
    if (int rc = vec(h, "logit_fc.3.bias", 2, &h->b_fc3)) return rc;
    return MMS_OK;
}

// ------------------------------------------------------------------------------------------------
// workspace
// ------------------------------------------------------------------------------------------------
int alloc_planes(mms_handle* h, std::vector<void*>& pool, Planes* p, int64_t elems) {
    void* q;
    if (int rc = dev_alloc(h, pool, &q, (size_t)elems * 2 * 2)) return rc;
    p->hi = (bf16*)q;
    p->lo = p->hi + MMS_PLANE_LO;      // hl32: one buffer of 2 * elems, 32-element blocks of hi and lo alternating (common.h)
    return MMS_OK;
}

int ensure_kparts(mms_handle* h, int64_t floats);
int64_t kparts_need(const mms_handle* h, int64_t pairs, int64_t rows);
int64_t kparts_side_need(const mms_handle* h, int64_t pairs);
// the lane-private buffers of the chain that is being enqueued (mms_handle::lane)
inline float* kp(const mms_handle* h) { return h->lane ? h->kparts_side : h->kparts; }
inline Planes midp(const mms_handle* h) { return h->lane ? h->mid.at(h->mid_side_off) : h->mid; }
// lxmert calls whose chunk has fewer token rows (pairs x (text_len + 10), padded) than this run their independent launch chains on two lanes
// LANE_ROWS_DEFAULT (regimes.h) = 400 000: measured (profiles/rd5_lanes.txt): +27 % at 256 pairs, +30 % at 1024, +28 % at 2048, +13 % at 4096, +8.6 % at 8192, +3 % at 16384 (= the query stage alone), +0.7 % at 30000 (query stage alone: +2 %)
int64_t lane_rows() {
#ifdef MMS_LAB
    static const int64_t v = getenv("MMS_LANE_ROWS") ? atoll(getenv("MMS_LANE_ROWS")) : LANE_ROWS_DEFAULT;      // A/B: 0 = one lane
    return v;
#else
    return LANE_ROWS_DEFAULT;
#endif
}
// ... and above that only the distinct-query stage runs on the side lane (beside the chunk's prologue and box-stream layers)
bool lane_query_stage() {
#ifdef MMS_LAB
    static const bool v = !getenv("MMS_LANE_QUERY") || atoi(getenv("MMS_LANE_QUERY")) != 0;
    return v;
#else
    return true;
#endif
}

int ensure_workspace(mms_handle* h, int64_t pairs) {
    if (pairs <= h->ws_pairs) return MMS_OK;
    free_pool(h->ws_allocs);
    h->ws_pairs = 0;
    h->ens_tok = nullptr;
    h->x.f8 = h->y.f8 = h->ctx.f8 = h->mid.f8 = nullptr;
    const mms_config& c = h->cfg;
    const int64_t T = c.text_len;
    int64_t rows, mid_rows;
    if (c.model == MMS_MODEL_ZK) rows = mid_rows = pairs * (T + MMS_NBOX);
    else if (c.model == MMS_MODEL_LDS) rows = mid_rows = pairs * (T + 2 * MMS_NBOX);
    else {      // (two lanes: the language and the vision stream's FFN intermediates are live at once)
        rows = pairs * (T + MMS_NBOX);
        mid_rows = rows < lane_rows() ? rows : pairs * (T > MMS_NBOX ? T : MMS_NBOX);
    }
    h->x_rows = rows / pairs;
    // mid also hosts the split 2048-d box features during the embedding stage
    int64_t mid_elems = mid_rows * c.inter;
    const int64_t feat_elems = pairs * MMS_NBOX * MMS_FEAT;
    if (mid_elems < feat_elems) mid_elems = feat_elems;
    void* p;
    if (int rc = alloc_planes(h, h->ws_allocs, &h->x, rows * H)) return rc;
    if (int rc = alloc_planes(h, h->ws_allocs, &h->ctx, rows * H)) return rc;
    if (int rc = alloc_planes(h, h->ws_allocs, &h->y, rows * H)) return rc;
    if (int rc = alloc_planes(h, h->ws_allocs, &h->mid, mid_elems)) return rc;
    h->mid_elems = mid_elems;
    if (h->f8) {   // e4m3 copies of the GEMM A operands (x / y: LayerNorm outputs, ctx: attention output, mid: FFN intermediate)
        // + 256 rows: gemm_mx8_kernel fetches whole 256-row panels (rows past the live count are multiplied and discarded, never clamped)
        if (int rc = dev_alloc(h, h->ws_allocs, &p, (size_t)(rows + 256) * H)) return rc;
        h->x.f8 = (unsigned char*)p;
        if (int rc = dev_alloc(h, h->ws_allocs, &p, (size_t)(rows + 256) * H)) return rc;
        h->y.f8 = (unsigned char*)p;
        if (int rc = dev_alloc(h, h->ws_allocs, &p, (size_t)(rows + 256) * H)) return rc;
        h->ctx.f8 = (unsigned char*)p;
        if (int rc = dev_alloc(h, h->ws_allocs, &p, (size_t)(mid_rows + 256) * c.inter)) return rc;
        h->mid.f8 = (unsigned char*)p;
    }
    if (int rc = dev_alloc(h, h->ws_allocs, &p, (size_t)rows * 3 * H * 4)) return rc;
    h->qkv = (float*)p;
    if (int rc = dev_alloc(h, h->ws_allocs, &p, (size_t)rows * H * 4)) return rc;
    h->t = (float*)p;
    if (int rc = dev_alloc(h, h->ws_allocs, &p, (size_t)rows * 4)) return rc;
    h->key_add = (float*)p;
    if (int rc = dev_alloc(h, h->ws_allocs, &p, (size_t)rows * 4)) return rc;
    h->key_add2 = (float*)p;
    if (int rc = dev_alloc(h, h->ws_allocs, &p, (size_t)pairs * H * 4)) return rc;
    h->pooled = (float*)p;
    if (int rc = dev_alloc(h, h->ws_allocs, &p, (size_t)pairs * 2 * H * 4)) return rc;
    h->hbuf = (float*)p;
    for (int s = 0; s < 2; ++s) {
        if (int rc = dev_alloc(h, h->ws_allocs, &p, (size_t)pairs * 4)) return rc;
        h->pk_off[s] = (int*)p;
        if (int rc = dev_alloc(h, h->ws_allocs, &p, (size_t)pairs * 4)) return rc;
        h->pk_cnt[s] = (int*)p;
        if (int rc = dev_alloc(h, h->ws_allocs, &p, (size_t)rows * 4)) return rc;
        h->pk_src[s] = (int*)p;
    }
    if (int rc = dev_alloc(h, h->ws_allocs, &p, 16)) return rc;
    h->pk_rows = (int*)p;
    for (int s = 0; s < (c.model == MMS_MODEL_LXMERT ? 4 : 2); ++s) {      // a sub-tile holds at least one pair
        if (int rc = dev_alloc(h, h->ws_allocs, &p, (size_t)(pairs + 2) * sizeof(int4))) return rc;
        h->qa_sub[s] = (int4*)p;
    }
    for (int s = 0; s < (c.model == MMS_MODEL_LXMERT ? 3 : 1); ++s) {
        if (int rc = dev_alloc(h, h->ws_allocs, &p, (size_t)(pairs + 2) * 4)) return rc;
        h->qa_rec[s] = (int*)p;
    }
    if (int rc = dev_alloc(h, h->ws_allocs, &p, (size_t)qkv_plan_scratch_ints((int)pairs) * 4)) return rc;
    h->qa_jump = (int*)p;
    if (int rc = dev_alloc(h, h->ws_allocs, &p, 16)) return rc;
    h->qa_nsub = (int*)p;
    if (int rc = dev_alloc(h, h->ws_allocs, &p, (size_t)(rows + 256) * 3 * 2 * 8)) return rc;
    h->ln_stats = (float*)p;
    HIP_TRY(h, hipMemset(p, 0, (size_t)(rows + 256) * 3 * 2 * 8));
    if (int rc = dev_alloc(h, h->ws_allocs, &p, (size_t)mms_handle::LN_SLOTS * 2 * 4)) return rc;
    h->ln_ctl = (int*)p;
    if (int rc = ensure_kparts(h, kparts_need(h, pairs, rows))) return rc;
    if (c.model == MMS_MODEL_LXMERT && lane_rows() > 0) {      // whichever call sizes the workspace: the side lane's partials come with it, never inside a scoring call
        h->lane = 1;
        const int rc = ensure_kparts(h, kparts_side_need(h, pairs));
        h->lane = 0;
        if (rc) return rc;
    }
    h->ws_pairs = pairs;
    return MMS_OK;
}

constexpr int64_t LAB_CHUNK = 2048;  // unique label texts encoded per GEMM wave (bounds the im2col buffer)

int ensure_label_ws(mms_handle* h, int64_t U) {
    if (U > h->lab_feat_cap || h->lab_cap == 0) {
        free_pool(h->lab_allocs);
        h->lab_cap = 0;
        const int64_t cap = U < LAB_CHUNK ? U : LAB_CHUNK;
        const int64_t kdim = h->cfg.model == MMS_MODEL_ZK ? (int64_t)MMS_LABEL_LEN * H : H;
        const int64_t rows = h->cfg.model == MMS_MODEL_ZK ? cap * MMS_LABEL_LEN : cap;
        void* p;
        if (int rc = alloc_planes(h, h->lab_allocs, &h->lab_planes, rows * kdim)) return rc;
        if (int rc = dev_alloc(h, h->lab_allocs, &p, (size_t)rows * H * 4)) return rc;
        h->lab_f32 = (float*)p;
        if (int rc = dev_alloc(h, h->lab_allocs, &p, (size_t)U * H * 4)) return rc;
        h->lab_feat = (float*)p;
        h->lab_cap = cap;
        h->lab_feat_cap = U;
    }
    return MMS_OK;
}

// ------------------------------------------------------------------------------------------------
// launch helpers
// ------------------------------------------------------------------------------------------------
struct GemmOut {
    int hm_rows = 0, hm_col0 = 0;                       // head-major [Q | K | V] blocks (kernels.h GemmParams::hm_rows)
    float* f32 = nullptr; int ldc = 0;
    Planes pl; int ldp = 0;
    RowMap cmap{0, 0, 0};
};

int ensure_kparts(mms_handle* h, int64_t floats);
// Launches of at most SKINNY_ROWS padded rows in precision modes 2 and 3 (the reference's zk call size: 1 pair = 30 token rows; up to 4 zk / 3 lds / 6 lxmert pairs,
// and the box-row projections of up to 12 pairs): gemm_skinny.hip -- one workgroup per 16 output columns, no LDS staging; K split over the workgroup's waves
// (wide projections: ONE launch where the split-K tile route needs two) or, where a LayerNorm / reduce launch follows anyway, over single-wave workgroups
// that leave the tile engine's partials.  128, measured (profiles/rd4r_skinny_gemm.txt): above it the rows a workgroup pulls through one CU's fill path cost
// more than the launch saved.
int64_t skinny_rows() {
#ifdef MMS_LAB
    static const int64_t v = getenv("MMS_SKINNY_ROWS") ? atoll(getenv("MMS_SKINNY_ROWS")) : SKINNY_ROWS_DEFAULT;      // A/B: 0 = the split-K routes of round 4
    return v;
#else
    return SKINNY_ROWS_DEFAULT;
#endif
}
bool skinny_shape(const mms_handle* h, int64_t M, int K) {
    return (h->nsplit == 2 || h->nsplit == 3) && !h->f8 && M <= skinny_rows() && K % 256 == 0 && (K < 2048 || K % 512 == 0) && h->resid_in_ln;
}
// Small launches (M < TINY_ROWS padded token rows: calls of up to 34 zk / 25 lds / 51 lxmert pairs) of the wide projections (QKV, K | V, FFN-up:
// N >= 1536, K = 768): the 128 x 256 tile grid is a few dozen workgroups walking K serially (25 .. 42 us at ANY such M: the time of one tile); four K slices
// + k_splitk_reduce (sum in fixed order, bias, activation, head-major fp32 or planes) take 10 + 6.  The N = 768 projections use proj_ln() below (no reduce
// launch at all).  1024, measured (profiles/rd4x_tile_engines_midsize.txt): zk 12 / 17 / 34 pairs -16 / -15 / -7 % against a bound of 256, lds 12 pairs -14 %;
// from ~2000 rows on the partial traffic and the second launch cost more than the shorter K walk saves (zk 68 pairs +10 %, lds 50 pairs +20 % at a bound of 2048).
int64_t tiny_rows() {
#ifdef MMS_LAB
    static const int64_t v = getenv("MMS_TINY_ROWS") ? atoll(getenv("MMS_TINY_ROWS")) : TINY_ROWS_DEFAULT;      // A/B: row bound of the wide split-K route (<= 4096: kparts)
    return v;
#else
    return TINY_ROWS_DEFAULT;
#endif
}
#define TINY_ROWS tiny_rows()
// padded row bound from which a stream's QKV projection + attention run as the fused kernel (qkv_attn.hip).  Rounds 3-4: 16384 (the persistent engines' bound); measured
// this round (profiles/rd5_fused_rows.txt): 16 row tiles x 12 heads fill 192 CUs at 256 zk pairs, one launch of ~32 us where the 128 x 256 tile grid of the QKV projection
// takes 47 (279 workgroups on 256 CUs: two rounds) and the attention kernel 16 -- zk 2.53 -> 2.34 ms at 256 pairs, lds 3.72 -> 3.38, lxmert 3.50 -> 3.28; down to 1024 rows
// (= TINY_ROWS, where the wide projections' split-K route takes over) never slower
int64_t fused_attn_rows() {
#ifdef MMS_LAB
    static const int64_t v = getenv("MMS_FUSED_ROWS") ? atoll(getenv("MMS_FUSED_ROWS")) : FUSED_ATTN_ROWS_DEFAULT;
    return v;
#else
    return FUSED_ATTN_ROWS_DEFAULT;
#endif
}

int gemm(mms_handle* h, hipStream_t st, Planes a, int lda, RowMap amap, const bf16* w, const float* bias, int64_t M,
         int N, int K, int act, const GemmOut& out, const Planes* resid = nullptr, const int* m_dev = nullptr,
         const int* a_index = nullptr, RowMap rmap = RowMap{0, 0, 0}, const int* r_index = nullptr, int cls_bit = 0, int force_ks = 0) {
    if (M <= 0) return MMS_OK;
    const int nsplit = (h->nsplit == 2 && (h->x1_mask & cls_bit)) ? 1 : h->nsplit;
    if (N % 128 || K % 64) return h->fail(MMS_ERR_ARG, "gemm: N % 128 or K % 64 != 0");
    const bool skinny = (nsplit == 2 || nsplit == 3) && skinny_shape(h, M, K);
    const bool splittable = h->nsplit >= 2 && !h->f8 && !resid && out.cmap.grp == 0 && !(h->x1_mask & cls_bit) && N % 256 == 0;
    const bool wide = splittable && M < TINY_ROWS && N >= 1536 && K == H;
    // ... and the two long-K projections in front of the encoder at any small M: kdd_conv1 as im2col (K = 6144, M = 8 x distinct label texts: 96 serial
    // K steps, 170 us per call whatever the batch) and kdd_conv2 / visn_fc / featureemb (K = 2048) below TALL_ROWS box rows: eight K slices
    const bool tall = splittable && !wide && M < TALL_ROWS && N == H && K >= 2048 && K % 512 == 0;
    // a long-K projection of a few rows (kdd_conv1 as im2col: 80 rows x K = 6144 in a 1-pair call): its K slices go to single-wave WORKGROUPS of the skinny
    // kernel and k_splitk_reduce sums them -- in one workgroup per 16 columns every workgroup pulls all rows x all of K (2 MB) through one CU: 40 us
    const bool skinny_tall = skinny && tall;
    const bool tiny = (wide || tall) && (!skinny || skinny_tall);
    // the skinny kernel slices K the way this projection's tile route does for launches below TINY_ROWS, so that it is bit-identical to it
    const int skinny_ks = force_ks ? force_ks : wide ? 4 : tall ? 8 : 1;
    GemmParams p{};
    p.a_hi = a.hi; p.a_lo = a.lo; p.lda = lda; p.amap = amap;
    p.w = w; p.bias = bias; p.M = (int)M; p.N = N; p.K = K;
    if (h->nsplit == 3) {   // locate the lo plane of (a view into) this weight matrix
        for (const auto& wp : h->w_planes)
            if (w >= wp.base && w < wp.base + wp.elems) { p.w_lo = w + wp.elems; break; }
        if (!p.w_lo) return h->fail(MMS_ERR_STATE, "gemm: weight has no lo plane");
    }
    p.act = act;
    p.out_kind = out.f32 ? OUT_F32 : OUT_PLANES;
    p.c_f32 = out.f32; p.ldc = out.ldc; p.hm_rows = out.hm_rows; p.hm_col0 = out.hm_col0;
    p.c_hi = out.pl.hi; p.c_lo = out.pl.lo; p.ldp = out.ldp; p.cmap = out.cmap;
    if (resid && !h->resid_in_ln) { p.r_hi = resid->hi; p.r_lo = resid->lo; p.ldr = H; }
    p.m_dev = m_dev; p.a_index = a_index; p.rmap = rmap; p.r_index = r_index;
    const int TINY_S = tall ? 8 : 4;
    const long long part_stride = (long long)M * N;      // the launch's own row bound: a 1-pair call needs a few MB of partials, not the 200 MB of the largest split launch
    if (tiny) {      // K slices into fp32 partials; the reduce kernel below applies what the epilogue would have
        if (int rc = ensure_kparts(h, TINY_S * part_stride)) return rc;      // (no-op: ensure_workspace sized it for every split launch of this workspace)
        (skinny_tall ? h->skinny_launches : h->splitk_launches) += 1;
        p.bias = nullptr; p.act = ACT_NONE; p.out_kind = OUT_F32; p.c_f32 = kp(h); p.ldc = N; p.hm_rows = 0; p.hm_col0 = 0;
        p.k_splits = TINY_S; p.c_split_stride = part_stride;
        if (skinny_tall) p.engine = ENG_SKINNY_PARTS;
    }
    auto tiny_reduce = [&]() {
        if (tiny) launch_splitk_reduce(kp(h), TINY_S, part_stride, (int)M, N, m_dev, bias, act, out.f32, out.ldc, out.hm_rows, out.hm_col0,
                                       out.pl.hi, out.pl.lo, out.ldp, st);
    };
    if (h->alternate) { p.reverse = h->flip; h->flip ^= 1; }
    if (skinny && !skinny_tall) { p.engine = ENG_SKINNY; p.wave_k_slices = skinny_ks; h->skinny_launches += 1; }
    if (h->timing && h->lane) p.flop_counter = h->flop_counter + 3;      // side lane: FLOPs only (slot 3, mms_side_lane_flops) -- a launch that shares the chip has no duration of its own
    if (h->timing && !h->lane) {      // (per-launch events and FLOP counts cover the main lane: a side-lane launch shares the chip with the launch beside it)
        if (h->ev_used + 2 > h->ev.size()) { if (int rc = grow_event_pair(h, h->ev)) return rc; }
        p.flop_counter = h->flop_counter;   // executed algorithmic FLOPs (2*M_live*N*K), counted on the device
        HIP_TRY(h, hipEventRecord(h->ev[h->ev_used], st));
        if (!launch_gemm(p, nsplit, st)) return h->fail(MMS_ERR_ARG, "gemm: no engine takes this shape");
        tiny_reduce();
        HIP_TRY(h, hipEventRecord(h->ev[h->ev_used + 1], st));
        h->ev_cls.resize(h->ev_used / 2 + 1); h->ev_cls[h->ev_used / 2] = 0;
        h->ev_used += 2;
        h->gemm_launches += 1;
    } else {
        if (!launch_gemm(p, nsplit, st)) return h->fail(MMS_ERR_ARG, "gemm: no engine takes this shape");
        tiny_reduce();
    }
    return MMS_OK;
}

// precision mode 4: C = act((A8 W8^T) * scale[n] + bias) on e4m3 operands, MX-scaled fp8 MFMA with the per-channel weight scale as the
// weight operand's hardware scale (gemm_mx.hip gemm_mx8_kernel); out: fp32 (out.f32) or e4m3 bytes (out.pl.f8).  a8 rows must be readable
// up to the next multiple of 256 (the workspace pads its byte planes)
int gemm_f8(mms_handle* h, hipStream_t st, const unsigned char* a8, int lda, const unsigned char* w8, const unsigned* wscale4,
            const float* bias, int64_t M, int N, int K, int act, const GemmOut& out, const int* m_dev) {
    if (M <= 0) return MMS_OK;
    if (N % 256 || K % 128) return h->fail(MMS_ERR_ARG, "gemm_f8: N % 256 or K % 128");
    GemmParams p{};
    p.a8 = a8; p.lda = lda; p.amap = RowMap{0, 0, 0};
    p.w8 = w8; p.w8_scale4 = wscale4; p.bias = bias; p.M = (int)M; p.N = N; p.K = K;
    p.act = act;
    if (out.f32) { p.out_kind = OUT_F32; p.c_f32 = out.f32; p.ldc = out.ldc; p.hm_rows = out.hm_rows; p.hm_col0 = out.hm_col0; }
    else { p.out_kind = OUT_F8; p.c_f8 = out.pl.f8; p.ldf8 = out.ldp; }
    p.cmap = out.cmap; p.rmap = RowMap{0, 0, 0};
    p.m_dev = m_dev;
    if (h->alternate) { p.reverse = h->flip; h->flip ^= 1; }
    if (h->timing && h->lane) p.flop_counter = h->flop_counter + 3;      // side lane: FLOPs only (slot 3, mms_side_lane_flops) -- a launch that shares the chip has no duration of its own
    if (h->timing && !h->lane) {
        if (h->ev_used + 2 > h->ev.size()) { if (int rc = grow_event_pair(h, h->ev)) return rc; }
        p.flop_counter = h->flop_counter;
        HIP_TRY(h, hipEventRecord(h->ev[h->ev_used], st));
        if (!launch_gemm_mx8(p, st)) return h->fail(MMS_ERR_ARG, "gemm_f8: shape not supported");
        HIP_TRY(h, hipEventRecord(h->ev[h->ev_used + 1], st));
        h->ev_cls.resize(h->ev_used / 2 + 1); h->ev_cls[h->ev_used / 2] = 0;
        h->ev_used += 2;
        h->gemm_launches += 1;
    } else if (!launch_gemm_mx8(p, st)) return h->fail(MMS_ERR_ARG, "gemm_f8: shape not supported");
    return MMS_OK;
}

void ln_resid(mms_handle* h, hipStream_t st, const float* t, const float* g, const float* b, Planes out, int64_t M, const int* m_dev,
              const Planes& resid, RowMap rmap, const int* r_index, const int* skip);

// out = LayerNorm(A W^T + bias + resid) for the N = 768 projections, the LayerNorm fused into the GEMM epilogue when the launch is big
// enough for the persistent ping-pong engine (gemm_pp_ln.h).  Returns MMS_OK with *fused = false when the caller has to take the
// two-kernel route itself (small M, precision modes 1 / 3, no control slot left).
// Padded row bound of the fused bias + residual + LayerNorm epilogue.  Rounds 2-4: 16 384 (with the persistent engines).  Its tile carries eight residual K stages and a statistics exchange across the
// three column tiles of a row panel; on a launch of a round or two that is latency the LayerNorm kernel does not add: with the two-kernel route zk takes 4.15 instead of 4.75 ms at 600 pairs, 6.33 / 6.99 at
// 1024, 11.4 / 12.0 at 2048 and 19.7 / 19.5 at 4096 (122 880 rows: the epilogue wins from here on); lds 5.69 / 6.39 at 512, 10.6 / 11.8 at 1024, a tie at 2048 (81 920 rows) -- profiles/rd5_fuse_ln_midsize.txt
int64_t lnf_rows() {
#ifdef MMS_LAB
    static const int64_t v = getenv("MMS_LNF_ROWS") ? atoll(getenv("MMS_LNF_ROWS")) : LNF_ROWS_DEFAULT;
    return v > pp_rows() ? v : pp_rows();
#else
    return LNF_ROWS_DEFAULT;
#endif
}

int gemm_ln(mms_handle* h, hipStream_t st, bool f8, const Planes& a, int lda, const bf16* w, const unsigned char* w8, const unsigned* wscale,
            const float* bias, int64_t M, int K, const Planes& resid, const float* g, const float* b, const Planes& out, float* t,
            const int* m_dev, bool* fused) {
    *fused = false;
    // mms_config.fuse_layernorm is a mask: bit 0 the attention-output projections (K = 768), bit 1 the FFN-down projections (K = inter)
    if (!(h->fuse_ln & (K == H ? 1 : 2)) || f8 || M < lnf_rows() || h->nsplit != 2 || !h->resid_in_ln || h->ln_slot >= mms_handle::LN_SLOTS) return MMS_OK;
    // two launch lanes: the fused epilogue needs its whole grid resident (the column tiles of a row panel exchange statistics) and falls back to the LayerNorm
    // kernel when it is not -- beside another lane's persistent kernel that would be decided by timing.  A call on two lanes takes the two-kernel route throughout.
    if (h->no_fused_ln || h->lanes_on || h->lane || h->lq_join_pending) return MMS_OK;      // (lq_join_pending: the distinct-query stage is running on the side lane)
    if (K % 64 != 0) return MMS_OK;
    (void)w8; (void)wscale;          // the fp8 mode always takes the two-kernel route (f8 returned above)
    GemmParams p{};
    p.a_hi = a.hi; p.a_lo = a.lo; p.lda = lda; p.w = w; p.K = K;
    p.amap = RowMap{0, 0, 0}; p.cmap = RowMap{0, 0, 0}; p.rmap = RowMap{0, 0, 0};
    p.bias = bias; p.M = (int)M; p.N = H; p.act = ACT_NONE;
    p.r_hi = resid.hi; p.r_lo = resid.lo; p.ldr = H;
    p.c_hi = out.hi; p.c_lo = out.lo; p.ldp = H; p.c_f8 = out.f8; p.ldf8 = H;
    p.out_kind = OUT_F32; p.c_f32 = t; p.ldc = H;                    // the plain route of the same launch
    p.m_dev = m_dev;
    p.ln_gamma = g; p.ln_beta = b; p.ln_stats = h->ln_stats; p.ln_tag = ++h->ln_tag; p.ln_ctl = h->ln_ctl + 2 * h->ln_slot;

    if (h->alternate) { p.reverse = h->flip; h->flip ^= 1; }
    const int* skip = p.ln_ctl + 1;
    h->ln_slot += 1;
    h->ln_fused_launches += 1;
    if (h->timing) {
        if (h->ev_used + 2 > h->ev.size()) { if (int rc = grow_event_pair(h, h->ev)) return rc; }
        p.flop_counter = h->flop_counter + 2;     // slot 2: the LayerNorm-fused launches (mms_gemm_timing adds it to the GEMM total)
        HIP_TRY(h, hipEventRecord(h->ev[h->ev_used], st));
        if (!launch_gemm_pp_ln(p, h->nsplit, st)) return h->fail(MMS_ERR_ARG, "gemm_ln: shape not supported");
        HIP_TRY(h, hipEventRecord(h->ev[h->ev_used + 1], st));
        h->ev_cls.resize(h->ev_used / 2 + 1); h->ev_cls[h->ev_used / 2] = 1;
        h->ev_used += 2;
        h->gemm_launches += 1;
    } else if (!launch_gemm_pp_ln(p, h->nsplit, st)) return h->fail(MMS_ERR_ARG, "gemm_ln: shape not supported");
    // queued behind the GEMM: does the LayerNorm only if that launch had to take the plain route (ln_ctl[1] != 1)
    ln_resid(h, st, t, g, b, out, M, m_dev, resid, RowMap{0, 0, 0}, nullptr, skip);
    *fused = true;
    return MMS_OK;
}

// every score call starts with fresh control words for its fused launches
int ln_begin_call(mms_handle* h, hipStream_t st) {
    h->ln_slot = 0;
    if (h->ln_ctl) HIP_TRY(h, hipMemsetAsync(h->ln_ctl, 0, (size_t)mms_handle::LN_SLOTS * 2 * 4, st));
    return MMS_OK;
}

GemmOut to_f32(float* p, int ldc) { GemmOut o; o.f32 = p; o.ldc = ldc; return o; }
// The [Q | K | V] projection of a token stream (first row row0, at most `rows` rows) in its own region of the qkv buffer, head-major:
// 36 blocks [rows][64] (Q heads, K heads, V heads).  The GEMM's 64-column wave tiles write whole blocks row after row, the attention
// kernel's per-(pair, head) wave reads S consecutive 256-byte rows -- contiguous on both sides instead of 256 bytes every 9216.
float* qkv_region(mms_handle* h, int64_t row0) { return h->qkv + row0 * 3 * H; }
GemmOut to_qkv(mms_handle* h, int64_t row0, int64_t rows, int col0 = 0) {
    GemmOut o; o.f32 = qkv_region(h, row0); o.ldc = 3 * H; o.hm_rows = (int)rows; o.hm_col0 = col0; return o;
}
void attn_q(AttnParams& a, mms_handle* h, int64_t row0, int64_t rows) { a.q = qkv_region(h, row0); a.ldq = MMS_HEAD_DIM; a.hs_q = rows * MMS_HEAD_DIM; }
void attn_kv(AttnParams& a, mms_handle* h, int64_t row0, int64_t rows) {
    a.k = qkv_region(h, row0) + 12 * rows * MMS_HEAD_DIM; a.v = qkv_region(h, row0) + 24 * rows * MMS_HEAD_DIM;
    a.ldkv = MMS_HEAD_DIM; a.hs_kv = rows * MMS_HEAD_DIM;
}
GemmOut to_planes(Planes p, int ldp, RowMap m = RowMap{0, 0, 0}) { GemmOut o; o.pl = p; o.ldp = ldp; o.cmap = m; return o; }
const RowMap ID{0, 0, 0};

// LayerNorm closing a residual sub-layer: out = LN(t + resid).  The residual rides here (default) or was already added
// by the producing GEMM's epilogue (h->resid_in_ln == 0); the arithmetic is the same fp32 (acc + bias) + resid either way.
void ln_resid(mms_handle* h, hipStream_t st, const float* t, const float* g, const float* b, Planes out, int64_t M, const int* m_dev,
              const Planes& resid, RowMap rmap = RowMap{0, 0, 0}, const int* r_index = nullptr, const int* skip = nullptr) {
    LnResid r;
    r.skip = skip;
    if (h->resid_in_ln) { r.hi = resid.hi; r.lo = resid.lo; r.ld = H; r.rmap = rmap; r.r_index = r_index; }
    if (h->alternate) { r.reverse = h->flip; h->flip ^= 1; }
    r.o_f8 = out.f8;
    launch_ln_to_planes(t, H, g, b, out.hi, out.lo, H, (int)M, st, m_dev, r);
}

// Small launches of the N = 768 projections (attention output, FFN down: M < SPLITK_ROWS padded token rows, i.e. calls of up to ~270 pairs --
// the reference's own call sizes, evaluate_normal.py:15, run_pretraining_predict_score.py:523, lxmert/src/param.py:46): a 128 x 256 tile
// grid has a handful of workgroups there and each walks its whole K serially (42 .. 100 us per launch at ANY M below ~4000,
// profiles/rd4g_small_batch_kernels.txt).  Split-K: S copies of the tile grid contract K / S columns each into fp32 partials, and the
// LayerNorm kernel that follows anyway sums them (fixed order: deterministic) and adds bias + residual -- no extra launch.  S depends on K
// alone, so launches in this regime stay bit-identical across batch sizes.
// SPLITK_ROWS (regimes.h): padded row bound of the launch (the live count is on the device): zk calls of <= 273 pairs
int splitk_for(const mms_handle* h, int64_t M, int K) {
    if (h->nsplit == 1 || h->f8) return 1;
    // 8192 .. 16383 rows: one pass over K -- except K = 3072 (FFN-down) from 11 264 rows on, in 2 slices: unsplit it is 130 .. 190 live workgroups walking 48 K steps on 256 CUs
    // (zk 400 / 520 pairs 3.63 -> 3.41 / 3.85 -> 3.71 ms, lxmert 600 pairs 3.85 -> 3.67; below 11 264 rows lds' fuller launches lose 3 %: profiles/rd5_splitk_midsize.txt)
    if (M >= SPLITK_ROWS) {
        int64_t lo = SPLITK2_LO_ROWS;
        int s2 = 2;
#ifdef MMS_LAB
        { static const int64_t e = getenv("MMS_SPLITK2_LO_ROWS") ? atoll(getenv("MMS_SPLITK2_LO_ROWS")) : 0; if (e) lo = e; }
#endif
        return (K >= 2048 && M >= lo && M < pp_rows() && K % (64 * s2) == 0) ? s2 : 1;
    }
    // K = 3072 (FFN-down): 8 slices up to 4095 rows, 4 from there on -- at 256 zk pairs (7680 padded rows) 8 slices are 744 workgroups of 6 K steps and 95 MB of partials for the
    // LayerNorm kernel to sum; 4 slices: zk 2.38 -> 2.19 ms per 256-pair call, lds 2.57 -> 2.38 at 150 pairs, neutral at 100 zk pairs (profiles/rd5_splitk_midsize.txt)
    int S = K >= 2048 ? (M >= SPLITK_HALF_ROWS ? 4 : 8) : 4;
#ifdef MMS_LAB
    if (M >= 1024) {      // A/B: slice counts of the upper half of the regime
        static const int s768 = getenv("MMS_SPLITK_768") ? atoi(getenv("MMS_SPLITK_768")) : 0, s3072 = getenv("MMS_SPLITK_3072") ? atoi(getenv("MMS_SPLITK_3072")) : 0;
        if (K < 2048 && s768) S = s768;
        if (K >= 2048 && s3072) S = s3072;
    }
#endif
    return K % (64 * S) == 0 ? S : 1;
}
int ensure_kparts(mms_handle* h, int64_t floats) {      // (of the lane that is being enqueued)
    float*& buf = h->lane ? h->kparts_side : h->kparts;
    int64_t& have = h->lane ? h->kparts_side_floats : h->kparts_floats;
    if (floats <= have) return MMS_OK;
    if (buf) { (void)hipFree(buf); buf = nullptr; have = 0; }      // (hipFree waits for the work that may still read it)
    void* p = nullptr;
    HIP_TRY(h, hipMalloc(&p, (size_t)floats * 4));
    buf = (float*)p; have = floats;
    return MMS_OK;
}
// Partials of every split-K launch a workspace of `pairs` pairs can issue (sized here, with the workspace, so that a scoring call on a warmed-up
// handle never allocates; a 1-pair handle holds ~2 MB instead of a fixed 200 MB): the N = 768 projections of launches below SPLITK_ROWS token rows (8 slices x rows x 768), the
// wide projections below TINY_ROWS (4 x rows x max(2304, inter)), the long-K projections in front of the encoder below 4096 box / label-text rows (8 x rows x 768)
int64_t kparts_need(const mms_handle* h, int64_t pairs, int64_t rows) {
    const int64_t r_ln = rows < SPLITK_ROWS ? rows : SPLITK_ROWS, r_wide = rows < TINY_ROWS ? rows : TINY_ROWS;
    const int64_t lab = pairs * MMS_NBOX * MMS_LABEL_LEN, r_tall = lab < 4096 ? lab : 4096;
    const int64_t nmax = h->cfg.inter > 3 * H ? h->cfg.inter : 3 * H;
    int64_t need = (int64_t)KSPLIT_MAX * r_ln * H;
    if (4 * r_wide * nmax > need) need = 4 * r_wide * nmax;
    if ((int64_t)KSPLIT_MAX * r_tall * H > need) need = (int64_t)KSPLIT_MAX * r_tall * H;
    return need;
}

// ... of the side lane (lxmert): it runs the box stream's chains (10 rows per pair) or the distinct-query stage (<= text_len rows per pair), never both streams' rows
int64_t kparts_side_need(const mms_handle* h, int64_t pairs) {
    const int64_t per = h->cfg.text_len > MMS_NBOX ? h->cfg.text_len : MMS_NBOX;
    return kparts_need(h, pairs, pairs * per);
}

// out = LayerNorm(A W^T + bias + resid) for a bf16 N = 768 projection on the two-kernel route: split-K partials (small M) or the plain fp32
// tensor t, then the LayerNorm kernel.  a_index / amap: A rows; rmap / r_index: residual rows.
int proj_ln(mms_handle* h, hipStream_t st, const Planes& a, int lda, RowMap amap, const int* a_index, const bf16* w, const float* bias, int64_t M,
            int K, const Planes& resid, RowMap rmap, const int* r_index, const float* g, const float* b, const Planes& out, float* t,
            const int* m_dev, int cls_bit) {
    const int S = splitk_for(h, M, K);
    const bool skinny = skinny_shape(h, M, K);      // <= 128 rows: the partials come from the skinny kernel, one single-wave workgroup per (16 columns, K slice)
    if (S > 1) {
        if (int rc = ensure_kparts(h, (int64_t)S * M * H)) return rc;
        (skinny ? h->skinny_launches : h->splitk_launches) += 1;
        GemmParams p{};
        p.a_hi = a.hi; p.a_lo = a.lo; p.lda = lda; p.amap = amap; p.a_index = a_index;
        p.w = w; p.M = (int)M; p.N = H; p.K = K; p.act = ACT_NONE;
        if (h->nsplit == 3) {
            for (const auto& wp : h->w_planes)
                if (w >= wp.base && w < wp.base + wp.elems) { p.w_lo = w + wp.elems; break; }
            if (!p.w_lo) return h->fail(MMS_ERR_STATE, "proj_ln: weight has no lo plane");
        }
        p.out_kind = OUT_F32; p.c_f32 = kp(h); p.ldc = H; p.cmap = RowMap{0, 0, 0}; p.rmap = RowMap{0, 0, 0};
        p.k_splits = S; p.c_split_stride = (long long)M * H;
        p.m_dev = m_dev;
        if (skinny) p.engine = ENG_SKINNY_PARTS;      // gemm_skinny.hip, K slices dealt to workgroups (same partials as the tile engine's: bit-identical)
        if (h->timing && h->lane) p.flop_counter = h->flop_counter + 3;      // side lane: FLOPs only (slot 3, mms_side_lane_flops) -- a launch that shares the chip has no duration of its own
        if (h->timing && !h->lane) {
            if (h->ev_used + 2 > h->ev.size()) { if (int rc = grow_event_pair(h, h->ev)) return rc; }
            p.flop_counter = h->flop_counter;
            HIP_TRY(h, hipEventRecord(h->ev[h->ev_used], st));
            if (!launch_gemm(p, h->nsplit, st)) return h->fail(MMS_ERR_ARG, "proj_ln: no engine takes this shape");
            HIP_TRY(h, hipEventRecord(h->ev[h->ev_used + 1], st));
            h->ev_cls.resize(h->ev_used / 2 + 1); h->ev_cls[h->ev_used / 2] = 0;
            h->ev_used += 2;
            h->gemm_launches += 1;
        } else if (!launch_gemm(p, h->nsplit, st)) return h->fail(MMS_ERR_ARG, "proj_ln: no engine takes this shape");
        LnResid r;
        r.hi = resid.hi; r.lo = resid.lo; r.ld = H; r.rmap = rmap; r.r_index = r_index;
        r.nparts = S; r.part_stride = p.c_split_stride; r.bias = bias;
        r.o_f8 = out.f8;
        launch_ln_to_planes(kp(h), H, g, b, out.hi, out.lo, H, (int)M, st, m_dev, r);
        return MMS_OK;
    }
    if (int rc = gemm(h, st, a, lda, amap, w, bias, M, H, K, ACT_NONE, to_f32(t, H), &resid, m_dev, a_index, rmap, r_index, cls_bit,
                      skinny_shape(h, M, K) ? splitk_for(h, M, K) : 0)) return rc;
    ln_resid(h, st, t, g, b, out, M, m_dev, resid, rmap, r_index);
    return MMS_OK;
}

// Packed-stream descriptor: per-pair first row / live count (relative to the stream's first row) and the
// device-side number of live rows.  off == nullptr: dense layout (row of (b, s) = b * S + s).
struct Pack { const int* off = nullptr; const int* cnt = nullptr; const int* rows = nullptr;
              const int4* sub = nullptr; const int* n_sub = nullptr;      // sub-tile table of the fused QKV + attention kernel (plan_tiles)
              const int4* sub2 = nullptr;                                // CROSS table (plan_cross_tiles): the sub-tiles' rows in the second stream
              const int* rec = nullptr; };                               // per-pair records of the table

// sub-tile table of one token stream (n pairs of at most S tokens; packed or dense) for qkv_attn.hip, in table slot `slot`
void plan_tiles(mms_handle* h, hipStream_t st, Pack& pk, int64_t n, int S, int slot, bool query_stage = false) {
    if (!h->fuse_attn || (h->nsplit != 2 && h->nsplit != 3) || h->f8 || S > 48) return;
    // att_block() takes the fused kernel for launches of >= FUSED_ATTN_ROWS (1024) padded rows (below, the wide split-K route + the attention kernel are faster; the plan
    // is ~20 us per call); the exact-fp32 attention route (fuse_attention = 1: one work item per pair and 16-query tile) only for pairs of >= 16 tokens -- the split-bf16
    // route works on 16-row tiles of the sub-tile whatever the pairs' lengths (lxmert's 10-token box stream included)
    if (n * S < fused_attn_rows() || (S < 16 && h->fuse_attn != 2)) return;
    if (query_stage) {      // lxmert's distinct-query stage has its own table and scratch: it may be planned on the side lane while the chunk's tables are
        if (!h->lq_qa_sub || !launch_qkv_tile_plan(pk.off, pk.cnt, (int)n, S, h->lq_qa_sub, h->qa_nsub + 3, h->nsplit, st, h->lq_qa_rec, h->lq_qa_jump)) return;
        pk.sub = h->lq_qa_sub; pk.n_sub = h->qa_nsub + 3; pk.rec = h->lq_qa_rec;
        return;
    }
    if (!launch_qkv_tile_plan(pk.off, pk.cnt, (int)n, S, h->qa_sub[slot], h->qa_nsub + slot, h->nsplit, st, h->qa_rec[slot], h->qa_jump)) return;
    pk.sub = h->qa_sub[slot]; pk.n_sub = h->qa_nsub + slot; pk.rec = h->qa_rec[slot];
}
// ... of a PAIR of streams for the fused cross-attention launches of lxmert's X layers (fuse_attention = 2): a sub-tile = the rows of its pairs in both streams
void plan_cross_tiles(mms_handle* h, hipStream_t st, Pack& px, const Pack& p1, const Pack& p2, int64_t n, int S1, int S2) {
    if (h->fuse_attn != 2 || (h->nsplit != 2 && h->nsplit != 3) || h->f8 || !h->qa_sub[2] || n * (S1 + S2) < fused_attn_rows()) return;
    if (!launch_qkv_cross_plan(p1.off, p1.cnt, p2.off, p2.cnt, (int)n, S1, S2, h->qa_sub[2], h->qa_sub[3], h->qa_nsub + 2, h->nsplit, st, h->qa_rec[2], h->qa_jump)) return;
    px.sub = h->qa_sub[2]; px.sub2 = h->qa_sub[3]; px.n_sub = h->qa_nsub + 2; px.rec = h->qa_rec[2];
}

// one fused QKV + attention launch, timed apart from the GEMM launches (its duration includes the attention of its pairs)
int fused_attn_launch(mms_handle* h, hipStream_t st, QkvAttnParams& q) {
    if (h->alternate) { q.reverse = h->flip; h->flip ^= 1; }
    if (h->timing && h->lane) q.flop_counter = h->flop_counter + 3;      // side lane: FLOPs only (slot 3, mms_side_lane_flops) -- a launch that shares the chip has no duration of its own
    if (h->timing && !h->lane) {
        if (h->ev_fused_used + 2 > h->ev_fused.size()) { if (int rc = grow_event_pair(h, h->ev_fused)) return rc; }
        q.flop_counter = h->flop_counter + 1;
        HIP_TRY(h, hipEventRecord(h->ev_fused[h->ev_fused_used], st));
        if (!launch_qkv_attn(q, st)) return h->fail(MMS_ERR_ARG, "qkv_attn: shape not supported");
        HIP_TRY(h, hipEventRecord(h->ev_fused[h->ev_fused_used + 1], st));
        h->ev_fused_used += 2;
        h->fused_timed += 1;
    } else if (!launch_qkv_attn(q, st)) return h->fail(MMS_ERR_ARG, "qkv_attn: shape not supported");
    h->fused_attn_launches += 1;
    return MMS_OK;
}

int attend(mms_handle* h, AttnParams& a, hipStream_t st) {
    if (h->alternate) { a.reverse = h->flip; h->flip ^= 1; }
    if (!launch_attention(a, st))
        return h->fail(MMS_ERR_ARG, "attention: sequence of " + std::to_string(a.Sq) + " x " + std::to_string(a.Sk) + " tokens exceeds the 48-token kernels");
    return MMS_OK;
}

// attention sub-layer: out = LN(dense(attn(in_q, in_kv)) + in_q)    (pixelbert.py:932-966, modeling.py:355-392)
// self-attention over one stream whose first row is row0; S = (maximum) tokens per pair.
int att_block(mms_handle* h, hipStream_t st, const AttW& w, Planes in, Planes out, int64_t row0, int S, int64_t B,
              const float* key_add, const Pack& pk = Pack()) {
    const int64_t M = B * S;
    const bool f8 = h->f8 && in.f8;
    // one kernel for projection + attention (qkv_attn.hip) from FUSED_ATTN_ROWS padded rows on, in two- or three-pass bf16
    // (streams of very short pairs -- lxmert's 10 box tokens -- stay on the two-kernel route: a dozen attention items per sub-tile make the
    // fused epilogue cost more than the attention launch it replaces, profiles/r03q_*)
    const bool fused_attn = pk.sub && w.wqkv_hm && !f8 && M >= fused_attn_rows() && (S >= 16 || h->fuse_attn == 2) && ((h->nsplit == 2 && !(h->x1_mask & 1)) || h->nsplit == 3);
    if (fused_attn) {
        QkvAttnParams q{};
        const Planes a_in = in.at(row0 * H), c_out = h->ctx.at(row0 * H);
        q.a_hi = a_in.hi; q.lda = H; q.w = w.wqkv_hm; q.bias = w.bqkv_hm; q.K = H;
        if (h->nsplit == 3) q.w_lo = w.wqkv_hm + (long long)3 * H * H;       // upload_mat keeps the lo plane right behind the hi plane
        q.sub = pk.sub; q.n_sub = pk.n_sub; q.pair_rec = pk.rec; q.pair_off = pk.off; q.pair_cnt = pk.cnt; q.S = S;
        q.key_add = key_add; q.o_hi = c_out.hi; q.o_lo = c_out.lo; q.ldo = H;
        q.M = (int)M; q.m_dev = pk.rows; q.fast = h->fuse_attn == 2;
        if (int rc = fused_attn_launch(h, st, q)) return rc;
    } else {
    if (f8) {
        if (int rc = gemm_f8(h, st, in.f8 + row0 * H, H, w.wqkv8, w.wqkvs, w.bqkv, M, 3 * H, H, ACT_NONE, to_qkv(h, row0, M), pk.rows)) return rc;
    } else if (int rc = gemm(h, st, in.at(row0 * H), H, ID, w.wqkv, w.bqkv, M, 3 * H, H, ACT_NONE,
                             to_qkv(h, row0, M), nullptr, pk.rows, nullptr, ID, nullptr, 1)) return rc;
    AttnParams a{};
    attn_q(a, h, row0, M); attn_kv(a, h, row0, M);
    a.q_base = 0; a.Sq = S; a.kv_base = 0; a.Sk = S;
    a.key_add = key_add;
    a.o_hi = h->ctx.at(row0 * H).hi; a.o_lo = h->ctx.at(row0 * H).lo; a.ldo = H;
    a.B = (int)B;
    a.q_off = a.kv_off = pk.off; a.q_cnt = a.kv_cnt = pk.cnt;
    if (f8) a.o_f8 = h->ctx.f8 + row0 * H;
    if (int rc = attend(h, a, st)) return rc;
    }
    const Planes resid = in.at(row0 * H);
    bool fused = false;
    if (int rc = gemm_ln(h, st, f8, h->ctx.at(row0 * H), H, w.wo, w.wo8, w.wos, w.bo, M, H, resid, w.g, w.b, out.at(row0 * H), h->t + row0 * H,
                         pk.rows, &fused)) return rc;
    if (fused) return MMS_OK;
    if (f8) {
        if (int rc = gemm_f8(h, st, h->ctx.f8 + row0 * H, H, w.wo8, w.wos, w.bo, M, H, H, ACT_NONE, to_f32(h->t + row0 * H, H), pk.rows)) return rc;
        ln_resid(h, st, h->t + row0 * H, w.g, w.b, out.at(row0 * H), M, pk.rows, resid);
        return MMS_OK;
    }
    return proj_ln(h, st, h->ctx.at(row0 * H), H, ID, nullptr, w.wo, w.bo, M, H, resid, ID, nullptr, w.g, w.b, out.at(row0 * H), h->t + row0 * H,
                   pk.rows, 2);
}

#ifdef MMS_LAB
// lab experiment (MMS_FFN_BLOCK=rows): FFN-up / FFN-down in row blocks that reuse ONE intermediate buffer of `rows` x inter, so the
// 12 KB / row intermediate could stay inside the 256 MB Infinity Cache instead of streaming through HBM
__global__ void k_block_counts(const int* rows, int blk, int nb, int* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nb) { const long long r = (long long)*rows - (long long)i * blk; out[i] = r < 0 ? 0 : r > blk ? blk : (int)r; }
}
static int* g_blk_counts = nullptr;
#endif

// feed-forward sub-layer: out = LN(dense(act(dense(in))) + in)      (pixelbert.py:969-985, modeling.py:395-420)
int ffn_block(mms_handle* h, hipStream_t st, const FfnW& w, Planes in, Planes out, int64_t row0, int64_t M, int act,
              const Pack& pk = Pack()) {
    const int I = h->cfg.inter;
    const Planes resid = in.at(row0 * H);
    bool fused = false;
    if (h->f8 && in.f8) {
        if (int rc = gemm_f8(h, st, in.f8 + row0 * H, H, w.wi8, w.wis, w.bi, M, I, H, act, to_planes(h->mid, I), pk.rows)) return rc;
        if (int rc = gemm_ln(h, st, true, h->mid, I, w.wd, w.wd8, w.wds, w.bd, M, I, resid, w.g, w.b, out.at(row0 * H), h->t + row0 * H, pk.rows, &fused)) return rc;
        if (fused) return MMS_OK;
        if (int rc = gemm_f8(h, st, h->mid.f8, I, w.wd8, w.wds, w.bd, M, H, I, ACT_NONE, to_f32(h->t + row0 * H, H), pk.rows)) return rc;
        ln_resid(h, st, h->t + row0 * H, w.g, w.b, out.at(row0 * H), M, pk.rows, resid);
        return MMS_OK;
    }
#ifdef MMS_LAB
    static const long blk = getenv("MMS_FFN_BLOCK") ? atol(getenv("MMS_FFN_BLOCK")) : 0;
    static const long blk_live = getenv("MMS_FFN_LIVE") ? atol(getenv("MMS_FFN_LIVE")) : 0;     // host-side hint: skip blocks past this row
    if (blk > 0 && M > blk && !h->fuse_ln) {
        const int nb = (int)((M + blk - 1) / blk);
        if (!g_blk_counts) HIP_TRY(h, hipMalloc((void**)&g_blk_counts, 4096 * 4));
        if (pk.rows) k_block_counts<<<(nb + 255) / 256, 256, 0, st>>>(pk.rows, (int)blk, nb, g_blk_counts);
        for (int i = 0; i < nb; ++i) {
            const int64_t r0 = (int64_t)i * blk, Mi = std::min<int64_t>(blk, M - r0);
            if (blk_live && r0 >= blk_live) break;
            const int* md = pk.rows ? g_blk_counts + i : nullptr;
            const Planes rs = in.at((row0 + r0) * H);
            if (int rc = gemm(h, st, in.at((row0 + r0) * H), H, ID, w.wi, w.bi, Mi, I, H, act, to_planes(h->mid, I), nullptr, md, nullptr, ID, nullptr, 4)) return rc;
            if (int rc = gemm(h, st, h->mid, I, ID, w.wd, w.bd, Mi, H, I, ACT_NONE, to_f32(h->t + (row0 + r0) * H, H), &rs, md, nullptr, ID, nullptr, 8)) return rc;
        }
        ln_resid(h, st, h->t + row0 * H, w.g, w.b, out.at(row0 * H), M, pk.rows, resid);
        return MMS_OK;
    }
#endif
    const Planes mid = midp(h);
    if (int rc = gemm(h, st, in.at(row0 * H), H, ID, w.wi, w.bi, M, I, H, act, to_planes(mid, I), nullptr, pk.rows, nullptr, ID, nullptr, 4)) return rc;
    if (int rc = gemm_ln(h, st, false, mid, I, w.wd, w.wd8, w.wds, w.bd, M, I, resid, w.g, w.b, out.at(row0 * H), h->t + row0 * H, pk.rows, &fused)) return rc;
    if (fused) return MMS_OK;
    return proj_ln(h, st, mid, I, ID, nullptr, w.wd, w.bd, M, I, resid, ID, nullptr, w.g, w.b, out.at(row0 * H), h->t + row0 * H, pk.rows, 8);
}

// Last self-attention + FFN block before the pooler: only the CLS row feeds the pooler (pixelbert.py:258-266,
// modeling.py:602-608), so K/V are projected for every live row of the stream but Q, attention, the attention-output
// dense, both LayerNorms and the FFN run on the n CLS rows only.  `in` holds the block input (stream rows start at
// row 0), `tmp` is the other hidden-state buffer; the result (compact [n,768]) is left in `in` rows 0..n.
int last_block_cls(mms_handle* h, hipStream_t st, const AttW& att, const FfnW& ffn, int act, Planes in, Planes tmp, int S,
                   int64_t n, const float* key_add, const Pack& pk) {
    const int64_t M = n * S;
    const RowMap cls = RowMap{1, S, 0};                 // dense: CLS of pair b is row b*S; packed: row pk.off[b]
    // K,V for every live row -> columns [768, 2304) of the qkv buffer
    if (int rc = gemm(h, st, in, H, ID, att.wqkv + (long long)H * H, att.bqkv + H, M, 2 * H, H, ACT_NONE,
                      to_qkv(h, 0, M, H), nullptr, pk.rows)) return rc;
    // Q for the CLS rows only -> compact fp32 [n,768] (the pooled buffer is idle until the pooler)
    if (int rc = gemm(h, st, in, H, cls, att.wqkv, att.bqkv, n, H, H, ACT_NONE, to_f32(h->pooled, H), nullptr, nullptr, pk.off)) return rc;
    AttnParams a{};
    a.q = h->pooled; a.ldq = H; a.hs_q = MMS_HEAD_DIM; a.q_stride = 1; a.Sq = 1; a.o_compact = 1;      // compact fp32 [n,768] CLS queries
    attn_kv(a, h, 0, M); a.Sk = S;
    a.key_add = key_add; a.kv_off = pk.off; a.kv_cnt = pk.cnt;
    a.o_hi = h->ctx.hi; a.o_lo = h->ctx.lo; a.ldo = H; a.B = (int)n;
    if (int rc = attend(h, a, st)) return rc;
    if (int rc = proj_ln(h, st, h->ctx, H, ID, nullptr, att.wo, att.bo, n, H, in, cls, pk.off, att.g, att.b, tmp, h->t, nullptr, 0)) return rc;
    const int I = h->cfg.inter;
    if (int rc = gemm(h, st, tmp, H, ID, ffn.wi, ffn.bi, n, I, H, act, to_planes(h->mid, I))) return rc;
    return proj_ln(h, st, h->mid, I, ID, nullptr, ffn.wd, ffn.bd, n, I, tmp, ID, nullptr, ffn.g, ffn.b, in, h->t, nullptr, 0);
}

int check_ready(mms_handle* h, int model, const void* batch, const float* logits) {
    if (!h) return MMS_ERR_ARG;
    if (!h->finalized) return h->fail(MMS_ERR_STATE, "mms_finalize has not been called");
    if (h->cfg.model != model) return h->fail(MMS_ERR_ARG, "handle was created for a different model");
    if (!batch) return h->fail(MMS_ERR_ARG, "null batch pointer");
    if (!logits && *(const int64_t*)batch != 0) return h->fail(MMS_ERR_ARG, "null logits pointer");  // n_pairs is the first field
    return MMS_OK;
}

int post_launch(mms_handle* h) {
    HIP_TRY(h, hipGetLastError());
    return MMS_OK;
}

// Device-side guard: restores the caller's current device when an ABI call returns (the handle may live on another GPU).
struct DeviceScope {
    int prev = -1;
    explicit DeviceScope(int dev) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; if (prev != dev) (void)hipSetDevice(dev); else prev = -1; }
    ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// ------------------------------------------------------------------------------------------------
// label-text tuples: dense [B*10, 8] ids -> distinct tuples + per-row index (batchops.hip)
// ------------------------------------------------------------------------------------------------
int ensure_dedup_ws(mms_handle* h, int64_t rows) {
    if (rows <= h->dd_rows) return MMS_OK;
    free_pool(h->dd_allocs);
    h->dd_rows = 0;
    int cap = 1024;
    while (cap < 2 * rows) cap <<= 1;
    void* p;
    if (int rc = dev_alloc(h, h->dd_allocs, &p, (size_t)cap * 4)) return rc;
    h->dd_slots = (int*)p;
    if (int rc = dev_alloc(h, h->dd_allocs, &p, (size_t)rows * 4)) return rc;
    h->dd_rep = (int*)p;
    if (int rc = dev_alloc(h, h->dd_allocs, &p, (size_t)rows * 4)) return rc;
    h->dd_uid = (int*)p;
    if (int rc = dev_alloc(h, h->dd_allocs, &p, (size_t)rows * 4)) return rc;
    h->dd_index = (int*)p;
    if (int rc = dev_alloc(h, h->dd_allocs, &p, 16)) return rc;
    h->dd_counter = (int*)p;
    if (int rc = dev_alloc(h, h->dd_allocs, &p, (size_t)rows * MMS_LABEL_LEN * 4)) return rc;
    h->dd_uniq32 = (int32_t*)p;
    if (int rc = dev_alloc(h, h->dd_allocs, &p, (size_t)rows * MMS_LABEL_LEN * 8)) return rc;
    h->dd_uniq64 = (int64_t*)p;
    h->dd_cap = cap;
    h->dd_rows = rows;
    return MMS_OK;
}

// Finds the distinct tuples of ids[rows][8] (T = int32_t: zk feed, int64_t: lxmert feed): h->dd_uniq32 / dd_uniq64 hold them,
// h->dd_index the row -> tuple index.  *U needs the count on the host: one 4-byte read + stream sync (the dense feed's price;
// callers that pass uniq_label_ids / label_index themselves stay fully asynchronous).
// async = true: no read-back -- *U is set to the upper bound `rows`, the count stays on the device (h->dd_counter[0]; zk small calls)
template <typename T>
int dedup_labels(mms_handle* h, hipStream_t st, const T* ids, int64_t rows, int64_t* U, bool async = false) {
    if (rows > (int64_t)1 << 30) return h->fail(MMS_ERR_ARG, "too many label tuples in one call");
    if (int rc = ensure_dedup_ws(h, rows)) return rc;
    if constexpr (sizeof(T) == 4)
        launch_label_dedup_i32((const int32_t*)ids, (int)rows, h->dd_slots, h->dd_cap, h->dd_rep, h->dd_uid, h->dd_counter, h->dd_uniq32, h->dd_uniq64, h->dd_index, st);
    else
        launch_label_dedup_i64((const int64_t*)ids, (int)rows, h->dd_slots, h->dd_cap, h->dd_rep, h->dd_uid, h->dd_counter, h->dd_uniq32, h->dd_uniq64, h->dd_index, st);
    if (async) { *U = rows; return MMS_OK; }
    int n = 0;
    HIP_TRY(h, hipMemcpyAsync(&n, h->dd_counter, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipStreamSynchronize(st));
    *U = n;
    return MMS_OK;
}

// ------------------------------------------------------------------------------------------------
// zk forward
// ------------------------------------------------------------------------------------------------
// U_dev != nullptr (small calls, U <= LAB_CHUNK): U is only an upper bound, the number of distinct tuples is on the device -- the kernels
// stop at it and the call never waits for the GPU (the dense-feed read-back of the large calls costs nothing there, but it kept every
// 1-pair call of the reference's zk driver from being asynchronous)
int zk_label_features(mms_handle* h, hipStream_t st, const int32_t* uniq_ids, int64_t U, const int* U_dev = nullptr) {
    if (int rc = ensure_label_ws(h, U)) return rc;
    h->n_labels = U;
    const int KD = MMS_LABEL_LEN * H;
    if (U_dev) {
        launch_zk_im2col(h->E, uniq_ids, (int)U, h->cfg.vocab, h->lab_planes.hi, h->lab_planes.lo, st, U_dev);
        launch_scale_count(U_dev, MMS_LABEL_LEN, h->dd_counter + 1, st);
        if (int rc = gemm(h, st, h->lab_planes, KD, ID, h->w_conv1, h->b_conv1, U * MMS_LABEL_LEN, H, KD, ACT_RELU, to_f32(h->lab_f32, H), nullptr,
                          h->dd_counter + 1)) return rc;
        launch_mean8(h->lab_f32, h->lab_feat, (int)U, st, U_dev);
        return MMS_OK;
    }
    for (int64_t u0 = 0; u0 < U; u0 += h->lab_cap) {
        const int64_t n = (U - u0) < h->lab_cap ? (U - u0) : h->lab_cap;
        launch_zk_im2col(h->E, uniq_ids + u0 * MMS_LABEL_LEN, (int)n, h->cfg.vocab, h->lab_planes.hi, h->lab_planes.lo, st);
        if (int rc = gemm(h, st, h->lab_planes, KD, ID, h->w_conv1, h->b_conv1, n * MMS_LABEL_LEN, H, KD, ACT_RELU,
                          to_f32(h->lab_f32, H))) return rc;
        launch_mean8(h->lab_f32, h->lab_feat + u0 * H, (int)n, st);
    }
    return MMS_OK;
}

// image tokens of pairs [p0, p0+n) (model_triple.py:189-195, pixelbert.py:449-452) -> fp32 [n*10,768] at h->qkv + n*10*768 (the
// QKV buffer is idle until the encoder starts).  The split box features sit in the (still unused) FFN buffer unless the caller
// already holds them (featp_shared: the fused three-model entry point splits them once for all members).
// compact = true (packed single-model path): only the LIVE boxes are split / projected / combined; tok then holds them consecutively, pair b's at
// row box_off[b] (h->pk_off[1]); their number is on the device (h->pk_rows + 1).
int zk_image_tokens(mms_handle* h, hipStream_t st, const mms_zk_batch* b, const int32_t* label_index, int64_t p0, int64_t n,
                    const Planes* featp_shared = nullptr, bool compact = false) {
    const int64_t NB = n * MMS_NBOX;
    Planes featp = h->mid;
    const int* box_idx = nullptr;
    const int* box_rows = nullptr;
    h->zk_plans_merged = false;
    if (compact) {      // stream 1's plan buffers are idle in a zk handle
        // small launch waves: the token plan of zk_encode() rides along (one launch instead of six; nothing in between touches its tables)
        h->zk_plans_merged = launch_zk_plans_small(b->len_query + p0, b->num_boxes + p0, h->cfg.text_len, (int)n, h->pk_cnt[1], h->pk_off[1], h->pk_src[1],
                                                   h->pk_cnt[0], h->pk_off[0], h->pk_src[0], h->key_add, h->pk_rows, st);
        if (!h->zk_plans_merged)
            launch_zk_box_plan(b->len_query + p0, b->num_boxes + p0, h->cfg.text_len, (int)n, h->pk_cnt[1], h->pk_off[1], h->pk_src[1], h->pk_rows + 1, st);
        box_idx = h->pk_src[1]; box_rows = h->pk_rows + 1;
        launch_split_f32_rows(b->feats + p0 * MMS_NBOX * MMS_FEAT, box_idx, box_rows, (int)NB, MMS_FEAT, featp.hi, featp.lo, st);
    } else if (featp_shared) featp = *featp_shared;
    else launch_split_f32(b->feats + p0 * MMS_NBOX * MMS_FEAT, featp.hi, featp.lo, NB * MMS_FEAT, st);
    float* img = h->qkv;                 // [NB,768] fp32
    float* tok = h->qkv + NB * H;        // [NB,768] fp32
    if (int rc = gemm(h, st, featp, MMS_FEAT, ID, h->w_conv2, h->b_conv2, NB, H, MMS_FEAT, ACT_RELU, to_f32(img, H), nullptr, box_rows)) return rc;
    launch_zk_tokpre(h->lab_feat, label_index + p0 * MMS_NBOX, (int)h->n_labels, b->boxes_5 + p0 * MMS_NBOX * 5, h->w_dense1, h->b_dense1,
                     img, h->ctx.hi, h->ctx.lo, (int)NB, st, box_idx, box_rows);
    return gemm(h, st, h->ctx, H, ID, h->w_femb, h->b_femb, NB, H, H, ACT_NONE, to_f32(tok, H), nullptr, box_rows);
}

// embeddings + encoder + pooler + AM-softmax head of pairs [p0, p0+n) on the image tokens `tok` [n*10,768]
int zk_encode(mms_handle* h, hipStream_t st, const mms_zk_batch* b, int64_t p0, int64_t n, const float* tok, float* logits, float* probs,
              const int* box_off = nullptr) {
    const mms_config& c = h->cfg;
    const int T = c.text_len, S = T + MMS_NBOX;
    // --- embeddings + mask (packed: live tokens only, laid out contiguously) ---
    Pack pk;
    if (c.pack_tokens) {
        if (!h->zk_plans_merged)
            launch_zk_pack_plan(b->len_query + p0, b->num_boxes + p0, T, (int)n, h->pk_off[0], h->pk_cnt[0], h->pk_src[0], h->key_add,
                                h->pk_rows, st);
        h->zk_plans_merged = false;
        pk.off = h->pk_off[0]; pk.cnt = h->pk_cnt[0]; pk.rows = h->pk_rows;
        launch_zk_embed_packed(h->E, h->type_tab, h->pos_tab, h->emb_g, h->emb_b, b->query_ids + p0 * T, b->segment_ids + p0 * S, tok,
                               T, c.vocab, h->pk_src[0], h->pk_rows, (int)(n * S), h->x.hi, h->x.lo, st, box_off);
    } else {
        launch_zk_embed(h->E, h->type_tab, h->pos_tab, h->emb_g, h->emb_b, b->query_ids + p0 * T, b->segment_ids + p0 * S, tok, T,
                        c.vocab, h->x.hi, h->x.lo, (int)n, st);
        launch_zk_mask(b->len_query + p0, b->num_boxes + p0, T, h->key_add, (int)n, st);
    }
    if (h->f8) launch_planes_to_f8(h->x.hi, h->x.lo, h->x.f8, n * S * H, st);   // layer 0's A operand (rows past the live count are never read)
    plan_tiles(h, st, pk, n, S, 0);
    // --- encoder ---
    const int nl = (c.stop_after >= 0 && c.stop_after < c.layers) ? c.stop_after : c.layers;
    const bool cls_only = c.stop_after < 0 && c.layers > 0;   // debug runs keep the full hidden state
    for (int i = 0; i < nl; ++i) {
        if (cls_only && i == nl - 1) {
            if (int rc = last_block_cls(h, st, h->layers[i].att, h->layers[i].ffn, ACT_GELU_TANH, h->x, h->y, S, n, h->key_add, pk)) return rc;
            break;
        }
        if (int rc = att_block(h, st, h->layers[i].att, h->x, h->y, 0, S, n, h->key_add, pk)) return rc;
        if (int rc = ffn_block(h, st, h->layers[i].ffn, h->y, h->x, 0, n * S, ACT_GELU_TANH, pk)) return rc;
    }
    // --- pooler on the CLS rows + AM-softmax head ---
    if (cls_only) {
        if (int rc = gemm(h, st, h->x, H, ID, h->w_pool, h->b_pool, n, H, H, ACT_TANH, to_f32(h->pooled, H))) return rc;
    } else if (int rc = gemm(h, st, h->x, H, RowMap{1, S, 0}, h->w_pool, h->b_pool, n, H, H, ACT_TANH, to_f32(h->pooled, H), nullptr,
                             nullptr, pk.off)) return rc;
    launch_zk_head(h->pooled, h->am_kernel, b->labels + p0, 30.0f, 0.35f, logits + p0 * 2, probs ? probs + p0 * 2 : nullptr, (int)n, st);
    return MMS_OK;
}

int zk_chunk(mms_handle* h, hipStream_t st, const mms_zk_batch* b, const int32_t* label_index, int64_t p0, int64_t n, float* logits, float* probs) {
    bool compact = h->cfg.pack_tokens != 0;
#ifdef MMS_LAB
    static const bool no_compact = getenv("MMS_NO_BOX_COMPACT") != nullptr;      // A/B: image-token stage on all 10 box rows of every pair
    if (no_compact) compact = false;
#endif
    if (int rc = zk_image_tokens(h, st, b, label_index, p0, n, nullptr, compact)) return rc;
    return zk_encode(h, st, b, p0, n, h->qkv + n * MMS_NBOX * H, logits, probs, compact ? h->pk_off[1] : nullptr);
}

int lds_chunk(mms_handle* h, hipStream_t st, const mms_lds_batch* b, int64_t p0, int64_t n, float* logits, float* probs,
              const Planes* featp_shared = nullptr) {
    const mms_config& c = h->cfg;
    const int T = c.text_len, S = T + 2 * MMS_NBOX;
    const int64_t NB = n * MMS_NBOX;
    // embedding stage in the dense [n, 40] layout; packed mode builds it in the idle y buffer and then keeps one representative of
    // every group of identical feature / label token rows (rowops.hip k_lds_plan_*: duplicates ride as log(multiplicity) on the key)
    Planes emb = c.pack_tokens ? h->y : h->x;
    launch_lds_embed_text(h->E, h->type_tab, h->pos_tab, h->emb_g, h->emb_b, b->input_ids + p0 * T, b->segment_ids + p0 * T, T, S,
                          c.vocab, emb.hi, emb.lo, (int)n, st);
    Planes featp = h->mid;
    if (featp_shared) featp = *featp_shared;
    else launch_split_f32(b->features + p0 * MMS_NBOX * MMS_FEAT, featp.hi, featp.lo, NB * MMS_FEAT, st);
    // featureemb (linear) written straight into rows b*S + T + n of the hidden state (pixelmodel.py:600-601)
    if (int rc = gemm(h, st, featp, MMS_FEAT, ID, h->w_feat, h->b_feat, NB, H, MMS_FEAT, ACT_NONE,
                      to_planes(emb, H, RowMap{MMS_NBOX, S, T}))) return rc;
    launch_lds_label(h->E, h->w_lab8, b->labelfeat + p0 * MMS_NBOX * MMS_LABEL_LEN, c.vocab, S, T + MMS_NBOX, emb.hi, emb.lo, (int)n, st);
    Pack pk;
    const float* key_add = nullptr;
    if (c.pack_tokens) {
        launch_lds_pack_plan(b->features + p0 * MMS_NBOX * MMS_FEAT, b->labelfeat + p0 * MMS_NBOX * MMS_LABEL_LEN, T, (int)n, h->pk_src[1],
                             h->pk_off[0], h->pk_cnt[0], h->pk_src[0], h->key_add, h->pk_rows, st);
        launch_rows_pick(emb.hi, emb.lo, h->pk_src[0], h->pk_rows, (int)(n * S), h->x.hi, h->x.lo, st);
        pk.off = h->pk_off[0]; pk.cnt = h->pk_cnt[0]; pk.rows = h->pk_rows;
        key_add = h->key_add;
    }
    plan_tiles(h, st, pk, n, S, 0);
    if (h->f8) launch_planes_to_f8(h->x.hi, h->x.lo, h->x.f8, n * S * H, st);
    const int nl = (c.stop_after >= 0 && c.stop_after < c.layers) ? c.stop_after : c.layers;
    const bool cls_only = c.stop_after < 0 && c.layers > 0;
    for (int i = 0; i < nl; ++i) {
        if (cls_only && i == nl - 1) {
            if (int rc = last_block_cls(h, st, h->layers[i].att, h->layers[i].ffn, ACT_GELU_TANH, h->x, h->y, S, n, key_add, pk)) return rc;
            break;
        }
        if (int rc = att_block(h, st, h->layers[i].att, h->x, h->y, 0, S, n, key_add, pk)) return rc;
        if (int rc = ffn_block(h, st, h->layers[i].ffn, h->y, h->x, 0, n * S, ACT_GELU_TANH, pk)) return rc;
    }
    if (int rc = gemm(h, st, h->x, H, cls_only ? ID : RowMap{1, S, 0}, h->w_pool, h->b_pool, n, H, H, ACT_TANH, to_f32(h->pooled, H), nullptr,
                      nullptr, cls_only ? nullptr : pk.off)) return rc;
    launch_lds_head(h->pooled, h->w_cls, h->b_cls, logits + p0 * 2, probs ? probs + p0 * 2 : nullptr, (int)n, st);
    return MMS_OK;
}

int lx_label_features(mms_handle* h, hipStream_t st, const int64_t* uniq_ids, int64_t U) {
    if (int rc = ensure_label_ws(h, U)) return rc;
    h->n_labels = U;
    for (int64_t u0 = 0; u0 < U; u0 += h->lab_cap) {
        const int64_t n = (U - u0) < h->lab_cap ? (U - u0) : h->lab_cap;
        launch_lx_label_emb(h->E, h->pos_tab, h->type_tab, h->emb_g, h->emb_b, h->w_lconv, h->b_lconv, uniq_ids + u0 * MMS_LABEL_LEN,
                            h->cfg.vocab, h->lab_planes.hi, h->lab_planes.lo, (int)n, st);
        if (int rc = gemm(h, st, h->lab_planes, H, ID, h->w_labfc, h->b_labfc, n, H, H, ACT_NONE, to_f32(h->lab_f32, H))) return rc;
        launch_ln_f32(h->lab_f32, h->g_lab, h->be_lab, h->lab_feat + u0 * H, (int)n, st);
    }
    return MMS_OK;
}

// Distinct-query stage (packed mode only).  In modeling.py:568-593 the language stream passes its l_layers BertLayers before it
// ever meets the image, so their output depends on (input_ids, input_mask) alone -- and a query arrives with its whole candidate
// set (8..30 pairs, run_pretraining_predict_score.py:566 / kdd_data.py).  The reference recomputes it per pair; here the distinct
// rows are found on the device (batchops.hip), embedded and run through the l_layers ONCE, stored dense by (query, token), and
// lx_chunk copies each pair's live language rows from the store.  Per-row arithmetic is unchanged (same kernels, same weights).
// Skipped (lq_active = false -> the per-pair path) when fewer than half of the pairs share their query with another pair.
// two launch lanes (mms_handle::side): fork = the side stream waits for everything enqueued on st so far; join = st waits for the side chain
int lanes_init(mms_handle* h) {
    if (!h->side) HIP_TRY(h, hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking));      // (non-blocking: ordered against the caller's stream -- the legacy default stream included -- by the two events alone)
    if (!h->ev_fork) HIP_TRY(h, hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
    if (!h->ev_join) HIP_TRY(h, hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
    return MMS_OK;
}
int lane_fork(mms_handle* h, hipStream_t st) {
    h->lane_forks += 1;
    HIP_TRY(h, hipEventRecord(h->ev_fork, st));
    HIP_TRY(h, hipStreamWaitEvent(h->side, h->ev_fork, 0));
    return MMS_OK;
}
int lane_join(mms_handle* h, hipStream_t st) {
    HIP_TRY(h, hipEventRecord(h->ev_join, h->side));
    HIP_TRY(h, hipStreamWaitEvent(st, h->ev_join, 0));
    return MMS_OK;
}
// enqueues `chain` on the side lane with the lane-private buffers selected
template <class F>
int on_side_lane(mms_handle* h, F&& chain) {
    h->lane = 1;
    const int rc = chain(h->side);
    h->lane = 0;
    return rc;
}

// A scoring call that may fork a side lane owns one of these: an error return between a fork and its join would leave the side stream running unordered against the
// caller's stream while it still touches x / y / ctx / mid / kparts_side, and lq_join_pending set until the next call (which would silently disable the fused LayerNorm
// epilogue).  Every exit that is not the successful one drains the side stream(s) and clears the lane state.
struct LaneGuard {
    mms_handle* hs[3] = {nullptr, nullptr, nullptr};
    int n = 0;
    bool ok = false;
    explicit LaneGuard(mms_handle* a, mms_handle* b = nullptr, mms_handle* c = nullptr) { hs[n++] = a; if (b) hs[n++] = b; if (c) hs[n++] = c; }
    int done(int rc) { ok = rc == MMS_OK; return rc; }
    ~LaneGuard() {
        if (ok) return;
        for (int i = 0; i < n; ++i) {
            mms_handle* h = hs[i];
            if (h->side) (void)hipStreamSynchronize(h->side);
            h->lane = 0; h->lq_join_pending = false; h->lanes_on = false; h->lanes_q = false;
        }
    }
};

int lx_query_stage(mms_handle* h, hipStream_t st, const mms_lxmert_batch* b, int64_t B, int cs) {
    const mms_config& c = h->cfg;
    h->lq_active = false;
    h->lq_join_pending = false;
    if (!c.pack_tokens || c.stop_after >= 0 || c.layers <= 0 || B < 2) return MMS_OK;
    const int T = c.text_len;
    if (B > h->lq_pairs) {
        free_pool(h->lq_allocs);
        free_pool(h->lq_sub_allocs);
        h->lq_pairs = 0; h->lq_store_q = 0; h->lq_sub = 0;
        int cap = 1024;
        while (cap < 2 * B) cap <<= 1;
        void* p;
        auto get = [&](size_t bytes, void** out) { int rc = dev_alloc(h, h->lq_allocs, &p, bytes); *out = p; return rc; };
        if (int rc = get((size_t)cap * 2 * 4, (void**)&h->lq_slots)) return rc;      // hash table + per-slot smallest row (launch_query_dedup)
        if (int rc = get((size_t)B * 4, (void**)&h->lq_rep)) return rc;
        if (int rc = get((size_t)B * 4, (void**)&h->lq_uid)) return rc;
        if (int rc = get((size_t)B * 4, (void**)&h->lq_rows_of)) return rc;
        if (int rc = get((size_t)B * 4, (void**)&h->lq_index)) return rc;
        if (int rc = get(16, (void**)&h->lq_counter)) return rc;
        h->lq_cap = cap;
        h->lq_pairs = B;
    }
    launch_query_dedup(b->input_ids, b->input_mask, T, (int)B, h->lq_slots, h->lq_cap, h->lq_rep, h->lq_uid, h->lq_counter, h->lq_rows_of,
                       h->lq_index, st);
    int nq = 0;
    HIP_TRY(h, hipMemcpyAsync(&nq, h->lq_counter, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipStreamSynchronize(st));
    const int64_t Q = nq;
    if (Q * 2 > B) return MMS_OK;                       // little sharing: not worth the extra copy
    if (Q > h->lq_store_q) {                            // store planes (dense [Q, T] rows) live outside the chunk workspace
        if (h->lq_store.hi) { (void)hipFree(h->lq_store.hi); h->lq_store.hi = nullptr; }
        void* p = nullptr;
        HIP_TRY(h, hipMalloc(&p, (size_t)Q * T * H * 4));
        h->lq_store.hi = (bf16*)p; h->lq_store.lo = h->lq_store.hi + MMS_PLANE_LO;
        h->lq_store_q = Q;
    }
    if (cs > h->lq_sub) {                               // gathered inputs of one sub-batch of distinct queries
        void* p;
        free_pool(h->lq_sub_allocs);                    // hipFree waits for the work that may still read the old pair
        if (int rc = dev_alloc(h, h->lq_sub_allocs, &p, (size_t)cs * T * 8)) return rc;
        h->lq_ids = (int64_t*)p;
        if (int rc = dev_alloc(h, h->lq_sub_allocs, &p, (size_t)cs * T * 8)) return rc;
        h->lq_mask = (int64_t*)p;
        if (int rc = dev_alloc(h, h->lq_sub_allocs, &p, (size_t)cs * 4)) return rc;
        h->lq_pk_off = (int*)p;
        if (int rc = dev_alloc(h, h->lq_sub_allocs, &p, (size_t)cs * 4)) return rc;
        h->lq_pk_cnt = (int*)p;
        if (int rc = dev_alloc(h, h->lq_sub_allocs, &p, (size_t)cs * T * 4)) return rc;
        h->lq_pk_src = (int*)p;
        if (int rc = dev_alloc(h, h->lq_sub_allocs, &p, (size_t)cs * T * 4)) return rc;
        h->lq_key_add = (float*)p;
        if (int rc = dev_alloc(h, h->lq_sub_allocs, &p, (size_t)(cs + 2) * sizeof(int4))) return rc;
        h->lq_qa_sub = (int4*)p;
        if (int rc = dev_alloc(h, h->lq_sub_allocs, &p, (size_t)(cs + 2) * 4)) return rc;
        h->lq_qa_rec = (int*)p;
        if (int rc = dev_alloc(h, h->lq_sub_allocs, &p, (size_t)qkv_plan_scratch_ints(cs) * 4)) return rc;
        h->lq_qa_jump = (int*)p;
        h->lq_sub = cs;
    }
    // One chunk, one sub-batch, a small call (two lanes): the stage runs on the side lane beside the chunk's box-stream prologue and r_layers (lx_chunk joins
    // before it copies the language rows out of lq_store).  The stage's rows [0, Q T) of x / y / ctx / qkv / t lie inside the chunk's language rows, which the
    // chunk does not touch before the join; its packed plan, its sub-tile table and the plan scratch are its own.
    // (mid: the main lane needs the box rows' share until the join -- split features, then the r_layers' FFN intermediate -- and the stage's rows follow it)
    const bool side = h->lanes_q && B <= cs && Q <= cs && (B * MMS_NBOX + Q * T) * (int64_t)c.inter <= h->mid_elems;
    hipStream_t qs = st;
    if (side) {
        if (int rc = lanes_init(h)) return rc;
        h->lane = 1;
        const int rc = ensure_kparts(h, kparts_side_need(h, B));      // (a no-op: sized with the workspace)
        h->lane = 0;
        if (rc) return rc;
        h->mid_side_off = B * MMS_NBOX * (int64_t)c.inter;
        if (int rc2 = lane_fork(h, st)) return rc2;
        qs = h->side;
        h->lane = 1;
    }
    int* const q_rows = h->pk_rows + 2;
    int rc_stage = MMS_OK;
    for (int64_t u0 = 0; u0 < Q && !rc_stage; u0 += cs) {
        const int64_t n = (Q - u0) < cs ? (Q - u0) : cs;
        launch_gather_i64_rows(b->input_ids, h->lq_rows_of + u0, T, n, h->lq_ids, qs);
        launch_gather_i64_rows(b->input_mask, h->lq_rows_of + u0, T, n, h->lq_mask, qs);
        launch_lx_pack_plan(h->lq_mask, nullptr, T, (int)n, h->lq_pk_off, h->lq_pk_cnt, h->lq_pk_src, h->lq_key_add, q_rows, nullptr, nullptr,
                            nullptr, nullptr, nullptr, qs);
        Pack pl;
        pl.off = h->lq_pk_off; pl.cnt = h->lq_pk_cnt; pl.rows = q_rows;
        launch_lx_embed_lang_packed(h->E, h->pos_tab, h->type_tab, h->emb_g, h->emb_b, h->lq_ids, T, c.vocab, h->lq_pk_src, q_rows,
                                    (int)(n * T), h->x.hi, h->x.lo, qs);
        plan_tiles(h, qs, pl, n, T, 0, true);
        if (h->f8) launch_planes_to_f8(h->x.hi, h->x.lo, h->x.f8, n * T * H, qs);
        for (int i = 0; i < c.layers && !rc_stage; ++i) {
            rc_stage = att_block(h, qs, h->layers[i].att, h->x, h->y, 0, T, n, h->lq_key_add, pl);
            if (!rc_stage) rc_stage = ffn_block(h, qs, h->layers[i].ffn, h->y, h->x, 0, n * T, ACT_GELU_ERF, pl);
        }
        // packed row r holds token pk_src[r] = u_local * T + t  ->  store row (u0 + u_local) * T + t
        launch_rows_scatter(h->x.hi, h->x.lo, h->lq_pk_src, q_rows, (int)(n * T), h->lq_store.at(u0 * T * H).hi, h->lq_store.at(u0 * T * H).lo, qs);
    }
    h->lane = 0;
    if (side) h->lq_join_pending = true;
    if (rc_stage) return rc_stage;
    h->lq_active = true;
    return MMS_OK;
}

int lx_chunk(mms_handle* h, hipStream_t st, const mms_lxmert_batch* b, const int32_t* label_index, int64_t p0, int64_t n, float* logits,
             float* probs, const Planes* featp_shared = nullptr) {
    const mms_config& c = h->cfg;
    const int T = c.text_len, V = MMS_NBOX;
    // language rows start at 0 (at most ML of them), vision rows start at ML (at most MV of them)
    const int64_t ML = n * T, MV = n * V, R = ML + MV;
    float* lang_add = h->key_add;
    float* visn_add = h->key_add2;
    Pack pl, pv;
    Planes featp = h->mid;
    float* xf = h->lq_join_pending ? h->t + ML * H : h->qkv;      // (the distinct-query stage on the side lane owns the first rows of qkv until the join)
    // packed single-model path: the vision plan comes first and visn_fc runs on the LIVE boxes only (features split by gather, projection with the
    // device-side row count): the masked boxes -- 62 % of the rows on the bench batch -- cannot reach a logit
    bool compact = c.pack_tokens && !featp_shared;
#ifdef MMS_LAB
    { static const bool no_compact = getenv("MMS_NO_BOX_COMPACT") != nullptr; if (no_compact) compact = false; }
#endif
    if (c.pack_tokens) {
        launch_lx_pack_plan(b->input_mask + p0 * T, b->visual_attention_mask + p0 * V, T, (int)n, h->pk_off[0], h->pk_cnt[0],
                            h->pk_src[0], lang_add, h->pk_rows, h->pk_off[1], h->pk_cnt[1], h->pk_src[1], visn_add, h->pk_rows + 1, st);
        pl.off = h->pk_off[0]; pl.cnt = h->pk_cnt[0]; pl.rows = h->pk_rows;
        pv.off = h->pk_off[1]; pv.cnt = h->pk_cnt[1]; pv.rows = h->pk_rows + 1;
    }
    if (compact) launch_split_f32_rows(b->feats + p0 * V * MMS_FEAT, h->pk_src[1], h->pk_rows + 1, (int)MV, MMS_FEAT, featp.hi, featp.lo, st);
    else if (featp_shared) featp = *featp_shared;
    else launch_split_f32(b->feats + p0 * V * MMS_FEAT, featp.hi, featp.lo, MV * MMS_FEAT, st);
    if (int rc = gemm(h, st, featp, MMS_FEAT, ID, h->w_visn, h->b_visn, MV, H, MMS_FEAT, ACT_NONE, to_f32(xf, H), nullptr, compact ? h->pk_rows + 1 : nullptr)) return rc;
    if (c.pack_tokens) {
        if (h->lq_join_pending) {}      // (the distinct-query stage is still running on the side lane: the copy follows the join, behind the r_layers)
        else if (h->lq_active)   // language rows after the l_layers, copied from the distinct-query store (pair p0 + src / T, token src % T)
            launch_rows_gather(h->lq_store.hi, h->lq_store.lo, h->pk_src[0], h->lq_index + p0, T, h->pk_rows, (int)ML, h->x.hi, h->x.lo, st);
        else
            launch_lx_embed_lang_packed(h->E, h->pos_tab, h->type_tab, h->emb_g, h->emb_b, b->input_ids + p0 * T, T, c.vocab, h->pk_src[0],
                                        h->pk_rows, (int)ML, h->x.hi, h->x.lo, st);
        launch_lx_visn(xf, h->g_visn, h->be_visn, b->boxes + p0 * V * 4, 4, h->w_box, h->b_box, h->g_box, h->be_box, h->lab_feat,
                       label_index + p0 * V, (int)h->n_labels, h->x.at(ML * H).hi, h->x.at(ML * H).lo, (int)MV, st, h->pk_src[1], h->pk_rows + 1, compact ? 1 : 0);
    } else {
        launch_lx_masks(b->input_mask + p0 * T, b->visual_attention_mask + p0 * V, T, lang_add, visn_add, (int)n, st);
        launch_lx_embed_lang(h->E, h->pos_tab, h->type_tab, h->emb_g, h->emb_b, b->input_ids + p0 * T, T, c.vocab, h->x.hi, h->x.lo, (int)n, st);
        launch_lx_visn(xf, h->g_visn, h->be_visn, b->boxes + p0 * V * 4, 4, h->w_box, h->b_box, h->g_box, h->be_box, h->lab_feat,
                       label_index + p0 * V, (int)h->n_labels, h->x.at(ML * H).hi, h->x.at(ML * H).lo, (int)MV, st);
    }
    plan_tiles(h, st, pl, n, T, 0);
    plan_tiles(h, st, pv, n, V, 1);
    Pack px;
    if (c.x_layers > 1 || (c.x_layers > 0 && c.stop_after >= 0)) plan_cross_tiles(h, st, px, pl, pv, n, T, V);      // (the last X layer of a full run is trimmed to what the pooler reads: two-kernel route)
    if (h->f8) launch_planes_to_f8(h->x.hi, h->x.lo, h->x.f8, R * H, st);
    int budget = c.stop_after >= 0 ? c.stop_after : (1 << 30);
    // small calls: two lanes (see mms_handle::side).  Not with a layer budget (debug runs count layers in stream order) and not in precision mode 4.
    const bool lanes = h->lanes_on && R * (int64_t)c.inter <= h->mid_elems;
    if (lanes) {
        if (int rc = lanes_init(h)) return rc;
        h->lane = 1;
        const int rc = ensure_kparts(h, kparts_side_need(h, n));      // (a no-op: sized with the workspace)
        h->lane = 0;
        if (rc) return rc;
        h->mid_side_off = ML * (int64_t)c.inter;
    }
    auto r_stage = [&](hipStream_t s) -> int {
        for (int i = 0; i < c.r_layers && budget > 0; ++i, --budget) {
            if (int rc = att_block(h, s, h->r_layers[i].att, h->x, h->y, ML, V, n, visn_add, pv)) return rc;
            if (int rc = ffn_block(h, s, h->r_layers[i].ffn, h->y, h->x, ML, MV, ACT_GELU_ERF, pv)) return rc;
        }
        return MMS_OK;
    };
    const bool lr_lanes = lanes && !h->lq_active && c.layers > 0 && c.r_layers > 0;      // language layers beside the box stream's layers
    if (lr_lanes) { if (int rc = lane_fork(h, st)) return rc; }
    for (int i = 0; i < c.layers && budget > 0 && !h->lq_active; ++i, --budget) {
        if (int rc = att_block(h, st, h->layers[i].att, h->x, h->y, 0, T, n, lang_add, pl)) return rc;
        if (int rc = ffn_block(h, st, h->layers[i].ffn, h->y, h->x, 0, ML, ACT_GELU_ERF, pl)) return rc;
    }
    if (lr_lanes) {
        if (int rc = on_side_lane(h, r_stage)) return rc;
        if (int rc = lane_join(h, st)) return rc;
    } else if (int rc = r_stage(st)) return rc;
    if (h->lq_join_pending) {      // the distinct-query stage ran beside the prologue and the r_layers above
        if (int rc = lane_join(h, st)) return rc;
        h->lq_join_pending = false;
        launch_rows_gather(h->lq_store.hi, h->lq_store.lo, h->pk_src[0], h->lq_index + p0, T, h->pk_rows, (int)ML, h->x.hi, h->x.lo, st);
    }
    const bool trim_last = c.stop_after < 0 && c.x_layers > 0;   // debug runs keep the full hidden state
    for (int i = 0; i < c.x_layers && budget > 0; ++i, --budget) {
        const XLayerW& w = h->x_layers[i];
        if (trim_last && i == c.x_layers - 1) {
            // Last cross layer: only the language CLS row is read afterwards (pooler).  The vision stream's outputs
            // (visn<-lang attention, vision self-attention, vision FFN) are dead; the language stream needs the
            // lang<-visn cross attention on all its rows (they are the keys of its self-attention), then a CLS-only block.
            if (int rc = gemm(h, st, h->x, H, ID, w.cross.wqkv, w.cross.bqkv, ML, H, H, ACT_NONE, to_qkv(h, 0, ML), nullptr, pl.rows)) return rc;
            if (int rc = gemm(h, st, h->x.at(ML * H), H, ID, w.cross.wqkv + (long long)H * H, w.cross.bqkv + H, MV, 2 * H, H, ACT_NONE,
                              to_qkv(h, ML, MV, H), nullptr, pv.rows)) return rc;
            AttnParams a{};
            a.ldo = H; a.B = (int)n; a.q_base = a.kv_base = 0;
            attn_q(a, h, 0, ML); a.Sq = T;
            attn_kv(a, h, ML, MV); a.Sk = V; a.key_add = visn_add;
            a.o_hi = h->ctx.hi; a.o_lo = h->ctx.lo;
            a.q_off = pl.off; a.q_cnt = pl.cnt; a.kv_off = pv.off; a.kv_cnt = pv.cnt;
            if (int rc = attend(h, a, st)) return rc;
            bool fl = false;      // (big launches: bias + residual + LayerNorm in the projection's epilogue, as in the layers before)
            if (!(h->f8 && h->x.f8)) {
                if (int rc = gemm_ln(h, st, false, h->ctx, H, w.cross.wo, nullptr, nullptr, w.cross.bo, ML, H, h->x, w.cross.g, w.cross.b, h->y, h->t, pl.rows, &fl)) return rc;
            }
            if (!fl) {
                if (int rc = gemm(h, st, h->ctx, H, ID, w.cross.wo, w.cross.bo, ML, H, H, ACT_NONE, to_f32(h->t, H), &h->x, pl.rows)) return rc;
                ln_resid(h, st, h->t, w.cross.g, w.cross.b, h->y, ML, pl.rows, h->x);
            }
            if (int rc = last_block_cls(h, st, w.lang_self, w.lang_ffn, ACT_GELU_ERF, h->y, h->x, T, n, lang_add, pl)) return rc;
            break;
        }
        // cross attention, both directions with the SAME weights (modeling.py:460-464): the QKV projection and
        // the output dense + LN run over both streams (one launch when dense, one per stream when packed)
        const bool cross_fused = px.sub && w.cross.wqkv_hm && !(h->x1_mask & 1);
        if (cross_fused) {
            // ONE launch: [Q | K | V] of both streams' rows with the shared weights, per head, K / V staged in LDS, both directions attended from there
            // (qkv_attn.hip CROSS mode) -- the fp32 [rows][2304] tensor and the two attention launches are gone
            QkvAttnParams q{};
            q.a_hi = h->x.hi; q.lda = H; q.w = w.cross.wqkv_hm; q.bias = w.cross.bqkv_hm; q.K = H;
            if (h->nsplit == 3) q.w_lo = w.cross.wqkv_hm + (long long)3 * H * H;
            q.sub = px.sub; q.sub2 = px.sub2; q.n_sub = px.n_sub; q.pair_rec = px.rec;
            q.pair_off = pl.off; q.pair_cnt = pl.cnt; q.S = T; q.pair_off2 = pv.off; q.pair_cnt2 = pv.cnt; q.S2 = V;
            q.key_add = lang_add; q.key_add2 = visn_add; q.row0_b = ML;
            q.o_hi = h->ctx.hi; q.o_lo = h->ctx.lo; q.ldo = H;
            q.M = (int)ML; q.m_dev = pl.rows; q.M2 = (int)MV; q.m_dev2 = pv.rows; q.fast = 1;
            if (int rc = fused_attn_launch(h, st, q)) return rc;
        } else {
        if (c.pack_tokens) {
            if (int rc = gemm(h, st, h->x, H, ID, w.cross.wqkv, w.cross.bqkv, ML, 3 * H, H, ACT_NONE, to_qkv(h, 0, ML), nullptr, pl.rows)) return rc;
            if (int rc = gemm(h, st, h->x.at(ML * H), H, ID, w.cross.wqkv, w.cross.bqkv, MV, 3 * H, H, ACT_NONE,
                              to_qkv(h, ML, MV), nullptr, pv.rows)) return rc;
        } else {
            // dense rows: one projection over both streams, written as the two streams' own regions would be (rows [0, ML) and [ML, R))
            if (int rc = gemm(h, st, h->x, H, ID, w.cross.wqkv, w.cross.bqkv, ML, 3 * H, H, ACT_NONE, to_qkv(h, 0, ML))) return rc;
            if (int rc = gemm(h, st, h->x.at(ML * H), H, ID, w.cross.wqkv, w.cross.bqkv, MV, 3 * H, H, ACT_NONE, to_qkv(h, ML, MV))) return rc;
        }
        AttnParams a{};
        a.ldo = H; a.B = (int)n; a.q_base = a.kv_base = 0;
        attn_q(a, h, 0, ML); a.Sq = T;                            // lang <- visn
        attn_kv(a, h, ML, MV); a.Sk = V; a.key_add = visn_add;
        a.o_hi = h->ctx.hi; a.o_lo = h->ctx.lo;
        a.q_off = pl.off; a.q_cnt = pl.cnt; a.kv_off = pv.off; a.kv_cnt = pv.cnt;
        if (int rc = attend(h, a, st)) return rc;
        attn_q(a, h, ML, MV); a.Sq = V;                           // visn <- lang
        attn_kv(a, h, 0, ML); a.Sk = T; a.key_add = lang_add;
        a.o_hi = h->ctx.at(ML * H).hi; a.o_lo = h->ctx.at(ML * H).lo;
        a.q_off = pv.off; a.q_cnt = pv.cnt; a.kv_off = pl.off; a.kv_cnt = pl.cnt;
        if (int rc = attend(h, a, st)) return rc;
        }
        if (lanes) {
            // the two streams' chains up to the next cross attention (attention output + LayerNorm, self-attention block, FFN block), side by side
            if (int rc = lane_fork(h, st)) return rc;
            if (int rc = gemm(h, st, h->ctx, H, ID, w.cross.wo, w.cross.bo, ML, H, H, ACT_NONE, to_f32(h->t, H), &h->x, pl.rows)) return rc;
            ln_resid(h, st, h->t, w.cross.g, w.cross.b, h->y, ML, pl.rows, h->x);
            if (int rc = att_block(h, st, w.lang_self, h->y, h->x, 0, T, n, lang_add, pl)) return rc;
            if (int rc = ffn_block(h, st, w.lang_ffn, h->x, h->x, 0, ML, ACT_GELU_ERF, pl)) return rc;
            if (int rc = on_side_lane(h, [&](hipStream_t s) -> int {
                    const Planes rv = h->x.at(ML * H);
                    if (int rc = gemm(h, s, h->ctx.at(ML * H), H, ID, w.cross.wo, w.cross.bo, MV, H, H, ACT_NONE, to_f32(h->t + ML * H, H), &rv, pv.rows)) return rc;
                    ln_resid(h, s, h->t + ML * H, w.cross.g, w.cross.b, h->y.at(ML * H), MV, pv.rows, rv);
                    if (int rc = att_block(h, s, w.visn_self, h->y, h->x, ML, V, n, visn_add, pv)) return rc;
                    return ffn_block(h, s, w.visn_ffn, h->x, h->x, ML, MV, ACT_GELU_ERF, pv);
                })) return rc;
            if (int rc = lane_join(h, st)) return rc;
            continue;
        }
        if (c.pack_tokens) {
            // per stream: the fused bias + residual + LayerNorm epilogue on big launches (mms_config.fuse_layernorm bit 0), else GEMM -> LayerNorm kernel
            const Planes rv = h->x.at(ML * H);
            bool fl = false, fv = false;
            if (!(h->f8 && h->x.f8)) {
                if (int rc = gemm_ln(h, st, false, h->ctx, H, w.cross.wo, nullptr, nullptr, w.cross.bo, ML, H, h->x, w.cross.g, w.cross.b, h->y, h->t, pl.rows, &fl)) return rc;
                if (int rc = gemm_ln(h, st, false, h->ctx.at(ML * H), H, w.cross.wo, nullptr, nullptr, w.cross.bo, MV, H, rv, w.cross.g, w.cross.b, h->y.at(ML * H),
                                     h->t + ML * H, pv.rows, &fv)) return rc;
            }
            if (!fl) {
                if (int rc = gemm(h, st, h->ctx, H, ID, w.cross.wo, w.cross.bo, ML, H, H, ACT_NONE, to_f32(h->t, H), &h->x, pl.rows)) return rc;
                ln_resid(h, st, h->t, w.cross.g, w.cross.b, h->y, ML, pl.rows, h->x);
            }
            if (!fv) {
                if (int rc = gemm(h, st, h->ctx.at(ML * H), H, ID, w.cross.wo, w.cross.bo, MV, H, H, ACT_NONE, to_f32(h->t + ML * H, H), &rv, pv.rows)) return rc;
                ln_resid(h, st, h->t + ML * H, w.cross.g, w.cross.b, h->y.at(ML * H), MV, pv.rows, rv);
            }
        } else {
            if (int rc = gemm(h, st, h->ctx, H, ID, w.cross.wo, w.cross.bo, R, H, H, ACT_NONE, to_f32(h->t, H), &h->x)) return rc;
            ln_resid(h, st, h->t, w.cross.g, w.cross.b, h->y, R, nullptr, h->x);
        }
        // per-stream self attention (y -> x), then per-stream FFN (x -> x)
        if (int rc = att_block(h, st, w.lang_self, h->y, h->x, 0, T, n, lang_add, pl)) return rc;
        if (int rc = att_block(h, st, w.visn_self, h->y, h->x, ML, V, n, visn_add, pv)) return rc;
        if (int rc = ffn_block(h, st, w.lang_ffn, h->x, h->x, 0, ML, ACT_GELU_ERF, pl)) return rc;
        if (int rc = ffn_block(h, st, w.visn_ffn, h->x, h->x, ML, MV, ACT_GELU_ERF, pv)) return rc;
    }
    // pooler (modeling.py:596-608) -> logit_fc (kdd_model.py:167-172)
    if (trim_last && budget > 0) {   // compact CLS rows were left in y by last_block_cls
        if (int rc = gemm(h, st, h->y, H, ID, h->w_pool, h->b_pool, n, H, H, ACT_TANH, to_planes(h->ctx, H))) return rc;
    } else if (int rc = gemm(h, st, h->x, H, RowMap{1, T, 0}, h->w_pool, h->b_pool, n, H, H, ACT_TANH, to_planes(h->ctx, H), nullptr, nullptr, pl.off)) return rc;
    if (b->x_norm) launch_xnorm(h->ctx.hi, h->ctx.lo, b->x_norm + p0 * H, (int)n, st);   // kdd_model.py:204-205
    if (int rc = gemm(h, st, h->ctx, H, ID, h->w_fc0, h->b_fc0, n, 2 * H, H, ACT_GELU_ERF, to_f32(h->hbuf, 2 * H))) return rc;
    launch_lx_head(h->hbuf, h->g_fc2, h->be_fc2, h->w_fc3, h->b_fc3, logits + p0 * 2, probs ? probs + p0 * 2 : nullptr, (int)n, st);
    return MMS_OK;
}

// Pairs per launch wave: at most chunk_pairs (default 32768: the 256-row GEMM tiles then run >= 25 full rounds over the
// 256 CUs per launch, +2.4 % zk / +4.8 % lxmert over 8192, profiles/r01c_gemm_variants.txt), and equal-sized chunks so that
// no launch wave is a short tail (B = 40000 -> 2 x 20000, not 32768 + 7232).  The workspace is sized for one chunk.
int chunk_size(const mms_handle* h, int64_t B) {
    const int64_t c = h->cfg.chunk_pairs > 0 ? h->cfg.chunk_pairs : 32768;
    const int64_t n = (B + c - 1) / c;
    return (int)(n <= 1 ? B : (B + n - 1) / n);
}

}  // namespace

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

int mms_version(void) { return MMS_ABI_VERSION; }
const char* mms_global_error(void) { return g_err.c_str(); }
const char* mms_last_error(const mms_handle* h) { return h ? h->err.c_str() : "null handle"; }

int mms_create(const mms_config* cfg, mms_handle** out) {
    if (!cfg || !out) { g_err = "null argument"; return MMS_ERR_ARG; }
    *out = nullptr;
    if (cfg->model < 0 || cfg->model > 2) { g_err = "bad model id"; return MMS_ERR_ARG; }
    if (cfg->inter <= 0 || cfg->inter % 128) { g_err = "inter must be a positive multiple of 128"; return MMS_ERR_ARG; }
    if (cfg->precision < 1 || cfg->precision > 4) { g_err = "precision must be 1, 2, 3 or 4"; return MMS_ERR_ARG; }
    if (cfg->fuse_attention < 0 || cfg->fuse_attention > 2) { g_err = "fuse_attention must be 0, 1 or 2"; return MMS_ERR_ARG; }
    if (cfg->fuse_layernorm < 0 || cfg->fuse_layernorm > 3) { g_err = "fuse_layernorm must be a mask in 0..3 (1: attention output, 2: FFN down)"; return MMS_ERR_ARG; }
    if (cfg->text_len <= 0 || cfg->text_len > 32 || cfg->text_len + 1 > cfg->max_pos) { g_err = "bad text_len"; return MMS_ERR_ARG; }
    if (cfg->layers < 0 || cfg->vocab <= 0 || cfg->type_vocab < 2) { g_err = "bad layer/vocab config (type_vocab must be >= 2: segment ids 0 / 1)"; return MMS_ERR_ARG; }
    {   // the attention kernels cover sequences of up to 48 tokens (3 x 3 tiles of 16): zk text+10 boxes, lds text+10+10, lxmert text | 10
        const int seq = cfg->text_len + (cfg->model == MMS_MODEL_ZK ? MMS_NBOX : cfg->model == MMS_MODEL_LDS ? 2 * MMS_NBOX : 0);
        if (seq > 48) { g_err = "text_len too long: the sequence (" + std::to_string(seq) + " tokens) exceeds the 48-token attention kernels"; return MMS_ERR_ARG; }
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) { g_err = std::string("no HIP device: ") + hipGetErrorString(e); return MMS_ERR_HIP; }
    if (cfg->device < 0 || cfg->device >= ndev) { g_err = "bad device ordinal"; return MMS_ERR_ARG; }
    {   // the kernels are written for gfx950 alone (160 KiB LDS per workgroup, v_mfma_f32_16x16x32_bf16, MX-scaled fp8 MFMA, global_load_lds b128,
        // v_permlane16_swap): refuse any other device here instead of failing inside the first launch
        hipDeviceProp_t prop;
        e = hipGetDeviceProperties(&prop, cfg->device);
        if (e != hipSuccess) { g_err = std::string("hipGetDeviceProperties: ") + hipGetErrorString(e); return MMS_ERR_HIP; }
        if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) { g_err = std::string("device ") + std::to_string(cfg->device) + " is " + prop.gcnArchName + ": libmmscore is built for gfx950 (MI355X) only"; return MMS_ERR_HIP; }
    }
    mms_handle* h = new mms_handle();
    h->cfg = *cfg;
    h->f8 = cfg->precision == 4;
    h->fuse_ln = cfg->fuse_layernorm;
    h->fuse_attn = cfg->fuse_attention;
    h->nsplit = h->f8 ? 2 : cfg->precision;
#ifdef MMS_LAB   // A/B knobs exist in libmmscore_lab.so only; the product library reads no environment variable
    if (const char* e = getenv("MMS_FUSE_LN")) h->fuse_ln = atoi(e);
    if (const char* e = getenv("MMS_X1_MASK")) h->x1_mask = atoi(e);
    if (const char* e = getenv("MMS_RESID_IN_LN")) h->resid_in_ln = atoi(e);
    if (const char* e = getenv("MMS_ALTERNATE")) h->alternate = atoi(e);
#endif
    *out = h;
    return MMS_OK;
}

void mms_destroy(mms_handle* h) {
    if (!h) return;
    DeviceScope dev(h->cfg.device);
    free_pool(h->w_allocs);
    free_pool(h->ws_allocs);
    free_pool(h->lab_allocs);
    free_pool(h->dd_allocs);
    free_pool(h->ens_allocs);
    free_pool(h->ens_c_allocs);
    free_pool(h->lq_allocs);
    free_pool(h->lq_sub_allocs);
    if (h->lq_store.hi) (void)hipFree(h->lq_store.hi);
    if (h->kparts) (void)hipFree(h->kparts);
    if (h->kparts_side) (void)hipFree(h->kparts_side);
    if (h->ens_featp.hi) (void)hipFree(h->ens_featp.hi);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    if (h->side) (void)hipStreamDestroy(h->side);
    for (auto e : h->ev) (void)hipEventDestroy(e);
    for (auto e : h->ev_fused) (void)hipEventDestroy(e);
    delete h;
}

int mms_load_weight(mms_handle* h, const char* name, const float* host_data, const int64_t* shape, int32_t rank) {
    if (!h) return MMS_ERR_ARG;
    if (!name || !host_data || !shape || rank < 1 || rank > 4) return h->fail(MMS_ERR_ARG, "mms_load_weight: bad argument");
    if (h->finalized) return h->fail(MMS_ERR_STATE, "mms_load_weight after mms_finalize");
    HostTensor t;
    t.shape.assign(shape, shape + rank);
    const int64_t n = t.numel();
    if (n <= 0) return h->fail(MMS_ERR_ARG, std::string("mms_load_weight: empty tensor ") + name);
    t.data.assign(host_data, host_data + n);
    h->host[name] = std::move(t);
    return MMS_OK;
}

int mms_finalize(mms_handle* h) {
    if (!h) return MMS_ERR_ARG;
    if (h->finalized) return h->fail(MMS_ERR_STATE, "already finalized");
    DeviceScope dev(h->cfg.device);
    int rc = h->cfg.model == MMS_MODEL_ZK ? finalize_zk(h) : h->cfg.model == MMS_MODEL_LDS ? finalize_lds(h) : finalize_lxmert(h);
    if (rc) {   // leave nothing dangling: a retry (after loading the missing tensor) starts from a clean handle
        free_pool(h->w_allocs);
        h->w_planes.clear(); h->flop_counter = nullptr; h->timing = false;
        h->layers.clear(); h->r_layers.clear(); h->x_layers.clear();
        return rc;
    }
    h->host.clear();
    h->finalized = true;
    return MMS_OK;
}

int mms_score_zk(mms_handle* h, const mms_zk_batch* b, float* logits, float* probs, void* stream) {
    if (int rc = check_ready(h, MMS_MODEL_ZK, b, logits)) return rc;
    const int64_t B = b->n_pairs;
    if (B < 0 || b->n_uniq_labels < 0) return h->fail(MMS_ERR_ARG, "negative batch size");
    if (B == 0) return MMS_OK;
    if (!b->num_boxes || !b->boxes_5 || !b->feats || !b->query_ids || !b->len_query || !b->labels || !b->segment_ids)
        return h->fail(MMS_ERR_ARG, "mms_score_zk: null batch field");
    if (b->uniq_label_ids ? (!b->label_index || b->n_uniq_labels == 0) : !b->label_ids)
        return h->fail(MMS_ERR_ARG, "mms_score_zk: pass label_ids (dense) or uniq_label_ids + label_index");
    h->lanes_on = false; h->no_fused_ln = false;
    DeviceScope dev(h->cfg.device);
    hipStream_t st = (hipStream_t)stream;
    const int cs = chunk_size(h, B);
    if (int rc = ensure_workspace(h, cs)) return rc;
    const int32_t* uniq = b->uniq_label_ids;
    const int32_t* index = b->label_index;
    int64_t U = b->n_uniq_labels;
    const int* U_dev = nullptr;
    if (!uniq) {
        const bool async = B * MMS_NBOX <= 512;      // calls of <= 51 pairs (the label workspace is sized for the upper bound): device-side count, no read-back
        if (int rc = dedup_labels<int32_t>(h, st, b->label_ids, B * MMS_NBOX, &U, async)) return rc;
        uniq = h->dd_uniq32; index = h->dd_index;
        if (async) U_dev = h->dd_counter;
    }
    if (int rc = ln_begin_call(h, st)) return rc;
    if (int rc = zk_label_features(h, st, uniq, U, U_dev)) return rc;
    for (int64_t p0 = 0; p0 < B; p0 += cs)
        if (int rc = zk_chunk(h, st, b, index, p0, (B - p0) < cs ? (B - p0) : cs, logits, probs)) return rc;
    return post_launch(h);
}

int mms_score_lds(mms_handle* h, const mms_lds_batch* b, float* logits, float* probs, void* stream) {
    if (int rc = check_ready(h, MMS_MODEL_LDS, b, logits)) return rc;
    const int64_t B = b->n_pairs;
    if (B < 0) return h->fail(MMS_ERR_ARG, "negative batch size");
    if (B == 0) return MMS_OK;
    if (!b->input_ids || !b->segment_ids || !b->features || !b->labelfeat) return h->fail(MMS_ERR_ARG, "mms_score_lds: null batch field");
    h->lanes_on = false; h->no_fused_ln = false;
    DeviceScope dev(h->cfg.device);
    hipStream_t st = (hipStream_t)stream;
    const int cs = chunk_size(h, B);
    if (int rc = ensure_workspace(h, cs)) return rc;
    if (int rc = ln_begin_call(h, st)) return rc;
    for (int64_t p0 = 0; p0 < B; p0 += cs)
        if (int rc = lds_chunk(h, st, b, p0, (B - p0) < cs ? (B - p0) : cs, logits, probs)) return rc;
    return post_launch(h);
}

int mms_score_lxmert(mms_handle* h, const mms_lxmert_batch* b, float* logits, float* probs, void* stream) {
    if (int rc = check_ready(h, MMS_MODEL_LXMERT, b, logits)) return rc;
    const int64_t B = b->n_pairs;
    if (B < 0 || b->n_uniq_labels < 0) return h->fail(MMS_ERR_ARG, "negative batch size");
    if (B == 0) return MMS_OK;
    if (!b->input_ids || !b->input_mask || !b->feats || !b->boxes || !b->visual_attention_mask)
        return h->fail(MMS_ERR_ARG, "mms_score_lxmert: null batch field");
    if (b->uniq_label_ids ? (!b->label_index || b->n_uniq_labels == 0) : !b->label_ids)
        return h->fail(MMS_ERR_ARG, "mms_score_lxmert: pass label_ids (dense) or uniq_label_ids + label_index");
    DeviceScope dev(h->cfg.device);
    hipStream_t st = (hipStream_t)stream;
    const int cs = chunk_size(h, B);
    if (int rc = ensure_workspace(h, cs)) return rc;
    const int64_t* uniq = b->uniq_label_ids;
    const int32_t* index = b->label_index;
    int64_t U = b->n_uniq_labels;
    if (!uniq) {
        if (int rc = dedup_labels<int64_t>(h, st, b->label_ids, B * MMS_NBOX, &U)) return rc;
        uniq = h->dd_uniq64; index = h->dd_index;
    }
    if (int rc = ln_begin_call(h, st)) return rc;
    // two launch lanes (mms_handle::side) for calls whose launch wave has fewer than LANE_ROWS token rows; not for debug runs (they count layers in stream
    // order) or precision mode 4; with per-launch timing (bench.py's roofline pass) only the distinct-query stage keeps its side lane, and its launches stay out of the per-launch sums
    h->lanes_q = h->cfg.pack_tokens && h->cfg.stop_after < 0 && !h->f8 && lane_rows() > 0 && lane_query_stage();
    const bool small_wave = h->lanes_q && (int64_t)cs * (h->cfg.text_len + MMS_NBOX) < lane_rows();
    h->lanes_on = small_wave && !h->timing;
    h->no_fused_ln = small_wave;      // with or without timing: instrumentation must not change the LayerNorm route, i.e. the logits
    LaneGuard guard(h);
    if (int rc = lx_label_features(h, st, uniq, U)) return rc;
    if (int rc = lx_query_stage(h, st, b, B, cs)) return rc;
    for (int64_t p0 = 0; p0 < B; p0 += cs)
        if (int rc = lx_chunk(h, st, b, index, p0, (B - p0) < cs ? (B - p0) : cs, logits, probs)) return rc;
    return guard.done(post_launch(h));
}

// ---- the three models on the same pairs in one call (BASELINE.json config 5; merge of code/main.py:59) ----
static int ensure_ens_ws(mms_handle* z, int64_t B, int T, int TL) {
    if (B <= z->ens_pairs) return MMS_OK;
    free_pool(z->ens_allocs);
    free_pool(z->ens_c_allocs);
    z->ens_pairs = 0;
    void* p;
    auto get = [&](size_t bytes, void** out) { int rc = dev_alloc(z, z->ens_allocs, &p, bytes); *out = p; return rc; };
    if (int rc = get((size_t)B * (T + MMS_NBOX) * 4, (void**)&z->ens_seg)) return rc;
    if (int rc = get((size_t)B * T * 8, (void**)&z->ens_ids64)) return rc;
    if (int rc = get((size_t)B * T * 8, (void**)&z->ens_seg64)) return rc;
    if (int rc = get((size_t)B * MMS_NBOX * MMS_LABEL_LEN * 8, (void**)&z->ens_lab64)) return rc;
    if (int rc = get((size_t)B * TL * 8 * 2, (void**)&z->ens_mask64)) return rc;     // lxmert ids, then mask
    if (int rc = get((size_t)B * MMS_NBOX * 4, (void**)&z->ens_vmask)) return rc;
    if (int rc = get((size_t)B * MMS_NBOX * 4 * 4, (void**)&z->ens_boxes4)) return rc;
    if (int rc = get((size_t)B * 2 * 4 * 4, (void**)&z->ens_logits)) return rc;
    if (int rc = get((size_t)B * 2 * 4 * 4, (void**)&z->ens_probs)) return rc;
    if (int rc = get((size_t)B * 4, (void**)&z->ens_diff)) return rc;
    if (int rc = get((size_t)B * 4, (void**)&z->ens_doff)) return rc;
    if (int rc = get((size_t)B * 4, (void**)&z->ens_dlist)) return rc;
    if (int rc = get((size_t)(B / 64 + 16) * 4, (void**)&z->ens_dcount)) return rc;
    z->ens_cpairs = 0;
    z->ens_pairs = B;
    return MMS_OK;
}

int mms_score_ensemble(mms_handle* z, mms_handle* l, mms_handle* x, const mms_ensemble_batch* b, const float* weights4, float* merged,
                       float* member_scores, void* stream) {
    if (!z || !l || !x) return MMS_ERR_ARG;
    if (int rc = check_ready(z, MMS_MODEL_ZK, b, merged)) return rc;
    if (int rc = check_ready(l, MMS_MODEL_LDS, b, merged)) return z->fail(rc, "lds handle: " + l->err);
    if (int rc = check_ready(x, MMS_MODEL_LXMERT, b, merged)) return z->fail(rc, "lxmert handle: " + x->err);
    if (z->cfg.device != l->cfg.device || z->cfg.device != x->cfg.device) return z->fail(MMS_ERR_ARG, "mms_score_ensemble: the three handles must live on one device");
    if (z->cfg.text_len != l->cfg.text_len) return z->fail(MMS_ERR_ARG, "mms_score_ensemble: zk and lds must share text_len (one query feed)");
    if (z->cfg.stop_after >= 0 || l->cfg.stop_after >= 0 || x->cfg.stop_after >= 0) return z->fail(MMS_ERR_ARG, "mms_score_ensemble: debug handles (stop_after) not supported");
    const int64_t B = b->n_pairs;
    if (B < 0) return z->fail(MMS_ERR_ARG, "negative batch size");
    if (B == 0) return MMS_OK;
    if (!weights4 || !b->feats || !b->boxes_5 || !b->num_boxes || !b->label_ids || !b->query_ids || !b->len_query || !b->s2f_query_ids ||
        !b->s2f_len_query || !b->labels || !b->lx_input_ids || !b->lx_input_mask)
        return z->fail(MMS_ERR_ARG, "mms_score_ensemble: null batch field");
    DeviceScope dev(z->cfg.device);
    hipStream_t st = (hipStream_t)stream;
    const int T = z->cfg.text_len, TL = x->cfg.text_len;
    // one launch-wave size for the three members (the smallest of their chunk settings)
    int cs = chunk_size(z, B);
    { const int c2 = chunk_size(l, B), c3 = chunk_size(x, B); cs = c2 < cs ? c2 : cs; cs = c3 < cs ? c3 : cs; }
    if (int rc = ensure_workspace(z, cs)) return rc;
    if (int rc = ensure_workspace(l, cs)) return z->fail(rc, "lds handle: " + l->err);
    if (int rc = ensure_workspace(x, cs)) return z->fail(rc, "lxmert handle: " + x->err);
    if (int rc = ensure_ens_ws(z, B, T, TL)) return rc;
    if (int rc = ln_begin_call(z, st)) return rc;
    if (int rc = ln_begin_call(l, st)) return z->fail(rc, "lds handle: " + l->err);
    if (int rc = ln_begin_call(x, st)) return z->fail(rc, "lxmert handle: " + x->err);
    // ---- feeds in each member's own dtypes ----
    launch_zk_segment_ids(z->ens_seg, B, T, st);                                   // load_data_v4.py:204
    launch_i32_to_i64(b->query_ids, z->ens_ids64, B * T, st);                      // lds reads the zk query as int64 (run_pretraining_predict_score.py:526-548)
    launch_fill_i64(z->ens_seg64, B * T, 0, st);
    launch_i32_to_i64(b->label_ids, z->ens_lab64, B * MMS_NBOX * MMS_LABEL_LEN, st);
    int64_t* lx_ids = z->ens_mask64;
    int64_t* lx_mask = z->ens_mask64 + B * TL;
    launch_i32_to_i64(b->lx_input_ids, lx_ids, B * TL, st);
    launch_i32_to_i64(b->lx_input_mask, lx_mask, B * TL, st);
    launch_box_mask(b->num_boxes, z->ens_vmask, B, st);
    launch_corners(b->boxes_5, z->ens_boxes4, B, st);
    // ---- label text: distinct tuples once, encoded by zk's conv stack and by lxmert's label encoder ----
    int64_t U = 0;
    if (int rc = dedup_labels<int32_t>(z, st, b->label_ids, B * MMS_NBOX, &U)) return rc;
    if (int rc = zk_label_features(z, st, z->dd_uniq32, U)) return rc;
    if (int rc = lx_label_features(x, st, z->dd_uniq64, U)) return z->fail(rc, "lxmert handle: " + x->err);
    const int32_t* index = z->dd_index;

    mms_zk_batch zb{};
    zb.n_pairs = B; zb.num_boxes = b->num_boxes; zb.boxes_5 = b->boxes_5; zb.feats = b->feats; zb.query_ids = b->query_ids;
    zb.len_query = b->len_query; zb.labels = b->labels; zb.segment_ids = z->ens_seg;
    mms_zk_batch zb2 = zb;
    zb2.query_ids = b->s2f_query_ids; zb2.len_query = b->s2f_len_query;
    mms_lds_batch lb{};
    lb.n_pairs = B; lb.input_ids = z->ens_ids64; lb.segment_ids = z->ens_seg64; lb.features = b->feats; lb.labelfeat = z->ens_lab64;
    mms_lxmert_batch xb{};
    xb.n_pairs = B; xb.input_ids = lx_ids; xb.input_mask = lx_mask; xb.feats = b->feats; xb.boxes = z->ens_boxes4;
    xb.visual_attention_mask = z->ens_vmask;
    // Waves of fewer than ENS_LANE_ROWS token rows in the longest member (lds: text + 20 per pair; 5000 pairs): the three members are independent once the feeds exist, and
    // each is a serial queue of launches that do not fill the chip.  They run side by side: zk (both passes) on the caller's stream, lds and lxmert on two more (fork after the
    // wave's feature split, join in front of the merge); same kernels, same operands -- and, as with lxmert's own lanes, every LayerNorm by its own kernel (the fused epilogue
    // needs its whole grid resident), which only matters above 409 pairs per wave.  Measured (profiles/rd5_lanes.txt): 3.52 -> 2.27 ms at 5 pairs, 10.3 -> 7.8 at 256,
    // 18.1 -> 13.9 at 600, 27.6 -> 22.6 at 1024, 46.3 -> 40.7 at 2048, 80.1 -> 77.8 at 4096.
    int64_t ens_rows = ENS_LANE_ROWS_DEFAULT;
#ifdef MMS_LAB
    { static const int64_t v = getenv("MMS_ENS_LANE_ROWS") ? atoll(getenv("MMS_ENS_LANE_ROWS")) : 0; if (v) ens_rows = v; }
#endif
    const bool ens_small = lane_rows() > 0 && (int64_t)cs * (T + 2 * MMS_NBOX) < ens_rows && (int64_t)cs * (TL + MMS_NBOX) < ens_rows;
    const bool ens_lanes = ens_small && !z->timing && !l->timing && !x->timing;
    z->lanes_on = l->lanes_on = ens_lanes;
    // no fused LayerNorm epilogue in ANY member beside another member's lane (its grid-residency check would be decided by timing) -- the lxmert member included, whose own
    // lanes_on is off with pack_tokens = 0 --, and not in the timed run of such a wave either
    z->no_fused_ln = l->no_fused_ln = ens_small;
    LaneGuard guard(z, l, x);
    if (ens_lanes) {
        if (int rc = lanes_init(z)) return rc;
        if (int rc = lanes_init(l)) return z->fail(rc, "lds handle: " + l->err);
        if (cs > z->ens_featp_pairs) {
            if (z->ens_featp.hi) { (void)hipFree(z->ens_featp.hi); z->ens_featp.hi = nullptr; z->ens_featp_pairs = 0; }
            void* q = nullptr;
            HIP_TRY(z, hipMalloc(&q, (size_t)cs * MMS_NBOX * MMS_FEAT * 4));
            z->ens_featp.hi = (bf16*)q; z->ens_featp.lo = z->ens_featp.hi + MMS_PLANE_LO;
            z->ens_featp_pairs = cs;
        }
    }
    // two lanes inside the lxmert member's chunks as in mms_score_lxmert; its distinct-query stage goes to the side lane only in the small-wave regime above: beside
    // the other members' big launches with fused LayerNorm epilogues it would break those (gemm_ln)
    const bool x_small = x->cfg.pack_tokens && !x->f8 && lane_rows() > 0 && (int64_t)cs * (TL + MMS_NBOX) < lane_rows();
    x->lanes_on = x_small && !x->timing;
    x->no_fused_ln = ens_small || x_small;
    x->lanes_q = ens_lanes && x->lanes_on && lane_query_stage();
    if (int rc = lx_query_stage(x, st, &xb, B, cs)) return z->fail(rc, "lxmert member: " + x->err);
    float* lg[4]; float* pr[4];
    for (int k = 0; k < 4; ++k) { lg[k] = z->ens_logits + (int64_t)k * B * 2; pr[k] = z->ens_probs + (int64_t)k * B * 2; }

    // image tokens of a wave survive zk's first encoder pass in a side buffer (the QKV buffer they are built in gets reused)
    if (!z->ens_tok) {
        void* p;
        if (int rc = dev_alloc(z, z->ws_allocs, &p, (size_t)z->ws_pairs * MMS_NBOX * H * 4)) return rc;   // lives and dies with the workspace
        z->ens_tok = (float*)p;
    }
    // second zk member: the rewrite (load_data_v4.py:153-154) touches only queries that contain the phrase; every other pair's second
    // forward would repeat the first one bit for bit, so only the changed pairs are encoded again (per launch wave: flags -> scan ->
    // compact list; the per-wave counts come back in one small read)
    launch_query_differs(b->query_ids, b->len_query, b->s2f_query_ids, b->s2f_len_query, T, (int)B, z->ens_diff, st);
    const int n_waves = (int)((B + cs - 1) / cs);
    if (n_waves > (int)(B / 64 + 16)) return z->fail(MMS_ERR_ARG, "mms_score_ensemble: chunk_pairs too small for this batch");
    for (int wv = 0; wv < n_waves; ++wv) {
        const int64_t p0 = (int64_t)wv * cs, n = (B - p0) < cs ? (B - p0) : cs;
        launch_plan_scan(z->ens_diff + p0, (int)n, z->ens_doff + p0, z->ens_dcount + wv, st);
        launch_compact_list(z->ens_diff + p0, z->ens_doff + p0, (int)n, z->ens_dlist + p0, st);
    }
    std::vector<int> changed(n_waves);
    HIP_TRY(z, hipMemcpyAsync(changed.data(), z->ens_dcount, (size_t)n_waves * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(z, hipStreamSynchronize(st));
    {
        int64_t need = 0;
        for (int v : changed) need = v > need ? v : need;
        if (need > z->ens_cpairs) {
            void* p;
            free_pool(z->ens_c_allocs);
            const int S = T + MMS_NBOX; (void)S;
            if (int rc = dev_alloc(z, z->ens_c_allocs, &p, (size_t)need * T * 4)) return rc;
            z->ens_cq = (int32_t*)p;
            if (int rc = dev_alloc(z, z->ens_c_allocs, &p, (size_t)need * 4)) return rc;
            z->ens_clen = (int32_t*)p;
            if (int rc = dev_alloc(z, z->ens_c_allocs, &p, (size_t)need * 4)) return rc;
            z->ens_cnb = (int32_t*)p;
            if (int rc = dev_alloc(z, z->ens_c_allocs, &p, (size_t)need * 8)) return rc;
            z->ens_clab = (int64_t*)p;
            if (int rc = dev_alloc(z, z->ens_c_allocs, &p, (size_t)need * MMS_NBOX * H * 4)) return rc;
            z->ens_ctok = (float*)p;
            if (int rc = dev_alloc(z, z->ens_c_allocs, &p, (size_t)need * 2 * 4)) return rc;
            z->ens_clog = (float*)p;
            if (int rc = dev_alloc(z, z->ens_c_allocs, &p, (size_t)need * 2 * 4)) return rc;
            z->ens_cprob = (float*)p;
            z->ens_cpairs = need;
        }
    }
    for (int64_t p0 = 0; p0 < B; p0 += cs) {
        const int64_t n = (B - p0) < cs ? (B - p0) : cs;
        const int64_t NB = n * MMS_NBOX;
        // the 2048-d box features become operand planes ONCE per wave (zk's FFN buffer; zk itself runs last)
        Planes featp = ens_lanes ? z->ens_featp : z->mid;
        launch_split_f32(b->feats + p0 * MMS_NBOX * MMS_FEAT, featp.hi, featp.lo, NB * MMS_FEAT, st);
        hipStream_t st_l = st, st_x = st;
        if (ens_lanes) {      // lds on its handle's side stream, lxmert on zk's (lxmert's own two lanes fork from there)
            st_l = l->side; st_x = z->side;
            z->lane_forks += 1;
            HIP_TRY(z, hipEventRecord(z->ev_fork, st));
            HIP_TRY(z, hipStreamWaitEvent(st_l, z->ev_fork, 0));
            HIP_TRY(z, hipStreamWaitEvent(st_x, z->ev_fork, 0));
        }
        if (int rc = lds_chunk(l, st_l, &lb, p0, n, lg[2], pr[2], &featp)) return z->fail(rc, "lds member: " + l->err);
        if (int rc = lx_chunk(x, st_x, &xb, index, p0, n, lg[3], pr[3], &featp)) return z->fail(rc, "lxmert member: " + x->err);
        if (int rc = zk_image_tokens(z, st, &zb, index, p0, n, &featp)) return rc;
        HIP_TRY(z, hipMemcpyAsync(z->ens_tok, z->qkv + NB * H, (size_t)NB * H * 4, hipMemcpyDeviceToDevice, st));
        if (int rc = zk_encode(z, st, &zb, p0, n, z->ens_tok, lg[0], pr[0])) return rc;
        const int64_t n2 = changed[p0 / cs];
        if (n2 > 0) {   // the changed pairs of this wave, compacted: rewritten query + their rows of the shared image tokens
            const int* list = z->ens_dlist + p0;
            launch_gather_rows_i32(b->s2f_query_ids + p0 * T, list, T, n2, z->ens_cq, st);
            launch_gather_rows_i32(b->s2f_len_query + p0, list, 1, n2, z->ens_clen, st);
            launch_gather_rows_i32(b->num_boxes + p0, list, 1, n2, z->ens_cnb, st);
            launch_gather_rows_i64(b->labels + p0, list, 1, n2, z->ens_clab, st);
            launch_gather_rows_f32x4(z->ens_tok, list, MMS_NBOX * H, n2, z->ens_ctok, st);
            mms_zk_batch zc = zb2;
            zc.n_pairs = n2; zc.query_ids = z->ens_cq; zc.len_query = z->ens_clen; zc.num_boxes = z->ens_cnb; zc.labels = z->ens_clab;
            zc.segment_ids = z->ens_seg;                                  // the same 0 x T, 1 x 10 pattern for every pair
            if (int rc = zk_encode(z, st, &zc, 0, n2, z->ens_ctok, z->ens_clog, z->ens_cprob)) return rc;
        }
        launch_select_rows2(z->ens_diff + p0, z->ens_doff + p0, lg[0] + p0 * 2, z->ens_clog, (int)n, lg[1] + p0 * 2, st);
        launch_select_rows2(z->ens_diff + p0, z->ens_doff + p0, pr[0] + p0 * 2, z->ens_cprob, (int)n, pr[1] + p0 * 2, st);
        if (ens_lanes) {      // the next wave's feature split and the merge wait for the other two members
            HIP_TRY(z, hipEventRecord(l->ev_join, st_l));
            HIP_TRY(z, hipStreamWaitEvent(st, l->ev_join, 0));
            HIP_TRY(z, hipEventRecord(z->ev_join, st_x));
            HIP_TRY(z, hipStreamWaitEvent(st, z->ev_join, 0));
        }
    }
    const float* prc[4] = {pr[0], pr[1], pr[2], pr[3]};
    launch_merge4(prc, weights4, merged, member_scores, B, st);
    return guard.done(post_launch(z));
}

int mms_gemm_timing(mms_handle* h, int32_t enable, int32_t reset, double* ms_out, int64_t* launches_out, double* flops_out) {
    if (!h) return MMS_ERR_ARG;
    DeviceScope dev(h->cfg.device);
    if (!h->flop_counter) {
        void* p;
        if (int rc = dev_alloc(h, h->w_allocs, &p, 32)) return rc;      // [0]: GEMM launches, [1]: fused QKV + attention launches, [2]: LayerNorm-fused GEMM launches, [3]: side-lane launches (untimed)
        h->flop_counter = (unsigned long long*)p;
        HIP_TRY(h, hipMemset(p, 0, 32));
    }
    if (ms_out || launches_out || flops_out) {
        double ms = 0;
        if (h->ev_used) HIP_TRY(h, hipEventSynchronize(h->ev[h->ev_used - 1]));
        for (size_t i = 0; i + 1 < h->ev_used; i += 2) {
            float t = 0;
            HIP_TRY(h, hipEventElapsedTime(&t, h->ev[i], h->ev[i + 1]));
            ms += t;
        }
        unsigned long long fl[3] = {0, 0, 0};
        HIP_TRY(h, hipMemcpy(fl, h->flop_counter, 24, hipMemcpyDeviceToHost));
        if (ms_out) *ms_out = ms;
        if (launches_out) *launches_out = h->gemm_launches;
        if (flops_out) *flops_out = (double)fl[0] + (double)fl[2];
    }
    if (reset) { h->ev_used = 0; h->gemm_launches = 0; h->ev_fused_used = 0; h->fused_timed = 0; HIP_TRY(h, hipMemset(h->flop_counter, 0, 32)); }
    h->timing = enable != 0;
    return MMS_OK;
}

// the LayerNorm-fused share of what mms_gemm_timing reports (those launches' duration includes the residual stages and the LayerNorm work):
// duration, launches and executed FLOPs of the timed launches of class `cls` (0: plain epilogue, 1: fused LayerNorm epilogue)
int mms_gemm_timing_class(mms_handle* h, int32_t cls, double* ms_out, int64_t* launches_out, double* flops_out) {
    if (!h || cls < 0 || cls > 1) return MMS_ERR_ARG;
    DeviceScope dev(h->cfg.device);
    double ms = 0;
    int64_t n = 0;
    if (h->ev_used) HIP_TRY(h, hipEventSynchronize(h->ev[h->ev_used - 1]));
    for (size_t i = 0; i + 1 < h->ev_used; i += 2) {
        if (h->ev_cls[i / 2] != cls) continue;
        float t = 0;
        HIP_TRY(h, hipEventElapsedTime(&t, h->ev[i], h->ev[i + 1]));
        ms += t; n += 1;
    }
    unsigned long long fl = 0;
    if (h->flop_counter) HIP_TRY(h, hipMemcpy(&fl, h->flop_counter + (cls ? 2 : 0), 8, hipMemcpyDeviceToHost));
    if (ms_out) *ms_out = ms;
    if (launches_out) *launches_out = n;
    if (flops_out) *flops_out = (double)fl;
    return MMS_OK;
}

// the fused QKV + attention launches (mms_config.fuse_attention) of the calls timed by mms_gemm_timing, which neither counts nor resets
// them separately: total duration, launches, executed projection FLOPs (2 * M_live * 2304 * 768 per launch)
int mms_fused_timing(mms_handle* h, double* ms_out, int64_t* launches_out, double* flops_out) {
    if (!h) return MMS_ERR_ARG;
    DeviceScope dev(h->cfg.device);
    double ms = 0;
    if (h->ev_fused_used) HIP_TRY(h, hipEventSynchronize(h->ev_fused[h->ev_fused_used - 1]));
    for (size_t i = 0; i + 1 < h->ev_fused_used; i += 2) {
        float t = 0;
        HIP_TRY(h, hipEventElapsedTime(&t, h->ev_fused[i], h->ev_fused[i + 1]));
        ms += t;
    }
    unsigned long long fl = 0;
    if (h->flop_counter) HIP_TRY(h, hipMemcpy(&fl, h->flop_counter + 1, 8, hipMemcpyDeviceToHost));
    if (ms_out) *ms_out = ms;
    if (launches_out) *launches_out = h->fused_timed;
    if (flops_out) *flops_out = (double)fl;
    return MMS_OK;
}

// executed FLOPs of the launches that ran on the side lane since the last mms_gemm_timing reset (lxmert's distinct-query stage beside the box stream's layers):
// part of the step's work, in no per-launch sum
int mms_side_lane_flops(mms_handle* h, double* flops_out) {
    if (!h || !flops_out) return MMS_ERR_ARG;
    DeviceScope dev(h->cfg.device);
    unsigned long long fl = 0;
    if (h->flop_counter) HIP_TRY(h, hipMemcpy(&fl, h->flop_counter + 3, 8, hipMemcpyDeviceToHost));
    *flops_out = (double)fl;
    return MMS_OK;
}

int mms_debug_read_x(mms_handle* h, float* dst_dev, int64_t rows, void* stream) {
    if (!h || !dst_dev) return MMS_ERR_ARG;
    if (!h->ws_pairs) return h->fail(MMS_ERR_STATE, "no forward has run yet");
    if (rows < 0 || rows > h->ws_pairs * h->x_rows) return h->fail(MMS_ERR_ARG, "rows exceeds the workspace");
    launch_planes_to_f32(h->x.hi, h->x.lo, dst_dev, rows * H, (hipStream_t)stream);
    return post_launch(h);
}

// ---- kernel-level test hooks: fp32 in / fp32 out around the production kernels --------------------
static int dbg_fail(const char* m) { g_err = m; return MMS_ERR_HIP; }
#define DBG_TRY(expr) do { if ((expr) != hipSuccess) return dbg_fail(#expr); } while (0)

int mms_dbg_gemm(const float* a_f32, int64_t M, int64_t K, int64_t lda, const float* w_f32_nk, int64_t N, const float* bias,
                 const float* resid_f32, int32_t act, int32_t nsplit, int32_t out_planes, int32_t engine, float* c_f32, void* stream) {
    if (!a_f32 || !w_f32_nk || !c_f32 || M <= 0 || N % 128 || K % 64 || lda < K || lda % 32 || engine < 0) { g_err = "mms_dbg_gemm: bad argument"; return MMS_ERR_ARG; }
    hipStream_t st = (hipStream_t)stream;
    bf16 *ap = nullptr, *wp = nullptr, *rp = nullptr, *cp = nullptr;
    float* wtmp = nullptr;
    DBG_TRY(hipMalloc((void**)&ap, (size_t)M * lda * 4));
    DBG_TRY(hipMalloc((void**)&wp, (size_t)N * K * 4));
    launch_split_f32(a_f32, ap, ap + MMS_PLANE_LO, M * lda, st);
    launch_tile_weights(w_f32_nk, wp, wp + N * K, N, K, st);  // hi plane == RNE bf16 of the weights (tiled), lo plane behind it
    GemmParams p{};
    p.a_hi = ap; p.a_lo = ap + MMS_PLANE_LO; p.lda = (int)lda; p.amap = RowMap{0, 0, 0}; p.cmap = RowMap{0, 0, 0};
    p.w = wp; p.w_lo = wp + N * K; p.bias = bias; p.M = (int)M; p.N = (int)N; p.K = (int)K; p.act = act;
    p.engine = engine;      // enum Engine (regimes.h): ENG_AUTO = the per-shape choice of the forward; else ONE named engine
    if (resid_f32) {
        DBG_TRY(hipMalloc((void**)&rp, (size_t)M * N * 4));
        launch_split_f32(resid_f32, rp, rp + MMS_PLANE_LO, M * N, st);
        p.r_hi = rp; p.r_lo = rp + MMS_PLANE_LO; p.ldr = (int)N;
    }
    if (out_planes) {
        DBG_TRY(hipMalloc((void**)&cp, (size_t)M * N * 4));
        p.out_kind = OUT_PLANES; p.c_hi = cp; p.c_lo = cp + MMS_PLANE_LO; p.ldp = (int)N;
    } else {
        p.out_kind = OUT_F32; p.c_f32 = c_f32; p.ldc = (int)N;
    }
    const bool taken = launch_gemm(p, nsplit, st);
    if (taken && out_planes) launch_planes_to_f32(cp, cp + MMS_PLANE_LO, c_f32, M * N, st);
    DBG_TRY(hipStreamSynchronize(st));
    DBG_TRY(hipGetLastError());
    (void)hipFree(ap); (void)hipFree(wp); (void)hipFree(rp); (void)hipFree(cp); (void)wtmp;
    if (!taken) { g_err = "mms_dbg_gemm: no engine takes this shape"; return MMS_ERR_ARG; }
    return MMS_OK;
}

int mms_dbg_gemm_f8(const float* a_f32, int64_t M, int64_t K, const float* w_f32_nk, int64_t N, const float* bias, int32_t act,
                    int32_t out_f8, float* c_f32, void* stream) {
    if (!a_f32 || !w_f32_nk || !c_f32 || M <= 0 || N % 256 || K % 128) { g_err = "mms_dbg_gemm_f8: bad argument (N % 256, K % 128)"; return MMS_ERR_ARG; }
    hipStream_t st = (hipStream_t)stream;
    const int64_t Mp = (M + 255) / 256 * 256;
    unsigned char *a8 = nullptr, *w8 = nullptr, *c8 = nullptr;
    unsigned* ws4 = nullptr;
    DBG_TRY(hipMalloc((void**)&a8, (size_t)Mp * K));
    DBG_TRY(hipMemsetAsync(a8, 0, (size_t)Mp * K, st));
    DBG_TRY(hipMalloc((void**)&w8, (size_t)N * K));
    DBG_TRY(hipMalloc((void**)&ws4, (size_t)N));
    launch_f32_to_f8(a_f32, a8, M * K, st);
    launch_quant_rows_f8(w_f32_nk, w8, nullptr, ws4, (int)N, (int)K, st);
    GemmParams p{};
    p.a8 = a8; p.lda = (int)K; p.amap = RowMap{0, 0, 0}; p.cmap = RowMap{0, 0, 0}; p.rmap = RowMap{0, 0, 0};
    p.w8 = w8; p.w8_scale4 = ws4; p.bias = bias; p.M = (int)M; p.N = (int)N; p.K = (int)K; p.act = act;
    if (out_f8) {
        DBG_TRY(hipMalloc((void**)&c8, (size_t)M * N));
        p.out_kind = OUT_F8; p.c_f8 = c8; p.ldf8 = (int)N;
    } else { p.out_kind = OUT_F32; p.c_f32 = c_f32; p.ldc = (int)N; }
    if (!launch_gemm_mx8(p, st)) { g_err = "mms_dbg_gemm_f8: shape not supported"; return MMS_ERR_ARG; }
    if (out_f8) launch_f8_to_f32(c8, c_f32, M * N, st);
    DBG_TRY(hipStreamSynchronize(st));
    DBG_TRY(hipGetLastError());
    (void)hipFree(a8); (void)hipFree(w8); (void)hipFree(ws4); (void)hipFree(c8);
    return MMS_OK;
}

#ifdef MMS_LAB
// precision mode 5 GEMM (gemm_mx.hip) on fp32 operands: A goes through the h3 split (fp16 + e4m3 residual), W through the weight
// preparation of the forward (fp16 copy + e4m3 copy + per-channel scales); out_h3: the result leaves as h3 planes (and is converted back)
int mms_dbg_gemm_mx(const float* a_f32, int64_t M, int64_t K, const float* w_f32_nk, int64_t N, const float* bias, int32_t act,
                    int32_t out_h3, float* c_f32, void* stream) {
    if (!a_f32 || !w_f32_nk || !c_f32 || M <= 0 || N % 256 || K % 256) { g_err = "mms_dbg_gemm_mx: bad argument (N % 256, K % 256)"; return MMS_ERR_ARG; }
    hipStream_t st = (hipStream_t)stream;
    const int64_t Mp = (M + 255) / 256 * 256;
    f16 *a16 = nullptr, *w16 = nullptr, *c16 = nullptr;
    unsigned char *a8 = nullptr, *w8 = nullptr, *c8 = nullptr;
    unsigned* ws4 = nullptr;
    float* cs = nullptr;
    DBG_TRY(hipMalloc((void**)&a16, (size_t)Mp * K * 2)); DBG_TRY(hipMalloc((void**)&a8, (size_t)Mp * K));
    DBG_TRY(hipMemsetAsync(a16, 0, (size_t)Mp * K * 2, st)); DBG_TRY(hipMemsetAsync(a8, 0, (size_t)Mp * K, st));
    DBG_TRY(hipMalloc((void**)&w16, (size_t)N * K * 2)); DBG_TRY(hipMalloc((void**)&w8, (size_t)N * K));
    DBG_TRY(hipMalloc((void**)&ws4, (size_t)N)); DBG_TRY(hipMalloc((void**)&cs, (size_t)N * 4));
    launch_split_h3(a_f32, a16, a8, M * K, st);
    launch_prep_w_mx(w_f32_nk, w16, w8, ws4, cs, (int)N, (int)K, st);
    GemmParams p{};
    p.a_hi = (const bf16*)a16; p.a8 = a8; p.lda = (int)K; p.amap = RowMap{0, 0, 0}; p.cmap = RowMap{0, 0, 0}; p.rmap = RowMap{0, 0, 0};
    p.w = (const bf16*)w16; p.w8 = w8; p.w8_scale4 = ws4; p.col_scale = cs; p.bias = bias; p.M = (int)M; p.N = (int)N; p.K = (int)K; p.act = act;
    if (out_h3) {
        DBG_TRY(hipMalloc((void**)&c16, (size_t)M * N * 2)); DBG_TRY(hipMalloc((void**)&c8, (size_t)M * N));
        p.out_kind = OUT_H3; p.c_h16 = c16; p.c_l8 = c8; p.ldh = (int)N;
    } else { p.out_kind = OUT_F32; p.c_f32 = c_f32; p.ldc = (int)N; }
    if (!launch_gemm_mx(p, st)) { g_err = "mms_dbg_gemm_mx: shape not supported"; return MMS_ERR_ARG; }
    if (out_h3) launch_h3_to_f32(c16, c8, c_f32, M * N, st);
    DBG_TRY(hipStreamSynchronize(st));
    DBG_TRY(hipGetLastError());
    for (void* q : {(void*)a16, (void*)a8, (void*)w16, (void*)w8, (void*)ws4, (void*)cs, (void*)c16, (void*)c8}) (void)hipFree(q);
    return MMS_OK;
}

#endif  // MMS_LAB

#ifdef MMS_LAB
unsigned long long* g_ln_dbg = nullptr;   // lab: device buffer for the per-tile phase stamps of the next mms_dbg_gemm_ln (tools/ln_trace.py)
int mms_lab_ln_trace(unsigned long long* dev_buf) { g_ln_dbg = dev_buf; return MMS_OK; }
#endif
int mms_dbg_gemm_ln(const float* a_f32, int64_t M, int64_t K, const float* w_f32_nk, const float* bias, const float* resid_f32,
                    const float* gamma, const float* beta, int32_t f8, float* c_f32, int32_t* mode_out, void* stream) {
    const int64_t N = H;
    if (!a_f32 || !w_f32_nk || !resid_f32 || !gamma || !beta || !c_f32 || M <= 0 || K % 128) { g_err = "mms_dbg_gemm_ln: bad argument"; return MMS_ERR_ARG; }
    hipStream_t st = (hipStream_t)stream;
    bf16 *ap = nullptr, *wp = nullptr, *rp = nullptr, *cp = nullptr;
    unsigned char *a8 = nullptr, *w8 = nullptr;
    float *ws = nullptr, *t = nullptr, *stats = nullptr;
    int* ctl = nullptr;
    DBG_TRY(hipMalloc((void**)&ap, (size_t)M * K * 4)); DBG_TRY(hipMalloc((void**)&wp, (size_t)N * K * 4));
    DBG_TRY(hipMalloc((void**)&rp, (size_t)M * N * 4)); DBG_TRY(hipMalloc((void**)&cp, (size_t)M * N * 4));
    DBG_TRY(hipMalloc((void**)&t, (size_t)M * N * 4)); DBG_TRY(hipMalloc((void**)&stats, (size_t)(M + 256) * 48));
    DBG_TRY(hipMalloc((void**)&ctl, 8));
    DBG_TRY(hipMemsetAsync(stats, 0, (size_t)(M + 256) * 48, st)); DBG_TRY(hipMemsetAsync(ctl, 0, 8, st));
    launch_split_f32(a_f32, ap, ap + MMS_PLANE_LO, M * K, st);
    launch_tile_weights(w_f32_nk, wp, wp + N * K, N, K, st);
    launch_split_f32(resid_f32, rp, rp + MMS_PLANE_LO, M * N, st);
    GemmParams p{};
    if (f8) { g_err = "mms_dbg_gemm_ln: the fused LayerNorm epilogue exists for the bf16 two-pass engine only"; return MMS_ERR_ARG; }
    p.a_hi = ap; p.a_lo = ap + MMS_PLANE_LO; p.lda = (int)K; p.w = wp; p.K = (int)K;
    p.amap = RowMap{0, 0, 0}; p.cmap = RowMap{0, 0, 0}; p.rmap = RowMap{0, 0, 0};
    p.bias = bias; p.M = (int)M; p.N = (int)N; p.act = ACT_NONE;
    p.r_hi = rp; p.r_lo = rp + MMS_PLANE_LO; p.ldr = (int)N;
    p.c_hi = cp; p.c_lo = cp + MMS_PLANE_LO; p.ldp = (int)N;
    p.out_kind = OUT_F32; p.c_f32 = t; p.ldc = (int)N;
    p.ln_gamma = gamma; p.ln_beta = beta; p.ln_stats = stats; p.ln_tag = 0x5EED0001u; p.ln_ctl = ctl;
#ifdef MMS_LAB
    p.ln_dbg = g_ln_dbg;
#endif
    if (!launch_gemm_pp_ln(p, 2, st)) { g_err = "mms_dbg_gemm_ln: shape not supported"; return MMS_ERR_ARG; }
    LnResid r;
    r.hi = rp; r.lo = rp + MMS_PLANE_LO; r.ld = (int)N; r.skip = ctl + 1;
    launch_ln_to_planes(t, (int)N, gamma, beta, cp, cp + MMS_PLANE_LO, (int)N, (int)M, st, nullptr, r);
    launch_planes_to_f32(cp, cp + MMS_PLANE_LO, c_f32, M * N, st);
    int ctl_h[2] = {0, 0};
    DBG_TRY(hipMemcpyAsync(ctl_h, ctl, 8, hipMemcpyDeviceToHost, st));
    DBG_TRY(hipStreamSynchronize(st));
    DBG_TRY(hipGetLastError());
    if (mode_out) *mode_out = ctl_h[1];
    for (void* q : {(void*)ap, (void*)wp, (void*)rp, (void*)cp, (void*)a8, (void*)w8, (void*)ws, (void*)t, (void*)stats, (void*)ctl}) (void)hipFree(q);
    return MMS_OK;
}

// out = LayerNorm(A W^T + bias + resid) over N = 768 on the SMALL-launch route: the register-staged tile engine with the contraction split
// `splits` ways (1, or a divisor of K / 64: fp32 partials) and the LayerNorm kernel that sums the partials, adds bias + residual and normalises
// -- the kernel pair proj_ln() launches for calls of a few hundred pairs
int mms_dbg_proj_ln_splitk(const float* a_f32, int64_t M, int64_t K, const float* w_f32_nk, const float* bias, const float* resid_f32,
                           const float* gamma, const float* beta, int32_t splits, float* c_f32, void* stream) {
    const int64_t N = H;
    const bool skinny_parts = splits < 0;      // negative: the partials come from gemm_skinny.hip (K slices dealt to workgroups; M <= 512) instead of the tile engine
    if (skinny_parts) splits = -splits;
    if (!a_f32 || !w_f32_nk || !resid_f32 || !gamma || !beta || !c_f32 || M <= 0 || splits == 0 || K % (64 * (splits < 0 ? -splits : splits))) { g_err = "mms_dbg_proj_ln_splitk: bad argument"; return MMS_ERR_ARG; }
    hipStream_t st = (hipStream_t)stream;
    bf16 *ap = nullptr, *wp = nullptr, *rp = nullptr, *cp = nullptr;
    float* parts = nullptr;
    DBG_TRY(hipMalloc((void**)&ap, (size_t)M * K * 4)); DBG_TRY(hipMalloc((void**)&wp, (size_t)N * K * 4));
    DBG_TRY(hipMalloc((void**)&rp, (size_t)M * N * 4)); DBG_TRY(hipMalloc((void**)&cp, (size_t)M * N * 4));
    DBG_TRY(hipMalloc((void**)&parts, (size_t)splits * M * N * 4));
    launch_split_f32(a_f32, ap, ap + MMS_PLANE_LO, M * K, st);
    launch_tile_weights(w_f32_nk, wp, wp + N * K, N, K, st);
    launch_split_f32(resid_f32, rp, rp + MMS_PLANE_LO, M * N, st);
    GemmParams p{};
    p.a_hi = ap; p.a_lo = ap + MMS_PLANE_LO; p.lda = (int)K; p.amap = RowMap{0, 0, 0}; p.cmap = RowMap{0, 0, 0}; p.rmap = RowMap{0, 0, 0};
    p.w = wp; p.M = (int)M; p.N = (int)N; p.K = (int)K; p.act = ACT_NONE;
    p.out_kind = OUT_F32; p.c_f32 = parts; p.ldc = (int)N;
    p.k_splits = splits; p.c_split_stride = (long long)M * N;
    p.engine = skinny_parts && splits > 1 ? ENG_SKINNY_PARTS : ENG_TILE;
    if (splits == 1) p.bias = bias;
    launch_gemm(p, 2, st);
    LnResid r;
    r.hi = rp; r.lo = rp + MMS_PLANE_LO; r.ld = (int)N;
    if (splits > 1) { r.nparts = splits; r.part_stride = p.c_split_stride; r.bias = bias; }
    launch_ln_to_planes(parts, (int)N, gamma, beta, cp, cp + MMS_PLANE_LO, (int)N, (int)M, st, nullptr, r);
    launch_planes_to_f32(cp, cp + MMS_PLANE_LO, c_f32, M * N, st);
    DBG_TRY(hipStreamSynchronize(st));
    DBG_TRY(hipGetLastError());
    for (void* q : {(void*)ap, (void*)wp, (void*)rp, (void*)cp, (void*)parts}) (void)hipFree(q);
    return MMS_OK;
}

__global__ void k_fill_random(float* p, long long n, unsigned seed) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        unsigned x = (unsigned)(i * 2654435761u) ^ seed;
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        p[i] = (float)(int)x * (1.0f / 2147483648.0f);   // uniform [-1, 1)
    }
}

int64_t mms_dbg_counter(mms_handle* h, int32_t which) {
    if (!h) return -1;
    return which == 0 ? h->fused_attn_launches : which == 1 ? h->ln_fused_launches : which == 2 ? h->splitk_launches : which == 3 ? h->skinny_launches : which == 4 ? h->lane_forks : -1;
}

// GEMM micro-benchmark on random operands: returns the average kernel time (ms) over `iters` launches.
// mms_dbg_gemm_bench's `what` beyond the engines of regimes.h: other kernels timed on the same random operands (tools/gemm_bench.py)
enum BenchMode : int {
    BENCH_MX = 50,               // lab: the fp16 + MX-fp8 "1.5 pass" engine (gemm_mx.hip)
    BENCH_MX_HI = 51,            // lab: its high pass alone
    BENCH_MX8 = 52,              // precision mode 4: gemm_mx8_kernel (e4m3 x e4m3)
    BENCH_LN_RESID = 60,         // the LayerNorm kernel with the residual planes on M rows
    BENCH_LN = 61,               // ... without
    BENCH_PROJ_LN_2K = 62,       // N = 768 projection + residual + LayerNorm as two kernels (fp32 tensor in between)
    BENCH_PROJ_LN_FUSED = 63,    // ... as ONE launch with the fused epilogue
};
int mms_dbg_gemm_bench(int64_t M, int64_t N, int64_t K, int32_t nsplit, int32_t act, int32_t out_planes, int32_t resid,
                       int32_t what, int32_t iters, float* ms_out) {      // what: an Engine (regimes.h) or a BenchMode below
    if (M <= 0 || N % 128 || K % 64 || iters <= 0 || !ms_out) { g_err = "mms_dbg_gemm_bench: bad argument"; return MMS_ERR_ARG; }
    float *af = nullptr, *wf = nullptr, *rf = nullptr, *bias = nullptr, *cf = nullptr;
    bf16 *ap = nullptr, *wp = nullptr, *rp = nullptr, *cp = nullptr;
    DBG_TRY(hipMalloc((void**)&af, (size_t)M * K * 4)); DBG_TRY(hipMalloc((void**)&wf, (size_t)N * K * 4));
    DBG_TRY(hipMalloc((void**)&rf, (size_t)M * N * 4)); DBG_TRY(hipMalloc((void**)&bias, (size_t)N * 4));
    DBG_TRY(hipMalloc((void**)&ap, (size_t)M * K * 4)); DBG_TRY(hipMalloc((void**)&wp, (size_t)N * K * 4));
    DBG_TRY(hipMalloc((void**)&rp, (size_t)M * N * 4)); DBG_TRY(hipMalloc((void**)&cp, (size_t)M * N * 4));
    DBG_TRY(hipMalloc((void**)&cf, (size_t)M * N * 4));
#ifdef MMS_LAB
    const bool zero_fill = getenv("MMS_GB_ZERO") != nullptr;   // DVFS probe: zero operands draw less power (never a bench number)
#else
    const bool zero_fill = false;
#endif
    if (zero_fill) { DBG_TRY(hipMemset(af, 0, (size_t)M * K * 4)); DBG_TRY(hipMemset(wf, 0, (size_t)N * K * 4)); }
    else
    hipLaunchKernelGGL(k_fill_random, dim3(4096), dim3(256), 0, 0, af, M * K, 1u);
    if (!zero_fill) hipLaunchKernelGGL(k_fill_random, dim3(4096), dim3(256), 0, 0, wf, N * K, 2u);
    hipLaunchKernelGGL(k_fill_random, dim3(4096), dim3(256), 0, 0, rf, M * N, 3u);
    hipLaunchKernelGGL(k_fill_random, dim3(64), dim3(256), 0, 0, bias, N, 4u);
    launch_split_f32(af, ap, ap + MMS_PLANE_LO, M * K, 0);
    launch_tile_weights(wf, wp, wp + N * K, N, K, 0);
    launch_split_f32(rf, rp, rp + MMS_PLANE_LO, M * N, 0);
    GemmParams p{};
    p.a_hi = ap; p.a_lo = ap + MMS_PLANE_LO; p.lda = (int)K; p.amap = RowMap{0, 0, 0}; p.cmap = RowMap{0, 0, 0};
    p.w = wp; p.w_lo = wp + N * K; p.bias = bias; p.M = (int)M; p.N = (int)N; p.K = (int)K; p.act = act;
    if (resid) { p.r_hi = rp; p.r_lo = rp + MMS_PLANE_LO; p.ldr = (int)N; }
    if (out_planes) { p.out_kind = OUT_PLANES; p.c_hi = cp; p.c_lo = cp + MMS_PLANE_LO; p.ldp = (int)N; }
    else { p.out_kind = OUT_F32; p.c_f32 = cf; p.ldc = (int)N; }
    p.engine = engine_is_named(what) ? what : ENG_AUTO;
    hipEvent_t e0, e1;
    DBG_TRY(hipEventCreate(&e0)); DBG_TRY(hipEventCreate(&e1));
    // BENCH_MX / BENCH_MX_HI: the precision-5 engine (gemm_mx.hip) / its high pass alone (lab) on the same random operands
    f16 *a16 = nullptr, *w16 = nullptr, *c16 = nullptr;
    unsigned char *a8 = nullptr, *w8 = nullptr, *c8 = nullptr;
    unsigned* ws4 = nullptr;
    float* cs = nullptr;
    GemmParams pm{};
    const bool mx = what == BENCH_MX || what == BENCH_MX_HI || what == BENCH_MX8;      // 52: precision mode 4 (gemm_mx8_kernel, e4m3 x e4m3)
    if (what == BENCH_MX8) {
        if (N % 256 || K % 128 || resid) { g_err = "mms_dbg_gemm_bench: mx8 engine needs N % 256 == 0, K % 128 == 0, no residual"; return MMS_ERR_ARG; }
        const int64_t Mp = (M + 255) / 256 * 256;
        DBG_TRY(hipMalloc((void**)&a8, (size_t)Mp * K)); DBG_TRY(hipMemset(a8, 0, (size_t)Mp * K));
        DBG_TRY(hipMalloc((void**)&w8, (size_t)N * K)); DBG_TRY(hipMalloc((void**)&ws4, (size_t)N)); DBG_TRY(hipMalloc((void**)&c8, (size_t)M * N));
        launch_f32_to_f8(af, a8, M * K, 0);
        launch_quant_rows_f8(wf, w8, nullptr, ws4, (int)N, (int)K, 0);
        pm = p;
        pm.a8 = a8; pm.w8 = w8; pm.w8_scale4 = ws4; pm.w_lo = nullptr;
        if (out_planes) { pm.out_kind = OUT_F8; pm.c_f8 = c8; pm.ldf8 = (int)N; }
    } else if (mx) {
        if (N % 256 || K % 256 || resid) { g_err = "mms_dbg_gemm_bench: mx engine needs N % 256 == 0, K % 256 == 0, no residual"; return MMS_ERR_ARG; }
        const int64_t Mp = (M + 255) / 256 * 256;
        DBG_TRY(hipMalloc((void**)&a16, (size_t)Mp * K * 2)); DBG_TRY(hipMalloc((void**)&a8, (size_t)Mp * K));
        DBG_TRY(hipMemset(a16, 0, (size_t)Mp * K * 2)); DBG_TRY(hipMemset(a8, 0, (size_t)Mp * K));
        DBG_TRY(hipMalloc((void**)&w16, (size_t)N * K * 2)); DBG_TRY(hipMalloc((void**)&w8, (size_t)N * K));
        DBG_TRY(hipMalloc((void**)&ws4, (size_t)N)); DBG_TRY(hipMalloc((void**)&cs, (size_t)N * 4));
        DBG_TRY(hipMalloc((void**)&c16, (size_t)M * N * 2)); DBG_TRY(hipMalloc((void**)&c8, (size_t)M * N));
        launch_split_h3(af, a16, a8, M * K, 0);
        launch_prep_w_mx(wf, w16, w8, ws4, cs, (int)N, (int)K, 0);
        pm = p;
        pm.a_hi = (const bf16*)a16; pm.a8 = a8; pm.w = (const bf16*)w16; pm.w8 = w8; pm.w8_scale4 = ws4; pm.col_scale = cs; pm.w_lo = nullptr;
        if (out_planes) { pm.out_kind = OUT_H3; pm.c_h16 = c16; pm.c_l8 = c8; pm.ldh = (int)N; }
    }
    // BENCH_PROJ_LN_2K / BENCH_PROJ_LN_FUSED: the N = 768 projection + residual + LayerNorm as two kernels (fp32 tensor in between) / as ONE launch with the
    // fused epilogue (gemm_pp_ln.h); both leave split planes
    float* ln_stats = nullptr; int* ln_ctl = nullptr; unsigned ln_tag = 0x1000u;
    if (what == BENCH_PROJ_LN_2K || what == BENCH_PROJ_LN_FUSED) {
        if (N != H || nsplit != 2) { g_err = "mms_dbg_gemm_bench: the projection + LayerNorm modes need N == 768, nsplit == 2"; return MMS_ERR_ARG; }
        DBG_TRY(hipMalloc((void**)&ln_stats, (size_t)(M + 256) * 48)); DBG_TRY(hipMemset(ln_stats, 0, (size_t)(M + 256) * 48));
        DBG_TRY(hipMalloc((void**)&ln_ctl, 8));
    }
    auto run = [&]() {
        if (what == BENCH_PROJ_LN_2K || what == BENCH_PROJ_LN_FUSED) {
            GemmParams q = p;
            q.out_kind = OUT_F32; q.c_f32 = cf; q.ldc = H; q.r_hi = nullptr; q.r_lo = nullptr;
            LnResid res;
            res.hi = rp; res.lo = rp + MMS_PLANE_LO; res.ld = H;
            if (what == BENCH_PROJ_LN_2K) {
                launch_gemm(q, 2, 0);
                launch_ln_to_planes(cf, H, bias, bias, cp, cp + MMS_PLANE_LO, H, (int)M, 0, nullptr, res);
                return true;
            }
            q.r_hi = rp; q.r_lo = rp + MMS_PLANE_LO; q.ldr = H; q.c_hi = cp; q.c_lo = cp + MMS_PLANE_LO; q.ldp = H;
            q.ln_gamma = bias; q.ln_beta = bias; q.ln_stats = ln_stats; q.ln_tag = ++ln_tag; q.ln_ctl = ln_ctl;
            if (hipMemsetAsync(ln_ctl, 0, 8, 0) != hipSuccess) return false;
            if (!launch_gemm_pp_ln(q, 2, 0)) return false;
            res.skip = ln_ctl + 1;
            launch_ln_to_planes(cf, H, bias, bias, cp, cp + MMS_PLANE_LO, H, (int)M, 0, nullptr, res);
            return true;
        }
        if (what == BENCH_LN_RESID || what == BENCH_LN) {     // the LayerNorm kernel (with / without the residual planes) on M rows: HBM-bound, 9 / 6 KB per row
            if (N != H) return false;
            LnResid res;
            if (what == BENCH_LN_RESID) { res.hi = rp; res.lo = rp + MMS_PLANE_LO; res.ld = H; }
            launch_ln_to_planes(cf, H, bias, bias, cp, cp + MMS_PLANE_LO, H, (int)M, 0, nullptr, res);
            return true;
        }
        if (!mx) { launch_gemm(p, nsplit, 0); return true; }
#ifdef MMS_LAB
        if (what == BENCH_MX_HI) return launch_gemm_mx_hi_only(pm, 0);
#endif
        if (what == BENCH_MX8) return launch_gemm_mx8(pm, 0);
#ifdef MMS_LAB
        return launch_gemm_mx(pm, 0);
#else
        return false;     // 50 / 51: lab build only
#endif
    };
    if (!run()) { g_err = "mms_dbg_gemm_bench: engine / mode not available"; return MMS_ERR_ARG; }
    run();
    DBG_TRY(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) run();
    DBG_TRY(hipEventRecord(e1, 0));
    DBG_TRY(hipEventSynchronize(e1));
    float ms = 0;
    DBG_TRY(hipEventElapsedTime(&ms, e0, e1));
    *ms_out = ms / iters;
    DBG_TRY(hipGetLastError());
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    for (void* q : {(void*)af, (void*)wf, (void*)rf, (void*)bias, (void*)cf, (void*)ap, (void*)wp, (void*)rp, (void*)cp, (void*)a16, (void*)a8,
                    (void*)w16, (void*)w8, (void*)ws4, (void*)cs, (void*)c16, (void*)c8, (void*)ln_stats, (void*)ln_ctl}) (void)hipFree(q);
    return MMS_OK;
}

int mms_dbg_attention(const float* q, const float* k, const float* v, int64_t B, int32_t Sq, int32_t Sk, const float* key_add,
                      float* out_f32, void* stream) {
    if (!q || !k || !v || !out_f32 || B <= 0 || Sq <= 0 || Sk <= 0 || Sq > 48 || Sk > 48) { g_err = "mms_dbg_attention: bad argument"; return MMS_ERR_ARG; }
    hipStream_t st = (hipStream_t)stream;
    bf16* op = nullptr;
    const int64_t n = B * Sq * H;
    DBG_TRY(hipMalloc((void**)&op, (size_t)n * 4));
    AttnParams a{};
    a.q = q; a.ldq = H; a.k = k; a.v = v; a.ldkv = H; a.hs_q = a.hs_kv = MMS_HEAD_DIM; a.q_base = 0; a.Sq = Sq; a.kv_base = 0; a.Sk = Sk;
    a.key_add = key_add; a.o_hi = op; a.o_lo = op + MMS_PLANE_LO; a.ldo = H; a.B = (int)B;
    if (!launch_attention(a, st)) { (void)hipFree(op); g_err = "mms_dbg_attention: no kernel for this (Sq, Sk)"; return MMS_ERR_ARG; }
    launch_planes_to_f32(op, op + MMS_PLANE_LO, out_f32, n, st);
    DBG_TRY(hipStreamSynchronize(st));
    DBG_TRY(hipGetLastError());
    (void)hipFree(op);
    return MMS_OK;
}

int mms_dbg_qkv_attn(const float* x, int64_t rows1, int64_t rows2, const int32_t* off1, const int32_t* cnt1, const int32_t* off2, const int32_t* cnt2,
                     int64_t n_pairs, int32_t S1, int32_t S2, const float* w_qkv, const float* bias_qkv, const float* key_add1, const float* key_add2,
                     int32_t mode, float* ctx_f32, int32_t* n_sub_out, void* stream) {
    if (!x || !w_qkv || !bias_qkv || !ctx_f32 || rows1 <= 0 || rows2 < 0 || n_pairs <= 0 || S1 <= 0 || (rows2 > 0 && S2 <= 0) || (mode != 1 && mode != 2) ||
        (off1 == nullptr) != (cnt1 == nullptr) || (off2 == nullptr) != (cnt2 == nullptr)) { g_err = "mms_dbg_qkv_attn: bad argument"; return MMS_ERR_ARG; }
    hipStream_t st = (hipStream_t)stream;
    const bool cross = rows2 > 0;
    const int64_t rows = rows1 + rows2;
    bf16 *xp = nullptr, *cp = nullptr, *wt = nullptr;
    float *whm = nullptr, *bhm = nullptr;
    int4 *sub = nullptr, *sub2 = nullptr;
    int *nsub = nullptr, *rec = nullptr, *scratch = nullptr;
    auto cleanup = [&]() {
        (void)hipFree(xp); (void)hipFree(cp); (void)hipFree(wt); (void)hipFree(whm); (void)hipFree(bhm); (void)hipFree(sub); (void)hipFree(sub2);
        (void)hipFree(nsub); (void)hipFree(rec); (void)hipFree(scratch);
    };
#define QA_TRY(expr) do { if ((expr) != hipSuccess) { cleanup(); return dbg_fail(#expr); } } while (0)
    QA_TRY(hipMalloc((void**)&xp, (size_t)rows * H * 4));
    QA_TRY(hipMalloc((void**)&cp, (size_t)rows * H * 4));
    QA_TRY(hipMalloc((void**)&wt, (size_t)3 * H * H * 2));
    QA_TRY(hipMalloc((void**)&whm, (size_t)3 * H * H * 4));
    QA_TRY(hipMalloc((void**)&bhm, (size_t)3 * H * 4));
    QA_TRY(hipMalloc((void**)&sub, (size_t)(n_pairs + 2) * sizeof(int4)));
    QA_TRY(hipMalloc((void**)&sub2, (size_t)(n_pairs + 2) * sizeof(int4)));
    QA_TRY(hipMalloc((void**)&nsub, 16));
    QA_TRY(hipMalloc((void**)&rec, (size_t)(n_pairs + 2) * 4));
    QA_TRY(hipMalloc((void**)&scratch, (size_t)qkv_plan_scratch_ints((int)n_pairs) * 4));
    QA_TRY(hipMemsetAsync(cp, 0, (size_t)rows * H * 4, st));
    launch_split_f32(x, xp, xp + MMS_PLANE_LO, rows * H, st);
    // [Wq; Wk; Wv] (torch Linear rows) -> head-major [12][Q 64 | K 64 | V 64] rows, as load_att() builds it for the scorers
    for (int hd = 0; hd < MMS_HEADS; ++hd)
        for (int i = 0; i < 3; ++i) {
            const size_t src = (size_t)i * H + (size_t)hd * MMS_HEAD_DIM, dst = ((size_t)hd * 3 + i) * MMS_HEAD_DIM;
            QA_TRY(hipMemcpyAsync(whm + dst * H, w_qkv + src * H, (size_t)MMS_HEAD_DIM * H * 4, hipMemcpyDeviceToDevice, st));
            QA_TRY(hipMemcpyAsync(bhm + dst, bias_qkv + src, (size_t)MMS_HEAD_DIM * 4, hipMemcpyDeviceToDevice, st));
        }
    launch_tile_weights(whm, wt, nullptr, 3 * H, H, st);
    bool planned;
    if (cross) planned = launch_qkv_cross_plan((const int*)off1, (const int*)cnt1, (const int*)off2, (const int*)cnt2, (int)n_pairs, S1, S2, sub, sub2, nsub, 2, st, rec, scratch);
    else planned = launch_qkv_tile_plan((const int*)off1, (const int*)cnt1, (int)n_pairs, S1, sub, nsub, 2, st, rec, scratch);
    if (!planned) { cleanup(); g_err = "mms_dbg_qkv_attn: too many pairs for one plan"; return MMS_ERR_ARG; }
    QkvAttnParams q{};
    q.a_hi = xp; q.lda = H; q.w = wt; q.bias = bhm; q.K = H;
    q.sub = sub; q.n_sub = nsub; q.pair_rec = rec; q.pair_off = (const int*)off1; q.pair_cnt = (const int*)cnt1; q.S = S1;
    q.key_add = key_add1; q.o_hi = cp; q.o_lo = cp + MMS_PLANE_LO; q.ldo = H; q.M = (int)rows1; q.fast = mode == 2;
    if (cross) { q.sub2 = sub2; q.pair_off2 = (const int*)off2; q.pair_cnt2 = (const int*)cnt2; q.S2 = S2; q.key_add2 = key_add2; q.row0_b = rows1; q.M2 = (int)rows2; }
    if (!launch_qkv_attn(q, st)) { cleanup(); g_err = "mms_dbg_qkv_attn: shape not supported by this route"; return MMS_ERR_ARG; }
    launch_planes_to_f32(cp, cp + MMS_PLANE_LO, ctx_f32, rows * H, st);
    QA_TRY(hipStreamSynchronize(st));
    QA_TRY(hipGetLastError());
    if (n_sub_out) QA_TRY(hipMemcpy(n_sub_out, nsub, 4, hipMemcpyDeviceToHost));
#undef QA_TRY
    cleanup();
    return MMS_OK;
}

int mms_dbg_layernorm(const float* x, const float* gamma, const float* beta, int64_t M, float* out_f32, void* stream) {
    if (!x || !gamma || !beta || !out_f32 || M <= 0) { g_err = "mms_dbg_layernorm: bad argument"; return MMS_ERR_ARG; }
    hipStream_t st = (hipStream_t)stream;
    bf16* op = nullptr;
    const int64_t n = M * H;
    DBG_TRY(hipMalloc((void**)&op, (size_t)n * 4));
    launch_ln_to_planes(x, H, gamma, beta, op, op + MMS_PLANE_LO, H, (int)M, st);
    launch_planes_to_f32(op, op + MMS_PLANE_LO, out_f32, n, st);
    DBG_TRY(hipStreamSynchronize(st));
    DBG_TRY(hipGetLastError());
    (void)hipFree(op);
    return MMS_OK;
}

}  // extern "C"
