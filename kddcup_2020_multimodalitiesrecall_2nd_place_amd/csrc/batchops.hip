// Per-batch bookkeeping kernels (HBM-bound integer work; no GEMM shapes here):
//
//  * label-text de-duplication.  The reference feeds the box class names as dense id tuples -- np_idx_class_labels [B,10,8]
//    (code/imagebert_zk/evaluate_normal.py:146,232), boxes_label_input_ids [B,10,8] (code/lxmert/src/tasks/kdd_model.py:97-100) -- and
//    runs the label-text encoder on every one of the B*10 tuples although a batch holds only as many DISTINCT tuples as there are
//    object classes (33 in the synthetic sets, a few hundred in multimodal_labels.txt).  The encoder depends on the 8-id tuple
//    only, so the library finds the distinct tuples itself: an open-addressing hash table keyed by the tuple (the slot holds the
//    row index of the tuple's first claimant, equality is checked on the ids themselves), then a numbering pass over the
//    claimants, then the per-row index.  Which claimant wins a slot (and so the ORDER of the unique table) depends on atomics;
//    every label-feature row is computed from its tuple alone, so the logits do not.
//  * the fused three-model entry point's feed conversions (mms_score_ensemble, include/mmscore.h): int32 -> int64 ids for the
//    lds / lxmert kernels, the constant zk segment ids 0 x T, 1 x 10 (load_data_v4.py:204), the lxmert visual mask from
//    num_boxes (utils.py:41-48: ones for the real boxes), and the weighted merge of the four member scores (main.py:59).
//  * x_norm of KDDModel.forward (kdd_model.py:204-205).
#include "kernels.h"

namespace {

template <typename T>
__device__ __forceinline__ unsigned tuple_hash(const T* t) {
    unsigned h = 0x9E3779B9u;
#pragma unroll
    for (int i = 0; i < MMS_LABEL_LEN; ++i) {
        h ^= (unsigned)t[i] + 0x7F4A7C15u + (h << 6) + (h >> 2);
        h *= 0x85EBCA6Bu;
        h ^= h >> 13;
    }
    return h;
}

template <typename T>
__device__ __forceinline__ bool tuple_eq(const T* a, const T* b) {
    bool e = true;
#pragma unroll
    for (int i = 0; i < MMS_LABEL_LEN; ++i) e = e && (a[i] == b[i]);
    return e;
}

// rep[r] = row index of the first claimant of r's tuple.  Two levels: the 256 rows of a workgroup (25 pairs' boxes: about as many
// distinct tuples as there are object classes) first agree on a representative through an LDS table, and only the representatives
// probe the global table -- 8x fewer walks over the handful of hot global slots every workgroup of the grid converges on
// (1.7 ms -> per 300 000-row call before, profiles/r02b_bench_ensemble_kernel_stats.csv).
template <typename T>
__global__ __launch_bounds__(256) void k_dedup_insert(const T* ids, int rows, int* slots, unsigned mask, int* rep) {
    __shared__ int tab[512];
    __shared__ int grep[256];
    const int tid = threadIdx.x, r = blockIdx.x * 256 + tid;
    tab[tid] = -1; tab[tid + 256] = -1;
    __syncthreads();
    const bool live = r < rows;
    const T* mine = ids + (long long)(live ? r : 0) * MMS_LABEL_LEN;
    const unsigned hv = tuple_hash(mine);
    int lrep = tid;                       // thread (of this workgroup) that represents my tuple
    if (live) {
        unsigned h = hv & 511u;
        for (;;) {
            int s = tab[h];
            if (s < 0) {
                const int old = atomicCAS(&tab[h], -1, tid);
                s = old < 0 ? tid : old;
            }
            if (s == tid || tuple_eq(ids + (long long)(blockIdx.x * 256 + s) * MMS_LABEL_LEN, mine)) { lrep = s; break; }
            h = (h + 1) & 511u;
        }
    }
    int g = r;
    if (live && lrep == tid) {            // representatives only: global table
        unsigned h = hv & mask;
        for (;;) {
            int s = slots[h];
            if (s < 0) {
                const int old = atomicCAS(&slots[h], -1, r);
                s = old < 0 ? r : old;
            }
            if (s == r || tuple_eq(ids + (long long)s * MMS_LABEL_LEN, mine)) { g = s; break; }
            h = (h + 1) & mask;
        }
    }
    grep[tid] = g;
    __syncthreads();
    if (live) rep[r] = grep[lrep];
}

// claimants take consecutive numbers and write their tuple into the unique tables
template <typename T>
__global__ __launch_bounds__(256) void k_dedup_number(const T* ids, int rows, const int* rep, int* uid, int* counter, int32_t* uniq32,
                                                      int64_t* uniq64) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= rows || rep[r] != r) return;
    const int u = atomicAdd(counter, 1);
    uid[r] = u;
#pragma unroll
    for (int i = 0; i < MMS_LABEL_LEN; ++i) {
        const T v = ids[(long long)r * MMS_LABEL_LEN + i];
        if (uniq32) uniq32[(long long)u * MMS_LABEL_LEN + i] = (int32_t)v;
        if (uniq64) uniq64[(long long)u * MMS_LABEL_LEN + i] = (int64_t)v;
    }
}

__global__ __launch_bounds__(256) void k_dedup_index(int rows, const int* rep, const int* uid, int* index) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r < rows) index[r] = uid[rep[r]];
}

template <typename T>
void dedup(const T* ids, int rows, int* slots, int cap, int* rep, int* uid, int* counter, int32_t* uniq32, int64_t* uniq64, int* index,
           hipStream_t st) {
    (void)hipMemsetAsync(slots, 0xFF, (size_t)cap * 4, st);
    (void)hipMemsetAsync(counter, 0, 4, st);
    const dim3 grid((rows + 255) / 256), block(256);
    hipLaunchKernelGGL((k_dedup_insert<T>), grid, block, 0, st, ids, rows, slots, (unsigned)(cap - 1), rep);
    hipLaunchKernelGGL((k_dedup_number<T>), grid, block, 0, st, ids, rows, rep, uid, counter, uniq32, uniq64);
    hipLaunchKernelGGL(k_dedup_index, grid, block, 0, st, rows, rep, uid, index);
}

// ---- lxmert query de-duplication: key = (input_ids[T], input_mask[T]) of a pair; the language stream of the first l_layers
// depends on nothing else (modeling.py:568-593: lang_feats only meets the image in the x_layers), and a query comes with 8..30
// candidates, so those layers run once per DISTINCT query and their rows are copied to the query's pairs.
__device__ __forceinline__ unsigned qhash(const int64_t* ids, const int64_t* mask, int T) {
    unsigned h = 0x9E3779B9u;
    for (int i = 0; i < T; ++i) {
        h ^= (unsigned)ids[i] * 2u + (unsigned)(mask[i] != 0) + 0x7F4A7C15u + (h << 6) + (h >> 2);
        h *= 0x85EBCA6Bu;
        h ^= h >> 13;
    }
    return h;
}
__device__ __forceinline__ bool qeq(const int64_t* a, const int64_t* am, const int64_t* b, const int64_t* bm, int T) {
    bool e = true;
    for (int i = 0; i < T; ++i) e = e && a[i] == b[i] && am[i] == bm[i];
    return e;
}
// The distinct queries are numbered in the order of their FIRST occurrence in the batch, whatever the order in which the threads reach the hash table: the
// table slot of a key remembers the smallest row that holds it (atomicMin), a prefix count over the "I am that row" flags gives the numbers.  (Numbering by an
// atomic counter made the row order of the distinct-query stage -- and with it the sub-tile a query lands in -- differ from call to call; the split-bf16
// attention route sums a query's keys in an order that depends on its place in the sub-tile, so identical calls were not bit-identical.)
__global__ __launch_bounds__(256) void k_qdedup_insert(const int64_t* ids, const int64_t* mask, int T, int rows, int* slots, unsigned cmask,
                                                       int* slot_of, int* min_row) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const int64_t *mi = ids + (long long)r * T, *mm = mask + (long long)r * T;
    unsigned h = qhash(mi, mm, T) & cmask;
    for (;;) {
        int s = slots[h];
        if (s < 0) {
            const int old = atomicCAS(&slots[h], -1, r);
            s = old < 0 ? r : old;
        }
        if (s == r || qeq(ids + (long long)s * T, mask + (long long)s * T, mi, mm, T)) { slot_of[r] = (int)h; atomicMin(&min_row[h], r); return; }
        h = (h + 1) & cmask;
    }
}
__global__ __launch_bounds__(256) void k_qdedup_flag(int rows, const int* slot_of, const int* min_row, int* flag) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r < rows) flag[r] = min_row[slot_of[r]] == r ? 1 : 0;
}
// index[] holds the exclusive prefix count of the flags: the number of a query's first row IS its distinct-query number; rows_of[u] = that row
__global__ __launch_bounds__(256) void k_qdedup_fill(int rows, const int* slot_of, const int* min_row, int* rows_of, int* index) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const int c = min_row[slot_of[r]];
    if (c == r) rows_of[index[r]] = r;
    else index[r] = index[c];      // (index[c] is the same value before and after row c's own pass)
}
__global__ __launch_bounds__(256) void k_gather_i64_rows(const int64_t* in, const int* rows_of, int T, long long n, int64_t* out) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) out[i] = in[(long long)rows_of[i / T] * T + i % T];
}
// plane rows: dst[dst_base + map(r)] = src[r]  (scatter, idx == nullptr)  or  dst[r] = src[idx[map(r) / T] * T + map(r) % T]  (gather)
__global__ __launch_bounds__(256) void k_rows_scatter(const bf16* s_hi, const bf16* s_lo, const int* map, const int* rows_dev, int max_rows,
                                                      bf16* d_hi, bf16* d_lo) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= max_rows || row >= *rows_dev) return;
    const long long so = (long long)row * MMS_HIDDEN, d = (long long)map[row] * MMS_HIDDEN;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        *reinterpret_cast<bf16x4*>(plane_ptr(d_hi, d + t * 256 + lane * 4)) = *reinterpret_cast<const bf16x4*>(plane_ptr(s_hi, so + t * 256 + lane * 4));
        *reinterpret_cast<bf16x4*>(plane_ptr(d_lo, d + t * 256 + lane * 4)) = *reinterpret_cast<const bf16x4*>(plane_ptr(s_lo, so + t * 256 + lane * 4));
    }
}
__global__ __launch_bounds__(256) void k_rows_pick(const bf16* s_hi, const bf16* s_lo, const int* map, const int* rows_dev, int max_rows,
                                                   bf16* d_hi, bf16* d_lo) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= max_rows || row >= *rows_dev) return;
    const long long so = (long long)map[row] * MMS_HIDDEN, d = (long long)row * MMS_HIDDEN;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        *reinterpret_cast<bf16x4*>(plane_ptr(d_hi, d + t * 256 + lane * 4)) = *reinterpret_cast<const bf16x4*>(plane_ptr(s_hi, so + t * 256 + lane * 4));
        *reinterpret_cast<bf16x4*>(plane_ptr(d_lo, d + t * 256 + lane * 4)) = *reinterpret_cast<const bf16x4*>(plane_ptr(s_lo, so + t * 256 + lane * 4));
    }
}
__global__ __launch_bounds__(256) void k_rows_gather(const bf16* s_hi, const bf16* s_lo, const int* map, const int* idx, int T, const int* rows_dev,
                                                     int max_rows, bf16* d_hi, bf16* d_lo) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= max_rows || row >= *rows_dev) return;
    const int m = map[row];
    const long long so = ((long long)idx[m / T] * T + m % T) * MMS_HIDDEN, d = (long long)row * MMS_HIDDEN;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        *reinterpret_cast<bf16x4*>(plane_ptr(d_hi, d + t * 256 + lane * 4)) = *reinterpret_cast<const bf16x4*>(plane_ptr(s_hi, so + t * 256 + lane * 4));
        *reinterpret_cast<bf16x4*>(plane_ptr(d_lo, d + t * 256 + lane * 4)) = *reinterpret_cast<const bf16x4*>(plane_ptr(s_lo, so + t * 256 + lane * 4));
    }
}

// ---- second zk member of the ensemble: only the pairs whose query the sen2forest rewrite actually changed are re-encoded ----
__global__ __launch_bounds__(256) void k_query_differs(const int32_t* q1, const int32_t* l1, const int32_t* q2, const int32_t* l2, int T, int n,
                                                       int* diff) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= n) return;
    bool d = l1[b] != l2[b];
    for (int i = 0; i < T; ++i) d = d || q1[(long long)b * T + i] != q2[(long long)b * T + i];
    diff[b] = d ? 1 : 0;
}
__global__ __launch_bounds__(256) void k_compact_list(const int* diff, const int* off, int n, int* list) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b < n && diff[b]) list[off[b]] = b;
}
template <typename T>
__global__ __launch_bounds__(256) void k_gather_rows(const T* in, const int* list, int width, long long n, T* out) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) out[i] = in[(long long)list[i / width] * width + i % width];
}
// out[b] = diff[b] ? changed[off[b]] : same[b]   for 2-wide rows (logits / probs)
__global__ __launch_bounds__(256) void k_select_rows2(const int* diff, const int* off, const float* same, const float* changed, int n, float* out) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= n) return;
    const float2 v = diff[b] ? reinterpret_cast<const float2*>(changed)[off[b]] : reinterpret_cast<const float2*>(same)[b];
    reinterpret_cast<float2*>(out)[b] = v;
}

__global__ __launch_bounds__(256) void k_i32_to_i64(const int32_t* in, int64_t* out, long long n) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) out[i] = in[i];
}

__global__ __launch_bounds__(256) void k_fill_i64(int64_t* out, long long n, int64_t v) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) out[i] = v;
}

__global__ __launch_bounds__(256) void k_zk_segment_ids(int32_t* out, long long n, int T, int S) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) out[i] = (int)(i % S) >= T ? 1 : 0;
}

__global__ __launch_bounds__(256) void k_box_mask(const int32_t* num_boxes, float* mask, long long n) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        mask[i] = (int)(i % MMS_NBOX) < num_boxes[i / MMS_NBOX] ? 1.0f : 0.0f;
}

__global__ __launch_bounds__(256) void k_corners(const float* boxes5, float* boxes4, long long n) {   // [B,10,5] -> [B,10,4]
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) boxes4[i] = boxes5[(i >> 2) * 5 + (i & 3)];
}

__global__ __launch_bounds__(256) void k_merge4(const float* p0, const float* p1, const float* p2, const float* p3, float w0, float w1,
                                                float w2, float w3, float* merged, float* members, long long n) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float s0 = p0[2 * i + 1], s1 = p1[2 * i + 1], s2 = p2[2 * i + 1], s3 = p3[2 * i + 1];   // probs[:, 1]
        // main.py:59 evaluates 0.2*a + 0.2*b + 0.3*c + 0.3*d left to right (in double; these are fp32 scores, so the fp32 sum
        // differs from it by < 1e-7, far below the 1e-5 tie tolerance of the uniqueness filter, main.py:83)
        merged[i] = ((w0 * s0 + w1 * s1) + w2 * s2) + w3 * s3;
        if (members) { members[i] = s0; members[n + i] = s1; members[2 * n + i] = s2; members[3 * n + i] = s3; }
    }
}

// x_norm = pooled / max(||pooled||_2, 1e-12)   (kdd_model.py:204-205); pooled arrives as split planes
__global__ __launch_bounds__(256) void k_xnorm(const bf16* hi, const bf16* lo, float* out, int rows) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    float v[12], ss = 0.f;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const bf16x4 h = *reinterpret_cast<const bf16x4*>(plane_ptr(hi, (long long)row * MMS_HIDDEN + t * 256 + lane * 4));
        const bf16x4 l = *reinterpret_cast<const bf16x4*>(plane_ptr(lo, (long long)row * MMS_HIDDEN + t * 256 + lane * 4));
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[t * 4 + e] = join_bf16(h[e], l[e]); ss += v[t * 4 + e] * v[t * 4 + e]; }
    }
    const float inv = 1.0f / fmaxf(sqrtf(wave_sum(ss)), 1e-12f);
#pragma unroll
    for (int t = 0; t < 3; ++t)
        *reinterpret_cast<float4*>(out + (long long)row * MMS_HIDDEN + t * 256 + lane * 4) =
            float4{v[t * 4] * inv, v[t * 4 + 1] * inv, v[t * 4 + 2] * inv, v[t * 4 + 3] * inv};
}

inline dim3 flat_grid(long long n) { long long g = (n + 255) / 256; return dim3((unsigned)(g < 1 ? 1 : (g > 65535 ? 65535 : g))); }

}  // namespace

void launch_label_dedup_i32(const int32_t* ids, int rows, int* slots, int cap, int* rep, int* uid, int* counter, int32_t* uniq32,
                            int64_t* uniq64, int* index, hipStream_t st) {
    if (rows > 0) dedup<int32_t>(ids, rows, slots, cap, rep, uid, counter, uniq32, uniq64, index, st);
}
void launch_label_dedup_i64(const int64_t* ids, int rows, int* slots, int cap, int* rep, int* uid, int* counter, int32_t* uniq32,
                            int64_t* uniq64, int* index, hipStream_t st) {
    if (rows > 0) dedup<int64_t>(ids, rows, slots, cap, rep, uid, counter, uniq32, uniq64, index, st);
}
void launch_query_dedup(const int64_t* ids, const int64_t* mask, int T, int rows, int* slots, int cap, int* rep, int* uid, int* counter,
                        int* rows_of, int* index, hipStream_t st) {
    if (rows <= 0) return;
    // slots: int[2 * cap] -- the hash table and, behind it, the smallest row of every slot's key
    int* min_row = slots + cap;
    (void)hipMemsetAsync(slots, 0xFF, (size_t)cap * 4, st);
    (void)hipMemsetAsync(min_row, 0x7F, (size_t)cap * 4, st);
    const dim3 grid((rows + 255) / 256), block(256);
    hipLaunchKernelGGL(k_qdedup_insert, grid, block, 0, st, ids, mask, T, rows, slots, (unsigned)(cap - 1), rep, min_row);
    hipLaunchKernelGGL(k_qdedup_flag, grid, block, 0, st, rows, rep, min_row, uid);
    launch_plan_scan(uid, rows, index, counter, st);
    hipLaunchKernelGGL(k_qdedup_fill, grid, block, 0, st, rows, rep, min_row, rows_of, index);
}
void launch_gather_i64_rows(const int64_t* in, const int* rows_of, int T, long long n_rows, int64_t* out, hipStream_t st) {
    if (n_rows > 0) hipLaunchKernelGGL(k_gather_i64_rows, flat_grid(n_rows * T), dim3(256), 0, st, in, rows_of, T, n_rows * T, out);
}
void launch_rows_scatter(const bf16* s_hi, const bf16* s_lo, const int* map, const int* rows_dev, int max_rows, bf16* d_hi, bf16* d_lo,
                         hipStream_t st) {
    if (max_rows > 0) hipLaunchKernelGGL(k_rows_scatter, dim3((max_rows + 3) / 4), dim3(256), 0, st, s_hi, s_lo, map, rows_dev, max_rows, d_hi, d_lo);
}
void launch_rows_pick(const bf16* s_hi, const bf16* s_lo, const int* map, const int* rows_dev, int max_rows, bf16* d_hi, bf16* d_lo,
                      hipStream_t st) {
    if (max_rows > 0) hipLaunchKernelGGL(k_rows_pick, dim3((max_rows + 3) / 4), dim3(256), 0, st, s_hi, s_lo, map, rows_dev, max_rows, d_hi, d_lo);
}
void launch_rows_gather(const bf16* s_hi, const bf16* s_lo, const int* map, const int* idx, int T, const int* rows_dev, int max_rows,
                        bf16* d_hi, bf16* d_lo, hipStream_t st) {
    if (max_rows > 0) hipLaunchKernelGGL(k_rows_gather, dim3((max_rows + 3) / 4), dim3(256), 0, st, s_hi, s_lo, map, idx, T, rows_dev, max_rows, d_hi, d_lo);
}
void launch_query_differs(const int32_t* q1, const int32_t* l1, const int32_t* q2, const int32_t* l2, int T, int n, int* diff, hipStream_t st) {
    if (n > 0) hipLaunchKernelGGL(k_query_differs, dim3((n + 255) / 256), dim3(256), 0, st, q1, l1, q2, l2, T, n, diff);
}
void launch_compact_list(const int* diff, const int* off, int n, int* list, hipStream_t st) {
    if (n > 0) hipLaunchKernelGGL(k_compact_list, dim3((n + 255) / 256), dim3(256), 0, st, diff, off, n, list);
}
void launch_gather_rows_i32(const int32_t* in, const int* list, int width, long long n_rows, int32_t* out, hipStream_t st) {
    if (n_rows > 0) hipLaunchKernelGGL((k_gather_rows<int32_t>), flat_grid(n_rows * width), dim3(256), 0, st, in, list, width, n_rows * width, out);
}
void launch_gather_rows_i64(const int64_t* in, const int* list, int width, long long n_rows, int64_t* out, hipStream_t st) {
    if (n_rows > 0) hipLaunchKernelGGL((k_gather_rows<int64_t>), flat_grid(n_rows * width), dim3(256), 0, st, in, list, width, n_rows * width, out);
}
void launch_gather_rows_f32x4(const float* in, const int* list, int width, long long n_rows, float* out, hipStream_t st) {   // width % 4 == 0
    if (n_rows > 0)
        hipLaunchKernelGGL((k_gather_rows<float4>), flat_grid(n_rows * (width / 4)), dim3(256), 0, st, (const float4*)in, list, width / 4,
                           n_rows * (width / 4), (float4*)out);
}
void launch_select_rows2(const int* diff, const int* off, const float* same, const float* changed, int n, float* out, hipStream_t st) {
    if (n > 0) hipLaunchKernelGGL(k_select_rows2, dim3((n + 255) / 256), dim3(256), 0, st, diff, off, same, changed, n, out);
}
void launch_i32_to_i64(const int32_t* in, int64_t* out, long long n, hipStream_t st) {
    if (n > 0) hipLaunchKernelGGL(k_i32_to_i64, flat_grid(n), dim3(256), 0, st, in, out, n);
}
void launch_fill_i64(int64_t* out, long long n, int64_t v, hipStream_t st) {
    if (n > 0) hipLaunchKernelGGL(k_fill_i64, flat_grid(n), dim3(256), 0, st, out, n, v);
}
void launch_zk_segment_ids(int32_t* out, long long B, int T, hipStream_t st) {
    const long long n = B * (T + MMS_NBOX);
    if (n > 0) hipLaunchKernelGGL(k_zk_segment_ids, flat_grid(n), dim3(256), 0, st, out, n, T, T + MMS_NBOX);
}
void launch_box_mask(const int32_t* num_boxes, float* mask, long long B, hipStream_t st) {
    if (B > 0) hipLaunchKernelGGL(k_box_mask, flat_grid(B * MMS_NBOX), dim3(256), 0, st, num_boxes, mask, B * MMS_NBOX);
}
void launch_corners(const float* boxes5, float* boxes4, long long B, hipStream_t st) {
    if (B > 0) hipLaunchKernelGGL(k_corners, flat_grid(B * MMS_NBOX * 4), dim3(256), 0, st, boxes5, boxes4, B * MMS_NBOX * 4);
}
void launch_merge4(const float* const probs[4], const float w[4], float* merged, float* members, long long n, hipStream_t st) {
    if (n > 0)
        hipLaunchKernelGGL(k_merge4, flat_grid(n), dim3(256), 0, st, probs[0], probs[1], probs[2], probs[3], w[0], w[1], w[2], w[3], merged,
                           members, n);
}
void launch_xnorm(const bf16* hi, const bf16* lo, float* out, int rows, hipStream_t st) {
    if (rows > 0) hipLaunchKernelGGL(k_xnorm, dim3((rows + 3) / 4), dim3(256), 0, st, hi, lo, out, rows);
}
