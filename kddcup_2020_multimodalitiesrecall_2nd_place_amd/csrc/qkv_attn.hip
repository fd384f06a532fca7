// Fused QKV projection + self-attention (precision modes 2 and 3): one workgroup computes [Q_h | K_h | V_h] = X W_h^T + b_h of ONE head for a
// 256-row tile of the token stream with the ping-pong engine of gemm_pp.hip, leaves the 192 result columns in LDS and runs the
// attention of the tile's pairs for that head straight from there -- the fp32 Q / K / V tensor ([rows][2304], 9.2 KB per row written
// by the projection and read back by attn.hip) never exists in HBM, and the attention launch disappears.
//
// Reference: pixelbert.py:767-836 / pixelmodel.py:770-831 (zk, lds) and lxrt/modeling.py:326-352 (lxmert self-attention): the three
// dense projections, scores = Q K^T / sqrt(64) + (1 - mask) * -10000, softmax over the keys, probs @ V, heads concatenated.
//
// Tile = two SUB-TILES of up to 128 rows, each holding whole pairs (a pair's tokens are consecutive rows, at most 48 of them):
// k_qkv_tile_plan packs the pairs greedily.  Wave grid 4(M) x 2(N), 64 x 96 outputs per wave; waves 0-3 own sub-tile 0, waves 4-7
// sub-tile 1 (the two staggered halves of the ping-pong).  K runs in 32-wide stages through the 3-slot LDS ring exactly as in
// gemm_pp.hip (A [256][hi 64 B | lo 64 B] + W [192][64 B] per stage, LDS-DMA, counted waits, swizzles on the source address); a stage
// is ONE phase of 48 MFMAs per wave (14 ds_read_b128).  Weights are stored head-major ([12][Q 64 | K 64 | V 64][768], tiled): a column
// tile is one head.
//
// Epilogue, per sub-tile: its four waves add the bias and write their accumulators to LDS as fp32 [128][196] (98 KB over the idle
// ring; row stride 784 B = 16 B mod 256 B: conflict-free for the row-strided fragment reads below) -> barrier -> the eight waves take
// the sub-tile's pairs round-robin and run attn.hip's register choreography with ds_read_b128 in place of the global loads (same
// v_mfma_f32_16x16x4_f32 sequence, same softmax: the context rows are BIT-IDENTICAL to the two-kernel route) -> split-bf16 context rows
// to HBM -> barrier.  The staging area overlays ring slots 1-2 (and the unused top of slot 0): the next tile's stage 0 is fetched into
// slot 0 during the epilogue, its stage 1 after the last barrier.
// FAST (mms_config.fuse_attention = 2): Q K^T and P V on split-bf16 MFMAs (hi + lo operands, three v_mfma_f32_16x16x32_bf16 products),
// v_exp_f32 / v_rcp_f32 softmax -- a fifth of the matrix-pipe time of the exact-fp32 form, ~2^-16 relative on the scores.
//
// Precision mode 3 (WPL = 2: the weights have a lo plane): the same kernel on a 192-row tile (sub-tiles of 96 rows, 48 x 96 per wave), so
// that A + W_hi + W_lo of a stage are 48 KiB and three slots still fit; passes a_hi w_hi, a_lo w_hi, a_hi w_lo in gemm_ppw.hip's order.
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "kernels.h"

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

namespace {

__device__ __forceinline__ int qa_swz(int r) { return (4 - ((r >> 2) & 3)) & 3; }

template <int N> __device__ __forceinline__ void qa_wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ void qa_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

constexpr int QA_SUB = 128;          // rows of a sub-tile: two-pass kernel (96 in the three-pass one: QA_SUB3)
constexpr int QA_SUB3 = 96;
constexpr int QA_LDROW = 196;        // floats per staged row: 192 + 4 (784 B = 16 B mod 256 B)
// FAST staging: a row is [Q_hi 64 | Q_lo 64 | K_hi 64 | K_lo 64 | V_hi 64 | V_lo 64] bf16 (split once, in the dump; Q pre-scaled by 1/8) + 16 B = the same 784 bytes
constexpr int QA_LDB = QA_LDROW * 4;
typedef short qa_s16x4 __attribute__((ext_vector_type(4)));
typedef short qa_s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

// min / max over the 16 lanes of a DPP row (every lane gets the result): quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror
__device__ __forceinline__ int qa_row16_min(int v) {
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false));
    return min(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false));
}
__device__ __forceinline__ int qa_row16_max(int v) {
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false));
    return max(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false));
}
// [4 keys][16 columns] bf16 block -> lane c of the 16-lane group gets column c (ds_read_b64_tr_b16: element j of lane c comes from the address of lane
// 4 j + c / 4 of the group, sub-element c % 4 -- tools/probes/tr16_probe.hip); every lane passes the address of ITS quad: row c / 4, columns 4 (c % 4) ..
__device__ __forceinline__ qa_s16x4 qa_tr16(const unsigned char* lds_quad) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) qa_s16x4*)(lds_quad));
}

}  // namespace

// ---- tile plan: pairs -> sub-tiles of <= 128 rows ----
// sub[u] = {first stream row, rows, first pair, pairs}; *n_sub = number of sub-tiles.  off == nullptr: dense stream (pair b = rows b*S ..).
// The packing is the greedy one in pair order (a sub-tile takes pairs until the next one would not fit, at most QA_SUB3 of them), computed in PARALLEL:
//   1. next[b] = the pair at which a sub-tile that starts at pair b ends (monotone in b: every thread sweeps its ~30 consecutive pairs with one running pointer);
//   2. pointer doubling in LDS: J_k[b] = next applied 2^k times (J_k+1[b] = J_k[J_k[b]], J[n] = n is the end; 16-bit entries, two levels ping-pong), and with every
//      level the known sub-tile starts double: start[i + 2^k] = J_k[start[i]] for i < 2^k;
//   3. one thread per sub-tile writes its record and the records of its pairs.
// ~30 us per plan of 30 000 pairs, whatever their lengths.  (Rounds 3-4 cut the stream into 8192-row segments and let one thread pack each segment sequentially:
// 60 us for zk's 15-row pairs, but 220-700 us for lxmert's 4-row box stream -- 1.8 % of its step once that stream and the cross plan were fused too -- and
// every segment ended in an underfull sub-tile.)
// CROSS plans (sub2 != nullptr; lxmert X layers): a pair brings the rows of BOTH its streams (cnt[b] + cnt2[b]) into the sub-tile;
// sub2[u] = {first row of stream 2, its rows, 0, 0}.
// pair_rec[b]: the pair's place inside its sub-tile, as the fused kernel's split-bf16 attention wants it (one dword, no arithmetic in front of the kernel's main
// loop): first row | rows << 8 in stream 1, first row << 16 | rows << 24 in stream 2 (rows relative to the sub-tile's part of the stream).
constexpr int QA_PLAN_THREADS = 1024;
constexpr int QA_SEG = 8192;         // (launch_qkv_attn's grid bound still allows for one underfull sub-tile per 8192 rows)
constexpr int QA_PLAN_MAXN = 40 * 1024 - 8;      // pairs per plan: two 16-bit jump tables of n + 1 entries in LDS (chunk_pairs <= 32768)

__global__ __launch_bounds__(QA_PLAN_THREADS) void k_qkv_tile_plan(const int* __restrict__ off, const int* __restrict__ cnt, int n, int S,
                                                                   int4* __restrict__ sub, int* __restrict__ n_sub, int sub_rows,
                                                                   const int* __restrict__ off2, const int* __restrict__ cnt2, int S2, int4* __restrict__ sub2,
                                                                   int* __restrict__ pair_rec, int* start) {
    __shared__ unsigned short jt[2][QA_PLAN_MAXN + 1];
    __shared__ int s_nsub;
    const int tid = threadIdx.x;
    const bool cross = sub2 != nullptr;
    auto first1 = [&](int b) { return off ? off[b] : b * S; };
    auto first2 = [&](int b) { return !cross ? 0 : off2 ? off2[b] : b * S2; };
    auto rows1 = [&](int b) { return cnt ? cnt[b] : S; };
    auto rows2 = [&](int b) { return !cross ? 0 : cnt2 ? cnt2[b] : S2; };
    auto cum = [&](int b) { return b < n ? first1(b) + first2(b) : first1(n - 1) + first2(n - 1) + rows1(n - 1) + rows2(n - 1); };   // rows in front of pair b
    // start[] lives in global memory and is written and read by this one workgroup: agent-scope accesses keep stale L1 lines out of the picture
    auto sl = [&](int i) { return __hip_atomic_load(start + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto ss = [&](int i, int v) { __hip_atomic_store(start + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    // 1. next[] -> level 0; start[] = "no sub-tile" (n) except start[0] = pair 0
    {
        const int per = (n + QA_PLAN_THREADS - 1) / QA_PLAN_THREADS, b0 = tid * per, b1 = min(b0 + per, n);
        int e = b0 + 1;
        for (int b = b0; b < b1; ++b) {
            const int base = cum(b), lim = min(n, b + QA_SUB3);      // (the kernel keeps at most 128 pair records per sub-tile)
            if (e <= b) e = b + 1;
            while (e < lim && cum(e + 1) - base <= sub_rows) ++e;
            jt[0][b] = (unsigned short)e;
        }
        if (tid == 0) jt[0][n] = (unsigned short)n;
        for (int i = tid; i <= n; i += QA_PLAN_THREADS) ss(i, i == 0 ? 0 : n);
    }
    __syncthreads();
    // 2. doubling
    int cur = 0;
    for (int k = 0; (1 << k) <= n; ++k) {
        for (int i = tid; i < (1 << k); i += QA_PLAN_THREADS) {
            const int b = sl(i);
            if (b < n) { const int t = jt[cur][b]; if (t < n) ss(i + (1 << k), t); }
        }
        for (int b = tid; b <= n; b += QA_PLAN_THREADS) jt[cur ^ 1][b] = jt[cur][jt[cur][b]];
        __syncthreads();
        cur ^= 1;
    }
    // 3. count, then the records
    for (int i = tid; i < n; i += QA_PLAN_THREADS)
        if (sl(i) < n && (i + 1 == n || sl(i + 1) >= n)) { s_nsub = i + 1; *n_sub = i + 1; }
    __syncthreads();
    const int nsub = s_nsub;
    for (int u = tid; u < nsub; u += QA_PLAN_THREADS) {
        const int s0 = sl(u), e = u + 1 < nsub ? sl(u + 1) : n;
        const int r1 = first1(s0), r2 = first2(s0);
        int a1 = 0, a2 = 0;
        for (int b = s0; b < e; ++b) {
            const int c = rows1(b), c2 = rows2(b);
            if (pair_rec) pair_rec[b] = a1 | (c << 8) | (a2 << 16) | (c2 << 24);
            a1 += c; a2 += c2;
        }
        sub[u] = make_int4(r1, a1, s0, e - s0);
        if (cross) sub2[u] = make_int4(r2, a2, 0, 0);
    }
}
bool launch_qkv_tile_plan(const int* off, const int* cnt, int n, int S, int4* sub, int* n_sub, int passes, hipStream_t st, int* pair_rec, int* scratch) {
    if (n <= 0) return true;
    if (n > QA_PLAN_MAXN) return false;
    hipLaunchKernelGGL(k_qkv_tile_plan, dim3(1), dim3(QA_PLAN_THREADS), 0, st, off, cnt, n, S, sub, n_sub, passes == 3 ? QA_SUB3 : QA_SUB,
                       (const int*)nullptr, (const int*)nullptr, 0, (int4*)nullptr, pair_rec, scratch);
    return true;
}
bool launch_qkv_cross_plan(const int* off, const int* cnt, const int* off2, const int* cnt2, int n, int S, int S2,
                           int4* sub, int4* sub2, int* n_sub, int passes, hipStream_t st, int* pair_rec, int* scratch) {
    if (n <= 0) return true;
    if (n > QA_PLAN_MAXN) return false;
    hipLaunchKernelGGL(k_qkv_tile_plan, dim3(1), dim3(QA_PLAN_THREADS), 0, st, off, cnt, n, S, sub, n_sub, passes == 3 ? QA_SUB3 : QA_SUB,
                       off2, cnt2, S2, sub2, pair_rec, scratch);
    return true;
}
int qkv_plan_scratch_ints(int n) { return n + 1; }

// MAXT: 16-token tiles per side of the attention (2: pairs of <= 32 tokens, 3: <= 48)
// WPL: weight planes.  1 = precision mode 2 (bf16 weights; a_hi w + a_lo w), sub-tiles of 128 rows, 64 x 96 per wave;
//      2 = precision mode 3 (w = w_hi + w_lo; a_hi w_hi + a_lo w_hi + a_hi w_lo in gemm_ppw.hip's order), sub-tiles of 96 rows (192-row tile,
//      48 x 96 per wave) so that A [192][128 B] + W_hi + W_lo [192][64 B] of a stage are 48 KiB and three slots still fit
template <int MAXT, bool FAST, int WPL>
__global__ __launch_bounds__(512) void qkv_attn_kernel(const QkvAttnParams p) {
    constexpr int SUB = WPL == 1 ? QA_SUB : QA_SUB3, BM = 2 * SUB;
    constexpr int NW = 8, WAVES_N = 2, TM = BM / 4, TN = 96, FM = TM / 16, FN = TN / 16, BN = 192;
    constexpr int AREG = BM * 128, WREG = BN * 64;  // A [BM][hi 64 B | lo 64 B], one W plane [192][64 B] of a stage
    constexpr int SLOT = 48 * 1024;                 // AREG + WPL * WREG = 44 / 48 KiB
    constexpr int NAP = BM / 64;                    // A pieces (8 rows x 128 B per wave) per wave and stage
    constexpr int NSLOT = 3, D = 2, P = NAP + 2 * WPL;   // + 2 W pieces per plane (waves 4-7 repeat waves 0-3's second one)
    static_assert(AREG + WPL * WREG <= SLOT && SUB * QA_LDROW * 4 <= NSLOT * SLOT, "stage fits a slot; Q | K | V staging overlays the ring");
    __shared__ __attribute__((aligned(16))) unsigned char smem[NSLOT * SLOT];
    // per-tile metadata of the epilogue, fetched with the tile's addresses (setup) and parked here after the main loop: a dependent
    // global load inside the attention phase costs a full memory latency with only eight waves on the CU (the phase took 16 k cycles per
    // sub-tile with the pair offsets / key mask / bias read from memory where they are used: profiles/r03n_qa_trace.txt)
    __shared__ __attribute__((aligned(16))) float m_keyadd[256 + 64];     // additive key mask by tile row (BM <= 256; + 64: a 32-key chunk may run past the sub-tile)
    __shared__ __attribute__((aligned(16))) float m_bias[192];       // this head's [Q | K | V] bias
    __shared__ int2 m_pair[FAST ? 1 : 256 + 64];                     // exact route: [sub-tile][pair]: first stream row, live tokens (+ 64: whole-wave reads)
    __shared__ int m_rowmeta[FAST ? 256 : 1];                        // FAST route: per tile row, the sub-tile rows [kbeg, kend) its query attends (kbeg | kend << 16)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int n_sub = *p.n_sub;
    const int nbm = (n_sub + 1) >> 1, nblk = nbm * MMS_HEADS;
    if (p.flop_counter && blockIdx.x == 0 && tid == 0) {
        int Meff = p.M;
        if (p.m_dev) { const int md = *p.m_dev; Meff = md < Meff ? md : Meff; }
        if (FAST && p.sub2) { int m2 = p.M2; if (p.m_dev2) { const int md = *p.m_dev2; m2 = md < m2 ? md : m2; } Meff += m2; }
        atomicAdd(p.flop_counter, 2ull * (unsigned long long)Meff * (unsigned long long)(3 * MMS_HIDDEN) * (unsigned long long)p.K);
    }
    int vb = blockIdx.x;
    if (vb >= nblk) return;
    if (tid < 64) m_keyadd[256 + tid] = 0.f;      // (read by chunks that run past a sub-tile's rows; those keys are masked by their range)

    const int gr_l = lane >> 2, gc = lane & 3;
    const bf16* a_src[NAP];
    const bf16* w_src[WPL][2];
    int head;
    int4 sub0, sub1;
    int2 sub0b = make_int2(0, 0), sub1b = sub0b;     // CROSS: {first row, rows} of the sub-tiles' stream-2 part
    int meta_a = 0, meta_b = 0;          // thread < 256: key mask of tile row tid, bias[tid]; else pair (tid - 256): first row, tokens
                                         // FAST, pair threads: meta_a = the plan's pair record (rows of the pair inside its sub-tile)
    const bool cross = FAST && p.sub2 != nullptr;
    // where virtual block v works: head and the two sub-tile records (uniform: scalar loads).  Done one tile ahead (top of the main loop), so
    // that the address set-up behind the loop has no dependent memory access in front of it
    int nhead = 0;
    int4 nsub0 = make_int4(0, 0, 0, 0), nsub1 = nsub0;
    int2 nsub0b = make_int2(0, 0), nsub1b = nsub0b;
    auto locate = [&](int v) {
        // bijective XCD remap (virtual block v runs on XCD v % 8): the twelve heads of a row tile stay on one XCD's L2
        const int q = nblk >> 3, r8 = nblk & 7, xcd = v & 7, loc = v >> 3;
        int bid = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + loc;
        if (p.reverse) bid = nblk - 1 - bid;
        const int nb = bid / MMS_HEADS;
        nhead = bid % MMS_HEADS;
        auto uni = [](int4 v) {      // wave-uniform by construction: keep the record in scalar registers
            return make_int4(__builtin_amdgcn_readfirstlane(v.x), __builtin_amdgcn_readfirstlane(v.y), __builtin_amdgcn_readfirstlane(v.z),
                             __builtin_amdgcn_readfirstlane(v.w));
        };
        nhead = __builtin_amdgcn_readfirstlane(nhead);
        nsub0 = uni(p.sub[2 * nb]);
        nsub1 = 2 * nb + 1 < n_sub ? uni(p.sub[2 * nb + 1]) : make_int4(nsub0.x, 0, 0, 0);
        if (cross) {
            const int4 a = uni(p.sub2[2 * nb]), b = 2 * nb + 1 < n_sub ? uni(p.sub2[2 * nb + 1]) : make_int4(a.x, 0, 0, 0);
            nsub0b = make_int2(a.x, a.y); nsub1b = make_int2(b.x, b.y);
        }
    };
    // row of the plane buffers (a_hi / o_hi) behind row lr of a sub-tile: its stream-1 rows first, then (CROSS) its stream-2 rows
    auto grow = [&](const int4& sa, const int2& sb2, int lr) -> long long {
        return lr < sa.y ? (long long)(sa.x + lr) : p.row0_b + (long long)(sb2.x + (lr - sa.y));
    };
    auto setup = [&]() {
        head = nhead; sub0 = nsub0; sub1 = nsub1; sub0b = nsub0b; sub1b = nsub1b;
#pragma unroll
        for (int q4 = 0; q4 < NAP; ++q4) {
            const int r = q4 * 64 + wave * 8 + (lane >> 3);          // tile row: sub-tile r / SUB, its row r % SUB (clamped to the last live one)
            const int4 sb = (r / SUB) ? sub1 : sub0;
            const int2 sb2 = make_int2((r / SUB) ? sub1b.x : sub0b.x, (r / SUB) ? sub1b.y : sub0b.y);     // (component selects: a struct select goes through scratch)
            const int nlive = sb.y + sb2.y;
            int lr = r % SUB;
            lr = lr < nlive ? lr : (nlive > 0 ? nlive - 1 : 0);
            const long long gr = nlive > 0 ? grow(sb, sb2, lr) : (long long)sb.x;
            a_src[q4] = p.a_hi + 2 * (gr * (long long)p.lda) + ((lane & 7) ^ ((r >> 1) & 7)) * 8;
        }
        {
            const int t = tid & 255;
            const int u = tid < 256 ? (t >= SUB) : (t >> 7), i = tid < 256 ? t - u * SUB : (t & 127);   // key-mask rows by tile row, pair records [sub-tile][128]
            const int4 sb = u ? sub1 : sub0;
            const int2 sb2 = make_int2(u ? sub1b.x : sub0b.x, u ? sub1b.y : sub0b.y);
            if (tid < 256) {
                meta_a = 0;
                if (t < BM && i < sb.y) { if (p.key_add) meta_a = __float_as_int(p.key_add[sb.x + i]); }
                else if (t < BM && i < sb.y + sb2.y) { if (p.key_add2) meta_a = __float_as_int(p.key_add2[sb2.x + (i - sb.y)]); }
                meta_b = tid < BN ? __float_as_int(p.bias[head * BN + tid]) : 0;
            } else if (i < sb.w) {
                const int b = sb.z + i;
                if constexpr (FAST) meta_a = p.pair_rec[b];      // launch_qkv_tile_plan's record: no arithmetic on a fresh load here (setup runs while the other waves wait at a barrier)
                else {
                    meta_a = p.pair_off ? p.pair_off[b] : b * p.S;
                    meta_b = p.pair_cnt ? p.pair_cnt[b] : p.S;
                }
            }
        }
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const int r = h2 ? 128 + (wave & 3) * 16 + gr_l : wave * 16 + gr_l;
            w_src[0][h2] = p.w + wtile_off(head * BN + r, 0, p.K) + (gc ^ qa_swz(r)) * 8;
            if constexpr (WPL == 2) w_src[1][h2] = w_src[0][h2] + (p.w_lo - p.w);
        }
    };
    locate(vb);
    setup();
    auto issue = [&](int q, int st, int slot) {
        unsigned char* d;
        const bf16* s;
        if (q < NAP) { d = smem + slot * SLOT + q * 8192 + wave * 1024; s = a_src[q] + st * 64; }
        else {
            const int pl = (q - NAP) >> 1, h2 = (q - NAP) & 1;
            d = smem + slot * SLOT + AREG + pl * WREG + (h2 ? 8192 + (wave & 3) * 1024 : wave * 1024);
            s = w_src[pl][h2] + st * 512;
        }
        __builtin_amdgcn_global_load_lds((glb_void*)s, (lds_void*)d, 16, 0, 0);
    };

    f32x4 acc[FM][FN];
    const int ns = p.K / 32;
    const int fr = lane & 15, fk = lane >> 4;
    const int laneA = (wm * TM + fr) * 128 + ((fk ^ ((fr >> 1) & 7)) << 4);          // hi fragment; lo: chunk ^ 4
    const int laneB = AREG + (wn * TN + fr) * 64 + ((fk ^ qa_swz(fr)) << 4);                 // W_hi fragment; W_lo: + WREG

    // one phase per stage: all fragment reads (+ the six LDS-DMA pieces of stage s+2), counted wait, reads retired, barrier, 48 MFMAs,
    // barrier.  Hazards as in gemm_pp.hip / gemm_ppw.hip: a slot's reads are retired before the first barrier of its stage, its refill is
    // issued after the second; a stage's data is waited for one stage before it is read.
    auto stage = [&](auto pre_tag, auto wait_tag, int s, int slot) {
        constexpr bool PRE = decltype(pre_tag)::value;
        constexpr int WAITN = decltype(wait_tag)::value;
        const unsigned char* sb = smem + slot * SLOT;
        const int nslot = slot == 0 ? NSLOT - 1 : slot - 1;
        bf16x8 a[2][FM], b[WPL][FN];
#pragma unroll
        for (int wp = 0; wp < WPL; ++wp)
#pragma unroll
            for (int j = 0; j < FN; ++j) b[wp][j] = *reinterpret_cast<const bf16x8*>(sb + laneB + wp * WREG + j * 16 * 64);
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int i = 0; i < FM; ++i) a[pl][i] = *reinterpret_cast<const bf16x8*>(sb + (laneA ^ (pl << 6)) + i * 16 * 128);
        if (PRE) {
#pragma unroll
            for (int q = 0; q < P; ++q) issue(q, s + D, nslot);
        }
        if (WAITN >= 0) qa_wait_vmcnt<(WAITN >= 0 ? WAITN : 0)>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        qa_barrier();
        __builtin_amdgcn_s_setprio(1);
        // per accumulator and stage: a_hi w_hi, a_lo w_hi (, a_hi w_lo) -- the order of gemm_pp.hip / gemm_ppw.hip, so the projection is bit-identical
        // to the two-kernel route's
#pragma unroll
        for (int pass = 0; pass < WPL + 1; ++pass)
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[pass == 2 ? WPL - 1 : 0][j], a[pass == 1 ? 1 : 0][i], acc[i][j], 0, 0, 0);   // swapped operands: C^T fragment
        __builtin_amdgcn_s_setprio(0);
        qa_barrier();
    };
    auto first_stages = [&]() {
#pragma unroll
        for (int d = 0; d < D; ++d) {
#pragma unroll
            for (int q = 0; q < P; ++q) issue(q, d, d);
        }
    };
    first_stages();
    qa_wait_vmcnt<P>();
    qa_barrier();
    if (wave >= NW / 2) qa_barrier();     // stagger the two halves by one barrier

    // the staging area sits at the TOP of the ring, above the 44 KiB that a stage occupies in slot 0: the next tile's first stage is
    // fetched into slot 0 while this tile's epilogue runs
    float* stg = reinterpret_cast<float*>(smem + NSLOT * SLOT - SUB * QA_LDROW * 4);
    static_assert(NSLOT * SLOT - SUB * QA_LDROW * 4 >= AREG + WPL * WREG, "slot 0's stage must lie below the staging area");
#ifdef MMS_LAB
    // lab (timing only, results WRONG): p.lab_flags bit 0 drops the P V MFMAs, 1 the Q K^T MFMAs, 3 the context stores, 4 all attention work
    const int QA_FLAGS = p.lab_flags;
#else
    constexpr int QA_FLAGS = 0;
#endif
#ifdef MMS_LAB
    // lab: per-tile timeline of thread 0 (shader-clock stamps: loop start, loop end, dump 0, attention 0, dump 1, attention 1, next
    // prologue done), first 8 tiles of every workgroup -> p.trace[(blockIdx.x * 8 + tile) * 8 + k]   (tools/qa_trace.py)
    unsigned long long tr[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};      // [8..]: wave 0's own progress inside dump 0 (8: body done, 9: stores landed) and attention 1 (10: range + Q fragments, 11: first step, 12: all steps, 13: stores issued)
#define QA_SUB_STAMP(k) do { if (p.trace) tr[k] = __builtin_readcyclecounter(); } while (0)
    int tile_i = 0;
#define QA_STAMP(k) do { if (p.trace) tr[k] = __builtin_readcyclecounter(); } while (0)
#else
#define QA_STAMP(k) do { } while (0)
#define QA_SUB_STAMP(k) do { } while (0)
#endif
    for (;;) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        QA_STAMP(0);
        const bool more = vb + (int)gridDim.x < nblk;
        if (more) locate(vb + (int)gridDim.x);
        int slot = 0, s = 0;
        for (; s + D < ns; ++s) {
            stage(std::true_type{}, std::integral_constant<int, (D - 1) * P>{}, s, slot);
            slot = slot == NSLOT - 1 ? 0 : slot + 1;
        }
        stage(std::false_type{}, std::integral_constant<int, 0>{}, s, slot);         // stage ns-2: the last stage must have landed
        slot = slot == NSLOT - 1 ? 0 : slot + 1;
        ++s;
        // park this tile's metadata in LDS; every wave passes a barrier between its own writes and the first read (waves 4-7: the last
        // stage's; waves 0-3: the re-aligning one)
        if (wave >= NW / 2) {
            if constexpr (FAST) {
                // the pair's rows learn which sub-tile rows their queries attend: self-attention -> the pair's own rows; CROSS -> its rows in the other stream
                const int t = tid - 256, u = t >> 7;
                const int4 sa = u ? sub1 : sub0;
                if ((t & 127) < sa.w) {
                    const int r1 = meta_a & 255, c1 = (meta_a >> 8) & 255;
                    if (cross) {
                        const int r2 = sa.y + ((meta_a >> 16) & 255), c2 = (meta_a >> 24) & 255;
                        for (int r = 0; r < c1; ++r) m_rowmeta[u * SUB + r1 + r] = r2 | ((r2 + c2) << 16);
                        for (int r = 0; r < c2; ++r) m_rowmeta[u * SUB + r2 + r] = r1 | ((r1 + c1) << 16);
                    } else {
                        for (int r = 0; r < c1; ++r) m_rowmeta[u * SUB + r1 + r] = r1 | ((r1 + c1) << 16);
                    }
                }
            } else m_pair[tid - 256] = make_int2(meta_a, meta_b);
        }
        stage(std::false_type{}, std::integral_constant<int, -1>{}, s, slot);        // stage ns-1
        if (wave < NW / 2) {
            m_keyadd[tid] = __int_as_float(meta_a);
            if constexpr (FAST) {      // rows behind the last pair of a sub-tile attend nothing
                const int u = tid >= SUB, i = tid - u * SUB;
                if (tid < BM && i >= (u ? sub1.y + sub1b.y : sub0.y + sub0b.y)) m_rowmeta[tid] = 0;
            }
            if (tid < BN) m_bias[tid] = __int_as_float(meta_b);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            qa_barrier();     // re-align the halves: nobody reads the ring any more
        }
        QA_STAMP(1);

        // this tile's identity for the epilogue; then the next tile's addresses and its stage 0 (slot 0 is idle and below the staging area)
        const int ehead = head;
        const int4 esub0 = sub0, esub1 = sub1;
        const int2 esub0b = sub0b, esub1b = sub1b;
        if (more) vb += (int)gridDim.x;
        // ---- epilogue: per sub-tile  accumulators -> LDS, attention of its pairs for this head ----
        const int mrow = lane & 15, nq = lane >> 4;
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            if ((wm >> 1) == half) {
                if constexpr (FAST) {
                    // split ONCE here: the attention reads MFMA operands (bf16x8) straight from LDS, a staged row = [Q_hi | Q_lo | K_hi | K_lo | V_hi | V_lo] (QA_LDB)
                    // One 16-byte store per plane: v_permlane16_swap between the lane rows nq = (0, 1) and (2, 3) gives every lane EIGHT consecutive d of ONE of the two
                    // column fragments j / j + 1 (lane rows 0, 2: fragment j, d 0..7 / 8..15; rows 1, 3: fragment j + 1).  8-byte stores (16 rows x 8 B per lane group:
                    // two-way bank conflicts at this row stride) made the dump LDS-write-bound at 4.5 k cycles per sub-tile; the arithmetic did not matter.
                    static_assert(FN % 2 == 0 && TN % 32 == 0, "column fragments are stored in pairs that share a Q / K / V section");
                    unsigned char* rowp = reinterpret_cast<unsigned char*>(stg) + ((wm & 1) * TM + mrow) * QA_LDB + (nq & 1) * 32 + (nq >> 1) * 16;
#pragma unroll
                    for (int j = 0; j < FN; j += 2) {
                        const int col = wn * TN + 16 * j;      // wave-uniform: section col / 64 (0 Q, 1 K, 2 V); the pair covers d = col % 64 + 0..31 of it
                        const f32x4 b0 = *reinterpret_cast<const f32x4*>(m_bias + col + nq * 4), b1 = *reinterpret_cast<const f32x4*>(m_bias + col + 16 + nq * 4);
                        unsigned char* dst = rowp + (col >> 6) * 256 + (col & 63) * 2;
#pragma unroll
                        for (int i = 0; i < FM; ++i) {
                            unsigned w[2][2][2];      // [plane][fragment of the pair][dword]
#pragma unroll
                            for (int t2 = 0; t2 < 2; ++t2) {
                                f32x4 v = acc[i][j + t2];
                                v += t2 ? b1 : b0;
                                bf16x4 hi, lo;
#pragma unroll
                                for (int e = 0; e < 4; ++e) { bf16 x, y; split_bf16(v[e], x, y); hi[e] = x; lo[e] = y; }
                                if (QA_FLAGS & 64) { hi = __builtin_bit_cast(bf16x4, f32x2_t{v[0], v[1]}); lo = __builtin_bit_cast(bf16x4, f32x2_t{v[2], v[3]}); }      // lab: no split arithmetic
                                const u32x2_t h2 = __builtin_bit_cast(u32x2_t, hi), l2 = __builtin_bit_cast(u32x2_t, lo);
                                w[0][t2][0] = h2[0]; w[0][t2][1] = h2[1]; w[1][t2][0] = l2[0]; w[1][t2][1] = l2[1];
                            }
#pragma unroll
                            for (int pl2 = 0; pl2 < 2; ++pl2) {
                                u32x4 out;
#pragma unroll
                                for (int dw = 0; dw < 2; ++dw) {
                                    const auto sw = __builtin_amdgcn_permlane16_swap(w[pl2][0][dw], w[pl2][1][dw], false, false);
                                    out[dw] = sw[0]; out[2 + dw] = sw[1];
                                }
                                if (!(QA_FLAGS & 32)) *reinterpret_cast<u32x4*>(dst + 16 * i * QA_LDB + 128 * pl2) = out;      // (lab flag 32: no staging stores)
                            }
                        }
                    }
                } else {
                const float* bias = m_bias + wn * TN + nq * 4;
                float* dst = stg + ((wm & 1) * TM + mrow) * QA_LDROW + wn * TN + nq * 4;
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias + 16 * j);
#pragma unroll
                    for (int i = 0; i < FM; ++i) {
                        f32x4 v = acc[i][j];
                        v += b4;
                        *reinterpret_cast<f32x4*>(dst + 16 * i * QA_LDROW + 16 * j) = v;
                    }
                }
                }
                if (half == 0) QA_SUB_STAMP(8);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (half == 0) QA_SUB_STAMP(9);
            } else if (more) {
                // the four waves that have nothing to write in this phase set up their next tile and send out its stage 0 (slot 0 lies below
                // the staging area)
                setup();
#pragma unroll
                for (int q = 0; q < P; ++q) issue(q, 0, 0);
            }
            qa_barrier();
#ifdef MMS_LAB
            if (p.trace) tr[2 + 2 * half] = __builtin_readcyclecounter();
#endif
            const int4 sb = half ? esub1 : esub0;
            if constexpr (FAST) {
            // ---- FAST: one 16-query tile of the sub-tile per wave, block-diagonal over the pairs it touches.  Query row i attends the sub-tile rows
            // [kbeg_i, kend_i) (m_rowmeta: its pair's rows; CROSS: its pair's rows in the other stream); the wave walks the union of its 16 queries' ranges in
            // chunks of 32 keys with an online softmax: S^T = K Q^T and O^T = V^T P^T on split-bf16 MFMAs (hi hi, hi lo, lo hi), K / Q fragments by ds_read_b128,
            // V^T fragments by ds_read_b64_tr_b16 from the row-major staged V.  Every lane's scores, probabilities and outputs belong to ONE query (i = lane & 15):
            // running maximum, sum and rescale are per-lane scalars. ----
            const int2 sb2 = make_int2(half ? esub1b.x : esub0b.x, half ? esub1b.y : esub0b.y);
            const int nlive = sb.y + sb2.y;
            if (wave * 16 < nlive && wave * 16 < SUB && !(QA_FLAGS & 16)) {
                const unsigned char* stgb = reinterpret_cast<const unsigned char*>(stg);
                const int i = wave * 16 + fr;
                const bool live = i < nlive;
                const int ic = live ? i : nlive - 1;
                const int rm = m_rowmeta[half * SUB + ic];
                const int kbeg = rm & 0xffff, kend = (int)((unsigned)rm >> 16);
                const int kb = __builtin_amdgcn_readfirstlane(qa_row16_min(live ? kbeg : 0x7fff)) & ~3;
                const int ke = __builtin_amdgcn_readfirstlane(qa_row16_max(live ? kend : 0));
                const unsigned char* qrow = stgb + ic * QA_LDB + 16 * fk;          // lane: d = 32 ds + 8 fk + 0..7
                const bf16x8 qh0 = *reinterpret_cast<const bf16x8*>(qrow), qh1 = *reinterpret_cast<const bf16x8*>(qrow + 64);
                const bf16x8 ql0 = *reinterpret_cast<const bf16x8*>(qrow + 128), ql1 = *reinterpret_cast<const bf16x8*>(qrow + 192);
                const float* kadd = m_keyadd + half * SUB;
                float m = -INFINITY, l = 0.f;
                f32x4 o[4];
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
                // one step = NH halves of 32 keys (compile-time: no branch inside, so the 2 NH score tiles' LDS reads and MFMA chains interleave -- with a
                // wave-uniform branch per tile every tile was its own basic block and the phase ran at the latency of one dependent chain after the other)
                auto step = [&](auto nh_tag, const int kc) {
                    constexpr int NH = decltype(nh_tag)::value;
                    // S^T tile t: first operand = K rows (key j = kc + 16 t + fr), second = the Q rows; lane gets keys kc + 16 t + 4 fk + r of query fr
                    f32x4 sc[2 * NH];
#pragma unroll
                    for (int t = 0; t < 2 * NH; ++t) {
                        f32x4 a4 = {0.f, 0.f, 0.f, 0.f};
                        if (!(QA_FLAGS & 2)) {
                            int j = kc + 16 * t + fr;
                            j = j < SUB ? j : SUB - 1;
                            const unsigned char* kr = stgb + j * QA_LDB + 256 + 16 * fk;
                            const bf16x8 kh0 = *reinterpret_cast<const bf16x8*>(kr), kh1 = *reinterpret_cast<const bf16x8*>(kr + 64);
                            const bf16x8 kl0 = *reinterpret_cast<const bf16x8*>(kr + 128), kl1 = *reinterpret_cast<const bf16x8*>(kr + 192);
                            a4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh0, qh0, a4, 0, 0, 0);
                            a4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh0, ql0, a4, 0, 0, 0);
                            a4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kl0, qh0, a4, 0, 0, 0);
                            a4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh1, qh1, a4, 0, 0, 0);
                            a4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh1, ql1, a4, 0, 0, 0);
                            a4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kl1, qh1, a4, 0, 0, 0);
                        }
                        sc[t] = a4;
                        if (t & 1) __builtin_amdgcn_sched_barrier(0);      // two tiles' fragments (32 registers) in flight at a time: the other sub-tile's accumulators are still live
                    }
                    // 1 / sqrt(64) and the additive key mask; keys outside the query's own range do not exist for it
                    const int j0 = kc + 4 * fk;
                    float sv[8 * NH];
                    float cm = -INFINITY;
#pragma unroll
                    for (int t = 0; t < 2 * NH; ++t) {
                        const f32x4 ka = *reinterpret_cast<const f32x4*>(kadd + j0 + 16 * t);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int j = j0 + 16 * t + r;
                            const float v = (j >= kbeg && j < kend) ? __builtin_fmaf(sc[t][r], 0.125f, ka[r]) : -INFINITY;
                            sv[4 * t + r] = v;
                            cm = fmaxf(cm, v);
                        }
                    }
                    cm = rows4_max(cm);
                    const float mn = fmaxf(m, cm);
                    const float ms = mn == -INFINITY ? 0.f : mn;          // (no key of this query so far)
                    const float alpha = __builtin_amdgcn_exp2f((m - ms) * 1.44269504088896340736f);
                    float ps = 0.f;
#pragma unroll
                    for (int e = 0; e < 8 * NH; ++e) {
                        sv[e] = __builtin_amdgcn_exp2f((sv[e] - ms) * 1.44269504088896340736f);
                        ps += sv[e];
                    }
                    ps = rows4_sum(ps);
                    l = l * alpha + ps;
                    m = mn;
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) o[dt] *= alpha;
                    // O^T += V^T P^T over each half's 32 key slots: slot (fk, e) = key kc + 32 h + 16 (e >> 2) + 4 fk + (e & 3) -- the lane's own probabilities are the
                    // second operand; first operand = V^T[d = 16 dt + fr][slot]: two transposing reads of the [4 keys][16 d] blocks at keys + 4 fk and + 16 + 4 fk
#pragma unroll
                    for (int h = 0; h < NH; ++h) {
                        if (QA_FLAGS & 1) continue;
                        bf16x8 ph, pl;
#pragma unroll
                        for (int e = 0; e < 8; ++e) { bf16 x, y; split_bf16(sv[8 * h + e], x, y); ph[e] = x; pl[e] = y; }
                        int r0 = kc + 32 * h + 4 * fk + (fr >> 2), r1 = r0 + 16;
                        r0 = r0 < SUB ? r0 : SUB - 1;      // (P is exactly 0 there)
                        r1 = r1 < SUB ? r1 : SUB - 1;
                        const unsigned char* v0 = stgb + r0 * QA_LDB + 512 + (fr & 3) * 8;
                        const unsigned char* v1 = stgb + r1 * QA_LDB + 512 + (fr & 3) * 8;
#pragma unroll
                        for (int dt = 0; dt < 4; ++dt) {
                            const bf16x8 vh = __builtin_bit_cast(bf16x8, __builtin_shufflevector(qa_tr16(v0 + 32 * dt), qa_tr16(v1 + 32 * dt), 0, 1, 2, 3, 4, 5, 6, 7));
                            const bf16x8 vl = __builtin_bit_cast(bf16x8, __builtin_shufflevector(qa_tr16(v0 + 128 + 32 * dt), qa_tr16(v1 + 128 + 32 * dt), 0, 1, 2, 3, 4, 5, 6, 7));
                            o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh, ph, o[dt], 0, 0, 0);
                            o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vl, ph, o[dt], 0, 0, 0);
                            o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh, pl, o[dt], 0, 0, 0);
                        }
                    }
                };
                if (half == 1) QA_SUB_STAMP(10);
#pragma unroll 1
                for (int kc = kb; kc < ke; kc += 32) {      // (32-key steps here: with the other sub-tile's 96 accumulators live, a 64-key step spills)
                    step(std::integral_constant<int, 1>{}, kc);
                    if (half == 1 && kc == kb) QA_SUB_STAMP(11);
                }
                if (half == 1) QA_SUB_STAMP(12);
                // lane holds O[query i][d = 16 dt + 4 fk + r].  v_permlane16_swap between the lane rows fk = (0, 1) and (2, 3) trades the dt = 2 c + 1 quad of the even row for
                // the dt = 2 c quad of the odd one: every lane then owns EIGHT consecutive d (32 c + {0, 16, 8, 24}[fk] + 0..7) = one 16-byte store per plane, the four
                // lanes of a query write one whole 64-byte block of the hl32 planes (8-byte stores of half blocks took 1.1 k of the phase's 6.3 k cycles and held up
                // the next tile's first stages behind them)
                {
                    const float inv = __builtin_amdgcn_rcpf(l);
                    const int dsel = (fk & 1) * 16 + (fk >> 1) * 8;
                    const long long off = (live ? grow(sb, sb2, i) : 0) * p.ldo + ehead * MMS_HEAD_DIM + dsel;
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        unsigned w[2][2][2];      // [plane][dt within the pair][dword]: two bf16 each
#pragma unroll
                        for (int t2 = 0; t2 < 2; ++t2) {
                            bf16x4 hi, lo;
#pragma unroll
                            for (int r = 0; r < 4; ++r) { bf16 x, y; split_bf16(o[2 * c + t2][r] * inv, x, y); hi[r] = x; lo[r] = y; }
                            const u32x2_t h2 = __builtin_bit_cast(u32x2_t, hi), l2 = __builtin_bit_cast(u32x2_t, lo);
                            w[0][t2][0] = h2[0]; w[0][t2][1] = h2[1]; w[1][t2][0] = l2[0]; w[1][t2][1] = l2[1];
                        }
#pragma unroll
                        for (int pl2 = 0; pl2 < 2; ++pl2) {
                            u32x4 out;
#pragma unroll
                            for (int dw = 0; dw < 2; ++dw) {
                                const auto sw = __builtin_amdgcn_permlane16_swap(w[pl2][0][dw], w[pl2][1][dw], false, false);
                                out[dw] = sw[0]; out[2 + dw] = sw[1];
                            }
                            if (live && !(QA_FLAGS & 8)) *reinterpret_cast<u32x4*>(plane_ptr(pl2 ? p.o_lo : p.o_hi, off + 32 * c)) = out;
                        }
                    }
                    if (half == 1) QA_SUB_STAMP(13);
                }
            }
            } else {
            // work items = (pair, 16-query tile), dealt round-robin to the eight waves in pair order: a pair of 17 .. 32 tokens is two items
            // (four times the MFMAs of a short pair), and the phase lasts as long as its slowest wave
            int item = 0;
            int2 recs = make_int2(0, 0);
#pragma unroll 1
            for (int unit = 0; unit < sb.w; ++unit) {
                if ((unit & 63) == 0) recs = m_pair[half * 128 + unit + lane];      // 64 pair records per read; the loop itself is scalar
                const int g0 = __builtin_amdgcn_readlane(recs.x, unit & 63);       // the pair's first stream row
                const int S = __builtin_amdgcn_readlane(recs.y, unit & 63);        // its live tokens (queries == keys)
                const int nqt = (S + 15) >> 4;
                const int first = item;
                item += nqt;
                // does any of this pair's query tiles fall to this wave?  tile qt belongs to wave (first + qt) % 8
                if (S <= 0 || (((wave - first) & (NW - 1)) >= nqt) || (QA_FLAGS & 16)) continue;
                const float* kadd = m_keyadd + half * SUB + (g0 - sb.x);
                const float* base = stg + (g0 - sb.x) * QA_LDROW;           // staged rows of the pair: [Q 0..63 | K 64..127 | V 128..191]
                float add[MAXT][4];
#pragma unroll
                for (int jt = 0; jt < MAXT; ++jt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int j = jt * 16 + fk * 4 + r;
                        add[jt][r] = j < S ? kadd[j] : -INFINITY;
                    }
#pragma unroll
                for (int qt = 0; qt < MAXT; ++qt) {
                    if (qt * 16 >= S || ((first + qt) & (NW - 1)) != wave) continue;
                    f32x4 sc[MAXT];
                    f32x4 o[4];
                    float4 qf[4];
                    {
                        int i = qt * 16 + fr;
                        i = i < S ? i : S - 1;
#pragma unroll
                        for (int s4 = 0; s4 < 4; ++s4) qf[s4] = *reinterpret_cast<const float4*>(base + i * QA_LDROW + s4 * 16 + fk * 4);
                    }
#pragma unroll
                    for (int jt = 0; jt < MAXT; ++jt) {
                        f32x4 a4 = {0.f, 0.f, 0.f, 0.f};
                        if (jt * 16 < S && !(QA_FLAGS & 2)) {
                            // K fragment of this key tile (attn.hip's layout: row j = 16 jt + fr, d = 16 s + 4 fk + 0..3), re-read per query
                            // tile: LDS reads are cheap here, registers are not (the other sub-tile's accumulators are still live)
                            int j = jt * 16 + fr;
                            j = j < S ? j : S - 1;
                            float4 kf[1][4];
#pragma unroll
                            for (int s4 = 0; s4 < 4; ++s4) kf[0][s4] = *reinterpret_cast<const float4*>(base + j * QA_LDROW + 64 + s4 * 16 + fk * 4);
#pragma unroll
                            for (int s4 = 0; s4 < 4; ++s4) {
                                a4 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[0][s4].x, qf[s4].x, a4, 0, 0, 0);
                                a4 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[0][s4].y, qf[s4].y, a4, 0, 0, 0);
                                a4 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[0][s4].z, qf[s4].z, a4, 0, 0, 0);
                                a4 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[0][s4].w, qf[s4].w, a4, 0, 0, 0);
                            }
                        }
                        sc[jt] = a4;
                    }
                    float m = -INFINITY;
#pragma unroll
                    for (int jt = 0; jt < MAXT; ++jt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float v = sc[jt][r] * 0.125f + add[jt][r];
                            sc[jt][r] = v;
                            m = fmaxf(m, v);
                        }
                    m = rows4_max(m);
                    float sum = 0.f;
#pragma unroll
                    for (int jt = 0; jt < MAXT; ++jt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float e = expf(sc[jt][r] - m);
                            sc[jt][r] = e;
                            sum += e;
                        }
                    sum = rows4_sum(sum);
                    const float inv = 1.0f / sum;
#pragma unroll
                    for (int jt = 0; jt < MAXT; ++jt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) sc[jt][r] *= inv;
                    // O = P V: V fragment rows j = 16 jt + 4 fk + r, columns d = 4 fr + 0..3
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int jt = 0; jt < MAXT; ++jt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if (jt * 16 >= S || (QA_FLAGS & 1)) continue;
                            int j = jt * 16 + fk * 4 + r;
                            j = j < S ? j : S - 1;       // P is exactly 0 there
                            const float4 vf = *reinterpret_cast<const float4*>(base + j * QA_LDROW + 128 + fr * 4);
                            const float pv = sc[jt][r];
                            o[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(pv, vf.x, o[0], 0, 0, 0);
                            o[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(pv, vf.y, o[1], 0, 0, 0);
                            o[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(pv, vf.z, o[2], 0, 0, 0);
                            o[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(pv, vf.w, o[3], 0, 0, 0);
                        }
                    // lane holds O[i = 16 qt + 4 fk + r][d = 4 fr + dt]
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = qt * 16 + fk * 4 + r;
                        if (i >= S || (QA_FLAGS & 8)) continue;
                        const long long off = (long long)(g0 + i) * p.ldo + ehead * MMS_HEAD_DIM + fr * 4;
                        bf16x4 hi, lo;
#pragma unroll
                        for (int dt = 0; dt < 4; ++dt) {
                            bf16 x, y;
                            split_bf16(o[dt][r], x, y);
                            hi[dt] = x;
                            lo[dt] = y;
                        }
                        *reinterpret_cast<bf16x4*>(plane_ptr(p.o_hi, off)) = hi;
                        *reinterpret_cast<bf16x4*>(plane_ptr(p.o_lo, off)) = lo;
                    }
                }
            }
            }
            qa_barrier();
#ifdef MMS_LAB
            if (p.trace) tr[3 + 2 * half] = __builtin_readcyclecounter();
#endif
        }
#ifdef MMS_LAB
        auto dump_trace = [&]() {
            if (p.trace && tid == 0 && tile_i < 8) {
#pragma unroll
                for (int k = 0; k < 16; ++k) p.trace[((long long)blockIdx.x * 8 + tile_i) * 16 + k] = tr[k];
            }
            ++tile_i;
        };
        if (!more) { tr[6] = tr[5]; dump_trace(); }
#endif
        if (!more) break;
#pragma unroll
        for (int q = 0; q < P; ++q) issue(q, 1, 1);       // slots 1 / 2 were under the staging area until the barrier above
        // stage 0 has landed once at most P operations are outstanding: loads complete in issue order, so the six pieces just issued are
        // all outstanding as long as any older load is (the context stores in between only add to the count)
        qa_wait_vmcnt<P>();
        qa_barrier();
#ifdef MMS_LAB
        QA_STAMP(6);
        dump_trace();
#endif
        if (wave >= NW / 2) qa_barrier();    // stagger again
    }
}

// =====================================================================================================================================
// qkv_attn2_kernel: the split-bf16 route (mms_config.fuse_attention = 2) of precision mode 2, wave grid 8(M) x 1(N).
//
// Same projection engine (256-row tile = two sub-tiles of whole pairs, one head = 192 columns, 3-slot LDS-DMA ring, one phase of 48 MFMAs per wave and
// stage), but every wave owns 16 rows of EACH sub-tile and all 192 columns of them: its accumulators hold Q, K and V of two 16-query tiles.  What that buys
// (tools/qa_trace.py, profiles/rd5*_qa_trace*.txt: in the 4 x 2 layout a sub-tile's staging was written by four waves, one per SIMD -- a single wave issues one
// VALU instruction per ~4.5 cycles, tools/probes/valu_rate_probe.hip -- and Q went through LDS like K and V):
//  * Q never leaves the registers: accumulator fragment pairs (2 b, 2 b + 1) of lane (row, nq) ARE the second MFMA operand of S^T = K Q^T for the contraction
//    slots d = 32 b + 16 h + 4 nq + e (h = fragment of the pair, e < 4), after one split into hi / lo bf16;
//  * K and V are staged in that slot order ("positions": p = 32 b + 8 nq + 4 h + e), so a lane's two fragments of a pair are ONE 16-byte store per plane (no
//    cross-lane exchange, conflict-free at the 528-byte row stride) and a K fragment is ONE ds_read_b128; all eight waves write, 16 rows each;
//  * per sub-tile: stage K / V (8 stores per lane) -> barrier -> every wave attends its own 16 queries (block-diagonal over the pairs they belong to, online
//    softmax over 64-key steps, V^T fragments by ds_read_b64_tr_b16) -> barrier.  The context comes out in position order: lane (query fr, fk) holds d = 32 b +
//    16 (fk & 1) + 8 (dt & 1) + 4 (fk >> 1) + r of fragment dt = 2 b + (dt & 1); v_permlane32_swap pairs them into 16-byte stores.
// Scores carry log2(e) / sqrt(64) (folded into Q before its split), the additive key mask log2(e): probabilities are one v_exp_f32 away.
// =====================================================================================================================================
__global__ __launch_bounds__(512) void qkv_attn2_kernel(const QkvAttnParams p) {
    constexpr int SUB = QA_SUB, BM = 2 * SUB, NW = 8, FM = 2, FN = 12, BN = 192;
    constexpr int AREG = BM * 128, WREG = BN * 64, SLOT = 48 * 1024, NAP = BM / 64, NSLOT = 3, D = 2, P = NAP + 2;
    constexpr int LDK = 528;      // staged row: [K_hi | K_lo | V_hi | V_lo] x 64 bf16 in position order + 16 B (4 banks mod 32: 16-byte stores of 8 rows and fragment reads are conflict-free)
    constexpr float QSCALE = 0.125f * 1.44269504088896340736f, LOG2E = 1.44269504088896340736f;
    static_assert(AREG + WREG <= SLOT && NSLOT * SLOT - SUB * LDK >= AREG + WREG, "slot 0's stage lies below the K / V staging area");
    __shared__ __attribute__((aligned(16))) unsigned char smem[NSLOT * SLOT];
    __shared__ __attribute__((aligned(16))) float m_keyadd[256 + 64];     // additive key mask (x log2 e) by tile row (+ 64: a step may run past the sub-tile)
    __shared__ __attribute__((aligned(16))) float m_bias[192];            // this head's [Q | K | V] bias
    __shared__ int m_rowmeta[256];                                        // per tile row: the sub-tile rows [kbeg, kend) its query attends (kbeg | kend << 16)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_sub = *p.n_sub;
    const int nbm = (n_sub + 1) >> 1, nblk = nbm * MMS_HEADS;
    const bool cross = p.sub2 != nullptr;
    if (p.flop_counter && blockIdx.x == 0 && tid == 0) {
        int Meff = p.M;
        if (p.m_dev) { const int md = *p.m_dev; Meff = md < Meff ? md : Meff; }
        if (cross) { int m2 = p.M2; if (p.m_dev2) { const int md = *p.m_dev2; m2 = md < m2 ? md : m2; } Meff += m2; }
        atomicAdd(p.flop_counter, 2ull * (unsigned long long)Meff * (unsigned long long)(3 * MMS_HIDDEN) * (unsigned long long)p.K);
    }
    int vb = blockIdx.x;
    if (vb >= nblk) return;
    if (tid < 64) m_keyadd[256 + tid] = 0.f;

    const int gr_l = lane >> 2, gc = lane & 3;
    const bf16* a_src[NAP];
    const bf16* w_src[2];
    int head;
    int4 sub0, sub1;
    int2 sub0b = make_int2(0, 0), sub1b = sub0b;
    int meta_a = 0, meta_b = 0;          // thread < 256: key mask of tile row tid, bias[tid]; else: the plan's record of pair (tid - 256) of the tile
    int nhead = 0;
    int4 nsub0 = make_int4(0, 0, 0, 0), nsub1 = nsub0;
    int2 nsub0b = make_int2(0, 0), nsub1b = nsub0b;
    auto locate = [&](int v) {      // bijective XCD remap (virtual block v runs on XCD v % 8): the twelve heads of a row tile stay on one XCD's L2
        const int q = nblk >> 3, r8 = nblk & 7, xcd = v & 7, loc = v >> 3;
        int bid = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + loc;
        if (p.reverse) bid = nblk - 1 - bid;
        const int nb = bid / MMS_HEADS;
        nhead = __builtin_amdgcn_readfirstlane(bid % MMS_HEADS);
        auto uni = [](int4 v) {
            return make_int4(__builtin_amdgcn_readfirstlane(v.x), __builtin_amdgcn_readfirstlane(v.y), __builtin_amdgcn_readfirstlane(v.z),
                             __builtin_amdgcn_readfirstlane(v.w));
        };
        nsub0 = uni(p.sub[2 * nb]);
        nsub1 = 2 * nb + 1 < n_sub ? uni(p.sub[2 * nb + 1]) : make_int4(nsub0.x, 0, 0, 0);
        if (cross) {
            const int4 a = uni(p.sub2[2 * nb]), b = 2 * nb + 1 < n_sub ? uni(p.sub2[2 * nb + 1]) : make_int4(a.x, 0, 0, 0);
            nsub0b = make_int2(a.x, a.y); nsub1b = make_int2(b.x, b.y);
        }
    };
    auto grow = [&](const int4& sa, const int2& sb2, int lr) -> long long {      // plane-buffer row behind row lr of a sub-tile (stream-1 rows first, then its stream-2 rows)
        return lr < sa.y ? (long long)(sa.x + lr) : p.row0_b + (long long)(sb2.x + (lr - sa.y));
    };
    auto setup = [&]() {
        head = nhead; sub0 = nsub0; sub1 = nsub1; sub0b = nsub0b; sub1b = nsub1b;
#pragma unroll
        for (int q4 = 0; q4 < NAP; ++q4) {
            const int r = q4 * 64 + wave * 8 + (lane >> 3);
            const int4 sb = (r / SUB) ? sub1 : sub0;
            const int2 sb2 = make_int2((r / SUB) ? sub1b.x : sub0b.x, (r / SUB) ? sub1b.y : sub0b.y);
            const int nlive = sb.y + sb2.y;
            int lr = r % SUB;
            lr = lr < nlive ? lr : (nlive > 0 ? nlive - 1 : 0);
            const long long gr = nlive > 0 ? grow(sb, sb2, lr) : (long long)sb.x;
            a_src[q4] = p.a_hi + 2 * (gr * (long long)p.lda) + ((lane & 7) ^ ((r >> 1) & 7)) * 8;
        }
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const int r = h2 ? 128 + (wave & 3) * 16 + gr_l : wave * 16 + gr_l;
            w_src[h2] = p.w + wtile_off(head * BN + r, 0, p.K) + (gc ^ qa_swz(r)) * 8;
        }
    };
    // ... and its metadata (key mask, bias, pair records): fetched at the END of the epilogue, so that the values are live across the main loop only -- requested in front
    // of the attention phases they were spilled to scratch at once, and every wave sat out the full latency of the loads (2.3 k cycles per tile)
    auto setup_meta = [&]() -> int2 {      // (returned, not written through the capture: as captured variables the two values lived on the stack)
        const int t = tid & 255;
        const int u = tid < 256 ? (t >= SUB) : (t >> 7), i = tid < 256 ? t - u * SUB : (t & 127);
        const int4 sb = u ? sub1 : sub0;
        const int2 sb2 = make_int2(u ? sub1b.x : sub0b.x, u ? sub1b.y : sub0b.y);
        int ma = 0, mb = 0;
        if (tid < 256) {
            if (i < sb.y) { if (p.key_add) ma = __float_as_int(p.key_add[sb.x + i]); }
            else if (i < sb.y + sb2.y) { if (p.key_add2) ma = __float_as_int(p.key_add2[sb2.x + (i - sb.y)]); }
            if (tid < BN) mb = __float_as_int(p.bias[head * BN + tid]);
        } else if (i < sb.w) ma = p.pair_rec[sb.z + i];
        return make_int2(ma, mb);
    };
    locate(vb);
    setup();
    { const int2 mm = setup_meta(); meta_a = mm.x; meta_b = mm.y; }
    auto issue = [&](int q, int st, int slot) {
        unsigned char* d;
        const bf16* s;
        if (q < NAP) { d = smem + slot * SLOT + q * 8192 + wave * 1024; s = a_src[q] + st * 64; }
        else {
            const int h2 = q - NAP;
            d = smem + slot * SLOT + AREG + (h2 ? 8192 + (wave & 3) * 1024 : wave * 1024);
            s = w_src[h2] + st * 512;
        }
        __builtin_amdgcn_global_load_lds((glb_void*)s, (lds_void*)d, 16, 0, 0);
    };

    f32x4 acc[FM][FN];
    const int ns = p.K / 32;
    const int fr = lane & 15, fk = lane >> 4;
    const int laneA = (wave * 16 + fr) * 128 + ((fk ^ ((fr >> 1) & 7)) << 4);      // rows 16 wave + fr of sub-tile 0; sub-tile 1: + SUB * 128; lo: chunk ^ 4
    const int laneB = AREG + fr * 64 + ((fk ^ qa_swz(fr)) << 4);                    // column fragment j: + j * 16 * 64

    auto stage = [&](auto pre_tag, auto wait_tag, int s, int slot) {
        constexpr bool PRE = decltype(pre_tag)::value;
        constexpr int WAITN = decltype(wait_tag)::value;
        const unsigned char* sb = smem + slot * SLOT;
        const int nslot = slot == 0 ? NSLOT - 1 : slot - 1;
        bf16x8 a[2][FM], b[FN];
#pragma unroll
        for (int j = 0; j < FN; ++j) b[j] = *reinterpret_cast<const bf16x8*>(sb + laneB + j * 16 * 64);
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int i = 0; i < FM; ++i) a[pl][i] = *reinterpret_cast<const bf16x8*>(sb + (laneA ^ (pl << 6)) + i * SUB * 128);
        if (PRE) {
#pragma unroll
            for (int q = 0; q < P; ++q) issue(q, s + D, nslot);
        }
        if (WAITN >= 0) qa_wait_vmcnt<(WAITN >= 0 ? WAITN : 0)>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        qa_barrier();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int pass = 0; pass < 2; ++pass)
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[pass][i], acc[i][j], 0, 0, 0);   // swapped operands: C^T fragment
        __builtin_amdgcn_s_setprio(0);
        qa_barrier();
    };
#pragma unroll
    for (int d = 0; d < D; ++d) {
#pragma unroll
        for (int q = 0; q < P; ++q) issue(q, d, d);
    }
    qa_wait_vmcnt<P>();
    qa_barrier();
    if (wave >= NW / 2) qa_barrier();     // stagger the two halves by one barrier

    unsigned char* stgb = smem + NSLOT * SLOT - SUB * LDK;
#ifdef MMS_LAB
    const int QA_FLAGS = p.lab_flags;     // timing only (results WRONG): 1 no P V MFMAs, 2 no Q K^T MFMAs, 8 no context stores, 16 no attention, 32 no staging stores
    unsigned long long tr[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int tile_i = 0;
#define QA2_STAMP(k) do { if (p.trace) tr[k] = __builtin_readcyclecounter(); } while (0)
#else
    constexpr int QA_FLAGS = 0;
#define QA2_STAMP(k) do { } while (0)
#endif
    for (;;) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        QA2_STAMP(0);
        const bool more = vb + (int)gridDim.x < nblk;
        if (more) locate(vb + (int)gridDim.x);
        int slot = 0, s = 0;
        for (; s + D < ns; ++s) {
            stage(std::true_type{}, std::integral_constant<int, (D - 1) * P>{}, s, slot);
            slot = slot == NSLOT - 1 ? 0 : slot + 1;
        }
        stage(std::false_type{}, std::integral_constant<int, 0>{}, s, slot);         // stage ns-2: the last stage must have landed
        slot = slot == NSLOT - 1 ? 0 : slot + 1;
        ++s;
        // park this tile's metadata in LDS (every wave passes a barrier between its own writes and the first read)
        if (wave >= NW / 2) {
            int tid_here = tid;
            asm volatile("" : "+v"(tid_here));      // opaque: the LDS addresses below are computed HERE -- hoisted above the main loop they were spilled to scratch
            const int t = tid_here - 256, u = t >> 7;
            const int4 sa = u ? sub1 : sub0;
            if ((t & 127) < sa.w) {      // the pair's rows learn which sub-tile rows their queries attend: its own rows, or (CROSS) its rows in the other stream
                const int r1 = meta_a & 255, c1 = (meta_a >> 8) & 255;
                // (plain loops of <= 48 trips: unrolled and vectorised they cost a VGPR spill whose reloads -- s_waitcnt vmcnt(0) -- drained the ring's LDS-DMA loads)
                if (cross) {
                    const int r2 = sa.y + ((meta_a >> 16) & 255), c2 = (meta_a >> 24) & 255;
#pragma clang loop unroll(disable) vectorize(disable)
                    for (int r = 0; r < c1; ++r) m_rowmeta[u * SUB + r1 + r] = r2 | ((r2 + c2) << 16);
#pragma clang loop unroll(disable) vectorize(disable)
                    for (int r = 0; r < c2; ++r) m_rowmeta[u * SUB + r2 + r] = r1 | ((r1 + c1) << 16);
                } else {
#pragma clang loop unroll(disable) vectorize(disable)
                    for (int r = 0; r < c1; ++r) m_rowmeta[u * SUB + r1 + r] = r1 | ((r1 + c1) << 16);
                }
            }
        }
        stage(std::false_type{}, std::integral_constant<int, -1>{}, s, slot);        // stage ns-1
        if (wave < NW / 2) {
            int tid_here = tid;
            asm volatile("" : "+v"(tid_here));      // (as above)
            m_keyadd[tid_here] = __int_as_float(meta_a) * LOG2E;
            const int u = tid_here >= SUB, i = tid_here - u * SUB;
            if (i >= (u ? sub1.y + sub1b.y : sub0.y + sub0b.y)) m_rowmeta[tid_here] = 0;      // rows behind the last pair attend nothing
            if (tid_here < BN) m_bias[tid_here] = __int_as_float(meta_b);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            qa_barrier();     // re-align the halves: nobody reads the ring any more
        }
        QA2_STAMP(1);

        // this tile's identity for the epilogue; then the next tile's addresses and its stage 0 (slot 0 is idle and lies below the staging area)
        const int ehead = head;
        const int4 esub0 = sub0, esub1 = sub1;
        const int2 esub0b = sub0b, esub1b = sub1b;
        if (more) vb += (int)gridDim.x;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int4 sa = s2 ? esub1 : esub0;
            const int2 sa2 = make_int2(s2 ? esub1b.x : esub0b.x, s2 ? esub1b.y : esub0b.y);
            const int nlive = sa.y + sa2.y;
            // ---- Q operands of this wave's 16 queries, from its own accumulators ----
            bf16x8 qh[2], ql[2];
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(m_bias + 16 * (2 * b + h) + 4 * fk);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { bf16 x, y; split_bf16((acc[s2][2 * b + h][e] + b4[e]) * QSCALE, x, y); qh[b][4 * h + e] = x; ql[b][4 * h + e] = y; }
                }
            // ---- K / V rows of this wave -> staging, position order ----
            {
                unsigned char* dst = stgb + (wave * 16 + fr) * LDK + fk * 16;
#pragma unroll
                for (int pr = 0; pr < 4; ++pr) {      // fragment pairs (4, 5) (6, 7): K blocks 0, 1;  (8, 9) (10, 11): V blocks 0, 1
                    bf16x8 hi, lo;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int j = 4 + 2 * pr + h;
                        const f32x4 b4 = *reinterpret_cast<const f32x4*>(m_bias + 16 * j + 4 * fk);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { bf16 x, y; split_bf16(acc[s2][j][e] + b4[e], x, y); hi[4 * h + e] = x; lo[4 * h + e] = y; }
                    }
                    if (QA_FLAGS & 32) continue;
                    *reinterpret_cast<bf16x8*>(dst + (pr >> 1) * 256 + (pr & 1) * 64) = hi;
                    *reinterpret_cast<bf16x8*>(dst + (pr >> 1) * 256 + (pr & 1) * 64 + 128) = lo;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            qa_barrier();
            QA2_STAMP(2 + 2 * s2);
            // the next tile's addresses and its stage 0 (slot 0 lies below the staging area), half of the pieces in front of each attention phase: 48 pieces
            // issued by all waves at once are a ~2.6 k cycle burst on the CU's one load path; spread, they land while the waves compute
            if (more) {
                if (s2 == 0) setup();
#pragma unroll
                for (int q = 0; q < P / 2; ++q) issue(s2 * (P / 2) + q, 0, 0);
            }
            // ---- attention of the wave's 16 queries: block-diagonal over the pairs they belong to ----
            if (wave * 16 < nlive && !(QA_FLAGS & 16)) {
                const int i = wave * 16 + fr;
                const bool live = i < nlive;
                const int rm = m_rowmeta[s2 * SUB + (live ? i : nlive - 1)];
                const int kbeg = rm & 0xffff, kend = (int)((unsigned)rm >> 16);
                const int kb = __builtin_amdgcn_readfirstlane(qa_row16_min(live ? kbeg : 0x7fff)) & ~3;
                const int ke = __builtin_amdgcn_readfirstlane(qa_row16_max(live ? kend : 0));
                const float* kadd = m_keyadd + s2 * SUB;
                float m = -INFINITY, l = 0.f;
                f32x4 o[4];
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
                // one step = NH halves of 32 keys (compile-time: no branch inside, so the score tiles' LDS reads and MFMA chains interleave)
                auto step = [&](auto nh_tag, const int kc) {
                    constexpr int NH = decltype(nh_tag)::value;
                    // S^T tile t: first operand = K rows (key j = kc + 16 t + fr), second = the Q rows; lane gets keys kc + 16 t + 4 fk + r of query fr
                    f32x4 sc[2 * NH];
#pragma unroll
                    for (int t = 0; t < 2 * NH; ++t) {
                        f32x4 a4 = {0.f, 0.f, 0.f, 0.f};
                        if (!(QA_FLAGS & 2)) {
                            int j = kc + 16 * t + fr;
                            j = j < SUB ? j : SUB - 1;
                            const unsigned char* kr = stgb + j * LDK + 16 * fk;
                            const bf16x8 kh0 = *reinterpret_cast<const bf16x8*>(kr), kh1 = *reinterpret_cast<const bf16x8*>(kr + 64);
                            const bf16x8 kl0 = *reinterpret_cast<const bf16x8*>(kr + 128), kl1 = *reinterpret_cast<const bf16x8*>(kr + 192);
                            a4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh0, qh[0], a4, 0, 0, 0);
                            a4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh0, ql[0], a4, 0, 0, 0);
                            a4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kl0, qh[0], a4, 0, 0, 0);
                            a4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh1, qh[1], a4, 0, 0, 0);
                            a4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh1, ql[1], a4, 0, 0, 0);
                            a4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kl1, qh[1], a4, 0, 0, 0);
                        }
                        sc[t] = a4;
                    }
                    // additive key mask; keys outside the query's own range do not exist for it
                    const int j0 = kc + 4 * fk;
                    float sv[8 * NH];
                    float cm = -INFINITY;
#pragma unroll
                    for (int t = 0; t < 2 * NH; ++t) {
                        const f32x4 ka = *reinterpret_cast<const f32x4*>(kadd + j0 + 16 * t);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int j = j0 + 16 * t + r;
                            const float v = (j >= kbeg && j < kend) ? sc[t][r] + ka[r] : -INFINITY;
                            sv[4 * t + r] = v;
                            cm = fmaxf(cm, v);
                        }
                    }
                    cm = rows4_max(cm);
                    const float mn = fmaxf(m, cm);
                    const float ms = mn == -INFINITY ? 0.f : mn;          // (no key of this query so far)
                    float ps = 0.f;
#pragma unroll
                    for (int e = 0; e < 8 * NH; ++e) {
                        sv[e] = __builtin_amdgcn_exp2f(sv[e] - ms);
                        ps += sv[e];
                    }
                    ps = rows4_sum(ps);
                    if (kc != kb) {      // (first step: the sums are still zero)
                        const float alpha = __builtin_amdgcn_exp2f(m - ms);
                        l *= alpha;
#pragma unroll
                        for (int dt = 0; dt < 4; ++dt) o[dt] *= alpha;
                    }
                    l += ps;
                    m = mn;
                    // O^T += V^T P^T over each half's 32 key slots: slot (fk, e) = key kc + 32 h + 16 (e >> 2) + 4 fk + (e & 3) -- the lane's own probabilities are the
                    // second operand; first operand = V^T[position 16 dt + fr][slot]: two transposing reads of the [4 keys][16 positions] blocks at keys + 4 fk, + 16 + 4 fk
#pragma unroll
                    for (int h = 0; h < NH; ++h) {
                        if (QA_FLAGS & 1) continue;
                        bf16x8 ph, pl;
#pragma unroll
                        for (int e = 0; e < 8; ++e) { bf16 x, y; split_bf16(sv[8 * h + e], x, y); ph[e] = x; pl[e] = y; }
                        int r0 = kc + 32 * h + 4 * fk + (fr >> 2), r1 = r0 + 16;
                        r0 = r0 < SUB ? r0 : SUB - 1;      // (P is exactly 0 there)
                        r1 = r1 < SUB ? r1 : SUB - 1;
                        const unsigned char* v0 = stgb + r0 * LDK + 256 + (fr & 3) * 8;
                        const unsigned char* v1 = stgb + r1 * LDK + 256 + (fr & 3) * 8;
#pragma unroll
                        for (int dt = 0; dt < 4; ++dt) {
                            const bf16x8 vh = __builtin_bit_cast(bf16x8, __builtin_shufflevector(qa_tr16(v0 + 32 * dt), qa_tr16(v1 + 32 * dt), 0, 1, 2, 3, 4, 5, 6, 7));
                            const bf16x8 vl = __builtin_bit_cast(bf16x8, __builtin_shufflevector(qa_tr16(v0 + 128 + 32 * dt), qa_tr16(v1 + 128 + 32 * dt), 0, 1, 2, 3, 4, 5, 6, 7));
                            o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh, ph, o[dt], 0, 0, 0);
                            o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vl, ph, o[dt], 0, 0, 0);
                            o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh, pl, o[dt], 0, 0, 0);
                        }
                    }
                };
#pragma unroll 1
                for (int kc = kb; kc < ke; kc += 64) {
                    if (kc + 32 < ke) step(std::integral_constant<int, 2>{}, kc);
                    else step(std::integral_constant<int, 1>{}, kc);
                }
                // lane holds O[query fr][position 16 dt + 4 fk + r] = d 32 (dt >> 1) + 16 (fk & 1) + 8 (dt & 1) + 4 (fk >> 1) + r.  v_permlane32_swap of the fragments
                // dt = 2 b (first) and 2 b + 1 (second) between the lane rows fk and fk + 2 leaves every lane EIGHT consecutive d: 32 b + 16 (fk & 1) + 8 (fk >> 1) + 0..7
                {
                    const float inv = __builtin_amdgcn_rcpf(l);
                    const long long off = (live ? grow(sa, sa2, i) : 0) * p.ldo + ehead * MMS_HEAD_DIM + (fk & 1) * 16 + (fk >> 1) * 8;
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        unsigned w[2][2][2];      // [plane][fragment of the pair][dword]: two bf16 each
#pragma unroll
                        for (int t2 = 0; t2 < 2; ++t2) {
                            bf16x4 hi, lo;
#pragma unroll
                            for (int r = 0; r < 4; ++r) { bf16 x, y; split_bf16(o[2 * b + t2][r] * inv, x, y); hi[r] = x; lo[r] = y; }
                            const u32x2_t h2 = __builtin_bit_cast(u32x2_t, hi), l2 = __builtin_bit_cast(u32x2_t, lo);
                            w[0][t2][0] = h2[0]; w[0][t2][1] = h2[1]; w[1][t2][0] = l2[0]; w[1][t2][1] = l2[1];
                        }
#pragma unroll
                        for (int pl2 = 0; pl2 < 2; ++pl2) {
                            u32x4 out;
#pragma unroll
                            for (int dw = 0; dw < 2; ++dw) {
                                const auto sw = __builtin_amdgcn_permlane32_swap(w[pl2][0][dw], w[pl2][1][dw], false, false);
                                out[dw] = sw[0]; out[2 + dw] = sw[1];
                            }
                            if (live && !(QA_FLAGS & 8)) *reinterpret_cast<u32x4*>(plane_ptr(pl2 ? p.o_lo : p.o_hi, off + 32 * b)) = out;
                        }
                    }
                }
            }
            qa_barrier();
            QA2_STAMP(3 + 2 * s2);
        }
#ifdef MMS_LAB
        auto dump_trace = [&]() {
            if (p.trace && tid == 0 && tile_i < 8) {
#pragma unroll
                for (int k = 0; k < 16; ++k) p.trace[((long long)blockIdx.x * 8 + tile_i) * 16 + k] = tr[k];
            }
            ++tile_i;
        };
        if (!more) { tr[6] = tr[5]; dump_trace(); }
#endif
        if (!more) break;
#pragma unroll
        for (int q = 0; q < P; ++q) issue(q, 1, 1);       // slots 1 / 2 were under the staging area until the barrier above
        qa_wait_vmcnt<P>();      // stage 0 has landed once at most P operations are outstanding (loads complete in issue order; the context stores only add to the count)
        qa_barrier();
        { const int2 mm = setup_meta(); meta_a = mm.x; meta_b = mm.y; }            // (behind the stage-1 pieces: the main loop's counted waits see these loads complete in front of every piece they wait for)
#ifdef MMS_LAB
        QA2_STAMP(6);
        dump_trace();
#endif
        if (wave >= NW / 2) qa_barrier();    // stagger again
    }
}

bool launch_qkv_attn(const QkvAttnParams& p, hipStream_t st) {
    if (p.M <= 0) return true;
    if (p.K % 64 || p.K < 128 || p.S <= 0 || !p.sub || !p.n_sub || (p.fast && !p.pair_rec)) return false;
    const bool cross = p.sub2 != nullptr;
    const int sub_rows = p.w_lo ? QA_SUB3 : QA_SUB;
    // exact-fp32 route: per-pair score tiles in registers (<= 48 tokens), one stream; fast route: any pair that fits a sub-tile, one stream or a pair of them
    if (p.fast ? (p.S + (cross ? p.S2 : 0) > sub_rows) : (p.S > 48 || cross)) return false;
    const int n_cu = device_cu_count();
    // an upper bound of the tile count (the live count is on the device): every sub-tile but the last of a stream holds > sub_rows - S rows
    const long long rows = (long long)p.M + (cross ? p.M2 : 0), smax = p.S + (cross ? p.S2 : 0);
    const long long max_sub = rows / (sub_rows - smax + 1) + p.M / QA_SEG + 3, max_blk = (max_sub + 1) / 2 * MMS_HEADS;
    const dim3 grid((unsigned)(max_blk < n_cu ? max_blk : n_cu)), block(512);
    auto go = [&](const QkvAttnParams& q) {
        const bool fast = q.fast != 0, big = q.S > 32;
        if (q.w_lo) {      // three-pass projection (precision mode 3)
            if (fast) hipLaunchKernelGGL((qkv_attn_kernel<2, true, 2>), grid, block, 0, st, q);
            else if (big) hipLaunchKernelGGL((qkv_attn_kernel<3, false, 2>), grid, block, 0, st, q);
            else hipLaunchKernelGGL((qkv_attn_kernel<2, false, 2>), grid, block, 0, st, q);
        } else if (fast) {      // 8 x 1 wave layout (qkv_attn_kernel<2, true, 1> is the same route in the 4 x 2 layout: lab A/B, MMS_QA_LAYOUT=42)
#ifdef MMS_LAB
            static const bool old_layout = getenv("MMS_QA_LAYOUT") && atoi(getenv("MMS_QA_LAYOUT")) == 42;
            if (old_layout) { hipLaunchKernelGGL((qkv_attn_kernel<2, true, 1>), grid, block, 0, st, q); return; }
#endif
            hipLaunchKernelGGL(qkv_attn2_kernel, grid, block, 0, st, q);
        }
        else if (big) hipLaunchKernelGGL((qkv_attn_kernel<3, false, 1>), grid, block, 0, st, q);
        else hipLaunchKernelGGL((qkv_attn_kernel<2, false, 1>), grid, block, 0, st, q);
    };
#ifdef MMS_LAB
    static const bool want_trace = getenv("MMS_QA_TRACE") != nullptr;      // first launch only: /tmp/qa_trace.bin = [workgroup][8 tiles][8] u64
    static bool traced = false;
    if (want_trace && !traced) {
        traced = true;
        const size_t bytes = (size_t)grid.x * 8 * 16 * 8;
        unsigned long long* buf = nullptr;
        if (hipMalloc((void**)&buf, bytes) != hipSuccess) return false;
        (void)hipMemsetAsync(buf, 0, bytes, st);
        QkvAttnParams q = p;
        q.trace = buf;
        if (const char* e = getenv("MMS_QA_FLAGS")) q.lab_flags = atoi(e);
        go(q);
        std::vector<unsigned long long> hst(bytes / 8);
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(hst.data(), buf, bytes, hipMemcpyDeviceToHost);
        (void)hipFree(buf);
        if (FILE* f = fopen("/tmp/qa_trace.bin", "wb")) { fwrite(hst.data(), 8, hst.size(), f); fclose(f); }
        return true;
    }
#endif
    go(p);
    return true;
}
