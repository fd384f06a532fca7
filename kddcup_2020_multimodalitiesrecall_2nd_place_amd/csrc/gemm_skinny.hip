// Skinny GEMM for the smallest calls: C[M][N] = act(A W^T + bias) on a handful of token rows (the reference's zk driver scores ONE pair per
// sess.run: 30 rows, evaluate_normal.py:15), precision modes 2 and 3 (A = hi + lo bf16 planes, W = tiled bf16 -- with a lo plane in mode 3).  api.hip takes it for launches of
// <= 128 padded rows; the kernel itself handles any M <= 512 in row blocks of 128 (tested to 256).
//
// Such a launch is pure latency: the 128x256 tile engine puts 3 .. 12 workgroups on the chip, each walking K serially through LDS
// (25 .. 42 us), and the split-K route that replaced it needs a second launch to sum the partials (10 + 6 us).  Here the chip is filled
// the other way round: one workgroup per 16 OUTPUT COLUMNS (N / 16 = 48 .. 192 workgroups), KS waves each contracting a K / KS slice of
// the whole row panel (KS = the split factor of the tile-engine route it replaces: bit-identical to it, see launch_gemm_skinny).  No LDS staging at all -- a 16 x 32 weight tile of the tiled layout (common.h wtile_off) IS one B fragment of
// v_mfma_f32_16x16x32_bf16 (lane l: row l & 15, k group l >> 4: the wave reads the tile's contiguous KiB), and a 32-wide K block of a
// row in the hl32 plane layout is [hi 64 B | lo 64 B], so the A fragments are 16-byte loads too; every operand byte is loaded by
// the lane that feeds it to the matrix pipe.  The KS partial accumulators meet in LDS (fixed order: deterministic), then all threads run
// the epilogue on float4 (bias, activation, fp32 row-major / head-major or split planes).  One launch, no partial buffer in HBM.
// zk at 1 pair: 0.99 -> 0.59 ms per call, lxmert 2.23 -> 1.31 ms (profiles/rd4r_skinny_gemm.txt).  Every workgroup reads the whole A row
// block, so the cost grows with M x N / 16: past ~128 rows the tile engine with split-K wins again (same file).
#include "kernels.h"

namespace {

constexpr int SKINNY_MAX_ROWS = 512;

// WPL: weight planes -- 1: precision mode 2 (a_hi w + a_lo w), 2: precision mode 3 (w = w_hi + w_lo: a_hi w_hi + a_lo w_hi + a_hi w_lo, the tile engine's order)
template <int FM, int KS, int ACT, int U, int WPL>
__global__ __launch_bounds__(64 * KS) void gemm_skinny_kernel(const GemmParams p) {
    __shared__ __attribute__((aligned(16))) float red[KS][FM * 16][16];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fk = lane >> 4;
    int Meff = p.M;
    if (p.m_dev) { const int md = *p.m_dev; Meff = md < Meff ? md : Meff; }
    if (p.flop_counter && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && tid == 0)
        atomicAdd(p.flop_counter, 2ull * (unsigned long long)Meff * (unsigned long long)p.N * (unsigned long long)p.K);
    const int r0 = blockIdx.y * (FM * 16);      // row block (launches of more than 128 rows: blocks of 128, FM = 8)
    if (Meff <= r0) return;
    const int n0 = blockIdx.x * 16;
    // K slices: KS waves of this workgroup x gridDim.z workgroups (gridDim.z > 1: the slice sums leave as fp32 partials, GemmParams::c_split_stride apart)
    const int nk = p.K >> 5, per = nk / (KS * (int)gridDim.z), kt0 = ((int)blockIdx.z * KS + wave) * per;

    f32x4 acc[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // the lane's A rows (row fragment i: logical row 16 i + fr, clamped to the last live row -- its products land in accumulator rows
    // the epilogue never reads) and its slot in the weight tiles of this column block
    const bf16* a_row[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        int r = r0 + 16 * i + fr;
        r = r < Meff ? r : Meff - 1;
        a_row[i] = p.a_hi + 2 * ((p.a_index ? (long long)p.a_index[r] : p.amap(r)) * (long long)p.lda) + 8 * fk;
    }
    const bf16* w_lane = p.w + (long long)(n0 >> 4) * nk * 512 + fr * 32 + fk * 8;
    const long long wlo_delta = WPL == 2 ? p.w_lo - p.w : 0;
    const int mrows = Meff - r0 < FM * 16 ? Meff - r0 : FM * 16;      // live rows of this block
    const int nfrag = (mrows + 15) >> 4;      // live row fragments (uniform)
    // no branch in the loop: the fragments past the live rows re-read the clamped row (one cache line for the whole wave), so the compiler is
    // free to put every load of a K step -- and of the next one -- in front of the MFMAs
    // U K steps per trip (per % U == 0: launch_gemm_skinny), every load of the trip in front of its first MFMA: a trip costs one memory latency
    // whatever U is, and these launches are nothing but a chain of trips (U = 6: a K = 768 slice of a quarter is ONE trip)
    for (int ktu = kt0; ktu < kt0 + per; ktu += U) {
        bf16x8 b[U], bl[WPL == 2 ? U : 1], a0[U][FM], a1[U][FM];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int kt = ktu + u;
            b[u] = *reinterpret_cast<const bf16x8*>(w_lane + (long long)kt * 512);
            if constexpr (WPL == 2) bl[u] = *reinterpret_cast<const bf16x8*>(w_lane + wlo_delta + (long long)kt * 512);
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                a0[u][i] = *reinterpret_cast<const bf16x8*>(a_row[i] + kt * 64);
                a1[u][i] = *reinterpret_cast<const bf16x8*>(a_row[i] + kt * 64 + MMS_PLANE_LO);
            }
        }
        __builtin_amdgcn_sched_barrier(0);      // keep every load of the trip ahead of its MFMAs (the scheduler otherwise pairs them up again to save registers)
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[u][i], b[u], acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[u][i], b[u], acc[i], 0, 0, 0);
                if constexpr (WPL == 2) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[u][i], bl[u], acc[i], 0, 0, 0);
            }
        }
    }
    // accumulator fragment i of lane l: C[16 i + 4 (l >> 4) + e][n0 + (l & 15)]
#pragma unroll
    for (int i = 0; i < FM; ++i)
        if (i < nfrag) {
#pragma unroll
            for (int e = 0; e < 4; ++e) red[wave][16 * i + 4 * fk + e][fr] = acc[i][e];
        }
    __syncthreads();
    // epilogue: one thread per (row, 4 consecutive columns)
    for (int q = tid; q < mrows * 4; q += 64 * KS) {
        const int row = q >> 2, c4 = (q & 3) * 4, col = n0 + c4;
        f32x4 v = *reinterpret_cast<const f32x4*>(&red[0][row][c4]);
#pragma unroll
        for (int w = 1; w < KS; ++w) v += *reinterpret_cast<const f32x4*>(&red[w][row][c4]);
        if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + col);
        v = apply_act4<ACT>(v);
        const long long orow = p.cmap(r0 + row);
        if (p.out_kind == OUT_F32) {
            float* dst = p.hm_rows ? p.c_f32 + ((long long)((p.hm_col0 + col) >> 6) * p.hm_rows + orow) * 64 + ((p.hm_col0 + col) & 63)
                                   : p.c_f32 + (long long)blockIdx.z * p.c_split_stride + orow * p.ldc + col;
            *reinterpret_cast<f32x4*>(dst) = v;
        } else {
            bf16x4 h, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) { bf16 a, c2; split_bf16(v[e], a, c2); h[e] = a; l[e] = c2; }
            *reinterpret_cast<bf16x4*>(plane_ptr(p.c_hi, orow * p.ldp + col)) = h;
            *reinterpret_cast<bf16x4*>(plane_ptr(p.c_lo, orow * p.ldp + col)) = l;
        }
    }
}

template <int FM, int KS, int U, int WPL>
void launch_act(const GemmParams& p, int slices, hipStream_t st) {
    const dim3 grid(p.N / 16, (p.M + FM * 16 - 1) / (FM * 16), slices), block(64 * KS);
    switch (p.act) {
        case ACT_RELU: hipLaunchKernelGGL((gemm_skinny_kernel<FM, KS, ACT_RELU, U, WPL>), grid, block, 0, st, p); break;
        case ACT_GELU_TANH: hipLaunchKernelGGL((gemm_skinny_kernel<FM, KS, ACT_GELU_TANH, U, WPL>), grid, block, 0, st, p); break;
        case ACT_GELU_ERF: hipLaunchKernelGGL((gemm_skinny_kernel<FM, KS, ACT_GELU_ERF, U, WPL>), grid, block, 0, st, p); break;
        case ACT_TANH: hipLaunchKernelGGL((gemm_skinny_kernel<FM, KS, ACT_TANH, U, WPL>), grid, block, 0, st, p); break;
        default: hipLaunchKernelGGL((gemm_skinny_kernel<FM, KS, ACT_NONE, U, WPL>), grid, block, 0, st, p); break;
    }
}
template <int FM, int KS, int WPL>
void launch_fm(const GemmParams& p, int slices, hipStream_t st) {
    // K steps in flight per wave: what the register file holds (operand registers at WPL = 1: FM = 2: 120, FM = 4: 216, FM = 6: 156, FM = 8: 136; the lo
    // weight fragment of mode 3 adds 4 per step: FM = 4 runs three steps per trip there)
    constexpr int DEEP = FM <= 2 ? 6 : FM <= 4 ? (WPL == 2 ? 3 : 6) : FM <= 6 ? 3 : 2;
    const int per = (p.K >> 5) / (KS * slices);
    if (per % DEEP == 0) launch_act<FM, KS, DEEP, WPL>(p, slices, st);
    else launch_act<FM, KS, 2, WPL>(p, slices, st);
}

template <int KS, int WPL>
void launch_ks(const GemmParams& p, int slices, hipStream_t st) {
    if (p.M <= 32) launch_fm<2, KS, WPL>(p, slices, st);
    else if (p.M <= 64) launch_fm<4, KS, WPL>(p, slices, st);
    else if (p.M <= 96) launch_fm<6, KS, WPL>(p, slices, st);      // (zk's label-text projection of a 1-pair call: 8 positions x 10 labels = 80 rows, K = 6144)
    else launch_fm<8, KS, WPL>(p, slices, st);      // more than 128 rows: row blocks of 128
}
template <int WPL>
bool launch_wpl(const GemmParams& p, int ks, hipStream_t st) {
    if (ks == 1) launch_ks<1, WPL>(p, 1, st);
    else if (ks == 4) launch_ks<4, WPL>(p, 1, st);
    else if (ks == 8) launch_ks<8, WPL>(p, 1, st);
    else return false;
    return true;
}

}  // namespace

// false: not a launch for this kernel (more than 512 rows, another precision mode or output kind, residual in the epilogue) -- the caller takes
// the tile engine.  GemmParams::wave_k_slices names the number of K slices (= waves per workgroup: 1, 4 or 8; 0 -> 1): api.hip passes the split factor
// the tile-engine route of the SAME projection uses for launches of up to 255 rows (4 for the wide and the LayerNorm-followed K = 768 projections, 8 for the
// long-K ones, 1 = one wave walking all of K for the rest).  A slice is accumulated in the tile engine's order (per 32-wide K block: hi plane, then lo plane)
// and the slices are summed in the same fixed order as k_splitk_reduce / k_ln_to_planes sum their partials, so a launch here is BIT-IDENTICAL to the
// split-K tile route it replaces: the 128-row bound is not a numerical regime boundary.
bool launch_gemm_skinny(const GemmParams& p, int nsplit, hipStream_t st) {
    if (p.M <= 0) return true;
    const int ks = p.wave_k_slices > 1 ? p.wave_k_slices : 1;
    if ((nsplit != 2 && nsplit != 3) || p.M > SKINNY_MAX_ROWS || p.N % 16 || p.K % (64 * ks) || p.f8 || p.r_hi || p.ln_gamma) return false;
    if (nsplit == 3 && !p.w_lo) return false;
    if (p.out_kind != OUT_F32 && p.out_kind != OUT_PLANES) return false;
    return nsplit == 3 ? launch_wpl<2>(p, ks, st) : launch_wpl<1>(p, ks, st);
}

// The same kernel with the K slices dealt to WORKGROUPS instead of waves (GemmParams::k_splits of them, one wave each: N / 16 x k_splits workgroups): slice s
// leaves as an fp32 partial at c_f32 + s * c_split_stride -- the split-K contract of gemm_tile.hip (no bias / activation / residual; the LayerNorm kernel behind
// it sums the partials).  For the N = 768 projections: 48 column blocks are too few workgroups to stream a K = 3072 weight matrix (98 KB each: 9 us); 384 single-wave
// workgroups of 12 KB each take half.  Same slices, same order, same sum in k_ln_to_planes: bit-identical to both the in-workgroup form and the tile route.
bool launch_gemm_skinny_parts(const GemmParams& p, int nsplit, hipStream_t st) {
    if (p.M <= 0) return true;
    const int S = p.k_splits;
    if (S < 2 || (nsplit != 2 && nsplit != 3) || p.M > SKINNY_MAX_ROWS || p.N % 16 || p.K % (64 * S) || p.f8 || p.r_hi || p.ln_gamma) return false;
    if (nsplit == 3 && !p.w_lo) return false;
    if (p.out_kind != OUT_F32 || p.hm_rows || p.bias || p.act != ACT_NONE || p.cmap.grp) return false;
    if (nsplit == 3) launch_ks<1, 2>(p, S, st); else launch_ks<1, 1>(p, S, st);
    return true;
}
