// Two-workgroup GEMM engine (gfx950): 128x256 tile, FOUR wavefronts as 2(M) x 2(N), 64x128 outputs per wave, TWO workgroups per CU
// (64 KiB of LDS and 256 registers each, one wave of each workgroup per SIMD).
//
// Why: with one 8-wave workgroup per CU (gemm_pp.hip) nothing issues MFMAs while that workgroup runs its epilogue -- for the FFN-up
// projection (GELU + plane split + 32 stores per lane and tile) that is ~40 % of a tile -- and every point of matrix-pipe duty is worth
// about half a point of throughput under the package power limit (LABBOOK R3.12).  Two independent workgroups drift apart by themselves:
// one's epilogue, fragment reads and barrier waits run under the other's MFMAs.  Price: each workgroup streams its own W stages (A 32 KiB
// + W 32 KiB of LDS-DMA per 2 x 64 MFMAs per SIMD instead of 32 + 16), and a 2-slot ring with one barrier per stage.
//
// Same contract / data layout / epilogue as gemm_pp.hip (GemmParams, hl32 planes, tiled weights, swizzles on the LDS-DMA source address,
// swapped MFMA operands + LDS-free epilogue of gemm_pp_epilogue.h).  Two-pass (hi + lo activation planes) only.
//
// Stage s of a tile (32 k; A [128][hi 64 B | lo 64 B] + W [256][64 B] = 32 KiB in slot s % 2):
//     s_waitcnt vmcnt(0)            my LDS-DMA pieces of stage s have landed
//     s_barrier                     ... everybody's have, and everybody is past its reads of stage s-1 (they fed the MFMAs before this point)
//     issue the 8 pieces of stage s+1 into the other slot
//     16 x ds_read_b128 ; s_waitcnt lgkmcnt(0)
//     64 x v_mfma_f32_16x16x32_bf16 (s_setprio 1)
// The next tile's stage 0 goes out before the epilogue (slot 0 was last read two barriers ago).
#include "kernels.h"
#include "gemm_pp_epilogue.h"

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

namespace {
__device__ __forceinline__ int dw_swz(int r) { return (4 - ((r >> 2) & 3)) & 3; }
__device__ __forceinline__ void dw_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
}  // namespace

template <int ACT>
__global__ __launch_bounds__(256, 2) void gemm_dw_kernel(const GemmParams p) {
    constexpr int BM = 128, BN = 256, WAVES_N = 2, TM = 64, TN = 128, FM = TM / 16, FN = TN / 16;
    constexpr int AREG = BM * 128, SLOT = AREG + BN * 64, NSLOT = 2;      // 16 KiB + 16 KiB per stage, 8 LDS-DMA pieces of 1 KiB per wave
    __shared__ __attribute__((aligned(16))) unsigned char smem[NSLOT * SLOT];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    int Meff = p.M;
    if (p.m_dev) { const int md = *p.m_dev; Meff = md < Meff ? md : Meff; }
    if (p.flop_counter && blockIdx.x == 0 && tid == 0)
        atomicAdd(p.flop_counter, 2ull * (unsigned long long)Meff * (unsigned long long)p.N * (unsigned long long)p.K);
    const int nbn = p.N / BN, nbm = (Meff + BM - 1) / BM, nblk = nbm * nbn;
    int vb = blockIdx.x;
    if (vb >= nblk) return;

    const int gr_l = lane >> 2, gc = lane & 3;
    const bf16* a_src[4];
    const bf16* w_src[4];
    int bm, bn;
    auto setup = [&](int v) {
        // bijective XCD remap over the live tiles (virtual block v runs on XCD v % 8): the column tiles of a row panel share one L2
        const int q = nblk >> 3, r8 = nblk & 7, xcd = v & 7, loc = v >> 3;
        int bid = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + loc;
        if (p.reverse) bid = nblk - 1 - bid;
        bm = bid / nbn; bn = bid % nbn;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const int r = q4 * 32 + wave * 8 + (lane >> 3);
            int gr = bm * BM + r;
            gr = gr < Meff ? gr : Meff - 1;
            const long long lrow = (p.a_index ? (long long)p.a_index[gr] : p.amap(gr)) * (long long)p.lda;
            a_src[q4] = p.a_hi + 2 * lrow + ((lane & 7) ^ ((r >> 1) & 7)) * 8;
        }
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int r = h * 64 + wave * 16 + gr_l;
            w_src[h] = p.w + wtile_off(bn * BN + r, 0, p.K) + (gc ^ dw_swz(r)) * 8;
        }
    };
    auto issue = [&](int st, int slot) {
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4)
            __builtin_amdgcn_global_load_lds((glb_void*)(a_src[q4] + st * 64), (lds_void*)(smem + slot * SLOT + q4 * 4096 + wave * 1024), 16, 0, 0);
#pragma unroll
        for (int h = 0; h < 4; ++h)
            __builtin_amdgcn_global_load_lds((glb_void*)(w_src[h] + st * 512), (lds_void*)(smem + slot * SLOT + AREG + h * 4096 + wave * 1024), 16, 0, 0);
    };
    setup(vb);
    issue(0, 0);

    const int ns = p.K / 32;
    const int fr = lane & 15, fk = lane >> 4;
    const int laneA = (wm * TM + fr) * 128 + ((fk ^ ((fr >> 1) & 7)) << 4);          // hi fragment; lo: chunk ^ 4
    const int laneB = AREG + (wn * TN + fr) * 64 + ((fk ^ dw_swz(fr)) << 4);
    f32x4 acc[FM][FN];
    for (;;) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        int slot = 0;
        for (int s = 0; s < ns; ++s) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            dw_barrier();
            if (s + 1 < ns) issue(s + 1, slot ^ 1);
            const unsigned char* sb = smem + slot * SLOT;
            bf16x8 a[2][FM], b[FN];
#pragma unroll
            for (int j = 0; j < FN; ++j) b[j] = *reinterpret_cast<const bf16x8*>(sb + laneB + j * 16 * 64);
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                for (int i = 0; i < FM; ++i) a[pl][i] = *reinterpret_cast<const bf16x8*>(sb + (laneA ^ (pl << 6)) + i * 16 * 128);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[pl][i], acc[i][j], 0, 0, 0);   // swapped operands: C^T fragment
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            slot ^= 1;
        }
        const int row0 = bm * BM + wm * TM, col0 = bn * BN + wn * TN;
        vb += (int)gridDim.x;
        const bool more = vb < nblk;
        if (more) {
            setup(vb);
            // slot 0's last reads (stage ns - 2, or the only stage's) lie behind a barrier every wave has passed -- unless the tile had a single
            // stage and an odd stage count leaves the LAST stage in slot 0: K % 64 == 0 keeps ns even
            issue(0, 0);
        }
        pp_epilogue<ACT, FM, FN>(p, acc, row0, col0, lane, Meff);
        if (!more) break;
    }
}

static int dw_cu_count() { return device_cu_count(); }

// false: shape not supported (N % 256, K % 64, not two-pass)
bool launch_gemm_dw(const GemmParams& p, int nsplit, hipStream_t st) {
    if (p.M <= 0) return true;
    if (nsplit != 2 || p.N % 256 || p.K % 64 || p.w_lo || p.a8) return false;
    const long long nblk = (long long)((p.M + 127) / 128) * (p.N / 256);
    const long long full = 2ll * dw_cu_count();
    const dim3 grid((unsigned)(nblk < full ? nblk : full)), block(256);
    switch (p.act) {
        case ACT_RELU: hipLaunchKernelGGL((gemm_dw_kernel<ACT_RELU>), grid, block, 0, st, p); break;
        case ACT_GELU_TANH: hipLaunchKernelGGL((gemm_dw_kernel<ACT_GELU_TANH>), grid, block, 0, st, p); break;
        case ACT_GELU_ERF: hipLaunchKernelGGL((gemm_dw_kernel<ACT_GELU_ERF>), grid, block, 0, st, p); break;
        case ACT_TANH: hipLaunchKernelGGL((gemm_dw_kernel<ACT_TANH>), grid, block, 0, st, p); break;
        default: hipLaunchKernelGGL((gemm_dw_kernel<ACT_NONE>), grid, block, 0, st, p); break;
    }
    return true;
}
