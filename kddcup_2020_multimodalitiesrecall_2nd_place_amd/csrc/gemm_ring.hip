// LDS-DMA ring GEMM (gfx950): 128x256 tile, 8 wavefronts, K consumed in 32-wide half-tiles through a 4-slot LDS
// ring filled by `global_load_lds_dwordx4`, waits are COUNTED (`s_waitcnt vmcnt(N)`, never 0 in steady state) so up to
// three half-tiles stay in flight across the per-half-tile barrier (the T3+T4 structure of the CDNA programming
// guide, applied to the split-plane A operand).  Same contract / epilogue as gemm_tile.hip.
//
// Ring slot = A_hi [128][64 B] (+ A_lo) + W [256][64 B]; 16-B chunk c of row r sits at chunk c ^ f(r),
// f = {0,3,2,1}[(r>>2)&3]: conflict-free for ds_read_b128 fragment reads on 64-byte rows.  LDS-DMA writes are
// lane-linear (one instruction = 16 rows x 4 chunks), so the swizzle is applied to the per-lane SOURCE address.
//
// Per half-tile h:   s_waitcnt vmcnt(pieces of half-tiles issued after h)   -> my pieces of h have landed
//                    s_barrier                                              -> everybody's have; slot (h-1)&3 is free
//                    issue half-tile h+3 into slot (h+3)&3
//                    12 ds_read_b128 + 32 MFMA from slot h&3
#include "kernels.h"
#include "gemm_epilogue.h"

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

__device__ __forceinline__ int swz32(int r) { return (4 - ((r >> 2) & 3)) & 3; }
__device__ __forceinline__ int lds_off32(int r, int c) { return r * 64 + ((c ^ swz32(r)) << 4); }

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else static_assert(N == 0, "unsupported count");
}

// DIAG (timing diagnostics only, results are WRONG): bit 0 drops the per-half-tile wait+barrier, bit 1 drops the LDS-DMA issue
template <int NSPLIT, int ACT, int NSLOT, int DIAG>
__global__ __launch_bounds__(512) void gemm_ring_kernel(const GemmParams p) {
    constexpr int BM = 128, BN = 256, WAVES_N = 4, NW = 8, TM = 64, TN = 64, FM = 4, FN = 4;
    constexpr int A_BYTES = BM * 64, B_BYTES = BN * 64, SLOT = NSPLIT * A_BYTES + B_BYTES;
    static_assert(NSLOT == 2 || NSLOT == 4, "ring of 2 or 4 half-tile slots");
    constexpr int P = NSPLIT + 2;                       // LDS-DMA pieces per wave per half-tile
    constexpr int EPI_BYTES = NW * 16 * (TN + 4) * 4;
    constexpr int SMEM_BYTES = NSLOT * SLOT > EPI_BYTES ? NSLOT * SLOT : EPI_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    int Meff = p.M;
    if (p.m_dev) { const int md = *p.m_dev; Meff = md < Meff ? md : Meff; }
    if (p.flop_counter && blockIdx.x == 0 && tid == 0)
        atomicAdd(p.flop_counter, 2ull * (unsigned long long)Meff * (unsigned long long)p.N * (unsigned long long)p.K);
    const int nbn = p.N / BN, nbm = (Meff + BM - 1) / BM, nblk = nbm * nbn;
    int bid = blockIdx.x;
    if (bid >= nblk) return;
    {
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, loc = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int bm = bid / nbn, bn = bid % nbn;
    const long long lo_delta = p.a_lo - p.a_hi;

    // LDS-DMA sources: A row group `wave` (16 rows), W row groups `wave` and `wave + 8`
    const int gr_l = lane >> 2, gc = lane & 3;
    const bf16* a_src;
    const bf16* w_src[2];
    {
        const int r = wave * 16 + gr_l;
        int gr = bm * BM + r;
        gr = gr < Meff ? gr : Meff - 1;
        a_src = p.a_hi + (p.a_index ? (long long)p.a_index[gr] : p.amap(gr)) * (long long)p.lda + (gc ^ swz32(r)) * 8;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int rw = (wave + 8 * s) * 16 + gr_l;
            w_src[s] = p.w + (long long)(bn * BN + rw) * p.K + (gc ^ swz32(rw)) * 8;
        }
    }
    auto issue = [&](int h) {
        unsigned char* sb = smem + (h & (NSLOT - 1)) * SLOT;
        const int ko = h * 32;
        __builtin_amdgcn_global_load_lds((glb_void*)(a_src + ko), (lds_void*)(sb + wave * 1024), 16, 0, 0);
        if (NSPLIT == 2)
            __builtin_amdgcn_global_load_lds((glb_void*)(a_src + lo_delta + ko), (lds_void*)(sb + A_BYTES + wave * 1024), 16, 0, 0);
#pragma unroll
        for (int s = 0; s < 2; ++s)
            __builtin_amdgcn_global_load_lds((glb_void*)(w_src[s] + ko), (lds_void*)(sb + NSPLIT * A_BYTES + (wave + 8 * s) * 1024), 16, 0, 0);
    };

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fk = lane >> 4;

    const int nh = p.K / 32;     // >= 2 (K % 64 == 0)
    constexpr int D = NSLOT - 2;   // half-tiles allowed to stay in flight behind the one being waited for
    issue(0);
    if (NSLOT == 4) { issue(1); if (nh > 2) issue(2); }
    for (int h = 0; h < nh; ++h) {
        const int ahead = (nh - 1 - h) < D ? (nh - 1 - h) : D;
        if (!(DIAG & 1)) {
            if (ahead == 2) wait_vmcnt<2 * P>();
            else if (ahead == 1) wait_vmcnt<P>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
        }
        asm volatile("" ::: "memory");
        if (!(DIAG & 2) && h + NSLOT - 1 < nh) issue(h + NSLOT - 1);
        const unsigned char* sb = smem + (h & (NSLOT - 1)) * SLOT;
        bf16x8 a0[FM], a1[FM], b[FN];
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int o = lds_off32(wm * TM + i * 16 + fr, fk);
            a0[i] = *reinterpret_cast<const bf16x8*>(sb + o);
            if (NSPLIT == 2) a1[i] = *reinterpret_cast<const bf16x8*>(sb + A_BYTES + o);
        }
#pragma unroll
        for (int j = 0; j < FN; ++j)
            b[j] = *reinterpret_cast<const bf16x8*>(sb + NSPLIT * A_BYTES + lds_off32(wn * TN + j * 16 + fr, fk));
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[i], b[j], acc[i][j], 0, 0, 0);
                if (NSPLIT == 2) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[i], b[j], acc[i][j], 0, 0, 0);
            }
    }
    gemm_epilogue<ACT, BM, BN, TM, TN, FM, FN>(p, acc, smem, bm, bn, wm, wn, wave, lane, Meff);
}

template <int NSPLIT, int NSLOT, int DIAG>
static void launch_ring_ns(const GemmParams& p, hipStream_t st) {
    const int nblk = ((p.M + 127) / 128) * (p.N / 256);
    const dim3 grid(nblk), block(512);
    switch (p.act) {
        case ACT_RELU: hipLaunchKernelGGL((gemm_ring_kernel<NSPLIT, ACT_RELU, NSLOT, DIAG>), grid, block, 0, st, p); break;
        case ACT_GELU_TANH: hipLaunchKernelGGL((gemm_ring_kernel<NSPLIT, ACT_GELU_TANH, NSLOT, DIAG>), grid, block, 0, st, p); break;
        case ACT_GELU_ERF: hipLaunchKernelGGL((gemm_ring_kernel<NSPLIT, ACT_GELU_ERF, NSLOT, DIAG>), grid, block, 0, st, p); break;
        case ACT_TANH: hipLaunchKernelGGL((gemm_ring_kernel<NSPLIT, ACT_TANH, NSLOT, DIAG>), grid, block, 0, st, p); break;
        default: hipLaunchKernelGGL((gemm_ring_kernel<NSPLIT, ACT_NONE, NSLOT, DIAG>), grid, block, 0, st, p); break;
    }
}

bool launch_gemm_ring(const GemmParams& p, int nsplit, int nslot, hipStream_t st) {
    if (p.M <= 0) return true;
    if (p.N % 256 || nsplit > 2) return false;
    if (nslot == 2) { if (nsplit == 2) launch_ring_ns<2, 2, 0>(p, st); else launch_ring_ns<1, 2, 0>(p, st); }
    else if (nslot >= 100) {   // diagnostics: 101 no barrier, 102 no DMA, 103 neither (2-slot geometry, nsplit 2 only)
        if (nsplit != 2) return false;
        if (nslot == 101) launch_ring_ns<2, 2, 1>(p, st); else if (nslot == 102) launch_ring_ns<2, 2, 2>(p, st); else launch_ring_ns<2, 2, 3>(p, st);
    }
    else { if (nsplit == 2) launch_ring_ns<2, 4, 0>(p, st); else launch_ring_ns<1, 4, 0>(p, st); }
    return true;
}
