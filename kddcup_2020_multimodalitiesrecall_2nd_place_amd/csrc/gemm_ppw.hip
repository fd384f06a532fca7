// Ping-pong phase GEMM for precision mode 3 (fp32-checkpoint-faithful): activations AND weights in split planes, three MFMA
// passes per product  a_hi*w_hi + a_lo*w_hi + a_hi*w_lo  (the a_lo*w_lo term is below 2^-17 of the product).
//
// The gemm_pp.hip engine re-cut for four operand planes: 256x128 tile (a 32-wide K stage = A [256][hi 64 B | lo 64 B] + W_hi, W_lo
// [128][64 B] = 48 KiB, so the 3-slot LDS-DMA ring still fits), 8 wavefronts as 2(M) x 4(N), 128x32 outputs per wave, ONE phase of
// 8 x 2 x 3 = 48 MFMAs per stage (round 3; before: two phases of 24, the schedule sketched below), wave rows staggered by one
// barrier, swapped MFMA operands + LDS-free epilogue (gemm_pp_epilogue.h), persistent workgroups.
//
//     phase 1:  ds_read W fragments (hi, lo) + A rows 0-63 (hi, lo)        | s_barrier | 24 MFMAs | s_barrier
//     phase 2:  ds_read A rows 64-127; issue the 6 LDS-DMA pieces of stage s+2; s_waitcnt vmcnt(6) | s_barrier | 24 MFMAs | s_barrier
//
// Hazards (as gemm_pp.hip): a slot is refilled two phases after its last ds_read (phase 2 of stage s-1 -> phase 2 of stage s);
// the counted wait precedes phase 2's first barrier and the data is first read in the next phase; the 6 newest operations at
// that wait are always LDS-DMA loads.
#include <type_traits>

#include "kernels.h"
#include "gemm_pp_epilogue.h"

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

namespace {

__device__ __forceinline__ int pw_swz(int r) { return (4 - ((r >> 2) & 3)) & 3; }

__device__ __forceinline__ void pw_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

}  // namespace

template <int ACT>
__global__ __launch_bounds__(512) void gemm_ppw_kernel(const GemmParams p) {
    constexpr int BM = 256, BN = 128, WAVES_N = 4, TM = 128, TN = 32, FM = 8, FN = 2;
    constexpr int APLANE = 256 * 64, WPLANE = 128 * 64;     // bytes of one plane of a stage
    constexpr int WOFF = 2 * APLANE;
    constexpr int SLOT = 2 * APLANE + 2 * WPLANE;            // 48 KiB
    constexpr int P = 6;                                     // LDS-DMA pieces per wave per stage: A_hi 2, A_lo 2, W_hi 1, W_lo 1
    __shared__ __attribute__((aligned(16))) unsigned char smem[3 * SLOT];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    int Meff = p.M;
    if (p.m_dev) { const int md = *p.m_dev; Meff = md < Meff ? md : Meff; }
    if (p.flop_counter && blockIdx.x == 0 && tid == 0)
        atomicAdd(p.flop_counter, 2ull * (unsigned long long)Meff * (unsigned long long)p.N * (unsigned long long)p.K);
    const int nbn = p.N / BN, nbm = (Meff + BM - 1) / BM, nblk = nbm * nbn;
    int vb = blockIdx.x;                 // virtual block id: the workgroup walks vb, vb + gridDim.x, ...
    if (vb >= nblk) return;
    const long long wlo = p.w_lo - p.w;

    // LDS-DMA sources.  A region of a slot: [256 rows][hi 64 B | lo 64 B]; piece q (0..3) of a wave = rows q*64 + wave*8 + lane/8, physical
    // 16-B chunk lane%8 <- logical chunk (lane%8) ^ ((row>>1)&7) of the row's stage line in the hl32 plane layout (common.h): eight whole
    // cache lines per piece.  W_hi / W_lo regions: [128][64 B] each, one piece per wave = one contiguous KiB of the tiled weights
    // (rows wave*16 + lane/4, physical chunk lane%4 <- logical chunk (lane%4) ^ pw_swz(row)).
    const int gr_l = lane >> 2, gc = lane & 3;
    const bf16* a_src[4];
    const bf16* w_src;
    int bm, bn;
    auto setup = [&](int v) {
        const int q = nblk >> 3, r8 = nblk & 7, xcd = v & 7, loc = v >> 3;
        int bid = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + loc;
        if (p.reverse) bid = nblk - 1 - bid;
        bm = bid / nbn; bn = bid % nbn;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const int r = q4 * 64 + wave * 8 + (lane >> 3);
            int gr = bm * BM + r;
            gr = gr < Meff ? gr : Meff - 1;
            a_src[q4] = p.a_hi + 2 * ((p.a_index ? (long long)p.a_index[gr] : p.amap(gr)) * (long long)p.lda) + ((lane & 7) ^ ((r >> 1) & 7)) * 8;
        }
        const int rw = wave * 16 + gr_l;
        w_src = p.w + wtile_off(bn * BN + rw, 0, p.K) + (gc ^ pw_swz(rw)) * 8;
    };
    setup(vb);
    auto issue = [&](int q, int st, int slot) {     // q: 0..3 = A row quarters (both planes), 4 = W_hi, 5 = W_lo
        unsigned char* d = smem + slot * SLOT + wave * 1024;
        const bf16* s;
        if (q < 4) { d += q * 8192; s = a_src[q] + st * 64; }
        else { d += WOFF + (q - 4) * WPLANE; s = w_src + (q == 5 ? wlo : 0) + st * 512; }
        __builtin_amdgcn_global_load_lds((glb_void*)s, (lds_void*)d, 16, 0, 0);
    };

    f32x4 acc[FM][FN];
    const int fr = lane & 15, fk = lane >> 4;
    const int laneA = (wm * TM + fr) * 128 + ((fk ^ ((fr >> 1) & 7)) << 4);      // hi fragment; the lo one: chunk ^ 4
    const int laneB = WOFF + (wn * TN + fr) * 64 + ((fk ^ pw_swz(fr)) << 4);
#ifndef MMS_PPW_PHASES
#define MMS_PPW_PHASES 1
#endif
    constexpr int AF = MMS_PPW_PHASES == 1 ? 8 : 4;
    bf16x8 a[2][AF], b[2][2];      // [plane][fragment]
    auto read_a = [&](const unsigned char* sb, int mh) {
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int i = 0; i < 4; ++i) a[pl][i] = *reinterpret_cast<const bf16x8*>(sb + (laneA ^ (pl << 6)) + (mh * 64 + i * 16) * 128);
    };
    auto read_b = [&](const unsigned char* sb) {
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int j = 0; j < 2; ++j) b[pl][j] = *reinterpret_cast<const bf16x8*>(sb + laneB + pl * WPLANE + j * 16 * 64);
    };
    // 24 MFMAs: a_hi*w_hi, a_lo*w_hi, a_hi*w_lo per fragment pair, one pass after the other (dependent MFMAs 8 apart);
    // swapped operands -> transposed result fragments (gemm_pp_epilogue.h)
    auto mma = [&](int mh) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int pass = 0; pass < 3; ++pass)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[mh * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[pass == 2 ? 1 : 0][j], a[pass == 1 ? 1 : 0][i], acc[mh * 4 + i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };

    const int ns = p.K / 32;   // >= 2
    // WAITN: outstanding pieces allowed at the phase-2 wait (-1: none in flight)
    auto stage = [&](auto pre_tag, auto wait_tag, int s, int slot) {
        constexpr bool PRE = decltype(pre_tag)::value;
        constexpr int WAITN = decltype(wait_tag)::value;
        const unsigned char* sb = smem + slot * SLOT;
        const int nslot = slot == 0 ? 2 : slot - 1;
#if MMS_PPW_PHASES == 1
        // ONE phase of 48 MFMAs per stage (round 3; rounds 1-2: two of 24): the wave tile is only 128x32, so all 16 A fragments and
        // 4 W fragments of a stage fit in registers (64 accumulator + 80 fragment VGPRs), and every barrier hand-over between the wave
        // rows costs ~90 idle matrix-pipe cycles (profiles/r03h_pp_phase_stamps.txt).  The reads are retired before the first barrier
        // (the slot is refilled from the next phase on); the counted wait for stage s+1 sits before it too (first read: next phase).
        read_b(sb);
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int i = 0; i < 8; ++i) a[pl][i] = *reinterpret_cast<const bf16x8*>(sb + (laneA ^ (pl << 6)) + i * 16 * 128);
        if (PRE) {
#pragma unroll
            for (int q = 0; q < P; ++q) issue(q, s + 2, nslot);
        }
        if (WAITN == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (WAITN == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        pw_barrier();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int pass = 0; pass < 3; ++pass)
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[pass == 2 ? 1 : 0][j], a[pass == 1 ? 1 : 0][i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        pw_barrier();
        return;
#endif
        read_b(sb);
        read_a(sb, 0);
        pw_barrier();
        mma(0);
        pw_barrier();
        read_a(sb, 1);
        if (PRE) {
#pragma unroll
            for (int q = 0; q < P; ++q) issue(q, s + 2, nslot);
        }
        if (WAITN == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (WAITN == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        pw_barrier();
        mma(1);
        pw_barrier();
    };

    // prologue: stages 0 and 1 in flight, stage 0 landed
#pragma unroll
    for (int q = 0; q < P; ++q) issue(q, 0, 0);
#pragma unroll
    for (int q = 0; q < P; ++q) issue(q, 1, 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    pw_barrier();
    if (wm == 1) pw_barrier();     // stagger the second wave row by one barrier

    for (;;) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        int slot = 0, s = 0;
        for (; s + 2 < ns; ++s) {
            stage(std::true_type{}, std::integral_constant<int, 6>{}, s, slot);
            slot = slot == 2 ? 0 : slot + 1;
        }
        stage(std::false_type{}, std::integral_constant<int, 0>{}, s, slot);
        slot = slot == 2 ? 0 : slot + 1;
        stage(std::false_type{}, std::integral_constant<int, -1>{}, s + 1, slot);
        if (wm == 0) pw_barrier();     // re-align the wave rows: nobody reads the ring any more

        // next tile: its first two stages go out before this tile's stores, one full drain covers both (gemm_pp.hip)
        const int row0 = bm * BM + wm * TM, col0 = bn * BN + wn * TN;
        const bool more = vb + (int)gridDim.x < nblk;
        if (more) {
            vb += gridDim.x;
            setup(vb);
#pragma unroll
            for (int q = 0; q < P; ++q) issue(q, 0, 0);
#pragma unroll
            for (int q = 0; q < P; ++q) issue(q, 1, 1);
        }
        pp_epilogue<ACT, FM, FN>(p, acc, row0, col0, lane, Meff);
        if (!more) break;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        pw_barrier();
        if (wm == 1) pw_barrier();
    }
}

bool launch_gemm_ppw(const GemmParams& p, hipStream_t st) {
    if (p.M <= 0) return true;
    if (p.N % 128 || p.K % 64 || !p.w_lo) return false;
    const int n_cu = device_cu_count();
    const int nblk = ((p.M + 255) / 256) * (p.N / 128);
    const dim3 grid(nblk > n_cu ? n_cu : nblk), block(512);
    switch (p.act) {
        case ACT_RELU: hipLaunchKernelGGL((gemm_ppw_kernel<ACT_RELU>), grid, block, 0, st, p); break;
        case ACT_GELU_TANH: hipLaunchKernelGGL((gemm_ppw_kernel<ACT_GELU_TANH>), grid, block, 0, st, p); break;
        case ACT_GELU_ERF: hipLaunchKernelGGL((gemm_ppw_kernel<ACT_GELU_ERF>), grid, block, 0, st, p); break;
        case ACT_TANH: hipLaunchKernelGGL((gemm_ppw_kernel<ACT_TANH>), grid, block, 0, st, p); break;
        default: hipLaunchKernelGGL((gemm_ppw_kernel<ACT_NONE>), grid, block, 0, st, p); break;
    }
    return true;
}
