// Generalised bf16 MFMA GEMM tile kernel (gfx950): configurable workgroup tile / wave grid, LDS-DMA
// (global_load_lds) double-buffered operand staging, LDS-staged vectorised epilogue.
//
// Same contract as gemm.hip (GemmParams): C = act(A W^T + bias (+ residual)), A in split planes.
//
//  * STAGE_GLDS = 1: operand tiles go HBM/L2 -> LDS directly with `global_load_lds_dwordx4` (no VGPR
//    round trip, no ds_write).  The LDS image is lane-linear per wave instruction (1 KiB = 8 rows of
//    128 B), so the bank-conflict swizzle is applied on the per-lane SOURCE address: LDS slot cpos of
//    row r is filled from global chunk cpos ^ ((r>>1)&7); fragment reads use the same involution.
//    Two stage buffers: tile k+1 streams in while tile k feeds the MFMAs; one barrier per K-step.
//  * STAGE_GLDS = 0: global -> VGPR -> LDS, single buffer (the round-1a structure, kept for A/B).
//  * Epilogue: each wave bounces its accumulators through a private LDS strip 16 rows at a time and
//    re-reads them row-contiguous, so bias / residual / activation / split run on float4 and global
//    stores are 16 B (fp32) or 8 B (bf16x4 per plane) per lane, 256 / 128 contiguous bytes per row.
#include "kernels.h"
#include "gemm_epilogue.h"

#define BK 64

__device__ __forceinline__ int lds_off(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

template <int NSPLIT, int ACT, int BM, int BN, int WAVES_M, int WAVES_N, int STAGE_GLDS, int OPT>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) void gemm_tile_kernel(const GemmParams p) {
    constexpr int NW = WAVES_M * WAVES_N, NT = NW * 64;
    constexpr int TM = BM / WAVES_M, TN = BN / WAVES_N, FM = TM / 16, FN = TN / 16;
    constexpr int NA = NSPLIT >= 2 ? 2 : 1, NB = NSPLIT == 3 ? 2 : 1;   // operand planes staged per K-tile
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
    constexpr int STAGE_BYTES = NA * A_BYTES + NB * B_BYTES;
    static_assert(!(STAGE_GLDS && NSPLIT == 3), "LDS-DMA path stages one weight plane only");
    constexpr int NBUF = STAGE_GLDS ? 2 : 1;
    constexpr int EPI_BYTES = NW * 16 * (TN + 4) * 4;
    constexpr int SMEM_BYTES = NBUF * STAGE_BYTES > EPI_BYTES ? NBUF * STAGE_BYTES : EPI_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    int Meff = p.M;
    if (p.m_dev) { const int md = *p.m_dev; Meff = md < Meff ? md : Meff; }
    if (p.flop_counter && blockIdx.x == 0 && tid == 0)
        atomicAdd(p.flop_counter, 2ull * (unsigned long long)Meff * (unsigned long long)p.N * (unsigned long long)p.K);
    // the grid is sized for the padded row bound; only the first nblk workgroups own live row panels
    const int nbn = p.N / BN, nbm = (Meff + BM - 1) / BM, nblk = nbm * nbn;
    int bid = blockIdx.x;
    // split-K: copy ksp of the tile grid contracts K columns [k0, k0 + klen) into its own fp32 partial (GemmParams::k_splits)
    const int S = p.k_splits > 1 ? p.k_splits : 1;
    if (bid >= nblk * S) return;   // block-uniform exit (packed mode: fewer live rows than the bound)
    const int ksp = bid / nblk;
    bid -= ksp * nblk;
    const int klen = p.K / S, k0 = ksp * klen;
    {   // bijective XCD remap over the LIVE workgroups (block b runs on XCD b % 8; speed only)
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, loc = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int bm = bid / nbn, bn = bid % nbn;
    const long long lo_delta = p.a_lo - p.a_hi;
    const long long wlo_delta = p.w_lo - p.w;

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = klen / BK;
    const int fr = lane & 15, fk = lane >> 4;

    // `hook(g)` runs after MFMA group g = ks * FM + i (used to interleave the next tile's LDS-DMA issue)
    auto compute = [&](const unsigned char* sb, auto&& hook) {
        const unsigned char* sA0 = sb;
        const unsigned char* sA1 = sb + A_BYTES;
        const unsigned char* sB = sb + NA * A_BYTES;
        const unsigned char* sB1 = sB + B_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 a0[FM], a1[FM], b[FN], b1[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int o = lds_off(wm * TM + i * 16 + fr, ks * 4 + fk);
                a0[i] = *reinterpret_cast<const bf16x8*>(sA0 + o);
                if (NA == 2) a1[i] = *reinterpret_cast<const bf16x8*>(sA1 + o);
            }
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int o = lds_off(wn * TN + j * 16 + fr, ks * 4 + fk);
                b[j] = *reinterpret_cast<const bf16x8*>(sB + o);
                if (NB == 2) b1[j] = *reinterpret_cast<const bf16x8*>(sB1 + o);
            }
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                if (OPT & 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[i], b[j], acc[i][j], 0, 0, 0);
                    if (NA == 2)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[i], b[j], acc[i][j], 0, 0, 0);
                    if (NB == 2)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[i], b1[j], acc[i][j], 0, 0, 0);
                }
                if (OPT & 2) __builtin_amdgcn_s_setprio(0);
                hook(ks * FM + i);
            }
        }
    };
    auto no_hook = [](int) {};

    if constexpr (STAGE_GLDS) {
        // one wave instruction fills one 1-KiB row group (8 rows x 128 B): lane -> (row g*8 + lane/8, slot lane%8)
        constexpr int GA = BM / 8 / NW, GB = BN / 8 / NW;  // row groups per wave
        static_assert(GA >= 1 && GB >= 1 && (BM / 8) % NW == 0 && (BN / 8) % NW == 0, "tile/wave mismatch");
        const bf16* a_src[GA];
        const bf16* w_src[GB];
#pragma unroll
        for (int s = 0; s < GA; ++s) {
            const int r = (wave + NW * s) * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            int gr = bm * BM + r;
            gr = gr < Meff ? gr : Meff - 1;
            a_src[s] = p.a_hi + 2 * ((p.a_index ? (long long)p.a_index[gr] : p.amap(gr)) * (long long)p.lda) + plane_off(c * 8) + 2 * k0;   // hl32 planes
        }
#pragma unroll
        for (int s = 0; s < GB; ++s) {
            const int r = (wave + NW * s) * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            w_src[s] = p.w + wtile_off(bn * BN + r, c * 8, p.K) + 16 * k0;                                                             // tiled weights
        }
        // piece q of a stage: q < GA*NSPLIT -> A planes, else B; one global_load_lds (1 KiB) per piece per wave
        constexpr int NPIECE = GA * NA + GB;
        auto issue_piece = [&](int q, int kt, unsigned char* sb) {
            static_assert(BK == 64, "K tile = two 32-column blocks");
            if (q < GA * NA) {
                const int s = q / NA, pl = q % NA;
                unsigned char* d = sb + pl * A_BYTES + (wave + NW * s) * 1024;
                __builtin_amdgcn_global_load_lds((glb_void*)(a_src[s] + (pl ? lo_delta : 0) + kt * (2 * BK)), (lds_void*)d, 16, 0, 0);
            } else {
                const int s = q - GA * NA;
                unsigned char* d = sb + NA * A_BYTES + (wave + NW * s) * 1024;
                __builtin_amdgcn_global_load_lds((glb_void*)(w_src[s] + kt * (16 * BK)), (lds_void*)d, 16, 0, 0);
            }
        };
#pragma unroll
        for (int q = 0; q < NPIECE; ++q) issue_piece(q, 0, smem);
        for (int kt = 0; kt < nk; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of tile kt have landed
            __syncthreads();                                    // ... everyone's have; buffer (kt+1)&1 is free
            unsigned char* nb = smem + ((kt + 1) & 1) * STAGE_BYTES;
            const bool more = kt + 1 < nk;
            if (OPT & 1) {
                // spread the next tile's LDS-DMA issue over the MFMA groups (issue cost hides under MFMA execution)
                constexpr int NG = 2 * FM;
                compute(smem + (kt & 1) * STAGE_BYTES, [&](int g) {
                    if (more) {
#pragma unroll
                        for (int q = 0; q < NPIECE; ++q)
                            if (q * NG / NPIECE == g) issue_piece(q, kt + 1, nb);
                    }
                });
            } else {
                if (more) {
#pragma unroll
                    for (int q = 0; q < NPIECE; ++q) issue_piece(q, kt + 1, nb);
                }
                compute(smem + (kt & 1) * STAGE_BYTES, no_hook);
            }
        }
    } else {
        constexpr int CA = BM * 8 / NT, CB = BN * 8 / NT;  // 16-B chunks per thread per tile
        static_assert(CA >= 1 && CB >= 1, "tile/thread mismatch");
        const int c = tid & 7, lr = tid >> 3;
        const bf16* a_row[CA];
        const bf16* w_row[CB];
#pragma unroll
        for (int s = 0; s < CA; ++s) {
            int r = bm * BM + lr + (NT / 8) * s;
            r = r < Meff ? r : Meff - 1;
            // hl32 planes (common.h): a row's K tile of 64 columns = [hi 32 | lo 32 | hi 32 | lo 32]; chunk c (8 columns) of the hi plane
            a_row[s] = p.a_hi + 2 * ((p.a_index ? (long long)p.a_index[r] : p.amap(r)) * (long long)p.lda) + plane_off(c * 8) + 2 * k0;
        }
        static_assert(BK == 64, "K tile = two 32-column blocks");
#pragma unroll
        for (int s = 0; s < CB; ++s) w_row[s] = p.w + wtile_off(bn * BN + lr + (NT / 8) * s, c * 8, p.K) + 16 * k0;      // tiled weights: a K tile = 2 tiles of 512
        u32x4 ra0[CA], ra1[CA], rb[CB], rb1[CB];
#pragma unroll
        for (int s = 0; s < CA; ++s) {
            ra0[s] = *reinterpret_cast<const u32x4*>(a_row[s]);
            if (NA == 2) ra1[s] = *reinterpret_cast<const u32x4*>(a_row[s] + lo_delta);
        }
#pragma unroll
        for (int s = 0; s < CB; ++s) {
            rb[s] = *reinterpret_cast<const u32x4*>(w_row[s]);
            if (NB == 2) rb1[s] = *reinterpret_cast<const u32x4*>(w_row[s] + wlo_delta);
        }
        for (int kt = 0; kt < nk; ++kt) {
            __syncthreads();
#pragma unroll
            for (int s = 0; s < CA; ++s) {
                const int o = lds_off(lr + (NT / 8) * s, c);
                *reinterpret_cast<u32x4*>(smem + o) = ra0[s];
                if (NA == 2) *reinterpret_cast<u32x4*>(smem + A_BYTES + o) = ra1[s];
            }
#pragma unroll
            for (int s = 0; s < CB; ++s) {
                const int o = NA * A_BYTES + lds_off(lr + (NT / 8) * s, c);
                *reinterpret_cast<u32x4*>(smem + o) = rb[s];
                if (NB == 2) *reinterpret_cast<u32x4*>(smem + B_BYTES + o) = rb1[s];
            }
            __syncthreads();
            const int kn = kt + 1 < nk ? kt + 1 : kt, koA = kn * (2 * BK), koW = kn * (16 * BK);
#pragma unroll
            for (int s = 0; s < CA; ++s) {
                ra0[s] = *reinterpret_cast<const u32x4*>(a_row[s] + koA);
                if (NA == 2) ra1[s] = *reinterpret_cast<const u32x4*>(a_row[s] + lo_delta + koA);
            }
#pragma unroll
            for (int s = 0; s < CB; ++s) {
                rb[s] = *reinterpret_cast<const u32x4*>(w_row[s] + koW);
                if (NB == 2) rb1[s] = *reinterpret_cast<const u32x4*>(w_row[s] + wlo_delta + koW);
            }
            compute(smem, no_hook);
        }
    }

    if (S > 1) {      // partial sums: plain fp32 rows of this split's own buffer (host: no bias / residual / activation on a split launch)
        GemmParams q = p;
        q.c_f32 = p.c_f32 + (long long)ksp * p.c_split_stride;
        gemm_epilogue<ACT, BM, BN, TM, TN, FM, FN>(q, acc, smem, bm, bn, wm, wn, wave, lane, Meff);
        return;
    }
    gemm_epilogue<ACT, BM, BN, TM, TN, FM, FN>(p, acc, smem, bm, bn, wm, wn, wave, lane, Meff);
}

template <int NSPLIT, int BM, int BN, int WM, int WN, int G, int OPT>
static void launch_cfg(const GemmParams& p, hipStream_t st) {
    const int nblk = ((p.M + BM - 1) / BM) * (p.N / BN) * (p.k_splits > 1 ? p.k_splits : 1);
    const dim3 grid(nblk), block(WM * WN * 64);
    switch (p.act) {
        case ACT_RELU: hipLaunchKernelGGL((gemm_tile_kernel<NSPLIT, ACT_RELU, BM, BN, WM, WN, G, OPT>), grid, block, 0, st, p); break;
        case ACT_GELU_TANH: hipLaunchKernelGGL((gemm_tile_kernel<NSPLIT, ACT_GELU_TANH, BM, BN, WM, WN, G, OPT>), grid, block, 0, st, p); break;
        case ACT_GELU_ERF: hipLaunchKernelGGL((gemm_tile_kernel<NSPLIT, ACT_GELU_ERF, BM, BN, WM, WN, G, OPT>), grid, block, 0, st, p); break;
        case ACT_TANH: hipLaunchKernelGGL((gemm_tile_kernel<NSPLIT, ACT_TANH, BM, BN, WM, WN, G, OPT>), grid, block, 0, st, p); break;
        default: hipLaunchKernelGGL((gemm_tile_kernel<NSPLIT, ACT_NONE, BM, BN, WM, WN, G, OPT>), grid, block, 0, st, p); break;
    }
}

template <int NSPLIT>
static bool launch_variant(const GemmParams& p, Engine tile, hipStream_t st) {
    switch (tile) {
        case ENG_TILE_128: launch_cfg<NSPLIT, 128, 128, 2, 2, 0, 0>(p, st); return true;                               // reg-staged 128x128 (N % 256 != 0)
        case ENG_TILE_DMA: if (p.N % 256) return false; launch_cfg<NSPLIT, 128, 256, 2, 4, 1, 0>(p, st); return true;  // LDS-DMA, double buffered, 1 WG/CU: launches of no more workgroups than CUs (gemm_dispatch.hip)
        case ENG_TILE: if (p.N % 256) return false; launch_cfg<NSPLIT, 128, 256, 2, 4, 0, 0>(p, st); return true;  // reg-staged 128x256 (default)
        case ENG_TILE_256: if (p.N % 256) return false; launch_cfg<NSPLIT, 256, 256, 4, 4, 0, 0>(p, st); return true; // 16 waves, 1 WG/CU, 25 % fewer operand bytes per flop
        default: return false;
    }
}

bool launch_gemm_tile(const GemmParams& p, int nsplit, Engine tile, hipStream_t st) {
    if (p.M <= 0) return true;
    if (p.k_splits > 1 && (p.K % (64 * p.k_splits) || p.out_kind != OUT_F32 || p.hm_rows || p.bias || p.r_hi || p.act != ACT_NONE)) return false;
    if (nsplit == 3) {   // A and W both split: 128x128 tile (4 operand planes x 16 KiB = 64 KiB, 2 workgroups / CU)
        if (!p.w_lo) return false;
        launch_cfg<3, 128, 128, 2, 2, 0, 0>(p, st);
        return true;
    }
    return nsplit == 2 ? launch_variant<2>(p, tile, st) : launch_variant<1>(p, tile, st);
}
