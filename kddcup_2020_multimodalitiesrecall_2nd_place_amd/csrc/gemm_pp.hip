// Ping-pong phase GEMM (gfx950): 256x256 workgroup tile, 8 wavefronts as 2(M) x 4(N), 128x64 outputs per wave
// (half the LDS fragment bytes per MFMA of the 64x64-per-wave kernels), one workgroup per CU, 2 waves per SIMD.
//
// Same contract / epilogue as gemm_tile.hip (GemmParams: C = act(A W^T + bias (+ residual)), A in split planes).
//
// K is consumed in 32-wide STAGES.  A stage = A [256][hi 64 B | lo 64 B] + W [256][64 B] (48 KiB at two passes; one pass: A [256][64 B]),
// kept in a 3-slot LDS ring filled by `global_load_lds_dwordx4` (LDS-DMA).  Every A piece of a wave is 8 rows x 128 B = eight whole
// cache lines of the hl32 plane layout (common.h), every W piece one contiguous KiB of the tiled weights -- with separate row-major
// planes a piece was sixteen half lines and each line was fetched twice (6-10 % of a launch, profiles/r03a_gemm_layout.txt).  Bank
// swizzles are applied on the per-lane SOURCE address (LDS-DMA writes lane-linear) and undone by the fragment reads' addresses:
// 128-byte A rows: 16-B chunk c of row r sits at chunk c ^ ((r >> 1) & 7); 64-byte rows (W, one-pass A): c ^ {0,3,2,1}[(r>>2)&3];
// both make every ds_read_b128 lane group hit 16 distinct 16-B slots.
// Each stage runs as 2 PHASES of two 64x32 quadrants of the wave's outputs (32 MFMAs at two passes; rounds 1-2: 4 phases of 16):
//
//     phase:   L: ds_read the fragments this phase needs (+ issue 3 LDS-DMA pieces of stage s+2)
//              s_barrier ; s_waitcnt lgkmcnt(0)
//              M: s_setprio 1 ; 32 x v_mfma_f32_16x16x32_bf16 ; s_setprio 0
//              s_barrier
//
// The two wave rows (wm = 0 / 1; one wave of each per SIMD) run STAGGERED by one barrier, so on every SIMD one wave is
// in its M section while the other is in its L section: the matrix pipe sees back-to-back MFMAs while LDS reads,
// DMA issue and address math ride in the other wave's slots.  Quadrant order (A0,B0) (A0,B1) | (A1,B1) (A1,B0) reuses
// one register set per A half: 12 / 8 ds_read_b128 per phase at two passes.
//
// DMA waits are COUNTED: stage s+2 is issued during stage s; its second phase waits `vmcnt(pieces per stage)`
// (= everything but the stage just issued), so one full stage stays in flight across every barrier.
// RAW: the wait sits before the second phase's first barrier and the first read of that data is in the next phase (one
// barrier later for the staggered wave row).  WAR: a slot's last reads (second phase) are retired (`lgkmcnt(0)`) before that
// phase's first barrier; its refill starts one phase later.
//
// Two shape parameters are compile-time (both measured, profiles/r02f_*; the defaults are what ships): MMS_PP_NSLOT1, the ring depth
// of the single-plane instantiations whose 32 KiB stages leave room for 4 or 5 slots (D = slots - 1 stages in flight; not faster),
// and MMS_PP_WN, the wave grid (4: as above; 2: 4(M) x 2(N), 64x128 outputs per wave, 16 instead of 20 ds_read_b128 per stage; equal).
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "kernels.h"
#include "gemm_pp_epilogue.h"
#include "gemm_pp_ln.h"

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

namespace {

__device__ __forceinline__ int pp_swz(int r) { return (4 - ((r >> 2) & 3)) & 3; }

template <int N> __device__ __forceinline__ void pp_wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ring depth of the single-plane instantiations (mode 1, fp8): their 32 KiB stages leave room for more than three slots
#ifndef MMS_PP_NSLOT1
#define MMS_PP_NSLOT1 3
#endif
// wave grid: 8 waves as (8 / WN)(M) x WN(N).  WN = 4: 128 x 64 outputs per wave; WN = 2: 64 x 128 (with two A planes the A fragments
// are the expensive ones: 4 x 2 + 8 = 16 ds_read_b128 per stage and wave instead of 8 x 2 + 4 = 20)
#ifndef MMS_PP_WN
#define MMS_PP_WN 4
#endif
// phases per K stage: 2 (product) or 4 (the rounds 1-2 schedule: one 64x32 quadrant of 16 MFMAs per phase)
#ifndef MMS_PP_PHASES
#define MMS_PP_PHASES 2
#endif


__device__ __forceinline__ void pp_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

}  // namespace

// DIAG (timing diagnostics only, results WRONG): bit 0 drops the phase barriers, bit 1 the in-loop LDS-DMA issue, bit 2 the in-loop ds_reads,
// bit 3 re-reads the first two K-stages (always cache hits)
// (Precision mode 4 ran here on the non-scaled v_mfma_f32_16x16x32_fp8_fp8 in round 2; it now runs on the MX-scaled instruction at
// twice that rate in gemm_mx.hip.)
// LNF: N == 768 with the fused bias + residual + LayerNorm epilogue of gemm_pp_ln.h (persistent launches only).
template <int NSPLIT, int ACT, int DIAG, bool PERSIST, bool LNF = false>
__global__ __launch_bounds__(512) void gemm_pp_kernel(const GemmParams p) {
    static_assert(!LNF || (PERSIST && ACT == ACT_NONE && DIAG == 0), "the LayerNorm epilogue belongs to the persistent, activation-free kernel");
    constexpr int BM = 256, BN = 256, NW = 8, WAVES_N = LNF ? 4 : MMS_PP_WN, WAVES_M = NW / WAVES_N;
    constexpr int TM = BM / WAVES_M, TN = BN / WAVES_N, FM = TM / 16, FN = TN / 16, HM = FM / 2, HN = FN / 2;     // HM x HN fragments per phase
    constexpr int PLANE = 256 * 64;                 // one operand plane of a stage: 256 rows x 32 bf16
    constexpr int SLOT = (NSPLIT + 1) * PLANE;
    constexpr int NSLOT = (NSPLIT == 1 && !LNF) ? MMS_PP_NSLOT1 : 3;
    constexpr int D = NSLOT - 1;                    // stages in flight ahead of the one being consumed
    constexpr int P = 2 * (NSPLIT + 1);             // LDS-DMA pieces per wave per stage (8 KiB = 128 rows per piece)
    static_assert(NSLOT >= 3 && NSLOT * SLOT <= 160 * 1024 && (D - 1) * P < 64, "ring must fit the LDS and the vmcnt counter");
    constexpr int EPI_BYTES = NW * 16 * (TN + 4) * 4;
    static_assert(NSLOT * SLOT >= EPI_BYTES, "epilogue strip must fit in the ring");
    // DIAG 64 (lab): per-phase barrier stamps of waves 0 and 4 of workgroup 0's first tile, kept in 8 KiB of LDS behind the ring (stores to
    // global memory would sit in vmcnt and disturb the counted LDS-DMA waits) and dumped after the tile: tools/pp_phase.py
    constexpr int STAMP_BYTES = (DIAG & 64) ? 8192 : 0;
    __shared__ __attribute__((aligned(16))) unsigned char smem[NSLOT * SLOT + STAMP_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    int stamp_i = 0;
    const bool stamping = (DIAG & 64) && blockIdx.x == 0 && (tid == 0 || tid == 256);
    auto stamp = [&]() {
        if constexpr ((DIAG & 64) != 0) {
            if (stamping && stamp_i < 512) { reinterpret_cast<unsigned long long*>(smem + NSLOT * SLOT)[(tid >> 8) * 512 + stamp_i] = __builtin_readcyclecounter(); ++stamp_i; }
        }
    };

    const unsigned long long tr0 = (DIAG & 32) ? wall_clock64() : 0;      // DIAG 32: per-workgroup timeline (100 MHz ticks) into p.flop_counter
    int Meff = p.M;
    if (p.m_dev) { const int md = *p.m_dev; Meff = md < Meff ? md : Meff; }
    if (!(DIAG & 96) && p.flop_counter && blockIdx.x == 0 && tid == 0)
        atomicAdd(p.flop_counter, 2ull * (unsigned long long)Meff * (unsigned long long)p.N * (unsigned long long)p.K);
    if (LNF && tid == 0) atomicAdd(&p.ln_ctl[0], 1);     // check-in: "this workgroup is resident" (gemm_pp_ln.h)
    const int nbn = p.N / BN, nbm = (Meff + BM - 1) / BM;
    // LNF: the three tiles of a row panel exchange row statistics, so they must run in the SAME persistent round, and should share an
    // XCD's L2 (one fetch of the A panel).  A workgroup keeps ONE seat for the whole launch: with c = gridDim.x / 8 workgroups per XCD,
    // seats loc = blockIdx.x / 8 < 3 (c / 3) of XCD blockIdx.x % 8 form c / 3 triples inside the XCD; the c % 3 seats left over per XCD
    // are pooled into triples that span XCDs (32 CUs per XCD: 80 + 5 panels per round on 255 of the 256 CUs; round 2 gave the left-over
    // seats no work at all: 6 % of the chip).  Round r of seat (panel pir, tile bn) is panel r * PR + pir.
    const int ln_c = (int)gridDim.x >> 3, ln_t = ln_c / 3, ln_l = ln_c - 3 * ln_t;
    const int PR = LNF ? 8 * ln_t + (8 * ln_l) / 3 : 1;      // panels per round
    int ln_pir = 0, ln_bn = 0;
    if constexpr (LNF) {
        const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
        if (loc < 3 * ln_t) { ln_pir = xcd * ln_t + loc / 3; ln_bn = loc % 3; }
        else { const int e = xcd * ln_l + (loc - 3 * ln_t); ln_pir = e < 3 * ((8 * ln_l) / 3) ? 8 * ln_t + e / 3 : nbm; ln_bn = e % 3; }     // pir = nbm: a seat without work
    }
    const int nblk = LNF ? (nbm + PR - 1) / PR : nbm * nbn;     // LNF: rounds
    auto valid = [&](int v) { return !LNF || v * PR + ln_pir < nbm; };
    const int vstep = LNF ? 1 : (int)gridDim.x;
    int vb = LNF ? 0 : (int)blockIdx.x;  // virtual block id; PERSIST: the workgroup walks vb, vb + gridDim.x, ... (gridDim.x % 8 == 0).  LNF: the round
    while (vb < nblk && !valid(vb)) vb += vstep;
    if (vb >= nblk) return;
    // fused or plain is settled BEFORE the first tile (the fused route appends the residual stages to its K loop): normally every
    // workgroup has checked in within a few microseconds of the first one
    int ln_decided = 0;
    if constexpr (LNF) ln_decided = ln_decide(p.ln_ctl, (int)gridDim.x, lane);
    // LDS-DMA sources.  A2 (two planes): the A region of a slot is [256 rows][128 B]; piece q (0..3) = rows q*64 + wave*8 + lane/8, physical
    // 16-B chunk lane%8 <- logical chunk (lane%8) ^ ((row>>1)&7) of the row's [hi 64 B | lo 64 B] stage line.  One plane (and W): [256][64 B];
    // piece h (0/1) = rows h*128 + wave*16 + lane/4, physical chunk lane%4 <- logical chunk (lane%4) ^ pp_swz(row).
    constexpr bool A2 = NSPLIT == 2;
    constexpr int NAP = A2 ? 4 : 2;                 // A pieces per wave and stage
    constexpr int ASTEP = 64;                       // elements between two K stages of an A row (hl32: a hi and a lo block per stage)
    constexpr int WSTEP = 512;                      // ... of a W row (one 1-KiB tile per stage)
    const int gr_l = lane >> 2, gc = lane & 3;
    const bf16* a_src[LNF ? 1 : NAP];
    const bf16* w_src[LNF ? 1 : 2];
    // LNF: the residual rides in as EIGHT extra K stages -- A = the residual planes' columns [bn*256, bn*256 + 256) of the tile's rows,
    // W = the 256 x 256 identity (built in registers, never fetched): acc += r_hi * I + r_lo * I = r_hi + r_lo, exact products, and the
    // bytes arrive through the LDS-DMA ring under the MFMAs like every other operand (fetched straight into the accumulators at the tile
    // boundary they cost 23 us per persistent round with nothing to overlap them: profiles/r02c_ln_fused_trace.txt)
    constexpr int NRES = LNF ? BN / 32 : 0;
    int r_row[LNF ? NAP : 1];      // residual row of each of the lane's A pieces (the address is formed at issue time: 4 registers, not 8)
    int bm, bn;
    auto setup = [&](int v) {
        // bijective XCD remap over the live tiles (virtual block v runs on XCD v % 8)
        if constexpr (LNF) {
            bm = v * PR + ln_pir;
            bn = ln_bn;
            if (p.reverse) bm = nbm - 1 - bm;
        } else {
        const int q = nblk >> 3, r8 = nblk & 7, xcd = v & 7, loc = v >> 3;
        int bid = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + loc;
        if (p.reverse) bid = nblk - 1 - bid;
        bm = bid / nbn; bn = bid % nbn;
        }
        auto a_row = [&](int r) {      // first element of tile row r's A row (clamped to the last live row), in 2-byte units
            int gr = bm * BM + r;
            gr = gr < Meff ? gr : Meff - 1;
            const long long lrow = (p.a_index ? (long long)p.a_index[gr] : p.amap(gr)) * (long long)p.lda;
            return p.a_hi + 2 * lrow;
        };
        if constexpr (LNF) {
            // rows only (the launcher guarantees identity row maps): both the A and the residual address of a piece are formed from
            // r_row[q] at issue time -- 4 registers instead of 16
        } else if constexpr (A2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = q * 64 + wave * 8 + (lane >> 3);
                a_src[q] = a_row(r) + ((lane & 7) ^ ((r >> 1) & 7)) * 8;
            }
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int r = h * 128 + wave * 16 + gr_l;
                a_src[h] = a_row(r) + (gc ^ pp_swz(r)) * 8;
            }
        }
        if constexpr (!LNF) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = h * 128 + wave * 16 + gr_l;
            w_src[h] = p.w + wtile_off(bn * BN + r, 0, p.K) + (gc ^ pp_swz(r)) * 8;
        }
        }
        if constexpr (LNF) {
#pragma unroll
            for (int q = 0; q < NAP; ++q) {
                const int r = q * 64 + wave * 8 + (lane >> 3);
                const int gr = bm * BM + r;
                r_row[q] = gr < Meff ? gr : Meff - 1;
            }
        }
    };
    setup(vb);
    // piece q of stage `st` into ring slot `slot`: the NAP A pieces first, then the two W halves; every piece is 1 KiB per wave
    const int nk_main = p.K / 32;      // stages of the product proper
    auto issue = [&](int q, int st, int slot) {
        const int kst = (DIAG & 8) ? (st & 1) : st;       // DIAG 8: L2-resident source
        unsigned char* d;
        const bf16* s;
        if (q < NAP) {
            d = smem + slot * SLOT + q * 8192 + wave * 1024;        // A2: rows q*64 + wave*8 .. of 128 B; else rows q*128 + wave*16 .. of 64 B
            if constexpr (LNF) {
                const int r = q * 64 + wave * 8 + (lane >> 3);
                int chunk = ((lane & 7) ^ ((r >> 1) & 7)) * 8;
                int rr = r_row[q];
                asm volatile("" : "+v"(rr), "+v"(chunk));      // formed HERE, every time (one 64-bit multiply-add): hoisted, the eight row products cost 16 registers
                if (kst >= nk_main) s = p.r_hi + 2 * ((long long)rr * p.ldr + bn * BN) + chunk + (kst - nk_main) * ASTEP;
                else s = p.a_hi + 2 * ((long long)rr * p.lda) + chunk + kst * ASTEP;
            } else s = a_src[q] + kst * ASTEP;
        } else {
            const int h = q - NAP;
            d = smem + slot * SLOT + NSPLIT * PLANE + h * 8192 + wave * 1024;
            if constexpr (LNF) {
                // tile row (bn * 16 + h * 8 + wave) of the tiled weights is wave-uniform; the lane's part of the address is one register.
                // (Residual stages fetch no W piece at all: stage(), NRES_NEXT)
                int w_lane = (gr_l << 5) + (gc ^ pp_swz(gr_l)) * 8;
                asm volatile("" : "+v"(w_lane));      // a seat's bn never changes: without this the W addresses of every piece are hoisted out of the tile loop (10 registers)
                s = p.w + (((long long)(bn * 16 + h * 8 + wave) * (p.K >> 5) + kst) << 9) + w_lane;
            } else s = w_src[h] + kst * WSTEP;
        }
        __builtin_amdgcn_global_load_lds((glb_void*)s, (lds_void*)d, 16, 0, 0);
    };

    f32x4 acc[FM][FN];
    const int ns = nk_main + (LNF && ln_decided == 1 ? NRES : 0);      // K stages (>= 2: K % 64 == 0); fused LayerNorm route: + the residual stages

    // fragment read offsets inside a slot (lane part; the rest are immediates)
    const int fr = lane & 15, fk = lane >> 4;
    constexpr int AROW = A2 ? 128 : 64;               // bytes of an A row in a slot
    const int laneA = (wm * TM + fr) * AROW + ((A2 ? (fk ^ ((fr >> 1) & 7)) : (fk ^ pp_swz(fr))) << 4);      // hi fragment; the lo one: chunk ^ 4
    const int laneB = NSPLIT * PLANE + (wn * TN + fr) * 64 + ((fk ^ pp_swz(fr)) << 4);

    bf16x8 a[NSPLIT][HM], b0[HN], b1[HN];
    auto read_a = [&](const unsigned char* sb, int mh) {
#pragma unroll
        for (int pl = 0; pl < NSPLIT; ++pl)
#pragma unroll
            for (int i = 0; i < HM; ++i) a[pl][i] = *reinterpret_cast<const bf16x8*>(sb + (laneA ^ (pl << 6)) + (mh * (TM / 2) + i * 16) * AROW);
    };
    auto read_b = [&](const unsigned char* sb, int nh, bf16x8 (&b)[HN]) {
#pragma unroll
        for (int j = 0; j < HN; ++j) b[j] = *reinterpret_cast<const bf16x8*>(sb + laneB + (nh * (TN / 2) + j * 16) * 64);
    };
    // M section: quadrant (mh, nh); hi-plane MFMAs first, then lo (dependent pairs 8 MFMAs apart).  The operands are SWAPPED
    // (W fragment first): each 16x16 result then sits transposed in the lane -- lane l holds row l&15, columns 4*(l>>4)..+3 --
    // so the epilogue stores float4s straight from the accumulators, no LDS transposition (pp_epilogue below).
    auto mma = [&](int mh, int nh, const bf16x8 (&b)[HN]) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int pl = 0; pl < NSPLIT; ++pl)
#pragma unroll
            for (int i = 0; i < HM; ++i)
#pragma unroll
                for (int j = 0; j < HN; ++j) {
                    acc[mh * HM + i][nh * HN + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[pl][i], acc[mh * HM + i][nh * HN + j], 0, 0, 0);   // swapped: C^T fragment
                }
        __builtin_amdgcn_s_setprio(0);
    };

    // PRE: stage s+2 exists and is issued during this stage; WAITN: outstanding pieces allowed at the phase-4 wait
    // RES (LNF): a residual stage -- W is the identity, so only the wave column that owns the stage's 32 output columns (stage j of the
    // eight: columns 32 j .. 32 j + 31 = half nh = j & 1 of wave column j >> 1) has anything to add: it runs the two quadrants (A0, B_nh),
    // (A1, B_nh) = 16 + 16 MFMAs; the other waves only keep the ring and the barriers going
    // NRES_NEXT (LNF): the stage issued during this one (s + D) is a residual stage: only its A pieces are fetched (its W is the identity built in
    // registers), and the counted wait allows exactly that many pieces to stay in flight
    auto stage = [&](auto pre_tag, auto wait_tag, auto res_tag, auto nres_tag, int s, int slot) {
        constexpr bool PRE = decltype(pre_tag)::value;
        constexpr int WAITN = decltype(wait_tag)::value;
        constexpr bool RES = decltype(res_tag)::value;
        constexpr int PI = decltype(nres_tag)::value ? NAP : P;      // pieces of stage s + D
        const int rj = RES ? s - nk_main : 0;
        const bool own = !RES || wn == (rj >> 1);
        const bool rh1 = RES && (rj & 1);      // owner of a residual stage: which half of its columns
        const unsigned char* sb = smem + slot * SLOT;
        const int nslot = slot == 0 ? NSLOT - 1 : slot - 1;     // (slot + D) % NSLOT: the slot stage s-1 just left
        constexpr bool DMA = PRE && !(DIAG & 2), RD = !(DIAG & 4);
        auto bar = [&]() { stamp(); if (DIAG & 1) { __builtin_amdgcn_sched_barrier(0); asm volatile("" ::: "memory"); } else pp_barrier(); stamp(); };
#if MMS_PP_PHASES == 2
        // TWO phases of 2 x 16 MFMAs per stage (round 3): every barrier hand-over between the wave rows costs ~90 cycles of idle matrix pipe
        // (profiles/r03h_pp_phase_stamps.txt), so 32 MFMAs per hand-over instead of 16 is worth ~5 % of a launch
        // (profiles/r03g_two_phases.txt; round 1 had measured "no difference" on an earlier engine).
        // phase A: quadrants (A0, B0), (A0, B1) -- 12 fragment reads at two passes, three LDS-DMA pieces of stage s+D
        if constexpr (RES) {
            // identity fragment j of the stage's 32 columns: lane (fr, fk) holds W[16 j + fr][8 fk + e], e = 0..7 -> 1.0 where 16 j + fr == 8 fk + e
            if (own) {
                int e0 = fr - 8 * fk;
                asm volatile("" : "+v"(e0));      // rebuilt per stage (a dozen VALU ops): hoisted out of the loop the two fragments cost 8 registers the kernel does not have
#pragma unroll
                for (int j = 0; j < HN; ++j) {
                    const int e = 16 * j + e0;
                    u32x4 v = u32x4{0u, 0u, 0u, 0u};
                    const unsigned one = (e & 1) ? 0x3F800000u : 0x00003F80u;
                    if (e >= 0 && e < 8) { v[0] = (e >> 1) == 0 ? one : 0u; v[1] = (e >> 1) == 1 ? one : 0u; v[2] = (e >> 1) == 2 ? one : 0u; v[3] = (e >> 1) == 3 ? one : 0u; }
                    b0[j] = __builtin_bit_cast(bf16x8, v);
                }
                if (RD) read_a(sb, 0);
            }
        } else if (RD) { read_b(sb, 0, b0); read_b(sb, 1, b1); read_a(sb, 0); }
        if (DMA) {
#pragma unroll
            for (int q = 0; q < PI / 2; ++q) issue(q, s + D, nslot);
        }
        bar();
        if constexpr (RES) {
            if (own) { if (rh1) mma(0, 1, b0); else mma(0, 0, b0); }
        } else { mma(0, 0, b0); mma(0, 1, b1); }
        bar();
        // phase B: quadrants (A1, B1), (A1, B0) -- 8 reads, the other pieces; my pieces of stage s+1 have landed before the first barrier
        // -> readable next phase.  The reads are RETIRED before that barrier too (lgkmcnt(0)): they are the last reads of this slot, and the
        // first piece that refills it is issued one phase later (phase A of the next stage) -- by then every wave of either row has passed a
        // barrier behind its reads.
        if (RD && own) read_a(sb, 1);
        if (DMA) {
#pragma unroll
            for (int q = PI / 2; q < PI; ++q) issue(q, s + D, nslot);
        }
        if (WAITN >= 0) pp_wait_vmcnt<(WAITN >= 0 && !(DIAG & 2) ? WAITN : 0)>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        bar();
        if constexpr (RES) {
            if (own) { if (rh1) mma(1, 1, b0); else mma(1, 0, b0); }
        } else { mma(1, 1, b1); mma(1, 0, b0); }
        bar();
#else
        // LDS-DMA pieces of stage s+D: two per phase in phases 1-3 (other placements measured no better, profiles/r01c_gemm_variants.txt)
        auto dma = [&](int ph) {
            if (!DMA) return;
#pragma unroll
            for (int q = 0; q < P; ++q)
                if (q / 2 == ph) issue(q, s + D, nslot);
        };
        // phase 1: (A0, B0)
        if (RD) { read_b(sb, 0, b0); read_a(sb, 0); }
        dma(0);
        bar();
        mma(0, 0, b0);
        bar();
        // phase 2: (A0, B1)
        if (RD) read_b(sb, 1, b1);
        dma(1);
        bar();
        mma(0, 1, b1);
        bar();
        // phase 3: (A1, B1)
        if (RD) read_a(sb, 1);
        dma(2);
        bar();
        mma(1, 1, b1);
        bar();
        // phase 4: (A1, B0); my pieces of stage s+1 have landed before the first barrier -> readable next phase
        dma(3);
        if (WAITN >= 0) pp_wait_vmcnt<(WAITN >= 0 && !(DIAG & 2) ? WAITN : 0)>();
        bar();
        mma(1, 0, b0);
        bar();
#endif
    };

    // prologue: stages 0 .. D-1 in flight (fewer when K is that short), stage 0 landed
    const int pre = ns < D ? ns : D;
    auto first_stages = [&]() {
#pragma unroll
        for (int d = 0; d < D; ++d)
            if (d < pre) {
#pragma unroll
                for (int q = 0; q < P; ++q) issue(q, d, d);
            }
    };
    first_stages();
    if constexpr (D == 2) pp_wait_vmcnt<P>();
    else {
        if (pre == D) pp_wait_vmcnt<(D - 1) * P>();
        else if (pre == 3) pp_wait_vmcnt<2 * P>();
        else pp_wait_vmcnt<P>();               // pre == 2 (K % 64 == 0: never fewer than two stages)
    }
    pp_barrier();
    if (wave >= NW / 2) pp_barrier();     // stagger the second half of the waves (one of each half per SIMD) by one barrier
    const unsigned long long tr1 = (DIAG & 32) ? wall_clock64() : 0;

    if (DIAG & 4) { read_b(smem, 0, b0); read_b(smem, 1, b1); read_a(smem, 0); }
    for (;;) {
#ifdef MMS_LAB
        const unsigned long long tr_loop = (LNF && p.ln_dbg) ? wall_clock64() : 0;
#else
        const unsigned long long tr_loop = 0; (void)tr_loop;
#endif
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        int slot = 0, s = 0;
        auto run_stages = [&](auto res_tag, auto nres_tag, int s_end) {      // stages [s, s_end) of this kind, every one followed by at least D more
            constexpr int PI = decltype(nres_tag)::value ? NAP : P;
            for (; s < s_end; ++s) {
                stage(std::true_type{}, std::integral_constant<int, (D - 1) * PI>{}, res_tag, nres_tag, s, slot);
                slot = slot == NSLOT - 1 ? 0 : slot + 1;
            }
        };
        // tail: R stages still to come after this one -> R-1 of them may stay in flight; the last stage waits for nothing
        auto tail = [&](auto r_tag, auto res_tag) {
            constexpr int R = decltype(r_tag)::value;
            constexpr int PT = decltype(res_tag)::value ? NAP : P;      // pieces of the stages still in flight (all of one kind inside a tail)
            if (ns - 1 - s == R) {
                stage(std::false_type{}, std::integral_constant<int, R >= 1 ? (R - 1) * PT : -1>{}, res_tag, res_tag, s, slot);
                slot = slot == NSLOT - 1 ? 0 : slot + 1;
                ++s;
            }
        };
        auto tails = [&](auto res_tag) {
            if constexpr (D >= 4) tail(std::integral_constant<int, 3>{}, res_tag);
            if constexpr (D >= 3) tail(std::integral_constant<int, 2>{}, res_tag);
            tail(std::integral_constant<int, 1>{}, res_tag);
            tail(std::integral_constant<int, 0>{}, res_tag);
        };
        if (LNF && ns > nk_main) {      // the product's stages, then the residual stages (NRES > D: the tail lies inside them)
            static_assert(!LNF || (NRES > D && D == 2), "the K loop's tail must lie inside the residual stages");
            constexpr auto T = std::integral_constant<bool, LNF>{};
            run_stages(std::false_type{}, std::false_type{}, nk_main - D);      // product stages that prefetch product stages
            run_stages(std::false_type{}, T, nk_main);                          // the last D of them prefetch residual stages: A pieces only
            run_stages(T, T, ns - D);
            tails(T);
        } else {
            run_stages(std::false_type{}, std::false_type{}, ns - D);
            tails(std::false_type{});
        }
        if (wave < NW / 2) pp_barrier();     // re-align the two halves: nobody reads the ring any more
        const unsigned long long tr2 = (DIAG & 32) ? wall_clock64() : 0;

        if (DIAG & 16) {   // no epilogue (keep the accumulators live)
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
            if (t == 12345.678f) p.c_f32[tid] = t;
            return;
        }
        // PERSIST: the workgroup's next tile.  Its first two stages go out BEFORE this tile's stores (the epilogue needs no LDS), so
        // their round trip -- the whole prologue of a fresh workgroup -- runs under the store drain; one full `vmcnt(0)` then covers
        // both, and neither a workgroup retirement nor a dispatch (~4.6 us) sits between two tiles.
        const int row0 = bm * BM + wm * TM, col0 = bn * BN + wn * TN;
        const int ebm = bm, ebn = bn, evb = LNF ? bm * 3 + bn : vb;      // this tile (setup() below moves bm / bn / vb on to the next one)
        int nvb = vb + vstep;
        while (nvb < nblk && !valid(nvb)) nvb += vstep;
        const bool more = PERSIST && nvb < nblk;
        if (more) {
            vb = nvb;
            setup(vb);
            first_stages();
        }
        if constexpr (LNF) {
#ifdef MMS_LAB
            if (p.ln_dbg && tid == 0) p.ln_dbg[(long long)evb * 6] = tr_loop;
#endif
            if (ln_decided == 1) pp_epilogue_ln<FM, FN>(p, acc, ebm, ebn, wm, wn, lane, tid, Meff, smem + 2 * SLOT, evb);   // ring slot 2 is idle here
            else pp_epilogue_plain_f32<FM, FN>(p, acc, row0, col0, lane, Meff);   // the LayerNorm kernel behind this launch finishes the job
        } else
        pp_epilogue<ACT, FM, FN>(p, acc, row0, col0, lane, Meff);
        if ((DIAG & 64) && stamping && p.flop_counter) {
            unsigned long long* out = p.flop_counter + (tid >> 8) * 512;
            for (int i = 0; i < 512; ++i) out[i] = i < stamp_i ? reinterpret_cast<unsigned long long*>(smem + NSLOT * SLOT)[(tid >> 8) * 512 + i] : 0ull;
            stamp_i = 512;       // first tile only
        }
        if ((DIAG & 32) && tid == 0 && p.flop_counter) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // epilogue stores acknowledged
            unsigned long long* t = p.flop_counter + 5ull * blockIdx.x;
            t[0] = tr0; t[1] = tr1; t[2] = tr2; t[3] = wall_clock64(); t[4] = ((unsigned long long)bm << 32) | (unsigned)bn;
        }
        if (!more) break;
        pp_wait_vmcnt<0>();
        pp_barrier();
        if (wave >= NW / 2) pp_barrier();     // stagger again
    }
}

static int pp_cu_count() {
#ifdef MMS_LAB
    static const int lab_grid = getenv("MMS_PP_GRID") ? atoi(getenv("MMS_PP_GRID")) / 8 * 8 : 0;   // lab: persistent GEMMs on part of the chip (tools/dual_stream.py)
    if (lab_grid) return lab_grid;
#endif
    return device_cu_count();
}

template <int NSPLIT, int DIAG, bool PERSIST = false>
static void launch_pp_ns(const GemmParams& p, hipStream_t st) {
    const int nblk = ((p.M + 255) / 256) * (p.N / 256);
    const dim3 grid(PERSIST && nblk > pp_cu_count() ? pp_cu_count() : nblk), block(512);
    switch (p.act) {
        case ACT_RELU: hipLaunchKernelGGL((gemm_pp_kernel<NSPLIT, ACT_RELU, DIAG, PERSIST>), grid, block, 0, st, p); break;
        case ACT_GELU_TANH: hipLaunchKernelGGL((gemm_pp_kernel<NSPLIT, ACT_GELU_TANH, DIAG, PERSIST>), grid, block, 0, st, p); break;
        case ACT_GELU_ERF: hipLaunchKernelGGL((gemm_pp_kernel<NSPLIT, ACT_GELU_ERF, DIAG, PERSIST>), grid, block, 0, st, p); break;
        case ACT_TANH: hipLaunchKernelGGL((gemm_pp_kernel<NSPLIT, ACT_TANH, DIAG, PERSIST>), grid, block, 0, st, p); break;
        default: hipLaunchKernelGGL((gemm_pp_kernel<NSPLIT, ACT_NONE, DIAG, PERSIST>), grid, block, 0, st, p); break;
    }
}

// N == 768 with the fused bias + residual + LayerNorm epilogue (nsplit 2, bf16 planes)
bool launch_gemm_pp_ln(const GemmParams& p, int nsplit, hipStream_t st) {
    if (p.M <= 0) return true;
    if (p.N != 768 || p.K % 64 || !p.ln_gamma || !p.ln_beta || !p.ln_stats || !p.ln_ctl || !p.r_hi || p.act != ACT_NONE) return false;
    if (p.a_index || p.amap.grp || p.r_index || p.rmap.grp || p.cmap.grp) return false;     // identity row maps only (the kernel forms A / residual addresses from the row number)
    // one workgroup per CU, every one a seat of a panel triple (gemm_pp_kernel); all of them check in, seats without work leave at once
    const dim3 grid(pp_cu_count()), block(512);
    if (nsplit != 2) return false;      // two-pass bf16 planes only
    hipLaunchKernelGGL((gemm_pp_kernel<2, ACT_NONE, 0, true, true>), grid, block, 0, st, p);
    return true;
}

bool launch_gemm_pp(const GemmParams& p, int nsplit, int diag, hipStream_t st, bool persist) {
    if (p.M <= 0) return true;
    if (p.N % 256 || p.K % 64 || nsplit > 2) return false;
#ifdef MMS_LAB
    if (diag) {   // timing diagnostics (two-pass only)
        if (nsplit != 2) return false;
        switch (diag) {
            case 1: launch_pp_ns<2, 1>(p, st); break; case 2: launch_pp_ns<2, 2>(p, st); break; case 3: launch_pp_ns<2, 3>(p, st); break;
            case 4: launch_pp_ns<2, 4>(p, st); break; case 6: launch_pp_ns<2, 6>(p, st); break; case 7: launch_pp_ns<2, 7>(p, st); break;
            case 32: {   // timeline dump: /tmp/pp_trace.bin = 5 x u64 per workgroup (entry, main loop start, main loop end, stores acknowledged, tile)
                const int nblk = ((p.M + 255) / 256) * (p.N / 256);
                unsigned long long* buf = nullptr;
                if (hipMalloc(&buf, (size_t)nblk * 40) != hipSuccess) return false;
                (void)hipMemsetAsync(buf, 0, (size_t)nblk * 40, st);
                GemmParams q = p; q.flop_counter = buf;
                launch_pp_ns<2, 32>(q, st);
                std::vector<unsigned long long> h((size_t)nblk * 5);
                (void)hipStreamSynchronize(st);
                (void)hipMemcpy(h.data(), buf, (size_t)nblk * 40, hipMemcpyDeviceToHost);
                (void)hipFree(buf);
                if (FILE* f = fopen("/tmp/pp_trace.bin", "wb")) { fwrite(h.data(), 8, h.size(), f); fclose(f); }
                break;
            }
            case 64: {   // phase stamps: /tmp/pp_phase.bin = 2 x 512 u64 shader-clock stamps (wave 0, wave 4) of workgroup 0's first tile
                unsigned long long* buf = nullptr;
                if (hipMalloc(&buf, 8192) != hipSuccess) return false;
                (void)hipMemsetAsync(buf, 0, 8192, st);
                GemmParams q = p; q.flop_counter = buf;
                launch_pp_ns<2, 64, true>(q, st);
                std::vector<unsigned long long> h(1024);
                (void)hipStreamSynchronize(st);
                (void)hipMemcpy(h.data(), buf, 8192, hipMemcpyDeviceToHost);
                (void)hipFree(buf);
                if (FILE* f = fopen("/tmp/pp_phase.bin", "wb")) { fwrite(h.data(), 8, h.size(), f); fclose(f); }
                break;
            }
            case 8: launch_pp_ns<2, 8>(p, st); break; case 16: launch_pp_ns<2, 16>(p, st); break; case 23: launch_pp_ns<2, 23>(p, st); break;
            default: return false;
        }
        return true;
    }
#else
    if (diag) return false;   // the timing-only (wrong-result) instantiations exist in libmmscore_lab.so only
#endif
    if (persist) { if (nsplit == 2) launch_pp_ns<2, 0, true>(p, st); else launch_pp_ns<1, 0, true>(p, st); }
    else if (nsplit == 2) launch_pp_ns<2, 0>(p, st); else launch_pp_ns<1, 0>(p, st);
    return true;
}
