// Shared GEMM epilogue (gemm_tile.hip): accumulators -> per-wave LDS strip (16 rows at a time) ->
// row-contiguous float4, so bias / residual / activation / split run vectorised and global stores are 16 B (fp32)
// or 8 B (bf16x4 per plane) per lane, 256 / 128 contiguous bytes per row.
#pragma once
#include "kernels.h"

// Orders this wave's strip writes against its strip reads (and the reverse).  LDS operations of one wavefront execute in
// order, so only the compiler has to be held back; NOT a wavefront-scope release fence: hipcc lowers that to
// `s_waitcnt vmcnt(0) lgkmcnt(0)`, which made every 16-row strip wait for its global stores to be acknowledged (a
// ~1 us round trip, 8-16 times per tile: measured 10-12 us of a 50 us tile, profiles/r01c_gemm_variants.txt).
__device__ __forceinline__ void strip_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// SYNC = false: the caller guarantees that no wave still reads the bytes behind `smem` (persistent kernel: a free ring slot)
template <int ACT, int BM, int BN, int TM, int TN, int FM, int FN, bool SYNC = true>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x4 (&acc)[FM][FN], unsigned char* smem, int bm, int bn,
                                              int wm, int wn, int wave, int lane, int Meff) {
    const int fr = lane & 15, fk = lane >> 4;
    if (SYNC) __syncthreads();  // all waves are done reading operand tiles
    constexpr int ES = TN + 4;                      // strip row stride in floats
    float* strip = reinterpret_cast<float*>(smem) + wave * 16 * ES;
    constexpr int V4_PER_ROW = TN / 4, ROWS_PER_IT = 64 / V4_PER_ROW, ITERS = 16 / ROWS_PER_IT;
    const int er = lane / V4_PER_ROW, ec = (lane % V4_PER_ROW) * 4;
    const int col = bn * BN + wn * TN + ec;
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) bias4 = *reinterpret_cast<const f32x4*>(p.bias + col);
    static_assert(ITERS == FN, "one float4 per lane per 16x16 fragment: the transposed tile reuses the accumulator registers");
    // Per 16-row strip: transpose it through the wave's LDS strip back INTO its own accumulator registers (acc[i][t] then
    // holds row t*ROWS_PER_IT + er, columns ec..ec+3), then bias / residual / activation / split and the global stores.
    // Every strip lives in registers of its own, so the next strip's ds_reads never target a register a pending store
    // still has to read (that hazard made hipcc wait `vmcnt(0)` -- a store round trip -- per strip), and the stores of
    // strip i drain while strip i+1 is being transposed.
#pragma unroll
    for (int i = 0; i < FM; ++i) {
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) strip[(fk * 4 + r) * ES + j * 16 + fr] = acc[i][j][r];
        strip_sync();
#pragma unroll
        for (int t = 0; t < ITERS; ++t) acc[i][t] = *reinterpret_cast<const f32x4*>(strip + (t * ROWS_PER_IT + er) * ES + ec);
        strip_sync();
#pragma unroll
        for (int t = 0; t < ITERS; ++t) {
            const int lrow = t * ROWS_PER_IT + er;
            const int row = bm * BM + wm * TM + i * 16 + lrow;
            f32x4 v = acc[i][t];
            if (row < Meff) {
                v += bias4;
                if (p.r_hi) {
                    const long long ro = (p.r_index ? (long long)p.r_index[row] : p.rmap(row)) * (long long)p.ldr + col;
                    const bf16x4 rh = *reinterpret_cast<const bf16x4*>(plane_ptr(p.r_hi, ro));
                    const bf16x4 rl = *reinterpret_cast<const bf16x4*>(plane_ptr(p.r_lo, ro));
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += join_bf16(rh[e], rl[e]);
                }
                v = apply_act4<ACT>(v);
                const long long orow = p.cmap(row);
                if (p.out_kind == OUT_F32) {
                    float* dst = p.hm_rows ? p.c_f32 + ((long long)((p.hm_col0 + col) >> 6) * p.hm_rows + orow) * 64 + ((p.hm_col0 + col) & 63)
                                           : p.c_f32 + orow * p.ldc + col;
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
    }
}
