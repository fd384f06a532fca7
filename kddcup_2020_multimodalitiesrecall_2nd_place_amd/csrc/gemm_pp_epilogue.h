// Epilogue of the ping-pong GEMMs (gemm_pp.hip, gemm_pp2.hip): their MFMAs run with swapped operands, so every 16x16 result
// fragment sits transposed in the lane and is stored straight from the accumulators.
#pragma once
#include "kernels.h"

// Epilogue for transposed accumulator fragments: acc[i][j] of lane l = C[row0 + 16 i + (l & 15)][col0 + 16 j + 4 (l >> 4) + 0..3].
// No LDS and no cross-lane traffic: 8 row pointers per lane, the four column fragments at immediate offsets; bias / residual /
// activation / split run on float4.  (The strip-transposing epilogue of gemm_epilogue.h spent ~11 us per 256x256 tile, almost all of
// it instruction issue -- 16 ds_write_b32 + 4 ds_read_b128 + per-store 64-bit address arithmetic and uniform branches -- while a CU
// can write the tile in 2.2 us: tools/probes/store_probe.hip.)
template <int ACT, int FM, int FN>
__device__ __forceinline__ void pp_epilogue(const GemmParams& p, f32x4 (&acc)[FM][FN], int row0, int col0, int lane, int Meff) {
    const int mrow = lane & 15, nq = lane >> 4;
    const int col = col0 + nq * 4;                       // + 16 j
    f32x4 bias4[FN];
#pragma unroll
    for (int j = 0; j < FN; ++j) bias4[j] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + col + 16 * j) : f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.col_scale) {   // fp8 weights: the accumulator is in units of the weight row's quantisation scale
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const f32x4 s4 = *reinterpret_cast<const f32x4*>(p.col_scale + col + 16 * j);
#pragma unroll
            for (int i = 0; i < FM; ++i) acc[i][j] *= s4;
        }
    }
    const bool f32_out = p.out_kind == OUT_F32, resid = p.r_hi != nullptr;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int row = row0 + 16 * i + mrow;
        if (row >= Meff) continue;
        const long long orow = p.cmap(row);
        if (resid) {
            const long long ro = (p.r_index ? (long long)p.r_index[row] : p.rmap(row)) * (long long)p.ldr + col;
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const bf16x4 rh = *reinterpret_cast<const bf16x4*>(p.r_hi + ro + 16 * j);
                const bf16x4 rl = *reinterpret_cast<const bf16x4*>(p.r_lo + ro + 16 * j);
                acc[i][j] += bias4[j];
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][e] += join_bf16(rh[e], rl[e]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] += bias4[j];
        }
        if (f32_out) {
            float* dst = p.c_f32 + orow * p.ldc + col;
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                f32x4 v = acc[i][j];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], ACT);
                *reinterpret_cast<f32x4*>(dst + 16 * j) = v;
            }
        } else if (p.out_kind == OUT_F8) {
            unsigned char* d8 = p.c_f8 + orow * p.ldf8 + col;
#pragma unroll
            for (int j = 0; j < FN; ++j)
                *reinterpret_cast<unsigned*>(d8 + 16 * j) = pack4_f8(apply_act(acc[i][j][0], ACT), apply_act(acc[i][j][1], ACT),
                                                                     apply_act(acc[i][j][2], ACT), apply_act(acc[i][j][3], ACT));
        } else {
            bf16* dh = p.c_hi + orow * p.ldp + col;
            bf16* dl = p.c_lo + orow * p.ldp + col;
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                bf16x4 h, l;
#pragma unroll
                for (int e = 0; e < 4; ++e) { bf16 a, c2; split_bf16(apply_act(acc[i][j][e], ACT), a, c2); h[e] = a; l[e] = c2; }
                *reinterpret_cast<bf16x4*>(dh + 16 * j) = h;
                *reinterpret_cast<bf16x4*>(dl + 16 * j) = l;
            }
        }
    }
}

