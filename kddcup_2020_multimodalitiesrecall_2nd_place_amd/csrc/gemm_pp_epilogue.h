// Epilogue of the ping-pong GEMMs (gemm_pp.hip, gemm_pp2.hip): their MFMAs run with swapped operands, so every 16x16 result
// fragment sits transposed in the lane and is stored straight from the accumulators.
#pragma once
#include "kernels.h"

// Epilogue for transposed accumulator fragments: acc[i][j] of lane l = C[row0 + 16 i + (l & 15)][col0 + 16 j + 4 (l >> 4) + 0..3].
// No LDS and no cross-lane traffic: 8 row pointers per lane, the four column fragments at immediate offsets; bias / residual /
// activation / split run on float4.  (The strip-transposing epilogue of gemm_epilogue.h spent ~11 us per 256x256 tile, almost all of
// it instruction issue -- 16 ds_write_b32 + 4 ds_read_b128 + per-store 64-bit address arithmetic and uniform branches -- while a CU
// can write the tile in 2.2 us: tools/probes/store_probe.hip.)
// Two column fragments (j, j + 1) of one output row as split planes, stored as ONE 16-byte access per plane and lane instead of two
// 8-byte ones: lane group nq holds columns 4 nq .. 4 nq + 3 of every 16-column fragment; v_permlane16_swap exchanges the odd 16-lane
// rows of the first operand with the even rows of the second, after which an even group owns 8 consecutive columns of fragment j and
// the odd group next to it 8 consecutive columns of fragment j + 1.  The plane epilogues are store-ISSUE bound (64 8-byte stores per
// lane and tile), so halving the store count is what shortens them (cdna_hip_programming.md T21).
typedef __attribute__((ext_vector_type(2))) unsigned pp_u32x2;
// `lane_row` = the lane's own pointer into the output row of an hl32 plane (common.h): physical address of logical column
// col0 + 16 (nq & 1) + 4 (nq & 2) -- col0 a multiple of 32, so that is 2 * (row offset + col0) + 16 (nq & 1) + 4 (nq & 2), inside the first
// 32-column block of the wave's columns; fragment pair j (even) then lies 32 (j / 2) logical = 64 (j / 2) physical elements further: an immediate.
__device__ __forceinline__ bf16* pp_plane_lane_row(bf16* plane, long long row_off, int col0, int nq) {
    return plane + 2 * (row_off + col0) + 16 * (nq & 1) + 4 * (nq & 2);
}
__device__ __forceinline__ void pp_store_plane_pair(bf16* lane_row, int j, bf16x4 f0, bf16x4 f1) {
    const pp_u32x2 x = __builtin_bit_cast(pp_u32x2, f0), y = __builtin_bit_cast(pp_u32x2, f1);
    const auto r0 = __builtin_amdgcn_permlane16_swap(x[0], y[0], false, false);
    const auto r1 = __builtin_amdgcn_permlane16_swap(x[1], y[1], false, false);
    const u32x4 v = {r0[0], r1[0], r0[1], r1[1]};      // {X'.lo, X'.hi, Y'.lo, Y'.hi} = 8 consecutive bf16
    *reinterpret_cast<u32x4*>(lane_row + 64 * (j >> 1)) = v;
}

// AGPR_ACC (gemm_mx.hip): the accumulators live in the accumulator half of the register file; an empty asm with an "+a" operand at the
// top of every strip keeps the compiler from copying all 128 of them into arch VGPRs before the first strip (it has only 128 of those)
// SCALED: the accumulators are in units of a per-column power-of-two scale (p.col_scale: the h3 / fp8 engines of gemm_mx.hip); the bf16
// engines pass false so that no registers are set aside for it
template <int ACT, int FM, int FN, bool AGPR_ACC = false, bool SCALED = false>
__device__ __forceinline__ void pp_epilogue(const GemmParams& p, f32x4 (&acc)[FM][FN], int row0, int col0, int lane, int Meff) {
    const int mrow = lane & 15, nq = lane >> 4;
    const int col = col0 + nq * 4;                       // + 16 j
    f32x4 bias4[FN];
#pragma unroll
    for (int j = 0; j < FN; ++j) bias4[j] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + col + 16 * j) : f32x4{0.f, 0.f, 0.f, 0.f};
    // fp8 / h3 weights: the accumulator is in units of the weight row's power-of-two quantisation scale.  Applied strip by strip (not to
    // all accumulators up front): gemm_mx.hip keeps its accumulators in the AGPR half of the register file and only one strip of them
    // should be in arch VGPRs at a time
    const bool scaled = SCALED && p.col_scale != nullptr;
    f32x4 scale4[SCALED ? FN : 1];
    if constexpr (SCALED) {
#pragma unroll
        for (int j = 0; j < FN; ++j) scale4[j] = scaled ? *reinterpret_cast<const f32x4*>(p.col_scale + col + 16 * j) : f32x4{1.f, 1.f, 1.f, 1.f};
    }
    const bool f32_out = p.out_kind == OUT_F32, resid = p.r_hi != nullptr;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        if constexpr (AGPR_ACC) {
#pragma unroll
            for (int j = 0; j < FN; ++j) asm volatile("" : "+a"(acc[i][j]));
        }
        const int row = row0 + 16 * i + mrow;
        if (row >= Meff) continue;
        const long long orow = p.cmap(row);
        if constexpr (SCALED) {
            if (scaled) {
#pragma unroll
                for (int j = 0; j < FN; ++j) acc[i][j] *= scale4[j];
            }
        }
        if (resid) {
            const long long ro = (p.r_index ? (long long)p.r_index[row] : p.rmap(row)) * (long long)p.ldr + col;
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const bf16x4 rh = *reinterpret_cast<const bf16x4*>(plane_ptr(p.r_hi, ro + 16 * j));
                const bf16x4 rl = *reinterpret_cast<const bf16x4*>(plane_ptr(p.r_lo, ro + 16 * j));
                acc[i][j] += bias4[j];
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][e] += join_bf16(rh[e], rl[e]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] += bias4[j];
        }
        if (f32_out) {
            // head-major: every 64 output columns are one head's block [rows][64].  A wave's columns start on a multiple of its own width:
            // 64 or 128 wide (gemm_pp.hip) they begin a head and fragment j lies (j >> 2) heads further; 32 wide (gemm_ppw.hip) they are
            // the lower or the upper half of ONE head -- (hc & 63) places them (without it the odd waves of gemm_ppw overwrote the even
            // ones' half).  Either way every fragment is an immediate offset from one pointer per row.
            const int hc = p.hm_col0 + col0;
            float* dst = p.hm_rows ? p.c_f32 + ((long long)(hc >> 6) * p.hm_rows + orow) * 64 + (hc & 63) + nq * 4
                                   : p.c_f32 + orow * p.ldc + col;
            const long long head_step = p.hm_rows ? (long long)p.hm_rows * 64 - 64 : 0;     // on top of the 64 columns themselves
            static_assert(FN <= 2 || FN % 4 == 0, "a wave covers half a head, one head or whole heads");
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                *reinterpret_cast<f32x4*>(dst + 16 * j + (j >> 2) * head_step) = apply_act4<ACT>(acc[i][j]);
            }
        } else if (p.out_kind == OUT_H3) {
            // h3 operand planes (common.h), row-major: the same pairing of column fragments as the bf16 planes -- after the exchange a lane
            // owns 8 consecutive columns: one 16-byte store of fp16 and one 8-byte store of e4m3 residuals per fragment pair
            static_assert(FN % 2 == 0, "plane stores pair up column fragments");
            const long long lrow = orow * p.ldh + col0 + 16 * (nq & 1) + 4 * (nq & 2);
            f16* dh = p.c_h16 + lrow;
            unsigned char* dl = p.c_l8 + lrow;
#pragma unroll
            for (int j = 0; j < FN; j += 2) {
                f16x4 h[2];
                unsigned l[2];
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const float v[4] = {apply_act(acc[i][j + jj][0], ACT), apply_act(acc[i][j + jj][1], ACT), apply_act(acc[i][j + jj][2], ACT),
                                        apply_act(acc[i][j + jj][3], ACT)};
                    split_h3(v, h[jj], l[jj]);
                }
                const pp_u32x2 x = __builtin_bit_cast(pp_u32x2, h[0]), y = __builtin_bit_cast(pp_u32x2, h[1]);
                const auto r0 = __builtin_amdgcn_permlane16_swap(x[0], y[0], false, false);
                const auto r1 = __builtin_amdgcn_permlane16_swap(x[1], y[1], false, false);
                *reinterpret_cast<u32x4*>(dh + 16 * j) = u32x4{r0[0], r1[0], r0[1], r1[1]};
                const auto rl = __builtin_amdgcn_permlane16_swap(l[0], l[1], false, false);
                *reinterpret_cast<pp_u32x2*>(dl + 16 * j) = pp_u32x2{rl[0], rl[1]};
            }
        } else if (p.out_kind == OUT_F8) {
            unsigned char* d8 = p.c_f8 + orow * p.ldf8 + col;
#pragma unroll
            for (int j = 0; j < FN; ++j)
            {
                const f32x4 g = apply_act4<ACT>(acc[i][j]);
                *reinterpret_cast<unsigned*>(d8 + 16 * j) = pack4_f8(g[0], g[1], g[2], g[3]);
            }
        } else {
            static_assert(FN % 2 == 0, "plane stores pair up column fragments");
            bf16* dh = pp_plane_lane_row(p.c_hi, orow * p.ldp, col0, nq);     // ldp % 32 == 0, col0 % 32 == 0
            bf16* dl = dh + (p.c_lo - p.c_hi);
            // note: every lane of the wave takes part in the exchange; rows past the live count were skipped above as whole 16-lane
            // groups of identical mrow across the four lane groups, so the partner lane (same mrow) is always present
#pragma unroll
            for (int j = 0; j < FN; j += 2) {
                bf16x4 h[2], l[2];
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const f32x4 g = apply_act4<ACT>(acc[i][j + jj]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { bf16 a, c2; split_bf16(g[e], a, c2); h[jj][e] = a; l[jj][e] = c2; }
                }
                pp_store_plane_pair(dh, j, h[0], h[1]);
                pp_store_plane_pair(dl, j, l[0], l[1]);
            }
        }
    }
}

