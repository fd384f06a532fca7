// LAB ONLY (libmmscore_lab.so, `make lab`): the first-generation GEMM kept for A/B runs -- not part of libmmscore.so.
//
// bf16 MFMA GEMM with split-plane activations and fused epilogues (gfx950).
//
// Covers every dense contraction of the three reference forwards:
// QKV / attention-output / FFN up / FFN down (pixelbert.py:767-788,960-985; modeling.py:326-328,
// 362-420), the 2048->768 box-feature projections (model_triple.py:192-194; pixelmodel.py:439-442;
// modeling.py:522), kdd_featureemb, kdd_conv1 (as im2col), poolers and logit_fc.0.
//
// Tiling: 128x128x64 per 256-thread workgroup (4 wavefronts as 2x2, 64x64 per wave = 4x4
// v_mfma_f32_16x16x32_bf16 accumulators).  Operand tiles are staged global -> VGPR -> LDS with an
// XOR swizzle (16-B chunk c of row r lives at chunk c ^ ((r>>1)&7)) that makes every ds_read_b128
// fragment read and every ds_write_b128 conflict-free for 128-B rows; the next K-tile's global loads
// are issued before the MFMA block so HBM/L2 latency hides under it.  In precision mode 2 the hi and
// lo activation planes are two MFMA passes sharing one weight fragment (2x MFMA, 1x weight traffic).
// Workgroup ids are remapped so each XCD (private L2) walks a contiguous range of tiles: the N/128
// column tiles that share one A row-panel hit the same L2.
#include <cstdlib>

#include "kernels.h"

#define BM 128
#define BN 128
#define BK 64
#define TILE_BYTES (128 * 128)

__device__ __forceinline__ int lds_off(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }

template <int NSPLIT, int ACT>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[(NSPLIT + 1) * TILE_BYTES];
    unsigned char* sA0 = smem;
    unsigned char* sA1 = smem + TILE_BYTES;          // only touched when NSPLIT == 2
    unsigned char* sB = smem + NSPLIT * TILE_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    const int nbn = p.N / BN, nbm = (p.M + BM - 1) / BM, nblk = nbm * nbn;
    int bid = blockIdx.x;
    {   // bijective XCD remap (block b runs on XCD b % 8; speed only, never correctness)
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, loc = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int bm = bid / nbn, bn = bid % nbn;

    // staging assignment: thread owns chunk column c of rows lr + 32*s
    const int c = tid & 7, lr = tid >> 3;
    const bf16* a_row[4];
    const bf16* w_row[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        int r = bm * BM + lr + 32 * s;
        r = r < p.M ? r : p.M - 1;
        a_row[s] = p.a_hi + (p.a_index ? (long long)p.a_index[r] : p.amap(r)) * (long long)p.lda + c * 8;
        w_row[s] = p.w + (long long)(bn * BN + lr + 32 * s) * p.K + c * 8;
    }
    const long long lo_delta = p.a_lo - p.a_hi;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / BK;
    const int fr = lane & 15, fk = lane >> 4;

    // staging registers (kept as plain unrolled locals: no lambdas / runtime indices -> no scratch)
    u32x4 ra0[4], ra1[4], rb[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        ra0[s] = *reinterpret_cast<const u32x4*>(a_row[s]);
        if (NSPLIT == 2) ra1[s] = *reinterpret_cast<const u32x4*>(a_row[s] + lo_delta);
        rb[s] = *reinterpret_cast<const u32x4*>(w_row[s]);
    }
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int o = lds_off(lr + 32 * s, c);
            *reinterpret_cast<u32x4*>(sA0 + o) = ra0[s];
            if (NSPLIT == 2) *reinterpret_cast<u32x4*>(sA1 + o) = ra1[s];
            *reinterpret_cast<u32x4*>(sB + o) = rb[s];
        }
        __syncthreads();
        {   // prefetch the next K-tile (the last iteration re-reads its own tile: branch-free)
            const int ko = (kt + 1 < nk ? kt + 1 : kt) * BK;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                ra0[s] = *reinterpret_cast<const u32x4*>(a_row[s] + ko);
                if (NSPLIT == 2) ra1[s] = *reinterpret_cast<const u32x4*>(a_row[s] + lo_delta + ko);
                rb[s] = *reinterpret_cast<const u32x4*>(w_row[s] + ko);
            }
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 a0[4], a1[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int o = lds_off(wm * 64 + i * 16 + fr, ks * 4 + fk);
                a0[i] = *reinterpret_cast<const bf16x8*>(sA0 + o);
                if (NSPLIT == 2) a1[i] = *reinterpret_cast<const bf16x8*>(sA1 + o);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                b[j] = *reinterpret_cast<const bf16x8*>(sB + lds_off(wn * 64 + j * 16 + fr, ks * 4 + fk));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[i], b[j], acc[i][j], 0, 0, 0);
                    if (NSPLIT == 2)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[i], b[j], acc[i][j], 0, 0, 0);
                }
        }
    }

    // epilogue: C/D layout col = lane & 15, row = (lane >> 4) * 4 + reg
    float bj[4];
    int colj[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        colj[j] = bn * BN + wn * 64 + j * 16 + fr;
        bj[j] = p.bias ? p.bias[colj[j]] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = bm * BM + wm * 64 + i * 16 + fk * 4 + r;
            if (row >= p.M) continue;
            const long long orow = p.cmap(row);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = acc[i][j][r] + bj[j];
                if (p.r_hi) {
                    const long long ro = (p.r_index ? (long long)p.r_index[row] : p.rmap(row)) * (long long)p.ldr + colj[j];
                    v += join_bf16(p.r_hi[ro], p.r_lo[ro]);
                }
                v = apply_act(v, ACT);
                if (p.out_kind == OUT_F32) {
                    p.c_f32[orow * p.ldc + colj[j]] = v;
                } else {
                    bf16 h, l;
                    split_bf16(v, h, l);
                    p.c_hi[orow * p.ldp + colj[j]] = h;
                    p.c_lo[orow * p.ldp + colj[j]] = l;
                }
            }
        }
}

template <int NSPLIT>
static void launch_ns(const GemmParams& p, int nblk, hipStream_t st) {
    switch (p.act) {
        case ACT_RELU: hipLaunchKernelGGL((gemm_kernel<NSPLIT, ACT_RELU>), dim3(nblk), dim3(256), 0, st, p); break;
        case ACT_GELU_TANH: hipLaunchKernelGGL((gemm_kernel<NSPLIT, ACT_GELU_TANH>), dim3(nblk), dim3(256), 0, st, p); break;
        case ACT_GELU_ERF: hipLaunchKernelGGL((gemm_kernel<NSPLIT, ACT_GELU_ERF>), dim3(nblk), dim3(256), 0, st, p); break;
        case ACT_TANH: hipLaunchKernelGGL((gemm_kernel<NSPLIT, ACT_TANH>), dim3(nblk), dim3(256), 0, st, p); break;
        default: hipLaunchKernelGGL((gemm_kernel<NSPLIT, ACT_NONE>), dim3(nblk), dim3(256), 0, st, p); break;
    }
}


// lab entry: the v0 128x128 register-staged kernel (no device-side row count)
void launch_gemm_v0(const GemmParams& p, int nsplit, hipStream_t st) {
    const int nblk = ((p.M + BM - 1) / BM) * (p.N / BN);
    if (nblk <= 0) return;
    if (nsplit == 2) launch_ns<2>(p, nblk, st);
    else launch_ns<1>(p, nblk, st);
}
