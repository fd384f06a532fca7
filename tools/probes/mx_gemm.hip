// PROTOTYPE (not on the product path): the gemm_pp.hip tile engine with the second pass made cheap.
//
//   C[M,N] = A16 * W16^T  +  2^-(SA+SB) * (A8 * W8^T)
//   A16 = fp16(a), A8 = e4m3((a - A16) * 2^SA)         activations: 11-bit high plane + 8-bit low plane (3 B / element)
//   W16 = fp16(w), W8 = e4m3(w * 2^SB)                  weights (bf16-exact w): fp16 copy + fp8 copy
//
// High pass: v_mfma_f32_16x16x32_f16 (the bf16 rate).  Low pass: v_mfma_scale_f32_16x16x128_f8f6f4 with constant e8m0
// scales -- one instruction per 128 k (layout + rates: tools/probes/mx_probe.hip, profiles/r01h_mx_probe.txt).
// Tile 256x256, 8 waves 2x4 (128x64 per wave), 32-wide high stages in a 3-slot LDS-DMA ring (32 KiB per slot) exactly as in
// gemm_pp.hip; the low operands of one 128-k super-stage (A8, W8: 256 rows x 128 B each = 64 KiB) sit in a single-buffered
// region next to the ring (96 + 64 = 160 KiB, the whole LDS): they are consumed during the LAST high stage of their
// super-stage (8 MX MFMAs per phase) and re-filled during the first stage of the next one.
//
// build + run (GPU box):  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I kddcup_2020_multimodalitiesrecall_2nd_place_amd/csrc \
//                               tools/probes/mx_gemm.hip -o /tmp/mx_gemm && /tmp/mx_gemm
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include "kernels.h"
#include "gemm_epilogue.h"

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int SA = 11, SB = 6;

struct MxParams {
    const _Float16* a16; const uint8_t* a8;   // [M][K]
    const _Float16* w16; const uint8_t* w8;   // [N][K]
    float* c;                                  // [M][N]
    int M, N, K;
    int use_lo;                                // 0: high pass only (timing reference)
};

__device__ __forceinline__ int sw64(int r) { return (4 - ((r >> 2) & 3)) & 3; }     // 64-B rows (high planes)
__device__ __forceinline__ int sw128(int r) { return (r >> 1) & 7; }                // 128-B rows (low planes)

__device__ __forceinline__ void xbarrier() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

template <int LO>
__global__ __launch_bounds__(512) void gemm_mx_kernel(const MxParams p) {
    constexpr int BM = 256, BN = 256, WAVES_N = 4, TM = 128, TN = 64, FM = 8, FN = 4;
    constexpr int PLANE = 256 * 64;             // high plane of a 32-k stage: 256 rows x 64 B
    constexpr int SLOT = 2 * PLANE;             // A16 + W16
    constexpr int RING = 3 * SLOT;              // 96 KiB
    constexpr int LPLANE = 256 * 128;           // low plane of a 128-k super-stage: 256 rows x 128 B
    __shared__ __attribute__((aligned(16))) unsigned char smem[RING + 2 * LPLANE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int nbn = p.N / BN, nbm = (p.M + BM - 1) / BM, nblk = nbm * nbn;
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, loc = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int bm = bid / nbn, bn = bid % nbn;
    const int K = p.K;

    // ---- LDS-DMA sources ----
    // high piece h (0/1) of an operand: rows h*128 + wave*16 + lane/4, 16-B chunk lane%4 of the 64-B stage row
    // (M % 256 == 0 in this prototype, so the pieces of an operand differ by a uniform row stride and one per-lane pointer each suffices)
    const int rh = wave * 16 + (lane >> 2), rl = wave * 8 + (lane >> 3);
    const _Float16* a_src = p.a16 + (long long)(bm * BM + rh) * K + ((lane & 3) ^ sw64(rh)) * 8;
    const _Float16* w_src = p.w16 + (long long)(bn * BN + rh) * K + ((lane & 3) ^ sw64(rh)) * 8;
    const uint8_t* a8_src = p.a8 + (long long)(bm * BM + rl) * K + ((lane & 7) ^ sw128(rl)) * 16;
    const uint8_t* w8_src = p.w8 + (long long)(bn * BN + rl) * K + ((lane & 7) ^ sw128(rl)) * 16;
    auto issue_hi = [&](int q, int st, int slot) {     // q: 0,1 = A halves, 2,3 = W halves
        const int o = q >> 1, h = q & 1;
        unsigned char* d = smem + slot * SLOT + o * PLANE + h * 8192 + wave * 1024;
        const _Float16* s = (o == 0 ? a_src : w_src) + (long long)h * 128 * K + st * 32;     // sw64(r + 128) == sw64(r)
        __builtin_amdgcn_global_load_lds((glb_void*)s, (lds_void*)d, 16, 0, 0);
    };
    auto issue_lo = [&](int q, int ss) {               // q: 0..3 = A8 quarters, 4..7 = W8 quarters; ss = super-stage
        const int o = q >> 2, h = q & 3;
        unsigned char* d = smem + RING + o * LPLANE + h * 8192 + wave * 1024;
        const uint8_t* s = (o == 0 ? a8_src : w8_src) + (long long)h * 64 * K + ss * 128;      // sw128(r + 64) == sw128(r)
        __builtin_amdgcn_global_load_lds((glb_void*)s, (lds_void*)d, 16, 0, 0);
    };

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fk = lane >> 4;
    const int laneA = (wm * TM + fr) * 64 + ((fk ^ sw64(fr)) << 4);
    const int laneB = PLANE + (wn * TN + fr) * 64 + ((fk ^ sw64(fr)) << 4);
    // low fragments: lane (row fr, g = fk) holds chunks g and 4+g of its 128-B row (k = 16g.. and 64+16g..)
    const int loA = RING + (wm * TM + fr) * 128, loB = RING + LPLANE + (wn * TN + fr) * 128;
    const int c0 = (fk ^ sw128(fr)) << 4, c1 = ((4 + fk) ^ sw128(fr)) << 4;      // sw128 depends on (r>>1)&7 = (fr>>1)&7: rows step by 16

    f16x8 a[4], b[4];
    v8i la[4], lb[4];
    auto read_a = [&](const unsigned char* sb, int mh) {
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const f16x8*>(sb + laneA + (mh * 64 + i * 16) * 64);
    };
    auto read_b = [&](const unsigned char* sb) {
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const f16x8*>(sb + laneB + j * 16 * 64);
    };
    auto read_lo = [&](int base, int rows_off, v8i& dst) {
        const v4i x = *reinterpret_cast<const v4i*>(smem + base + rows_off * 128 + c0);
        const v4i y = *reinterpret_cast<const v4i*>(smem + base + rows_off * 128 + c1);
        dst = v8i{x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
    };
    auto read_la = [&](int mh) {
#pragma unroll
        for (int i = 0; i < 4; ++i) read_lo(loA, mh * 64 + i * 16, la[i]);
    };
    auto read_lb = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j) read_lo(loB, j * 16, lb[j]);
    };
    // one phase = one 64x64 half of the wave's outputs: 16 MFMAs
    auto mma_hi = [&](int mh) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[mh * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[mh * 4 + i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    auto mma_lo = [&](int mh) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[mh * 4 + i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(la[i], lb[j], acc[mh * 4 + i][j], 0, 0, 0, 127 - SA, 0, 127 - SB);
        __builtin_amdgcn_s_setprio(0);
    };

    const int ns = K / 32;      // multiple of 4
    // One 32-k high stage = 2 phases (rows 0-63 / 64-127 of the wave tile x all 64 columns, 16 MFMAs each).  The last stage of a
    // 128-k super-stage is followed by 2 low phases (16 MX MFMAs each) whose fragments reuse the high fragments' registers.
    // LDS-DMA issue sits in phase 2 (>= 2 phases after the last read of the slot / of the low region); the counted wait follows it.
    auto stage = [&](auto ph_tag, int s, int slot) {
        constexpr int ph = decltype(ph_tag)::value;     // s % 4
        const unsigned char* sb = smem + slot * SLOT;
        const int nslot = slot == 0 ? 2 : slot - 1;
        const bool pre = s + 2 < ns;
        constexpr bool lo_use = LO && ph == 3;
        const bool lo_issue = LO && ph == 0 && s >= 4;
        // phase 1
        read_b(sb);
        read_a(sb, 0);
        xbarrier();
        mma_hi(0);
        xbarrier();
        // phase 2
        read_a(sb, 1);
        if (lo_issue) {
#pragma unroll
            for (int q = 0; q < 8; ++q) issue_lo(q, s >> 2);
        }
        if (pre) {
#pragma unroll
            for (int q = 0; q < 4; ++q) issue_hi(q, s + 2, nslot);
        }
        // pieces newer than stage s+1's: this stage's 4 high pieces (+ the 8 low pieces issued just before them)
        if (!pre) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (lo_issue) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        xbarrier();
        mma_hi(1);
        xbarrier();
        if (lo_use) {
            read_lb();
            read_la(0);
            xbarrier();
            mma_lo(0);
            xbarrier();
            read_la(1);
            xbarrier();
            mma_lo(1);
            xbarrier();
        }
    };

    // prologue: low operands of super-stage 0, then high stages 0 and 1; stage 0 (and everything older) landed
    if (LO) {
#pragma unroll
        for (int q = 0; q < 8; ++q) issue_lo(q, 0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) issue_hi(q, 0, 0);
#pragma unroll
    for (int q = 0; q < 4; ++q) issue_hi(q, 1, 1);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    xbarrier();
    if (wm == 1) xbarrier();

    int slot = 0;
    auto next = [&]() { slot = slot == 2 ? 0 : slot + 1; };
    for (int s = 0; s < ns; s += 4) {
        stage(std::integral_constant<int, 0>{}, s, slot); next();
        stage(std::integral_constant<int, 1>{}, s + 1, slot); next();
        stage(std::integral_constant<int, 2>{}, s + 2, slot); next();
        stage(std::integral_constant<int, 3>{}, s + 3, slot); next();
    }
    if (wm == 0) xbarrier();

    GemmParams g{};
    g.M = p.M; g.N = p.N; g.K = p.K; g.out_kind = OUT_F32; g.c_f32 = p.c; g.ldc = p.N; g.act = ACT_NONE;
    gemm_epilogue<ACT_NONE, BM, BN, TM, TN, FM, FN>(g, acc, smem, bm, bn, wm, wn, wave, lane, p.M);
}

// ---- host side ----
static uint8_t to_e4m3(float x) {      // round to nearest even, saturate to +-448
    if (x != x) return 0x7f;
    const uint8_t s = x < 0 ? 0x80 : 0;
    float a = fabsf(x);
    if (a >= 448.f) return s | 0x7e;
    if (a < ldexpf(1.f, -10)) return s;                 // below half the smallest subnormal (2^-9)
    int e;
    float m = frexpf(a, &e);                            // a = m * 2^e, m in [0.5, 1)
    int E = e - 1 + 7;                                  // biased exponent of 1.xxx * 2^(e-1)
    if (E <= 0) {                                       // subnormal: multiples of 2^-9
        const int q = (int)lrintf(a * 512.f);           // round to nearest even
        return s | (uint8_t)(q >= 8 ? 0x08 : q);
    }
    int q = (int)lrintf((m * 2.f - 1.f) * 8.f);         // 3 mantissa bits
    if (q == 8) { q = 0; ++E; }
    if (E > 15 || (E == 15 && q == 7)) return s | 0x7e;
    return s | (uint8_t)(E << 3) | (uint8_t)q;
}
static float from_e4m3(uint8_t v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    const float x = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.0f + m / 8.0f, e - 7);
    return s ? -x : x;
}
static uint64_t rng_state = 0x9e3779b97f4a7c15ull;
static float rnd_normal() {
    auto u = [&]() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (double)(rng_state >> 11) * (1.0 / 9007199254740992.0); };
    return (float)(sqrt(-2.0 * log(u() + 1e-300)) * cos(6.283185307179586 * u()));
}
static float to_bf16_exact(float x) { uint32_t u; memcpy(&u, &x, 4); u = (u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000u; memcpy(&x, &u, 4); return x; }

int main() {
    const int M = 122880;
    struct Shape { const char* name; int N, K; } shapes[] = {{"qkv", 2304, 768}, {"attout", 768, 768}, {"ffn_up", 3072, 768}, {"ffn_down", 768, 3072}};
    for (const Shape& sh : shapes) {
        const int N = sh.N, K = sh.K;
        const int Mh = 512;                                  // rows generated and checked on the host (device rows wrap around them)
        std::vector<float> a((size_t)Mh * K), w((size_t)N * K);
        for (auto& v : a) v = rnd_normal();
        for (auto& v : w) v = to_bf16_exact(rnd_normal() / sqrtf((float)K));
        std::vector<_Float16> a16((size_t)Mh * K), w16((size_t)N * K);
        std::vector<uint8_t> a8((size_t)Mh * K), w8((size_t)N * K);
        for (size_t i = 0; i < a.size(); ++i) { a16[i] = (_Float16)a[i]; a8[i] = to_e4m3((a[i] - (float)a16[i]) * ldexpf(1.f, SA)); }
        for (size_t i = 0; i < w.size(); ++i) { w16[i] = (_Float16)w[i]; w8[i] = to_e4m3(w[i] * ldexpf(1.f, SB)); }
        _Float16 *da16, *dw16; uint8_t *da8, *dw8; float* dc;
        CK(hipMalloc(&da16, (size_t)M * K * 2)); CK(hipMalloc(&da8, (size_t)M * K)); CK(hipMalloc(&dw16, (size_t)N * K * 2)); CK(hipMalloc(&dw8, (size_t)N * K));
        CK(hipMalloc(&dc, (size_t)M * N * 4));
        for (int r0 = 0; r0 < M; r0 += Mh) {                 // replicate the host rows over all M device rows (random data everywhere)
            CK(hipMemcpy(da16 + (size_t)r0 * K, a16.data(), (size_t)Mh * K * 2, hipMemcpyHostToDevice));
            CK(hipMemcpy(da8 + (size_t)r0 * K, a8.data(), (size_t)Mh * K, hipMemcpyHostToDevice));
        }
        CK(hipMemcpy(dw16, w16.data(), w16.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dw8, w8.data(), w8.size(), hipMemcpyHostToDevice));
        MxParams p{da16, da8, dw16, dw8, dc, M, N, K, 1};
        const int nblk = ((M + 255) / 256) * (N / 256);
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float ms[2] = {0, 0};
        for (int lo = 1; lo >= 0; --lo) {
            for (int rep = 0; rep < 12; ++rep) {
                if (rep == 2) CK(hipEventRecord(e0));
                if (lo) hipLaunchKernelGGL((gemm_mx_kernel<1>), dim3(nblk), dim3(512), 0, 0, p);
                else hipLaunchKernelGGL((gemm_mx_kernel<0>), dim3(nblk), dim3(512), 0, 0, p);
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms[lo], e0, e1)); ms[lo] /= 10;
            if (lo) {                                        // correctness of the two-operand product, rows 0..Mh-1 and the last tile's rows
                std::vector<float> c((size_t)Mh * N);
                CK(hipMemcpy(c.data(), dc + (size_t)(M - Mh) * N, c.size() * 4, hipMemcpyDeviceToHost));
                double e_exact = 0, e_true = 0, e_hi = 0, ref_max = 0;
                for (int i = 0; i < Mh; i += 7)
                    for (int j = 0; j < N; j += 13) {
                        double s_hi = 0, s_lo = 0, s_true = 0;
                        for (int k = 0; k < K; ++k) {
                            s_hi += (double)(float)a16[(size_t)i * K + k] * (double)(float)w16[(size_t)j * K + k];
                            s_lo += (double)from_e4m3(a8[(size_t)i * K + k]) * from_e4m3(w8[(size_t)j * K + k]);
                            s_true += (double)a[(size_t)i * K + k] * w[(size_t)j * K + k];
                        }
                        const double want = s_hi + ldexp(s_lo, -(SA + SB)), got = c[(size_t)i * N + j];
                        e_exact = fmax(e_exact, fabs(got - want)); e_true = fmax(e_true, fabs(got - s_true)); e_hi = fmax(e_hi, fabs(s_hi - s_true));
                        ref_max = fmax(ref_max, fabs(s_true));
                    }
                printf("%-8s N=%4d K=%4d  vs the specified two-operand product: %.2e   vs the fp64 product of the unsplit operands: %.2e  (high pass alone: %.2e)   [max |ref| %.2f]\n",
                       sh.name, N, K, e_exact / ref_max, e_true / ref_max, e_hi / ref_max, ref_max);
            }
        }
        const double fl = 2.0 * M * N * K;
        printf("%-8s fp16 + MX-fp8: %.3f ms = %4.0f TFLOP/s algorithmic   |   fp16 pass alone: %.3f ms = %4.0f TFLOP/s\n", sh.name, ms[1], fl / ms[1] / 1e9, ms[0], fl / ms[0] / 1e9);
        CK(hipFree(da16)); CK(hipFree(da8)); CK(hipFree(dw16)); CK(hipFree(dw8)); CK(hipFree(dc));
    }
    return 0;
}
