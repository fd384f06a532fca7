// lab probe: what the matrix pipe sustains on the WHOLE chip from registers alone (no LDS, no memory in the loop), by MFMA shape
// and operand data -- separates "the kernel starves the pipe" from "the chip's power / clock limit".
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_power_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
// Every wave keeps 8 A fragments + 4 B fragments and 128 accumulator registers (the ping-pong GEMM's wave tile, 128 x 64 outputs)
// and issues the same 524 288 flops per iteration as 32 x v_mfma_f32_16x16x32_bf16 or 16 x v_mfma_f32_32x32x16_bf16.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int SHAPE, int SLEEP = 0>
__global__ __launch_bounds__(512) void k_probe(const bf16x8* __restrict__ src, float* __restrict__ sink, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bf16x8 a[8], b[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = src[(wave * 12 + i) * 64 + lane];
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = src[(wave * 12 + 8 + j) * 64 + lane];
    float t = 0.f;
    if constexpr (SHAPE == 16) {
        f32x4 acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
            if constexpr (SLEEP > 0) __builtin_amdgcn_s_sleep(SLEEP);      // duty-cycle probe: 64 x SLEEP idle cycles per 32 MFMAs
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    } else {
        // 32x32x16: an A / B fragment is 32 rows x 16 k = 8 values per lane; 4 x 2 output tiles, two k-steps per iteration
        f32x16 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[ks * 2 + j], a[ks * 4 + i], acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) t += acc[i][j][e];
    }
    if (t == 12345.678f) sink[threadIdx.x] = t;
}

static float run(int shape, const bf16x8* src, float* sink, int grid, int iters) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {      // first launch warms the clocks
        (void)hipEventRecord(e0, 0);
        if (shape == 16) hipLaunchKernelGGL(k_probe<16>, dim3(grid), dim3(512), 0, 0, src, sink, iters);
        else hipLaunchKernelGGL(k_probe<32>, dim3(grid), dim3(512), 0, 0, src, sink, iters);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
    }
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    const size_t n = 8 * 12 * 64 * 8;      // bf16 values
    std::vector<unsigned short> h(n);
    bf16x8* src; float* sink;
    (void)hipMalloc((void**)&src, n * 2); (void)hipMalloc((void**)&sink, 4096);
    const int iters = 400000;              // ~0.2 s per launch at full rate
    const char* names[5] = {"zeros", "gaussian-like bf16 (|x| ~ 1, random mantissas)", "hi/lo mix: even fragments O(1), odd fragments O(2^-9)",
                            "hi/lo mix, lo mantissas cut to their top 3 bits", "hi/lo mix, lo fragments zero"};
    for (int data = 0; data < 5; ++data) {
        unsigned s = 12345u;
        for (size_t i = 0; i < n; ++i) {
            s = s * 1664525u + 1013904223u;
            const unsigned m = (s >> 9) & 0x7f, sg = (s >> 31) << 15, e = 125 + ((s >> 20) & 3);       // exponents 2^-2 .. 2^1
            const bool lo = data >= 2 && ((i / (64 * 8)) & 1);
            h[i] = data == 0 ? 0 : (unsigned short)(sg | ((lo ? e - 9 : e) << 7) | m);
            if (lo && data == 3) h[i] &= 0xfff0;
            if (lo && data == 4) h[i] = 0;
        }
        (void)hipMemcpy(src, h.data(), n * 2, hipMemcpyHostToDevice);
        for (int grid : {256, 128}) {
            for (int shape : {16, 32}) {
                const float ms = run(shape, src, sink, grid, iters);
                const double fl = 524288.0 * iters * 8.0 * grid;
                printf("%-58s grid %3d  mfma %dx%d  %8.2f ms  %7.1f TFLOP/s\n", names[data], grid, shape, shape, ms, fl / (ms * 1e-3) / 1e12);
            }
        }
    }
    // duty cycle: the same random operands, every wave sleeping 64 x SLEEP cycles after each burst of 32 MFMAs (2 waves per SIMD:
    // a burst pair keeps the pipe busy for 1024 cycles).  rate / 2049 against the duty tells whether the clock comes back up.
    {
        unsigned s2 = 12345u;
        for (size_t i = 0; i < n; ++i) {
            s2 = s2 * 1664525u + 1013904223u;
            h[i] = (unsigned short)(((s2 >> 31) << 15) | ((125 + ((s2 >> 20) & 3)) << 7) | ((s2 >> 9) & 0x7f));
        }
        (void)hipMemcpy(src, h.data(), n * 2, hipMemcpyHostToDevice);
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        const int it2 = 100000;
        auto go = [&](auto kern, int sleep) {
            float ms = 0.f;
            for (int rep = 0; rep < 2; ++rep) {
                (void)hipEventRecord(e0, 0);
                hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, src, sink, it2);
                (void)hipEventRecord(e1, 0);
                (void)hipEventSynchronize(e1);
                (void)hipEventElapsedTime(&ms, e0, e1);
            }
            const double fl = 524288.0 * it2 * 8.0 * 256;
            printf("random bf16, grid 256, 16x16x32, s_sleep %2d after every 32 MFMAs: %8.2f ms  %7.1f TFLOP/s\n", sleep, ms, fl / (ms * 1e-3) / 1e12);
        };
        go(k_probe<16, 0>, 0); go(k_probe<16, 2>, 2); go(k_probe<16, 4>, 4); go(k_probe<16, 8>, 8); go(k_probe<16, 16>, 16); go(k_probe<16, 32>, 32);
    }
    return 0;
}
