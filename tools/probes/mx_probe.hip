// Feasibility probe for the round-2 lever "fp16 high pass + MX low pass" (DESIGN.md section 6): operand layout of
// v_mfma_scale_f32_16x16x128_f8f6f4 (checked against a CPU product), and the issue rate of mixed bf16 / MX-scaled
// MFMA streams on all CUs.  Standalone: hipcc --offload-arch=gfx950 -O3 tools/probes/mx_probe.hip -o /tmp/mx_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// ---- layout check: one wave, C[16][16] = sum_k A[i][k] * B[j][k] * 2^(sa-127) * 2^(sb-127) ----
__global__ void k_layout(const v8i* a, const v8i* b, const int* sa, const int* sb, v4f* c) {
    v4f acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0, sa[threadIdx.x], 0, sb[threadIdx.x]);
    c[threadIdx.x] = acc;
}

static float fp8_e4m3(uint8_t v) {   // OCP e4m3fn
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float x = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.0f + m / 8.0f, e - 7);
    if (e == 15 && m == 7) x = NAN;
    return s ? -x : x;
}

// ---- rate kernels: MODE 0: 8 bf16 | 1: 4 bf16 + 1 MX(fmt) | 2: MX only (2 per iter) | 3: 4 bf16 only ----
template <int MODE, int FMT>
__global__ __launch_bounds__(512) void k_rate(const v8i* src, v4f* out, int iters) {
    const int t = threadIdx.x + blockIdx.x * blockDim.x;
    v8i a8 = src[t & 1023], b8 = src[(t + 17) & 1023];
    bf16x8 ah[4], bh[2];
    for (int i = 0; i < 4; ++i) { v8i x = src[(t + 31 * i + 3) & 1023]; ah[i] = *reinterpret_cast<bf16x8*>(&x); }
    for (int i = 0; i < 2; ++i) { v8i x = src[(t + 57 * i + 5) & 1023]; bh[i] = *reinterpret_cast<bf16x8*>(&x); }
    v4f acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = v4f{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i & 3], bh[i >> 2], acc[i], 0, 0, 0);
        } else if (MODE == 1) {          // 4 iterations of (4 bf16 + 1 MX) with static accumulator indices
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh[0], acc[i], 0, 0, 0);
                acc[4 + u] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, acc[4 + u], FMT, FMT, 0, 127, 0, 127);
            }
        } else if (MODE == 2) {          // 8 MX per iteration
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(u & 1 ? b8 : a8, u & 1 ? a8 : b8, acc[u], FMT, FMT, 0, 127, 0, 127);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh[0], acc[i], 0, 0, 0);
        }
    }
    v4f s = {0, 0, 0, 0};
    for (int i = 0; i < 8; ++i) s += acc[i];
    out[t] = s;
}

template <int MODE, int FMT>
static double rate(const v8i* src, v4f* out, int iters) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_rate<MODE, FMT>), dim3(256), dim3(512), 0, 0, src, out, 100);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_rate<MODE, FMT>), dim3(256), dim3(512), 0, 0, src, out, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e-3;
}

int main() {
    // ---- layout ----
    std::vector<uint8_t> A(16 * 128), B(16 * 128);
    srand(7);
    auto rnd8 = [&]() { uint8_t v; do { v = rand() & 255; } while (((v >> 3) & 15) == 15 || ((v >> 3) & 15) > 9); return v; };   // |x| <= 7.5
    for (auto& v : A) v = rnd8();
    for (auto& v : B) v = rnd8();
    std::vector<v8i> ra(64), rb(64);
    for (int l = 0; l < 64; ++l) {
        const int row = l & 15, kb = l >> 4;
        memcpy(&ra[l], &A[row * 128 + kb * 32], 32);
        memcpy(&rb[l], &B[row * 128 + kb * 32], 32);
    }
    v8i *da, *db; int *dsa, *dsb; v4f* dc;
    CK(hipMalloc(&da, 64 * 32)); CK(hipMalloc(&db, 64 * 32)); CK(hipMalloc(&dsa, 256)); CK(hipMalloc(&dsb, 256)); CK(hipMalloc(&dc, 64 * 16));
    CK(hipMemcpy(da, ra.data(), 64 * 32, hipMemcpyHostToDevice)); CK(hipMemcpy(db, rb.data(), 64 * 32, hipMemcpyHostToDevice));
    std::vector<float> C(64 * 4);
    auto run = [&](const std::vector<int>& sa, const std::vector<int>& sb) {
        CK(hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dc);
        CK(hipMemcpy(C.data(), dc, 64 * 16, hipMemcpyDeviceToHost));
    };
    // partial sums per (i, j, k-block) under the hypothesis "lane (row l&15, k-block l>>4) holds 32 consecutive k"
    auto part = [&](int i, int j, int kb) { double s = 0; for (int k = 0; k < 32; ++k) s += (double)fp8_e4m3(A[i * 128 + kb * 32 + k]) * fp8_e4m3(B[j * 128 + kb * 32 + k]); return s; };
    auto cval = [&](int i, int j) { return (double)C[(((i >> 2) << 4) | j) * 4 + (i & 3)]; };   // C: col = lane & 15, row = 4 * (lane >> 4) + reg
    {
        std::vector<int> one(64, 127);
        run(one, one);
        double maxerr = 0, maxref = 0;
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double r = 0; for (int kb = 0; kb < 4; ++kb) r += part(i, j, kb); maxerr = fmax(maxerr, fabs(r - cval(i, j))); maxref = fmax(maxref, fabs(r)); }
        printf("data layout (unit scales): lane = (row l&15, k-block l>>4), 32 consecutive fp8 per lane, C col = l&15 row = 4*(l>>4)+reg: max |err| %.3g of %.3g -> %s\n",
               maxerr, maxref, maxerr <= 1e-3 * maxref ? "OK" : "MISMATCH");
    }
    // which lane's scale byte applies to which 32-wide k-block of which row?  A = 1.0 inside ONE k-block (hypothesised layout), B = 1.0
    // everywhere, all scales 2^0 except one A lane at 2^3: C[i][*] = 32 x (scale applied to that block of row i)
    {
        std::vector<int> one(64, 127);
        for (int q = 0; q < 4; ++q) {
            for (int l = 0; l < 64; ++l) {
                uint8_t bytes[32];
                memset(bytes, (l >> 4) == q ? 0x38 : 0x00, 32);     // 0x38 = 1.0 in e4m3
                memcpy(&ra[l], bytes, 32);
                memset(bytes, 0x38, 32);
                memcpy(&rb[l], bytes, 32);
            }
            CK(hipMemcpy(da, ra.data(), 64 * 32, hipMemcpyHostToDevice)); CK(hipMemcpy(db, rb.data(), 64 * 32, hipMemcpyHostToDevice));
            for (int pg = 0; pg < 4; ++pg) {                 // scale x8 on all 16 lanes of lane group pg
                std::vector<int> sa(64, 127);
                for (int l = 16 * pg; l < 16 * pg + 16; ++l) sa[l] = 130;
                run(sa, one);
                bool same = true;
                for (int i = 1; i < 16; ++i) same = same && cval(i, 0) == cval(0, 0);
                if (fabs(cval(0, 0) - 32.0) > 0.5)
                    printf("  A = 1.0 in the registers of lane group %d: scale x8 in lane group %d -> every row sums to %.0f = 32 x %.2f  (%d of those 32 elements scaled)%s\n",
                           q, pg, cval(0, 0), cval(0, 0) / 32.0, (int)lround((cval(0, 0) - 32.0) / 7.0), same ? "" : "  [rows differ]");
            }
        }
        printf("  => element j (0..31) of lane (row = l&15, g = l>>4) is k = 64*(j>>4) + 16*g + (j&15); the scale byte of lane (row, b) applies to k in [32b, 32b+32)\n");
        // which byte does op_sel = 0 read?  put the factor in byte 1..3 of lane 0's scale register instead
        for (int byte = 1; byte < 4; ++byte) {
            std::vector<int> sa(64, 127 | (127 << 8) | (127 << 16) | (127 << 24));
            sa[0] = (sa[0] & ~(0xff << (8 * byte))) | (130 << (8 * byte));
            run(sa, one);
            printf("  factor in byte %d of lane 0's scale register (op_sel 0): row 0 sums to %.0f\n", byte, cval(0, 0));
        }
    }

    // ---- rates ----
    v8i* src; v4f* out;
    std::vector<uint32_t> h(1024 * 8);
    for (auto& v : h) { uint32_t x = 0; for (int b4 = 0; b4 < 4; ++b4) x |= (uint32_t)rnd8() << (8 * b4); v = (x & 0x7fff7fffu) % 0x40004000u | 0x38003800u; }   // sane bf16 AND fp8 values
    CK(hipMalloc(&src, h.size() * 4)); CK(hipMalloc(&out, 256 * 512 * 16));
    CK(hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    const int iters = 200000;
    const double waves = 256.0 * 8;
    const double f_bf = 2.0 * 16 * 16 * 32, f_mx = 2.0 * 16 * 16 * 128;
    double t;
    t = rate<0, 0>(src, out, iters); printf("8 x bf16 16x16x32 per iter            : %.3f s  %7.0f TFLOP/s issued  (%.1f cycles/iter/wave-pair at 2.4 GHz)\n", t, waves * iters * 8 * f_bf / t / 1e12, t / iters * 2.4e9 / 2);
    t = rate<3, 0>(src, out, iters); printf("4 x bf16 per iter                     : %.3f s  %7.0f TFLOP/s issued\n", t, waves * iters * 4 * f_bf / t / 1e12);
    const double t8 = rate<0, 0>(src, out, iters);
    t = rate<1, 0>(src, out, iters / 4); printf("4 x bf16 + 1 x MX-fp8 (K=128), same two-pass K=128 work as 8 x bf16: %.3f s -> %.2fx the 8 x bf16 time\n", t, t / t8);
    t = rate<1, 2>(src, out, iters / 4); printf("4 x bf16 + 1 x MX-fp6 (K=128)                                      : %.3f s -> %.2fx\n", t, t / t8);
    t = rate<1, 4>(src, out, iters / 4); printf("4 x bf16 + 1 x MX-fp4 (K=128)                                      : %.3f s -> %.2fx\n", t, t / t8);
    t = rate<2, 0>(src, out, iters / 4); printf("MX-fp8 only : %.3f s  %7.0f TFLOP/s issued\n", t, waves * (iters / 4) * 8 * f_mx / t / 1e12);
    t = rate<2, 2>(src, out, iters / 4); printf("MX-fp6 only : %.3f s  %7.0f TFLOP/s issued\n", t, waves * (iters / 4) * 8 * f_mx / t / 1e12);
    t = rate<2, 4>(src, out, iters / 4); printf("MX-fp4 only : %.3f s  %7.0f TFLOP/s issued\n", t, waves * (iters / 4) * 8 * f_mx / t / 1e12);
    return 0;
}
