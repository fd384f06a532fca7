// Probe: how fast can ONE workgroup per CU write a 256x256 fp32 tile (256 KB, the gemm_pp epilogue's job), as a function of the
// store shape and cache-policy hint?  (profiles/r01c_gemm_variants.txt: the epilogue drains at ~12 B/clk per CU.)
// hipcc --offload-arch=gfx950 -O3 tools/probes/store_probe.hip -o /tmp/store_probe && /tmp/store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// MODE 0: epilogue shape: per instruction a wave writes 4 rows x 256 B (row stride = ldc floats)
// MODE 1: per instruction a wave writes 1 KB contiguous (one full 256-float row of the tile)
// HINT 0: plain, 1: nontemporal
template <int MODE, int HINT>
__global__ __launch_bounds__(512) void k_store(float* c, int ldc, int tiles_per_block, int rows_total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int t = 0; t < tiles_per_block; ++t) {
        const long long tile = (long long)blockIdx.x * tiles_per_block + t;
        const long long row0 = (tile * 256) % rows_total;
        f32x4 v = {(float)lane, (float)wave, (float)t, 1.0f};
#pragma unroll 8
        for (int i = 0; i < 32; ++i) {
            long long off;
            if (MODE == 0) { const int wm = wave >> 2, wn = wave & 3; off = (row0 + wm * 128 + (i >> 2) * 16 + (i & 3) * 4 + (lane >> 4)) * ldc + wn * 64 + (lane & 15) * 4; }
            else off = (row0 + wave * 32 + i) * ldc + lane * 4;
            f32x4* p = reinterpret_cast<f32x4*>(c + off);
            if (HINT) __builtin_nontemporal_store(v, p); else *p = v;
            v[0] += 1.0f;
        }
    }
}

template <int MODE, int HINT>
static void run(const char* name, float* c, int ldc, int blocks, int tiles, int rows_total) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_store<MODE, HINT>), dim3(blocks), dim3(512), 0, 0, c, ldc, tiles, rows_total);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_store<MODE, HINT>), dim3(blocks), dim3(512), 0, 0, c, ldc, tiles, rows_total);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)blocks * tiles * 256 * 1024;
    printf("%-44s blocks %3d x %2d tiles: %7.1f us  %6.2f TB/s aggregate  %5.1f GB/s per CU  %5.1f us per 256 KB tile\n", name, blocks, tiles, ms * 1e3, bytes / ms / 1e9,
           bytes / blocks / ms / 1e6, ms * 1e3 / tiles);
}

int main() {
    const int ldc = 2304, rows = 256 * 1024;     // 2.4 GB buffer, rows wrap
    float* c; CK(hipMalloc(&c, (size_t)rows * ldc * 4));
    CK(hipMemset(c, 0, (size_t)rows * ldc * 4));
    for (int blocks : {32, 256}) {
        run<0, 0>("epilogue shape (4 rows x 256 B / instr)", c, ldc, blocks, 16, rows);
        run<0, 1>("epilogue shape, nontemporal", c, ldc, blocks, 16, rows);
        run<1, 0>("1 KB contiguous / instr", c, ldc, blocks, 16, rows);
        run<1, 1>("1 KB contiguous / instr, nontemporal", c, ldc, blocks, 16, rows);
    }
    return 0;
}
