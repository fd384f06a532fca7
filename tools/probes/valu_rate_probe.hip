// Issue cost of the VALU instructions the fused QKV + attention epilogue leans on (gfx950), one wave per SIMD and two waves per SIMD:
// v_cvt_pk_bf16_f32, v_permlane16_swap_b32, v_pk_add_f32, v_add_f32, v_lshlrev_b32, v_exp_f32, ds_write_b128, ds_read_b64_tr_b16.
// Each test = 8 independent streams x REP instructions, timed with s_memtime by lane 0 of wave 0.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 tools/probes/valu_rate_probe.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 64
template <int OP>
__global__ __launch_bounds__(512) void k(unsigned long long* out, float seed) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x * 0.001f + i; b[i] = seed * 0.5f + i; }
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = i;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            if (OP == 1) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            if (OP == 2) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a[i]), "+v"(b[i]));
            if (OP == 3) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(a[i]));
            if (OP == 4) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            if (OP == 5) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            if (OP == 6) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[i]), "+v"(b[i]));
            if (OP == 7) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i]));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + b[i];
    if (s == 12345.678f) out[100] = 1;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

// packed f32 add on register pairs
__global__ __launch_bounds__(512) void k_pk(unsigned long long* out, float seed) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = f2{seed + i, seed - i}; b[i] = f2{seed * 0.5f, 1.0f}; }
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int r = 0; r < REP; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
    if (s == 12345.678f) out[100] = 1;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

// LDS: 16-byte stores (row stride 784 B, the staging layout) and transposing 8-byte reads
template <int OP>
__global__ __launch_bounds__(512) void k_lds(unsigned long long* out, float seed) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[140 * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef short s4 __attribute__((ext_vector_type(4)));
    f4 v = {seed, seed + 1, seed + 2, seed + 3};
    unsigned char* wp = lds + ((wave & 7) * 16 + (lane & 15)) * 784 + (lane >> 4) * 16;
    const unsigned char* rp = lds + ((wave & 7) * 4 + (lane >> 4) * 4 + ((lane & 15) >> 2)) * 784 + 512 + (lane & 3) * 8;
    s4 acc = {0, 0, 0, 0};
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int r = 0; r < REP; ++r) {
        if (OP == 0) { *reinterpret_cast<f4*>(wp + (r & 7) * 64) = v; }
        if (OP == 1) { acc += __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(rp + (r & 3) * 32)); }
        if (OP == 2) { const f4 x = *reinterpret_cast<const f4*>(wp + (r & 7) * 64); v += x; }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345 || v[0] + v[1] == 54321.f) out[100] = 1;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

int main() {
    unsigned long long* d;
    hipMalloc(&d, 8 * 1024);
    auto run = [&](const char* name, auto launch, int per_rep) {
        for (int threads : {256, 512}) {
            launch(threads);
            hipDeviceSynchronize();
            launch(threads);
            std::vector<unsigned long long> h(256);
            hipMemcpy(h.data(), d, 256 * 8, hipMemcpyDeviceToHost);
            printf("%-28s %d waves/SIMD: %6.2f cycles per wave-instruction (wave 0's view, %d instr)\n", name, threads / 256, (double)h[0] / (REP * per_rep), REP * per_rep);
        }
    };
    run("v_add_f32", [&](int t) { hipLaunchKernelGGL(k<0>, dim3(256), dim3(t), 0, 0, d, 1.0f); }, 8);
    run("v_cvt_pk_bf16_f32", [&](int t) { hipLaunchKernelGGL(k<1>, dim3(256), dim3(t), 0, 0, d, 1.0f); }, 8);
    run("v_permlane16_swap_b32", [&](int t) { hipLaunchKernelGGL(k<2>, dim3(256), dim3(t), 0, 0, d, 1.0f); }, 8);
    run("v_permlane32_swap_b32", [&](int t) { hipLaunchKernelGGL(k<6>, dim3(256), dim3(t), 0, 0, d, 1.0f); }, 8);
    run("v_lshlrev_b32", [&](int t) { hipLaunchKernelGGL(k<3>, dim3(256), dim3(t), 0, 0, d, 1.0f); }, 8);
    run("v_exp_f32", [&](int t) { hipLaunchKernelGGL(k<4>, dim3(256), dim3(t), 0, 0, d, 1.0f); }, 8);
    run("v_sub_f32", [&](int t) { hipLaunchKernelGGL(k<5>, dim3(256), dim3(t), 0, 0, d, 1.0f); }, 8);
    run("v_cndmask_b32", [&](int t) { hipLaunchKernelGGL(k<7>, dim3(256), dim3(t), 0, 0, d, 1.0f); }, 8);
    run("v_pk_add_f32", [&](int t) { hipLaunchKernelGGL(k_pk, dim3(256), dim3(t), 0, 0, d, 1.0f); }, 8);
    run("ds_write_b128 (stride 784)", [&](int t) { hipLaunchKernelGGL(k_lds<0>, dim3(256), dim3(t), 0, 0, d, 1.0f); }, 1);
    run("ds_read_b64_tr_b16", [&](int t) { hipLaunchKernelGGL(k_lds<1>, dim3(256), dim3(t), 0, 0, d, 1.0f); }, 1);
    run("ds_read_b128 (stride 784)", [&](int t) { hipLaunchKernelGGL(k_lds<2>, dim3(256), dim3(t), 0, 0, d, 1.0f); }, 1);
    return 0;
}
