// ds_read_b64_tr_b16 mapping probe (gfx950): which lane's ADDRESS feeds which lane's result element, with arbitrary per-lane addresses.
// Every lane points at its own 4-element (8-byte) quad of an LDS image holding lds[i] = i; lane l's address is quad perm(l) (a scrambled,
// non-linear assignment), so the result shows the source lane and sub-element of every output element.
// Build:  hipcc --offload-arch=gfx950 -O2 tools/probes/tr16_probe.hip -o gpurun_out/tr16_probe   Run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void k(const int* quad_of_lane, short* out) {
    __shared__ __attribute__((aligned(16))) short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int q = quad_of_lane[threadIdx.x];
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + 4 * q));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}
int main() {
    int h_q[64]; short h_o[256];
    // lane l -> quad 7 * l + 3 (distinct quads, stride 7): element value v tells quad v / 4 -> source lane (v / 4 - 3) / 7, sub-element v % 4
    for (int l = 0; l < 64; ++l) h_q[l] = 7 * l + 3;
    int* d_q; short* d_o;
    hipMalloc(&d_q, sizeof h_q); hipMalloc(&d_o, sizeof h_o);
    hipMemcpy(d_q, h_q, sizeof h_q, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_q, d_o);
    hipMemcpy(h_o, d_o, sizeof h_o, hipMemcpyDeviceToHost);
    int ok = 1;
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int j = 0; j < 4; ++j) {
            const int v = h_o[l * 4 + j], src = (v / 4 - 3) / 7, sub = v % 4;
            printf("  e%d<-lane %2d.%d", j, src, sub);
            // hypothesis: within the 16-lane group g, lane c = l & 15, element j comes from lane 16 g + 4 j + (c >> 2), sub-element c & 3
            if (src != (l & ~15) + 4 * j + ((l & 15) >> 2) || sub != (l & 3)) ok = 0;
        }
        printf("\n");
    }
    printf("hypothesis (e_j of lane c <- address of lane 4 j + c / 4 of the same 16-lane group, sub-element c %% 4): %s\n", ok ? "CONFIRMED" : "REFUTED");
    return 0;
}
