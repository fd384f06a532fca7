// Row-wise (HBM-bound) kernels: embedding gathers + LayerNorm, feature splitting, box/label token
// assembly, masks and match heads.  One wavefront owns one 768-wide row: lane l covers columns
// t*256 + 4*l .. +3 (t = 0..2), i.e. three coalesced 16-B accesses per lane per fp32 row and three
// 8-B bf16x4 accesses per plane.  All arithmetic fp32; LayerNorm is two-pass (mean, then centred
// variance) with biased variance and eps 1e-12 inside the square root, exactly as
// tf.contrib.layers.layer_norm (pixelbert.py:414-417) and torch.nn.LayerNorm (modeling.py:266).
#include "kernels.h"

#define ROWS_PER_BLOCK 4

struct Row { float v[12]; };

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_row() { return blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6); }

__device__ __forceinline__ void row_zero(Row& x) {
#pragma unroll
    for (int i = 0; i < 12; ++i) x.v[i] = 0.f;
}
__device__ __forceinline__ void row_load(Row& x, const float* p) {
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const float4 f = *reinterpret_cast<const float4*>(p + t * 256 + lane_id() * 4);
        x.v[t * 4 + 0] = f.x; x.v[t * 4 + 1] = f.y; x.v[t * 4 + 2] = f.z; x.v[t * 4 + 3] = f.w;
    }
}
__device__ __forceinline__ void row_axpy(Row& x, float a, const float* p) {
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const float4 f = *reinterpret_cast<const float4*>(p + t * 256 + lane_id() * 4);
        x.v[t * 4 + 0] += a * f.x; x.v[t * 4 + 1] += a * f.y; x.v[t * 4 + 2] += a * f.z; x.v[t * 4 + 3] += a * f.w;
    }
}
__device__ __forceinline__ void row_add(Row& x, const float* p) {
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const float4 f = *reinterpret_cast<const float4*>(p + t * 256 + lane_id() * 4);
        x.v[t * 4 + 0] += f.x; x.v[t * 4 + 1] += f.y; x.v[t * 4 + 2] += f.z; x.v[t * 4 + 3] += f.w;
    }
}
__device__ __forceinline__ void row_store_f32(const Row& x, float* p) {
#pragma unroll
    for (int t = 0; t < 3; ++t)
        *reinterpret_cast<float4*>(p + t * 256 + lane_id() * 4) =
            make_float4(x.v[t * 4], x.v[t * 4 + 1], x.v[t * 4 + 2], x.v[t * 4 + 3]);
}
__device__ __forceinline__ void row_store_planes(const Row& x, bf16* hi, bf16* lo) {
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        bf16x4 h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            bf16 a, c;
            split_bf16(x.v[t * 4 + e], a, c);
            h[e] = a; l[e] = c;
        }
        *reinterpret_cast<bf16x4*>(plane_ptr(hi, t * 256 + lane_id() * 4)) = h;     // hi / lo: row pointers (plane_ptr of a 32-aligned index)
        *reinterpret_cast<bf16x4*>(plane_ptr(lo, t * 256 + lane_id() * 4)) = l;
    }
}
__device__ __forceinline__ void row_ln(Row& x, const float* gamma, const float* beta) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) s += x.v[i];
    const float mean = wave_sum(s) * (1.0f / MMS_HIDDEN);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) { const float d = x.v[i] - mean; q += d * d; }
    const float var = wave_sum(q) * (1.0f / MMS_HIDDEN);
    const float inv = 1.0f / sqrtf(var + MMS_LN_EPS);
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const float4 g = *reinterpret_cast<const float4*>(gamma + t * 256 + lane_id() * 4);
        const float4 b = *reinterpret_cast<const float4*>(beta + t * 256 + lane_id() * 4);
        x.v[t * 4 + 0] = (x.v[t * 4 + 0] - mean) * inv * g.x + b.x;
        x.v[t * 4 + 1] = (x.v[t * 4 + 1] - mean) * inv * g.y + b.y;
        x.v[t * 4 + 2] = (x.v[t * 4 + 2] - mean) * inv * g.z + b.z;
        x.v[t * 4 + 3] = (x.v[t * 4 + 3] - mean) * inv * g.w + b.w;
    }
}
__device__ __forceinline__ void row_store_f8(const Row& x, unsigned char* p) {   // e4m3 bytes, 4 per lane per 256-column third
#pragma unroll
    for (int t = 0; t < 3; ++t)
        *reinterpret_cast<unsigned*>(p + t * 256 + lane_id() * 4) = pack4_f8(x.v[t * 4], x.v[t * 4 + 1], x.v[t * 4 + 2], x.v[t * 4 + 3]);
}
__device__ __forceinline__ long long clamp_id(long long id, int vocab) {
    return id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
}

static inline dim3 row_grid(long long rows) { return dim3((unsigned)((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK)); }

// ------------------------------------------------------------------------------------------------
// generic
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void row_add_planes(Row& x, const bf16* hi, const bf16* lo) {
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const bf16x4 h = *reinterpret_cast<const bf16x4*>(plane_ptr(hi, t * 256 + lane_id() * 4));
        const bf16x4 l = *reinterpret_cast<const bf16x4*>(plane_ptr(lo, t * 256 + lane_id() * 4));
#pragma unroll
        for (int e = 0; e < 4; ++e) x.v[t * 4 + e] += join_bf16(h[e], l[e]);
    }
}
// out = LN(in (+ residual row)).  The residual add sits here rather than in the GEMM epilogue: this kernel streams whole
// rows at full HBM rate, the epilogue's per-strip residual reads were a latency chain (profiles/r01c_gemm_variants.txt).
// All of a row's reads precede its writes, so out may alias the residual planes row for row (ffn_block: x -> x).
__global__ __launch_bounds__(256) void k_ln_to_planes(const float* in, int ld, const float* gamma,
                                                      const float* beta, bf16* o_hi, bf16* o_lo, int ldo, int M,
                                                      const int* m_dev, LnResid res) {
    int row = wave_row();
    if (res.skip && *res.skip == 1) return;       // the producing GEMM ran its fused LayerNorm epilogue
    const int lim = m_dev ? min(M, *m_dev) : M;
    if (row >= lim) return;
    if (res.reverse) row = lim - 1 - row;
    Row x;
    row_load(x, in + (long long)row * ld);
    if (res.nparts > 1) {      // split-K producer: the row is the sum of its partials (fixed order) + bias
        for (int s = 1; s < res.nparts; ++s) row_add(x, in + s * res.part_stride + (long long)row * ld);
        if (res.bias) row_add(x, res.bias);
    }
    if (res.hi) {
        const long long ro = (res.r_index ? (long long)res.r_index[row] : res.rmap(row)) * (long long)res.ld;
        row_add_planes(x, plane_ptr(res.hi, ro), plane_ptr(res.lo, ro));
    }
    row_ln(x, gamma, beta);
    row_store_planes(x, plane_ptr(o_hi, (long long)row * ldo), plane_ptr(o_lo, (long long)row * ldo));
    if (res.o_f8) row_store_f8(x, res.o_f8 + (long long)row * ldo);
}
void launch_ln_to_planes(const float* in, int ld, const float* gamma, const float* beta, bf16* o_hi,
                         bf16* o_lo, int ldo, int M, hipStream_t st, const int* m_dev, LnResid res) {
    if (M > 0) hipLaunchKernelGGL(k_ln_to_planes, row_grid(M), dim3(256), 0, st, in, ld, gamma, beta, o_hi, o_lo, ldo, M, m_dev, res);
}

// Reduce + epilogue of a split-K GEMM launch whose consumer is NOT a LayerNorm (QKV, FFN-up at tiny M: api.hip gemm()): out = act(sum of the
// S fp32 partials (fixed order) + bias), written as plain / head-major fp32 (the [Q | K | V] blocks of GemmParams::hm_rows) or as split planes.
__global__ __launch_bounds__(256) void k_splitk_reduce(const float* parts, int S, long long stride, int M, int N, const int* m_dev, const float* bias, int act,
                                                       float* c_f32, int ldc, int hm_rows, int hm_col0, bf16* c_hi, bf16* c_lo, int ldp) {
    const int lim = m_dev ? min(M, *m_dev) : M;
    const long long n4 = (long long)lim * (N / 4);
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const int row = (int)(i / (N / 4)), col = (int)(i % (N / 4)) * 4;
        float4 v = *reinterpret_cast<const float4*>(parts + (long long)row * N + col);
        for (int s = 1; s < S; ++s) {
            const float4 u = *reinterpret_cast<const float4*>(parts + s * stride + (long long)row * N + col);
            v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
        if (bias) { const float4 b = *reinterpret_cast<const float4*>(bias + col); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
        const float y[4] = {apply_act(v.x, act), apply_act(v.y, act), apply_act(v.z, act), apply_act(v.w, act)};
        if (c_f32) {
            const int hc = hm_col0 + col;
            float* dst = hm_rows ? c_f32 + ((long long)(hc >> 6) * hm_rows + row) * 64 + (hc & 63) : c_f32 + (long long)row * ldc + col;
            *reinterpret_cast<float4*>(dst) = make_float4(y[0], y[1], y[2], y[3]);
        } else {
            bf16x4 h, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) { bf16 a, c; split_bf16(y[e], a, c); h[e] = a; l[e] = c; }
            const long long off = (long long)row * ldp + col;
            *reinterpret_cast<bf16x4*>(plane_ptr(c_hi, off)) = h;
            *reinterpret_cast<bf16x4*>(plane_ptr(c_lo, off)) = l;
        }
    }
}
void launch_splitk_reduce(const float* parts, int S, long long stride, int M, int N, const int* m_dev, const float* bias, int act, float* c_f32, int ldc,
                          int hm_rows, int hm_col0, bf16* c_hi, bf16* c_lo, int ldp, hipStream_t st) {
    const long long n4 = (long long)M * (N / 4);
    if (n4 <= 0) return;
    const long long blocks = (n4 + 255) / 256;
    hipLaunchKernelGGL(k_splitk_reduce, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st, parts, S, stride, M, N, m_dev, bias, act, c_f32, ldc,
                       hm_rows, hm_col0, c_hi, c_lo, ldp);
}

__global__ __launch_bounds__(256) void k_ln_f32(const float* in, const float* gamma, const float* beta,
                                                float* out, int M) {
    const int row = wave_row();
    if (row >= M) return;
    Row x;
    row_load(x, in + (long long)row * MMS_HIDDEN);
    row_ln(x, gamma, beta);
    row_store_f32(x, out + (long long)row * MMS_HIDDEN);
}
void launch_ln_f32(const float* in, const float* gamma, const float* beta, float* out, int M, hipStream_t st) {
    if (M > 0) hipLaunchKernelGGL(k_ln_f32, row_grid(M), dim3(256), 0, st, in, gamma, beta, out, M);
}

__global__ __launch_bounds__(256) void k_split_f32(const float* in, bf16* o_hi, bf16* o_lo, long long n4) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 f = reinterpret_cast<const float4*>(in)[i];
        const float v[4] = {f.x, f.y, f.z, f.w};
        bf16x4 h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) { bf16 a, c; split_bf16(v[e], a, c); h[e] = a; l[e] = c; }
        *reinterpret_cast<bf16x4*>(plane_ptr(o_hi, i * 4)) = h;
        *reinterpret_cast<bf16x4*>(plane_ptr(o_lo, i * 4)) = l;
    }
}
void launch_split_f32(const float* in, bf16* o_hi, bf16* o_lo, long long n, hipStream_t st) {
    const long long n4 = n / 4;
    if (n4 <= 0) return;
    const long long blocks = (n4 + 255) / 256;
    hipLaunchKernelGGL(k_split_f32, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, st, in, o_hi, o_lo, n4);
}

// fp32 W[N][K] -> bf16 tiles (wtile_off, common.h): hi = bf16(w) and, when o_lo != nullptr, lo = bf16(w - hi) as a second tiled matrix
__global__ __launch_bounds__(256) void k_tile_weights(const float* in, bf16* o_hi, bf16* o_lo, long long N, long long K) {
    const long long n4 = N * K / 4;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 f = reinterpret_cast<const float4*>(in)[i];
        const float v[4] = {f.x, f.y, f.z, f.w};
        bf16x4 h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) { bf16 a, c; split_bf16(v[e], a, c); h[e] = a; l[e] = c; }
        const long long at = wtile_off(i * 4 / K, i * 4 % K, K);
        *reinterpret_cast<bf16x4*>(o_hi + at) = h;
        if (o_lo) *reinterpret_cast<bf16x4*>(o_lo + at) = l;
    }
}
void launch_tile_weights(const float* in, bf16* o_hi, bf16* o_lo, long long N, long long K, hipStream_t st) {
    const long long n4 = N * K / 4;
    if (n4 <= 0) return;
    const long long blocks = (n4 + 255) / 256;
    hipLaunchKernelGGL(k_tile_weights, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, st, in, o_hi, o_lo, N, K);
}

__global__ __launch_bounds__(256) void k_planes_to_f32(const bf16* hi, const bf16* lo, float* out, long long n4) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const bf16x4 h = *reinterpret_cast<const bf16x4*>(plane_ptr(hi, i * 4));
        const bf16x4 l = *reinterpret_cast<const bf16x4*>(plane_ptr(lo, i * 4));
        reinterpret_cast<float4*>(out)[i] = make_float4(join_bf16(h[0], l[0]), join_bf16(h[1], l[1]),
                                                        join_bf16(h[2], l[2]), join_bf16(h[3], l[3]));
    }
}
void launch_planes_to_f32(const bf16* hi, const bf16* lo, float* out, long long n, hipStream_t st) {
    const long long n4 = n / 4;
    if (n4 <= 0) return;
    const long long blocks = (n4 + 255) / 256;
    hipLaunchKernelGGL(k_planes_to_f32, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, st, hi, lo, out, n4);
}

// ---- precision mode 4: e4m3 operand bytes ----
__global__ __launch_bounds__(256) void k_planes_to_f8(const bf16* hi, const bf16* lo, unsigned* out, long long n4) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const bf16x4 h = *reinterpret_cast<const bf16x4*>(plane_ptr(hi, i * 4));
        const bf16x4 l = *reinterpret_cast<const bf16x4*>(plane_ptr(lo, i * 4));
        out[i] = pack4_f8(join_bf16(h[0], l[0]), join_bf16(h[1], l[1]), join_bf16(h[2], l[2]), join_bf16(h[3], l[3]));
    }
}
void launch_planes_to_f8(const bf16* hi, const bf16* lo, unsigned char* out, long long n, hipStream_t st) {
    const long long n4 = n / 4;
    if (n4 > 0) hipLaunchKernelGGL(k_planes_to_f8, dim3((unsigned)((n4 + 255) / 256 > 16384 ? 16384 : (n4 + 255) / 256)), dim3(256), 0, st, hi, lo, (unsigned*)out, n4);
}
__global__ __launch_bounds__(256) void k_f32_to_f8(const float4* in, unsigned* out, long long n4) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 f = in[i];
        out[i] = pack4_f8(f.x, f.y, f.z, f.w);
    }
}
void launch_f32_to_f8(const float* in, unsigned char* out, long long n, hipStream_t st) {
    const long long n4 = n / 4;
    if (n4 > 0) hipLaunchKernelGGL(k_f32_to_f8, dim3((unsigned)((n4 + 255) / 256 > 16384 ? 16384 : (n4 + 255) / 256)), dim3(256), 0, st, (const float4*)in, (unsigned*)out, n4);
}
__device__ __forceinline__ float e4m3_to_f32(unsigned v) {   // OCP e4m3fn decode
    const unsigned s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float x = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.0f + m * 0.125f, (int)e - 7);
    if (e == 15 && m == 7) x = __builtin_nanf("");
    return s ? -x : x;
}
__global__ __launch_bounds__(256) void k_f8_to_f32(const unsigned char* in, float* out, long long n) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) out[i] = e4m3_to_f32(in[i]);
}
void launch_f8_to_f32(const unsigned char* in, float* out, long long n, hipStream_t st) {
    if (n > 0) hipLaunchKernelGGL(k_f8_to_f32, dim3((unsigned)((n + 255) / 256 > 16384 ? 16384 : (n + 255) / 256)), dim3(256), 0, st, in, out, n);
}
// one wavefront per weight row: scale = the smallest power of two with max|w| / scale <= 448 (division by it is exact), bytes = e4m3(w / scale)
// in 8-row x 128-byte tiles [N/8][K/128][8][128] (one LDS-DMA piece of gemm_mx8_kernel = 1 KiB of one tile), scale4 = the e8m0 byte
// 127 + log2(scale) of every channel, four per dword: {channels 64 g + r, + 16, + 32, + 48} at dword 16 g + r (the hardware scale of the MX
// instruction's weight operand); scale[n] = the same scale as fp32 (tests, tools)
__global__ __launch_bounds__(256) void k_quant_rows_f8(const float* w, unsigned char* out, float* scale, unsigned char* scale_bytes, int N, int K) {
    const int row = wave_row();
    if (row >= N) return;
    const float* src = w + (long long)row * K;
    float m = 0.f;
    for (int k = lane_id() * 4; k < K; k += 256) {
        const float4 f = *reinterpret_cast<const float4*>(src + k);
        m = fmaxf(m, fmaxf(fmaxf(fabsf(f.x), fabsf(f.y)), fmaxf(fabsf(f.z), fabsf(f.w))));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    int ex = 0;
    const float fr = frexpf(m, &ex);                       // m = fr * 2^ex, fr in [0.5, 1); 448 = 0.875 * 2^9
    const int e = m > 0.f ? (fr <= 0.875f ? ex - 9 : ex - 8) : 0;
    const float inv = ldexpf(1.0f, -e);
    if (lane_id() == 0) {
        if (scale) scale[row] = ldexpf(1.0f, e);
        scale_bytes[(((row >> 6) * 16 + (row & 15)) << 2) + ((row >> 4) & 3)] = (unsigned char)(127 + e);
    }
    for (int k = lane_id() * 4; k < K; k += 256) {
        const float4 f = *reinterpret_cast<const float4*>(src + k);
        *reinterpret_cast<unsigned*>(out + (((long long)(row >> 3) * (K >> 7) + (k >> 7)) << 10) + ((row & 7) << 7) + (k & 127)) =
            pack4_f8(f.x * inv, f.y * inv, f.z * inv, f.w * inv);
    }
}
void launch_quant_rows_f8(const float* w, unsigned char* out, float* scale, unsigned* scale4, int N, int K, hipStream_t st) {
    if (N > 0) hipLaunchKernelGGL(k_quant_rows_f8, row_grid(N), dim3(256), 0, st, w, out, scale, (unsigned char*)scale4, N, K);
}

// ---- precision mode 5: h3 operand planes (common.h) and the weight copies of gemm_mx.hip ----
__global__ __launch_bounds__(256) void k_split_h3(const float* in, f16* o_h, unsigned* o_l, long long n4) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 f = reinterpret_cast<const float4*>(in)[i];
        const float v[4] = {f.x, f.y, f.z, f.w};
        f16x4 h; unsigned l;
        split_h3(v, h, l);
        reinterpret_cast<f16x4*>(o_h)[i] = h;
        o_l[i] = l;
    }
}
void launch_split_h3(const float* in, f16* o_h, unsigned char* o_l, long long n, hipStream_t st) {
    const long long n4 = n / 4;
    if (n4 > 0) hipLaunchKernelGGL(k_split_h3, dim3((unsigned)((n4 + 255) / 256 > 16384 ? 16384 : (n4 + 255) / 256)), dim3(256), 0, st, in, o_h, (unsigned*)o_l, n4);
}
__global__ __launch_bounds__(256) void k_planes_to_h3(const bf16* hi, const bf16* lo, f16* o_h, unsigned* o_l, long long n4) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const bf16x4 a = *reinterpret_cast<const bf16x4*>(plane_ptr(hi, i * 4));
        const bf16x4 c = *reinterpret_cast<const bf16x4*>(plane_ptr(lo, i * 4));
        const float v[4] = {join_bf16(a[0], c[0]), join_bf16(a[1], c[1]), join_bf16(a[2], c[2]), join_bf16(a[3], c[3])};
        f16x4 h; unsigned l;
        split_h3(v, h, l);
        reinterpret_cast<f16x4*>(o_h)[i] = h;
        o_l[i] = l;
    }
}
void launch_planes_to_h3(const bf16* hi, const bf16* lo, f16* o_h, unsigned char* o_l, long long n, hipStream_t st) {
    const long long n4 = n / 4;
    if (n4 > 0) hipLaunchKernelGGL(k_planes_to_h3, dim3((unsigned)((n4 + 255) / 256 > 16384 ? 16384 : (n4 + 255) / 256)), dim3(256), 0, st, hi, lo, o_h, (unsigned*)o_l, n4);
}
__global__ __launch_bounds__(256) void k_h3_to_f32(const f16* h, const unsigned char* l, float* out, long long n) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        out[i] = (float)h[i] + e4m3_to_f32(l[i]) * (1.0f / (float)(1 << MMS_H3_SA));
}
void launch_h3_to_f32(const f16* h, const unsigned char* l, float* out, long long n, hipStream_t st) {
    if (n > 0) hipLaunchKernelGGL(k_h3_to_f32, dim3((unsigned)((n + 255) / 256 > 16384 ? 16384 : (n + 255) / 256)), dim3(256), 0, st, h, l, out, n);
}
// one wavefront per weight row n of W[N][K] (fp32, bf16-exact values): m = max|w|.
//   w16 = fp16(w 2^e16), e16 the exponent that puts m into [2^13, 2^14): exact for every bf16 weight down to 2^-28 m; 16 x 32 tiles (wtile_off)
//   w8  = e4m3(w 2^-e),  2^e the smallest power of two with m / 2^e <= 448 (as launch_quant_rows_f8);           8-row x 128-byte tiles
//   scale byte (e8m0) = 127 + e16 + e: the MX low pass then accumulates in the high pass's units; col_scale = 2^-e16 brings both back
__global__ __launch_bounds__(256) void k_prep_w_mx(const float* w, f16* w16, unsigned char* w8, unsigned char* scale_bytes, float* col_scale, int N, int K) {
    const int row = wave_row();
    if (row >= N) return;
    const float* src = w + (long long)row * K;
    float m = 0.f;
    for (int k = lane_id() * 4; k < K; k += 256) {
        const float4 f = *reinterpret_cast<const float4*>(src + k);
        m = fmaxf(m, fmaxf(fmaxf(fabsf(f.x), fabsf(f.y)), fmaxf(fabsf(f.z), fabsf(f.w))));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    int ex = 0;
    const float fr = frexpf(m, &ex);                       // m = fr * 2^ex, fr in [0.5, 1)
    const int e16 = m > 0.f ? 14 - ex : 0;
    const int e = m > 0.f ? (fr <= 0.875f ? ex - 9 : ex - 8) : 0;
    const float s16 = ldexpf(1.0f, e16), s8 = ldexpf(1.0f, -e);
    if (lane_id() == 0) {
        col_scale[row] = ldexpf(1.0f, -e16);
        scale_bytes[(((row >> 6) * 16 + (row & 15)) << 2) + ((row >> 4) & 3)] = (unsigned char)(127 + e16 + e);
    }
    for (int k = lane_id() * 4; k < K; k += 256) {
        const float4 f = *reinterpret_cast<const float4*>(src + k);
        f16x4 h;
        h[0] = (f16)(f.x * s16); h[1] = (f16)(f.y * s16); h[2] = (f16)(f.z * s16); h[3] = (f16)(f.w * s16);
        *reinterpret_cast<f16x4*>(w16 + wtile_off(row, k, K)) = h;
        *reinterpret_cast<unsigned*>(w8 + (((long long)(row >> 3) * (K >> 7) + (k >> 7)) << 10) + ((row & 7) << 7) + (k & 127)) =
            pack4_f8(f.x * s8, f.y * s8, f.z * s8, f.w * s8);
    }
}
void launch_prep_w_mx(const float* w, f16* w16, unsigned char* w8, unsigned* w8_scale4, float* col_scale, int N, int K, hipStream_t st) {
    if (N > 0) hipLaunchKernelGGL(k_prep_w_mx, row_grid(N), dim3(256), 0, st, w, w16, w8, (unsigned char*)w8_scale4, col_scale, N, K);
}

__global__ __launch_bounds__(256) void k_mean8(const float* in, float* out, int U, const int* U_dev) {
    const int row = wave_row();
    if (row >= U || (U_dev && row >= *U_dev)) return;
    Row x;
    row_zero(x);
    for (int p = 0; p < MMS_LABEL_LEN; ++p) row_add(x, in + ((long long)row * MMS_LABEL_LEN + p) * MMS_HIDDEN);
#pragma unroll
    for (int i = 0; i < 12; ++i) x.v[i] *= (1.0f / MMS_LABEL_LEN);
    row_store_f32(x, out + (long long)row * MMS_HIDDEN);
}
void launch_mean8(const float* in, float* out, int U, hipStream_t st, const int* U_dev) {
    if (U > 0) hipLaunchKernelGGL(k_mean8, row_grid(U), dim3(256), 0, st, in, out, U, U_dev);
}
__global__ void k_scale_count(const int* in, int mul, int* out) { *out = *in * mul; }
void launch_scale_count(const int* in, int mul, int* out, hipStream_t st) { hipLaunchKernelGGL(k_scale_count, dim3(1), dim3(1), 0, st, in, mul, out); }

// ------------------------------------------------------------------------------------------------
// zk
// ------------------------------------------------------------------------------------------------
// im2col for kdd_conv1 (model_triple.py:189): row (u, p), column block k holds E[ids[u][p+k-3]] or 0
// (SAME padding of an 8-tap window over 8 positions: 3 left / 4 right).
__global__ __launch_bounds__(256) void k_zk_im2col(const float* E, const int* uniq_ids, int U, int vocab,
                                                   bf16* o_hi, bf16* o_lo, const int* U_dev) {
    const int w = wave_row();
    if (w >= U * MMS_LABEL_LEN * MMS_LABEL_LEN || (U_dev && w >= *U_dev * MMS_LABEL_LEN * MMS_LABEL_LEN)) return;
    const int k = w % MMS_LABEL_LEN, p = (w / MMS_LABEL_LEN) % MMS_LABEL_LEN, u = w / (MMS_LABEL_LEN * MMS_LABEL_LEN);
    const int src = p + k - 3;
    Row x;
    row_zero(x);
    if (src >= 0 && src < MMS_LABEL_LEN)
        row_load(x, E + clamp_id(uniq_ids[u * MMS_LABEL_LEN + src], vocab) * MMS_HIDDEN);
    const long long off = ((long long)u * MMS_LABEL_LEN + p) * (MMS_LABEL_LEN * MMS_HIDDEN) + k * MMS_HIDDEN;
    row_store_planes(x, plane_ptr(o_hi, off), plane_ptr(o_lo, off));
}
void launch_zk_im2col(const float* E, const int* uniq_ids, int U, int vocab, bf16* o_hi, bf16* o_lo, hipStream_t st, const int* U_dev) {
    if (U > 0)
        hipLaunchKernelGGL(k_zk_im2col, row_grid((long long)U * 64), dim3(256), 0, st, E, uniq_ids, U, vocab, o_hi, o_lo, U_dev);
}

// model_triple.py:190-195: mean(relu(conv1)) [by unique label] + kdd_dense1(boxes_5) + relu(conv2(feats))
__global__ __launch_bounds__(256) void k_zk_tokpre(const float* labfeat, const int* lab_index, int n_labels, const float* boxes5,
                                                   const float* Wd, const float* bd, const float* img,
                                                   bf16* o_hi, bf16* o_lo, int rows, const int* src, const int* rows_dev) {
    const int row = wave_row();
    if (row >= rows || (rows_dev && row >= *rows_dev)) return;
    const int box = src ? src[row] : row;      // compact rows (live boxes only): img / output by row, label and geometry by the box it stands for
    Row x;
    row_load(x, labfeat + clamp_id(lab_index[box], n_labels) * MMS_HIDDEN);   // a bad index must not read outside the table
    row_add(x, bd);
#pragma unroll
    for (int k = 0; k < 5; ++k) row_axpy(x, boxes5[(long long)box * 5 + k], Wd + k * MMS_HIDDEN);
    row_add(x, img + (long long)row * MMS_HIDDEN);
    row_store_planes(x, plane_ptr(o_hi, (long long)row * MMS_HIDDEN), plane_ptr(o_lo, (long long)row * MMS_HIDDEN));
}
void launch_zk_tokpre(const float* labfeat, const int* lab_index, int n_labels, const float* boxes5, const float* Wd,
                      const float* bd, const float* img, bf16* o_hi, bf16* o_lo, int rows, hipStream_t st, const int* src, const int* rows_dev) {
    if (rows > 0)
        hipLaunchKernelGGL(k_zk_tokpre, row_grid(rows), dim3(256), 0, st, labfeat, lab_index, n_labels, boxes5, Wd, bd, img, o_hi, o_lo, rows, src, rows_dev);
}

// fp32 rows [*][width] -> split planes of the rows listed in idx (compact: output row r = input row idx[r]), r < *rows_dev; width % 256 == 0
__global__ __launch_bounds__(256) void k_split_f32_rows(const float* in, const int* idx, const int* rows_dev, int max_rows, int width, bf16* o_hi, bf16* o_lo) {
    const int row = wave_row();
    if (row >= max_rows || row >= *rows_dev) return;
    const float* src = in + (long long)idx[row] * width;
    for (int t = 0; t < width / 256; ++t) {
        const float4 f = *reinterpret_cast<const float4*>(src + t * 256 + lane_id() * 4);
        const float v[4] = {f.x, f.y, f.z, f.w};
        bf16x4 h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) { bf16 a, c; split_bf16(v[e], a, c); h[e] = a; l[e] = c; }
        const long long off = (long long)row * width + t * 256 + lane_id() * 4;
        *reinterpret_cast<bf16x4*>(plane_ptr(o_hi, off)) = h;
        *reinterpret_cast<bf16x4*>(plane_ptr(o_lo, off)) = l;
    }
}
void launch_split_f32_rows(const float* in, const int* idx, const int* rows_dev, int max_rows, int width, bf16* o_hi, bf16* o_lo, hipStream_t st) {
    if (max_rows > 0) hipLaunchKernelGGL(k_split_f32_rows, row_grid(max_rows), dim3(256), 0, st, in, idx, rows_dev, max_rows, width, o_hi, o_lo);
}

// pixelbert.py:580-621: concat text || image tokens, + token_type[segment_ids], + positions
// [0..T-1] + [T]*10, LayerNorm over every row.
__global__ __launch_bounds__(256) void k_zk_embed(const float* E, const float* type_tab, const float* pos_tab,
                                                  const float* gamma, const float* beta, const int* query_ids,
                                                  const int* segment_ids, const float* tok, int T, int vocab,
                                                  bf16* o_hi, bf16* o_lo, int B) {
    const int S = T + MMS_NBOX;
    const int row = wave_row();
    if (row >= B * S) return;
    const int b = row / S, s = row % S;
    Row x;
    if (s < T) row_load(x, E + clamp_id(query_ids[b * T + s], vocab) * MMS_HIDDEN);
    else row_load(x, tok + ((long long)b * MMS_NBOX + (s - T)) * MMS_HIDDEN);
    row_add(x, type_tab + clamp_id(segment_ids[row], 2) * MMS_HIDDEN);
    row_add(x, pos_tab + (s < T ? s : T) * MMS_HIDDEN);
    row_ln(x, gamma, beta);
    row_store_planes(x, plane_ptr(o_hi, (long long)row * MMS_HIDDEN), plane_ptr(o_lo, (long long)row * MMS_HIDDEN));
}
void launch_zk_embed(const float* E, const float* type_tab, const float* pos_tab, const float* gamma,
                     const float* beta, const int* query_ids, const int* segment_ids, const float* tok,
                     int T, int vocab, bf16* o_hi, bf16* o_lo, int B, hipStream_t st) {
    if (B > 0)
        hipLaunchKernelGGL(k_zk_embed, row_grid((long long)B * (T + MMS_NBOX)), dim3(256), 0, st, E, type_tab, pos_tab,
                           gamma, beta, query_ids, segment_ids, tok, T, vocab, o_hi, o_lo, B);
}

// model_triple.py:198-201 + pixelbert.py:813: additive key mask (1 - mask) * -10000
__global__ void k_zk_mask(const int* len_query, const int* num_boxes, int T, float* key_add, int B) {
    const int S = T + MMS_NBOX;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * S) return;
    const int b = i / S, s = i % S;
    const bool keep = s < T ? (s < len_query[b]) : ((s - T) < num_boxes[b]);
    key_add[i] = keep ? 0.f : -10000.f;
}
void launch_zk_mask(const int* len_query, const int* num_boxes, int T, float* key_add, int B, hipStream_t st) {
    if (B > 0)
        hipLaunchKernelGGL(k_zk_mask, dim3((B * (T + MMS_NBOX) + 255) / 256), dim3(256), 0, st, len_query, num_boxes, T, key_add, B);
}

// amsoftmax_loss, model_triple.py:56-86 (label-dependent margin at inference)
__global__ __launch_bounds__(256) void k_zk_head(const float* pooled, const float* am_kernel, const int64_t* labels,
                                                 float scale, float margin, float* logits, float* probs, int B) {
    const int row = wave_row();
    if (row >= B) return;
    Row x;
    row_load(x, pooled + (long long)row * MMS_HIDDEN);
    float ss = 0.f, d0 = 0.f, d1 = 0.f, k0 = 0.f, k1 = 0.f;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int col = t * 256 + lane_id() * 4 + e;
            const float v = x.v[t * 4 + e], a = am_kernel[col * 2], c = am_kernel[col * 2 + 1];
            ss += v * v; d0 += v * a; d1 += v * c; k0 += a * a; k1 += c * c;
        }
    ss = wave_sum(ss); d0 = wave_sum(d0); d1 = wave_sum(d1); k0 = wave_sum(k0); k1 = wave_sum(k1);
    const float xn = 1.0f / sqrtf(fmaxf(ss, 1e-12f));
    float c0 = d0 * xn * (1.0f / sqrtf(fmaxf(k0, 1e-10f)));
    float c1 = d1 * xn * (1.0f / sqrtf(fmaxf(k1, 1e-10f)));
    c0 = fminf(fmaxf(c0, -1.f), 1.f);
    c1 = fminf(fmaxf(c1, -1.f), 1.f);
    const int lab = labels[row] != 0;
    const float gt = lab ? c1 : c0;
    const float m = gt > margin ? margin : 0.f;
    const float l0 = (c0 - (lab ? 0.f : m)) * scale, l1 = (c1 - (lab ? m : 0.f)) * scale;
    if (lane_id() == 0) {
        const float mx = fmaxf(l0, l1), e0 = expf(l0 - mx), e1 = expf(l1 - mx), inv = 1.0f / (e0 + e1);
        logits[row * 2] = l0; logits[row * 2 + 1] = l1;
        if (probs) { probs[row * 2] = e0 * inv; probs[row * 2 + 1] = e1 * inv; }
    }
}
void launch_zk_head(const float* pooled, const float* am_kernel, const int64_t* labels, float scale, float margin,
                    float* logits, float* probs, int B, hipStream_t st) {
    if (B > 0) hipLaunchKernelGGL(k_zk_head, row_grid(B), dim3(256), 0, st, pooled, am_kernel, labels, scale, margin, logits, probs, B);
}

// ------------------------------------------------------------------------------------------------
// lds
// ------------------------------------------------------------------------------------------------
// pixelmodel.py:506-598: text = E[id] + type[segment] + pos[t] -> LN, written to rows b*S + t
__global__ __launch_bounds__(256) void k_lds_embed_text(const float* E, const float* type_tab, const float* pos_tab,
                                                        const float* gamma, const float* beta, const int64_t* input_ids,
                                                        const int64_t* segment_ids, int T, int S, int vocab,
                                                        bf16* o_hi, bf16* o_lo, int B) {
    const int row = wave_row();
    if (row >= B * T) return;
    const int b = row / T, t = row % T;
    Row x;
    row_load(x, E + clamp_id(input_ids[row], vocab) * MMS_HIDDEN);
    row_add(x, type_tab + clamp_id(segment_ids[row], 2) * MMS_HIDDEN);
    row_add(x, pos_tab + t * MMS_HIDDEN);
    row_ln(x, gamma, beta);
    const long long off = ((long long)b * S + t) * MMS_HIDDEN;
    row_store_planes(x, plane_ptr(o_hi, off), plane_ptr(o_lo, off));
}
void launch_lds_embed_text(const float* E, const float* type_tab, const float* pos_tab, const float* gamma,
                           const float* beta, const int64_t* input_ids, const int64_t* segment_ids, int T, int S,
                           int vocab, bf16* o_hi, bf16* o_lo, int B, hipStream_t st) {
    if (B > 0)
        hipLaunchKernelGGL(k_lds_embed_text, row_grid((long long)B * T), dim3(256), 0, st, E, type_tab, pos_tab, gamma, beta,
                           input_ids, segment_ids, T, S, vocab, o_hi, o_lo, B);
}

// pixelmodel.py:489-498 raw reshape-matmul: out[b,box,j] = sum_k wl[k] * E[ids[b,box,j/96]][8*(j%96)+k]
__global__ __launch_bounds__(256) void k_lds_label(const float* E, const float* wl, const int64_t* labelfeat, int vocab,
                                                   int S, int row_off, bf16* o_hi, bf16* o_lo, int B) {
    const int row = wave_row();
    if (row >= B * MMS_NBOX) return;
    const int b = row / MMS_NBOX, n = row % MMS_NBOX;
    float w[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) w[k] = wl[k];
    Row x;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = t * 256 + lane_id() * 4 + e;
            const float* src = E + clamp_id(labelfeat[(long long)row * MMS_LABEL_LEN + j / 96], vocab) * MMS_HIDDEN + 8 * (j % 96);
            const float4 f0 = *reinterpret_cast<const float4*>(src);
            const float4 f1 = *reinterpret_cast<const float4*>(src + 4);
            // same summation order as a row-vector x [8,1] matmul: k ascending
            float a = f0.x * w[0];
            a += f0.y * w[1]; a += f0.z * w[2]; a += f0.w * w[3];
            a += f1.x * w[4]; a += f1.y * w[5]; a += f1.z * w[6]; a += f1.w * w[7];
            x.v[t * 4 + e] = a;
        }
    const long long off = ((long long)b * S + row_off + n) * MMS_HIDDEN;
    row_store_planes(x, plane_ptr(o_hi, off), plane_ptr(o_lo, off));
}
void launch_lds_label(const float* E, const float* wl, const int64_t* labelfeat, int vocab, int S, int row_off,
                      bf16* o_hi, bf16* o_lo, int B, hipStream_t st) {
    if (B > 0)
        hipLaunchKernelGGL(k_lds_label, row_grid((long long)B * MMS_NBOX), dim3(256), 0, st, E, wl, labelfeat, vocab, S, row_off, o_hi, o_lo, B);
}

// run_pretraining_predict_score.py:479-501: logits = pooled W^T + b, W [2,768]
__global__ __launch_bounds__(256) void k_lds_head(const float* pooled, const float* W, const float* bias,
                                                  float* logits, float* probs, int B) {
    const int row = wave_row();
    if (row >= B) return;
    Row x, w0, w1;
    row_load(x, pooled + (long long)row * MMS_HIDDEN);
    row_load(w0, W);
    row_load(w1, W + MMS_HIDDEN);
    float d0 = 0.f, d1 = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) { d0 += x.v[i] * w0.v[i]; d1 += x.v[i] * w1.v[i]; }
    d0 = wave_sum(d0) + bias[0];
    d1 = wave_sum(d1) + bias[1];
    if (lane_id() == 0) {
        const float mx = fmaxf(d0, d1), e0 = expf(d0 - mx), e1 = expf(d1 - mx), inv = 1.0f / (e0 + e1);
        logits[row * 2] = d0; logits[row * 2 + 1] = d1;
        if (probs) { probs[row * 2] = e0 * inv; probs[row * 2 + 1] = e1 * inv; }
    }
}
void launch_lds_head(const float* pooled, const float* W, const float* b, float* logits, float* probs, int B, hipStream_t st) {
    if (B > 0) hipLaunchKernelGGL(k_lds_head, row_grid(B), dim3(256), 0, st, pooled, W, b, logits, probs, B);
}

// ------------------------------------------------------------------------------------------------
// lxmert
// ------------------------------------------------------------------------------------------------
// BertEmbeddings, modeling.py:283-297: word + position(0..T-1) + token_type(0) -> LN
__global__ __launch_bounds__(256) void k_lx_embed_lang(const float* E, const float* pos_tab, const float* type_tab,
                                                       const float* gamma, const float* beta, const int64_t* input_ids,
                                                       int T, int vocab, bf16* o_hi, bf16* o_lo, int B) {
    const int row = wave_row();
    if (row >= B * T) return;
    Row x;
    row_load(x, E + clamp_id(input_ids[row], vocab) * MMS_HIDDEN);
    row_add(x, pos_tab + (row % T) * MMS_HIDDEN);
    row_add(x, type_tab);
    row_ln(x, gamma, beta);
    row_store_planes(x, plane_ptr(o_hi, (long long)row * MMS_HIDDEN), plane_ptr(o_lo, (long long)row * MMS_HIDDEN));
}
void launch_lx_embed_lang(const float* E, const float* pos_tab, const float* type_tab, const float* gamma,
                          const float* beta, const int64_t* input_ids, int T, int vocab, bf16* o_hi, bf16* o_lo,
                          int B, hipStream_t st) {
    if (B > 0)
        hipLaunchKernelGGL(k_lx_embed_lang, row_grid((long long)B * T), dim3(256), 0, st, E, pos_tab, type_tab, gamma, beta,
                           input_ids, T, vocab, o_hi, o_lo, B);
}

// modeling.py:915 + 526: per unique label text, BertEmbeddings over its 8 tokens, then
// Conv2d(8 -> 1, k = 1) over the token-position axis: z = sum_t cw[t] * emb[t] + cb
__global__ __launch_bounds__(256) void k_lx_label_emb(const float* E, const float* pos_tab, const float* type_tab,
                                                      const float* gamma, const float* beta, const float* conv_w,
                                                      const float* conv_b, const int64_t* uniq_ids, int vocab,
                                                      bf16* o_hi, bf16* o_lo, int U) {
    const int u = wave_row();
    if (u >= U) return;
    Row acc;
    row_zero(acc);
    for (int t = 0; t < MMS_LABEL_LEN; ++t) {
        Row x;
        row_load(x, E + clamp_id(uniq_ids[(long long)u * MMS_LABEL_LEN + t], vocab) * MMS_HIDDEN);
        row_add(x, pos_tab + t * MMS_HIDDEN);
        row_add(x, type_tab);
        row_ln(x, gamma, beta);
        const float w = conv_w[t];
#pragma unroll
        for (int i = 0; i < 12; ++i) acc.v[i] += w * x.v[i];
    }
    const float cb = conv_b[0];
#pragma unroll
    for (int i = 0; i < 12; ++i) acc.v[i] += cb;
    row_store_planes(acc, plane_ptr(o_hi, (long long)u * MMS_HIDDEN), plane_ptr(o_lo, (long long)u * MMS_HIDDEN));
}
void launch_lx_label_emb(const float* E, const float* pos_tab, const float* type_tab, const float* gamma,
                         const float* beta, const float* conv_w, const float* conv_b, const int64_t* uniq_ids,
                         int vocab, bf16* o_hi, bf16* o_lo, int U, hipStream_t st) {
    if (U > 0)
        hipLaunchKernelGGL(k_lx_label_emb, row_grid(U), dim3(256), 0, st, E, pos_tab, type_tab, gamma, beta, conv_w, conv_b,
                           uniq_ids, vocab, o_hi, o_lo, U);
}

// VisualFeatEncoder, modeling.py:519-531: (LN(Wf f) + LN(Wb b) + LN(Wl conv(label))) / 3
__global__ __launch_bounds__(256) void k_lx_visn(const float* xf, const float* g_x, const float* b_x, const float* boxes,
                                                 int box_dim, const float* Wb, const float* bb, const float* g_y,
                                                 const float* b_y, const float* z, const int* lab_index, int n_labels,
                                                 bf16* o_hi, bf16* o_lo, int rows, const int* src_map, const int* rows_dev, int xf_compact) {
    const int orow = wave_row();
    if (orow >= rows || (rows_dev && orow >= *rows_dev)) return;
    const int row = src_map ? src_map[orow] : orow;   // packed mode: output row orow <- box row src_map[orow]
    Row x, y;
    row_load(x, xf + (long long)(xf_compact ? orow : row) * MMS_HIDDEN);      // xf_compact: the visn_fc projection ran on the live boxes only, in output order
    row_ln(x, g_x, b_x);
    row_load(y, bb);
    for (int k = 0; k < box_dim; ++k) {
        const float bv = boxes[(long long)row * box_dim + k];
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                y.v[t * 4 + e] += bv * Wb[(t * 256 + lane_id() * 4 + e) * box_dim + k];  // torch [out, in]
    }
    row_ln(y, g_y, b_y);
    Row zz;
    row_load(zz, z + clamp_id(lab_index[row], n_labels) * MMS_HIDDEN);
#pragma unroll
    for (int i = 0; i < 12; ++i) x.v[i] = (x.v[i] + y.v[i] + zz.v[i]) / 3.0f;
    row_store_planes(x, plane_ptr(o_hi, (long long)orow * MMS_HIDDEN), plane_ptr(o_lo, (long long)orow * MMS_HIDDEN));
}
void launch_lx_visn(const float* xf, const float* g_x, const float* b_x, const float* boxes, int box_dim,
                    const float* Wb, const float* bb, const float* g_y, const float* b_y, const float* z,
                    const int* lab_index, int n_labels, bf16* o_hi, bf16* o_lo, int rows, hipStream_t st, const int* src,
                    const int* rows_dev, int xf_compact) {
    if (rows > 0)
        hipLaunchKernelGGL(k_lx_visn, row_grid(rows), dim3(256), 0, st, xf, g_x, b_x, boxes, box_dim, Wb, bb, g_y, b_y, z,
                           lab_index, n_labels, o_hi, o_lo, rows, src, rows_dev, xf_compact);
}

// modeling.py:890-910: additive masks (1 - m) * -10000 for language and visual keys
__global__ void k_lx_masks(const int64_t* input_mask, const float* visual_mask, int T, float* lang_add,
                           float* visn_add, int B) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < B * T) lang_add[i] = (1.0f - (float)input_mask[i]) * -10000.f;
    if (i < B * MMS_NBOX) visn_add[i] = (1.0f - visual_mask[i]) * -10000.f;
}
void launch_lx_masks(const int64_t* input_mask, const float* visual_mask, int T, float* lang_add, float* visn_add,
                     int B, hipStream_t st) {
    const int n = B * (T > MMS_NBOX ? T : MMS_NBOX);
    if (B > 0) hipLaunchKernelGGL(k_lx_masks, dim3((n + 255) / 256), dim3(256), 0, st, input_mask, visual_mask, T, lang_add, visn_add, B);
}

// logit_fc tail, kdd_model.py:167-172: LayerNorm(1536) -> Linear(1536, 2); h = GeLU(Linear(768,1536)(pooled))
__global__ __launch_bounds__(256) void k_lx_head(const float* h, const float* gamma, const float* beta, const float* W,
                                                 const float* bias, float* logits, float* probs, int B) {
    const int row = wave_row();
    if (row >= B) return;
    const int N = 2 * MMS_HIDDEN;
    float v[24];
#pragma unroll
    for (int t = 0; t < 6; ++t) {
        const float4 f = *reinterpret_cast<const float4*>(h + (long long)row * N + t * 256 + lane_id() * 4);
        v[t * 4] = f.x; v[t * 4 + 1] = f.y; v[t * 4 + 2] = f.z; v[t * 4 + 3] = f.w;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 24; ++i) s += v[i];
    const float mean = wave_sum(s) / N;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 24; ++i) { const float d = v[i] - mean; q += d * d; }
    const float inv = 1.0f / sqrtf(wave_sum(q) / N + MMS_LN_EPS);
    float d0 = 0.f, d1 = 0.f;
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int col = t * 256 + lane_id() * 4 + e;
            const float hn = (v[t * 4 + e] - mean) * inv * gamma[col] + beta[col];
            d0 += hn * W[col];
            d1 += hn * W[N + col];
        }
    d0 = wave_sum(d0) + bias[0];
    d1 = wave_sum(d1) + bias[1];
    if (lane_id() == 0) {
        const float mx = fmaxf(d0, d1), e0 = expf(d0 - mx), e1 = expf(d1 - mx), iv = 1.0f / (e0 + e1);
        logits[row * 2] = d0; logits[row * 2 + 1] = d1;
        if (probs) { probs[row * 2] = e0 * iv; probs[row * 2 + 1] = e1 * iv; }
    }
}
void launch_lx_head(const float* h, const float* gamma, const float* beta, const float* W, const float* b, float* logits,
                    float* probs, int B, hipStream_t st) {
    if (B > 0) hipLaunchKernelGGL(k_lx_head, row_grid(B), dim3(256), 0, st, h, gamma, beta, W, b, logits, probs, B);
}

// ------------------------------------------------------------------------------------------------
// packed (ragged) execution plans
// ------------------------------------------------------------------------------------------------
// One workgroup scans the per-pair live-token counts and lays the live tokens out contiguously.
// A token is dropped only if it is masked as a KEY and is not the CLS row; its own outputs are
// never read by a live row (keys masked, pooler reads CLS), so logits are unchanged.  A pair whose
// keys are ALL masked keeps every token (the reference's softmax is then uniform over all of them).
#define PLAN_THREADS 1024

__device__ __forceinline__ int plan_scan(int local, int* sh, int* total) {
    const int tid = threadIdx.x;
    sh[tid] = local;
    __syncthreads();
    for (int o = 1; o < PLAN_THREADS; o <<= 1) {
        const int v = tid >= o ? sh[tid - o] : 0;
        __syncthreads();
        sh[tid] += v;
        __syncthreads();
    }
    const int incl = sh[tid];
    if (total) *total = sh[PLAN_THREADS - 1];
    __syncthreads();
    return incl - local;
}

// Three launches per plan: per-pair live counts (one thread per pair, whole grid), one-block exclusive scan of the counts
// (-> first row of every pair, device-side live-row total), per-pair fill of the gather / mask tables (whole grid).  The first
// version did all of it in ONE 1024-thread block: 0.4 ms (zk) / 3 ms (lxmert) per 30 000-pair launch wave.
__global__ __launch_bounds__(PLAN_THREADS) void k_plan_scan(const int* cnt, int n, int* off, int* rows_dev) {
    __shared__ int sh[PLAN_THREADS];
    const int tid = threadIdx.x;
    const int per = (n + PLAN_THREADS - 1) / PLAN_THREADS;
    const int b0 = tid * per, b1 = (b0 + per) < n ? (b0 + per) : n;
    int local = 0;
    for (int b = b0; b < b1; ++b) local += cnt[b];
    int total;
    int base = plan_scan(local, sh, &total);
    for (int b = b0; b < b1; ++b) { off[b] = base; base += cnt[b]; }
    if (tid == 0) *rows_dev = total;
}

void launch_plan_scan(const int* cnt, int n, int* off, int* total_dev, hipStream_t st) {
    if (n > 0) hipLaunchKernelGGL(k_plan_scan, dim3(1), dim3(PLAN_THREADS), 0, st, cnt, n, off, total_dev);
}

__global__ __launch_bounds__(256) void k_zk_plan_count(const int* len_query, const int* num_boxes, int T, int n, int* cnt) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= n) return;
    const int lq = min(max(len_query[b], 0), T), nb = min(max(num_boxes[b], 0), MMS_NBOX);
    cnt[b] = (lq + nb == 0) ? T + MMS_NBOX : (max(lq, 1) + nb);
}
__global__ __launch_bounds__(256) void k_zk_plan_fill(const int* len_query, const int* num_boxes, int T, int n, const int* off,
                                                      int* tok_src, float* key_add) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= n) return;
    const int S = T + MMS_NBOX;
    const int lq = min(max(len_query[b], 0), T), nb = min(max(num_boxes[b], 0), MMS_NBOX);
    const bool dense = (lq + nb == 0);
    const int nt = dense ? T : max(lq, 1), nv = dense ? MMS_NBOX : nb;
    const int base = off[b];
    for (int s = 0; s < nt; ++s) { tok_src[base + s] = b * S + s; key_add[base + s] = s < lq ? 0.f : -10000.f; }
    for (int j = 0; j < nv; ++j) { tok_src[base + nt + j] = b * S + T + j; key_add[base + nt + j] = j < nb ? 0.f : -10000.f; }
}
void launch_zk_pack_plan(const int* len_query, const int* num_boxes, int T, int n, int* off, int* cnt, int* tok_src,
                         float* key_add, int* rows_dev, hipStream_t st) {
    if (n <= 0) return;
    const dim3 grid((n + 255) / 256);
    hipLaunchKernelGGL(k_zk_plan_count, grid, dim3(256), 0, st, len_query, num_boxes, T, n, cnt);
    hipLaunchKernelGGL(k_plan_scan, dim3(1), dim3(PLAN_THREADS), 0, st, cnt, n, off, rows_dev);
    hipLaunchKernelGGL(k_zk_plan_fill, grid, dim3(256), 0, st, len_query, num_boxes, T, n, off, tok_src, key_add);
}

// Live boxes of a zk launch wave (packed mode): pair b keeps nv(b) image tokens -- min(num_boxes, 10), or all 10 when nothing of the pair is
// live (k_zk_plan_fill's rule) -- so only those boxes' 2048-d features are split, projected (kdd_conv2, kdd_featureemb) and combined; the
// padded ones (62 % of the rows on the bench batch) cannot reach a logit.  box_idx[r] = b * 10 + j for compact row r = box_off[b] + j.
__global__ __launch_bounds__(256) void k_zk_box_count(const int* len_query, const int* num_boxes, int T, int n, int* cnt) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= n) return;
    const int lq = min(max(len_query[b], 0), T), nb = min(max(num_boxes[b], 0), MMS_NBOX);
    cnt[b] = (lq + nb == 0) ? MMS_NBOX : nb;
}
__global__ __launch_bounds__(256) void k_zk_box_fill(const int* cnt, const int* off, int n, int* box_idx) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= n) return;
    for (int j = 0; j < cnt[b]; ++j) box_idx[off[b] + j] = b * MMS_NBOX + j;
}
void launch_zk_box_plan(const int* len_query, const int* num_boxes, int T, int n, int* cnt, int* off, int* box_idx, int* rows_dev, hipStream_t st) {
    if (n <= 0) return;
    const dim3 grid((n + 255) / 256);
    hipLaunchKernelGGL(k_zk_box_count, grid, dim3(256), 0, st, len_query, num_boxes, T, n, cnt);
    hipLaunchKernelGGL(k_plan_scan, dim3(1), dim3(PLAN_THREADS), 0, st, cnt, n, off, rows_dev);
    hipLaunchKernelGGL(k_zk_box_fill, grid, dim3(256), 0, st, cnt, off, n, box_idx);
}
// Small launch waves (n <= 1024 pairs: the reference's call sizes): the box plan and the token plan above -- six launches of ~4.5 us each in a chain that is
// all latency -- as ONE single-block kernel (one thread per pair, two block scans).  Same tables, same totals.
__global__ __launch_bounds__(PLAN_THREADS) void k_zk_plans_small(const int* len_query, const int* num_boxes, int T, int n, int* b_cnt, int* b_off, int* box_idx,
                                                                 int* t_cnt, int* t_off, int* tok_src, float* key_add, int* rows_dev) {
    __shared__ int sh[PLAN_THREADS];
    const int b = threadIdx.x;
    const bool valid = b < n;
    const int S = T + MMS_NBOX;
    const int lq = valid ? min(max(len_query[b], 0), T) : 0, nb = valid ? min(max(num_boxes[b], 0), MMS_NBOX) : 0;
    const bool dense = (lq + nb == 0);
    const int nt = dense ? T : max(lq, 1), nv = dense ? MMS_NBOX : nb;      // k_zk_plan_fill's / k_zk_box_count's rule: a pair with nothing live keeps everything
    int total;
    const int boff = plan_scan(valid ? nv : 0, sh, &total);
    if (b == 0) rows_dev[1] = total;
    const int toff = plan_scan(valid ? nt + nv : 0, sh, &total);
    if (b == 0) rows_dev[0] = total;
    if (!valid) return;
    b_cnt[b] = nv; b_off[b] = boff;
    for (int j = 0; j < nv; ++j) box_idx[boff + j] = b * MMS_NBOX + j;
    t_cnt[b] = nt + nv; t_off[b] = toff;
    for (int s = 0; s < nt; ++s) { tok_src[toff + s] = b * S + s; key_add[toff + s] = s < lq ? 0.f : -10000.f; }
    for (int j = 0; j < nv; ++j) { tok_src[toff + nt + j] = b * S + T + j; key_add[toff + nt + j] = j < nb ? 0.f : -10000.f; }
}
bool launch_zk_plans_small(const int* len_query, const int* num_boxes, int T, int n, int* b_cnt, int* b_off, int* box_idx, int* t_cnt, int* t_off,
                           int* tok_src, float* key_add, int* rows_dev, hipStream_t st) {
    if (n <= 0 || n > PLAN_THREADS) return false;
    hipLaunchKernelGGL(k_zk_plans_small, dim3(1), dim3(PLAN_THREADS), 0, st, len_query, num_boxes, T, n, b_cnt, b_off, box_idx, t_cnt, t_off, tok_src, key_add, rows_dev);
    return true;
}
__global__ __launch_bounds__(256) void k_zk_embed_packed(const float* E, const float* type_tab, const float* pos_tab,
                                                         const float* gamma, const float* beta, const int* query_ids,
                                                         const int* segment_ids, const float* tok, int T, int vocab,
                                                         const int* tok_src, const int* rows_dev, int max_rows,
                                                         bf16* o_hi, bf16* o_lo, const int* box_off) {
    const int S = T + MMS_NBOX;
    const int row = wave_row();
    if (row >= max_rows || row >= *rows_dev) return;
    const int src = tok_src[row], b = src / S, s = src % S;
    Row x;
    if (s < T) row_load(x, E + clamp_id(query_ids[b * T + s], vocab) * MMS_HIDDEN);
    else row_load(x, tok + ((box_off ? (long long)box_off[b] : (long long)b * MMS_NBOX) + (s - T)) * MMS_HIDDEN);      // box_off: image tokens of live boxes only (compact)
    row_add(x, type_tab + clamp_id(segment_ids[src], 2) * MMS_HIDDEN);
    row_add(x, pos_tab + (s < T ? s : T) * MMS_HIDDEN);
    row_ln(x, gamma, beta);
    row_store_planes(x, plane_ptr(o_hi, (long long)row * MMS_HIDDEN), plane_ptr(o_lo, (long long)row * MMS_HIDDEN));
}
void launch_zk_embed_packed(const float* E, const float* type_tab, const float* pos_tab, const float* gamma,
                            const float* beta, const int* query_ids, const int* segment_ids, const float* tok, int T,
                            int vocab, const int* tok_src, const int* rows_dev, int max_rows, bf16* o_hi, bf16* o_lo,
                            hipStream_t st, const int* box_off) {
    if (max_rows > 0)
        hipLaunchKernelGGL(k_zk_embed_packed, row_grid(max_rows), dim3(256), 0, st, E, type_tab, pos_tab, gamma, beta, query_ids,
                           segment_ids, tok, T, vocab, tok_src, rows_dev, max_rows, o_hi, o_lo, box_off);
}

// lds has no attention mask (pixelmodel.py:189-190): all 40 tokens of a pair attend and are attended.  But its 10 feature tokens and
// 10 label tokens carry NO position / type embedding (they are concatenated after the LayerNorm, pixelmodel.py:600-601), so two boxes
// with identical inputs give identical token rows at EVERY layer: the zero-padded boxes (featureemb(0) = bias; label ids 0) and any
// two boxes of the same class.  A softmax over keys with duplicates equals a softmax over the distinct keys with exp(s) multiplied by
// the multiplicity, i.e. an additive log(m) on the score, and P V sums identical value rows the same way -- so the duplicates are
// dropped and their representative carries key_add = log(multiplicity).  Exact up to fp32 round-off; with 3.8 boxes per image on
// average 40 rows become ~29.  Feature rows are merged only when they are all-zero (flag kernel below); label rows by tuple equality.
__global__ __launch_bounds__(256) void k_row_nonzero(const float* feats, int* flag, int rows) {
    const int row = wave_row();
    if (row >= rows) return;
    const float4* p = reinterpret_cast<const float4*>(feats + (long long)row * MMS_FEAT);
    bool nz = false;
#pragma unroll
    for (int t = 0; t < MMS_FEAT / 256; ++t) {
        const float4 f = p[t * 64 + lane_id()];
        nz = nz || f.x != 0.f || f.y != 0.f || f.z != 0.f || f.w != 0.f;
    }
    const unsigned long long any = __ballot(nz);
    if (lane_id() == 0) flag[row] = any != 0ull;
}
// One WAVEFRONT per pair (lane j < 10 owns box j): the label tuples are compared through v_readlane broadcasts instead of 45 x 8 dependent global loads per
// thread -- the thread-per-pair form took 16 us (count) + 30 us (fill) on a 5-pair call, all of it load latency.  Same tables, same order, same logf.
__device__ __forceinline__ void lds_pair_masks(const int* nz, const int64_t* labelfeat, int b, int lane, unsigned& nz_mask, unsigned& first_mask, int& m_own) {
    const bool lj = lane < MMS_NBOX;
    int lo[MMS_LABEL_LEN], hi[MMS_LABEL_LEN];
#pragma unroll
    for (int k = 0; k < MMS_LABEL_LEN; ++k) {
        const long long v = lj ? (long long)labelfeat[((long long)b * MMS_NBOX + lane) * MMS_LABEL_LEN + k] : 0;
        lo[k] = (int)(v & 0xffffffffll); hi[k] = (int)(v >> 32);
    }
    const bool z = lj && nz[b * MMS_NBOX + lane] != 0;
    bool first = lj;
    int m = 1;
#pragma unroll
    for (int i = 0; i < MMS_NBOX; ++i) {
        bool same = true;
#pragma unroll
        for (int k = 0; k < MMS_LABEL_LEN; ++k)
            same = same && __builtin_amdgcn_readlane(lo[k], i) == lo[k] && __builtin_amdgcn_readlane(hi[k], i) == hi[k];
        if (i < lane) first = first && !same;
        if (i > lane) m += same ? 1 : 0;
    }
    nz_mask = (unsigned)__ballot(z);
    first_mask = (unsigned)__ballot(lj && first);
    m_own = m;
}
__global__ __launch_bounds__(256) void k_lds_plan_count(const int* nz, const int64_t* labelfeat, int T, int n, int* cnt) {
    const int b = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (b >= n) return;      // wave-uniform
    unsigned nzm, fm; int m;
    lds_pair_masks(nz, labelfeat, b, lane, nzm, fm, m);
    const int nnz = __popc(nzm);
    if (lane == 0) cnt[b] = T + nnz + (nnz < MMS_NBOX ? 1 : 0) + __popc(fm);
}
__global__ __launch_bounds__(256) void k_lds_plan_fill(const int* nz, const int64_t* labelfeat, int T, int n, const int* off, int* tok_src,
                                                       float* key_add) {
    const int b = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (b >= n) return;
    const int S = T + 2 * MMS_NBOX;
    unsigned nzm, fm; int m;
    lds_pair_masks(nz, labelfeat, b, lane, nzm, fm, m);
    const int r0 = off[b];
    if (lane < T) { tok_src[r0 + lane] = b * S + lane; key_add[r0 + lane] = 0.f; }      // T <= 32
    // feature tokens in box order: every non-zero box, and the FIRST all-zero box standing for all of them (key bias log(#zero boxes))
    const int nnz = __popc(nzm), zeros = MMS_NBOX - nnz;
    const int jz = zeros ? __ffs((int)(~nzm & ((1u << MMS_NBOX) - 1))) - 1 : -1;
    if (lane < MMS_NBOX) {
        const unsigned below = (1u << lane) - 1;
        const bool own_nz = (nzm >> lane) & 1;
        if (own_nz || lane == jz) {
            const int pos = __popc(nzm & below) + ((jz >= 0 && jz < lane) ? 1 : 0);
            tok_src[r0 + T + pos] = b * S + T + lane;
            key_add[r0 + T + pos] = own_nz ? 0.f : logf((float)zeros);
        }
        if ((fm >> lane) & 1) {      // label tokens: the first box of every distinct label tuple, key bias log(multiplicity)
            const int pos = T + nnz + (zeros ? 1 : 0) + __popc(fm & below);
            tok_src[r0 + pos] = b * S + T + MMS_NBOX + lane;
            key_add[r0 + pos] = logf((float)m);
        }
    }
}
void launch_lds_pack_plan(const float* feats, const int64_t* labelfeat, int T, int n, int* nz_flags, int* off, int* cnt, int* tok_src,
                          float* key_add, int* rows_dev, hipStream_t st) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_row_nonzero, row_grid((long long)n * MMS_NBOX), dim3(256), 0, st, feats, nz_flags, n * MMS_NBOX);
    const dim3 grid((n + 3) / 4);      // one wavefront per pair
    hipLaunchKernelGGL(k_lds_plan_count, grid, dim3(256), 0, st, nz_flags, labelfeat, T, n, cnt);
    hipLaunchKernelGGL(k_plan_scan, dim3(1), dim3(PLAN_THREADS), 0, st, cnt, n, off, rows_dev);
    hipLaunchKernelGGL(k_lds_plan_fill, grid, dim3(256), 0, st, nz_flags, labelfeat, T, n, off, tok_src, key_add);
}

// lxmert: language stream keeps positions with input_mask != 0 plus position 0 (CLS feeds the pooler);
// vision stream keeps boxes with visual_attention_mask != 0; an all-masked stream is kept whole.
__global__ __launch_bounds__(256) void k_lx_plan_count(const int64_t* input_mask, const float* visual_mask, int T, int n, int* l_cnt,
                                                       int* v_cnt) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= n) return;
    int live = 0;
    for (int s = 0; s < T; ++s) live += input_mask[(long long)b * T + s] != 0;
    l_cnt[b] = live == 0 ? T : live + (input_mask[(long long)b * T] == 0 ? 1 : 0);
    if (!visual_mask) return;   // language-only plan (distinct-query stage)
    int lv = 0;
    for (int j = 0; j < MMS_NBOX; ++j) lv += visual_mask[(long long)b * MMS_NBOX + j] != 0.f;
    v_cnt[b] = lv == 0 ? MMS_NBOX : lv;
}
__global__ __launch_bounds__(256) void k_lx_plan_fill(const int64_t* input_mask, const float* visual_mask, int T, int n, const int* l_off,
                                                      int* l_src, float* l_add, const int* v_off, int* v_src, float* v_add) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= n) return;
    const int V = MMS_NBOX;
    int live = 0;
    for (int s = 0; s < T; ++s) live += input_mask[(long long)b * T + s] != 0;
    int base = l_off[b], c = 0;
    for (int s = 0; s < T; ++s) {
        const int64_t m = input_mask[(long long)b * T + s];
        if (live == 0 || m != 0 || s == 0) {
            l_src[base + c] = b * T + s;
            l_add[base + c] = (1.0f - (float)m) * -10000.f;
            ++c;
        }
    }
    if (!visual_mask) return;
    live = 0;
    for (int j = 0; j < V; ++j) live += visual_mask[(long long)b * V + j] != 0.f;
    base = v_off[b]; c = 0;
    for (int j = 0; j < V; ++j) {
        const float m = visual_mask[(long long)b * V + j];
        if (live == 0 || m != 0.f) {
            v_src[base + c] = b * V + j;
            v_add[base + c] = (1.0f - m) * -10000.f;
            ++c;
        }
    }
}
void launch_lx_pack_plan(const int64_t* input_mask, const float* visual_mask, int T, int n, int* l_off, int* l_cnt, int* l_src,
                         float* l_add, int* l_rows, int* v_off, int* v_cnt, int* v_src, float* v_add, int* v_rows, hipStream_t st) {
    if (n <= 0) return;
    const dim3 grid((n + 255) / 256);
    hipLaunchKernelGGL(k_lx_plan_count, grid, dim3(256), 0, st, input_mask, visual_mask, T, n, l_cnt, v_cnt);
    hipLaunchKernelGGL(k_plan_scan, dim3(1), dim3(PLAN_THREADS), 0, st, l_cnt, n, l_off, l_rows);
    if (visual_mask) hipLaunchKernelGGL(k_plan_scan, dim3(1), dim3(PLAN_THREADS), 0, st, v_cnt, n, v_off, v_rows);
    hipLaunchKernelGGL(k_lx_plan_fill, grid, dim3(256), 0, st, input_mask, visual_mask, T, n, l_off, l_src, l_add, v_off, v_src, v_add);
}

__global__ __launch_bounds__(256) void k_lx_embed_lang_packed(const float* E, const float* pos_tab, const float* type_tab,
                                                              const float* gamma, const float* beta, const int64_t* input_ids,
                                                              int T, int vocab, const int* src, const int* rows_dev,
                                                              int max_rows, bf16* o_hi, bf16* o_lo) {
    const int row = wave_row();
    if (row >= max_rows || row >= *rows_dev) return;
    const int sp = src[row];
    Row x;
    row_load(x, E + clamp_id(input_ids[sp], vocab) * MMS_HIDDEN);
    row_add(x, pos_tab + (sp % T) * MMS_HIDDEN);
    row_add(x, type_tab);
    row_ln(x, gamma, beta);
    row_store_planes(x, plane_ptr(o_hi, (long long)row * MMS_HIDDEN), plane_ptr(o_lo, (long long)row * MMS_HIDDEN));
}
void launch_lx_embed_lang_packed(const float* E, const float* pos_tab, const float* type_tab, const float* gamma,
                                 const float* beta, const int64_t* input_ids, int T, int vocab, const int* src,
                                 const int* rows_dev, int max_rows, bf16* o_hi, bf16* o_lo, hipStream_t st) {
    if (max_rows > 0)
        hipLaunchKernelGGL(k_lx_embed_lang_packed, row_grid(max_rows), dim3(256), 0, st, E, pos_tab, type_tab, gamma, beta, input_ids,
                           T, vocab, src, rows_dev, max_rows, o_hi, o_lo);
}
