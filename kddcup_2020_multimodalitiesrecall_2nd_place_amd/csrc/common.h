// Shared device helpers for libmmscore (gfx950 / CDNA4 only).
//
// Activation format between kernels: "split planes".  A logical fp32 activation x[M][K] is kept in
// HBM as two bf16 planes hi = bf16(x), lo = bf16(x - hi) (same bytes as fp32).  hi+lo carries
// 16-17 significant bits; GEMMs feed MFMA with hi (precision mode 1) or hi and lo as two MFMA
// passes that share the weight fragment (mode 2, the parity mode: SURVEY.md Appendix C row
// "split-bf16 activations").  Residual / LayerNorm / softmax / GELU always run in fp32.
//
// Memory layout of a plane pair ("hl32"): ONE buffer in which 32-element blocks of hi and lo alternate -- logical element i
// (flat index over [M][K], K % 32 == 0) of the hi plane sits at (i >> 5) * 64 + (i & 31), the same element of the lo plane 32
// further.  A 32-wide K stage of one row is then 128 contiguous, 128-byte-aligned bytes [hi 64 B | lo 64 B]: every LDS-DMA piece
// of the GEMM engines (8 rows x 128 B) covers whole cache lines.  With two separate planes a piece was 16 rows x 64 B = sixteen
// HALF lines, each line fetched twice (once per K stage): measured 6-10 % of the big GEMMs' time (profiles/r03a_gemm_layout.txt).
// Every kernel addresses planes through plane_ptr(); `lo == hi + 32` always; leading dimensions and row maps stay LOGICAL.
//
// Weight matrices W[N][K] (bf16, K contiguous) are stored as 16-row x 32-column tiles of 1 KiB, tile (n / 16, k / 32) at
// ((n / 16) * (K / 32) + k / 32) * 512, row-major inside (wtile_off): an LDS-DMA piece of W is one contiguous KiB.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#define MMS_HIDDEN 768
#define MMS_HEADS 12
#define MMS_HEAD_DIM 64
#define MMS_NBOX 10
#define MMS_LABEL_LEN 8
#define MMS_FEAT 2048
#define MMS_LN_EPS 1e-12f

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU_TANH = 2, ACT_GELU_ERF = 3, ACT_TANH = 4 };
enum { OUT_F32 = 0, OUT_PLANES = 1, OUT_F8 = 2, OUT_H3 = 3 };

// "h3" operand format of precision mode 5 (gemm_mx.hip): an activation x as an fp16 high plane h = fp16(x) [rows][ld] and an e4m3 low
// plane l = e4m3((x - h) * 2^MMS_H3_SA) [rows][ld] bytes, both row-major: 3 bytes per element, 11 + 4 significant bits.  The constant
// exponent keeps the residual of |x| in [2^-8, 64] inside e4m3's normal range (|x - h| <= 2^-11 |x|); smaller values lose only bits
// below 2^-24 of the row scale, larger ones saturate towards plain fp16 precision.
#define MMS_H3_SA 13
typedef _Float16 f16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;

__device__ __forceinline__ void split_bf16(float v, bf16& hi, bf16& lo) {
    hi = (bf16)v;                 // v_cvt_pk_bf16_f32: round-to-nearest-even
    lo = (bf16)(v - (float)hi);   // exact remainder, rounded once
}

__device__ __forceinline__ float join_bf16(bf16 hi, bf16 lo) { return (float)hi + (float)lo; }
// hl32 plane addressing: `base` is the hi (or lo = hi + 32) pointer of a plane pair, `i` a logical flat element index.  Vector
// accesses of 4 / 8 elements stay inside one 32-element block as long as i is 4- / 8-aligned.
#define MMS_PLANE_LO 32
__host__ __device__ __forceinline__ long long plane_off(long long i) { return ((i >> 5) << 6) + (i & 31); }
template <typename T> __host__ __device__ __forceinline__ T* plane_ptr(T* base, long long i) { return base + plane_off(i); }
// tiled weights: element (n, k) of W[N][K]
__host__ __device__ __forceinline__ long long wtile_off(long long n, long long k, long long K) {
    return (((n >> 4) * (K >> 5) + (k >> 5)) << 9) + ((n & 15) << 5) + (k & 31);
}

// fp32 -> OCP e4m3fn, round-to-nearest-even, saturating at +-448 (precision mode 4).  v_cvt_pk_fp8_f32 packs two values into
// one 16-bit half of a dword; the clamp makes the result independent of the instruction's overflow convention.
__device__ __forceinline__ unsigned pack4_f8(float a, float b, float c, float d) {
    a = fminf(fmaxf(a, -448.f), 448.f); b = fminf(fmaxf(b, -448.f), 448.f);
    c = fminf(fmaxf(c, -448.f), 448.f); d = fminf(fmaxf(d, -448.f), 448.f);
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return (unsigned)w;
}

// h3 split of four values: hi = fp16 (RNE), lo = the four e4m3 bytes of (v - hi) * 2^SA packed into one dword
__device__ __forceinline__ void split_h3(const float (&v)[4], f16x4& hi, unsigned& lo) {
    float r[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { hi[e] = (f16)v[e]; r[e] = (v[e] - (float)hi[e]) * (float)(1 << MMS_H3_SA); }
    lo = pack4_f8(r[0], r[1], r[2], r[3]);
}


// Fast transcendental forms for the GEMM epilogues: v_exp_f32 / v_rcp_f32 directly (both ~1 ulp; `__fdividef` and `x / y` compile
// to the full IEEE division sequence, 10 instructions per element).  Absolute error <= ~2e-7, entering GELU only through a factor
// in [0, 1]: far inside the 1e-3 logit budget.  A 256x256 tile's GELU + plane split is ~1400 VALU instructions per lane with
// these forms against ~2800 with libm-style division -- the FFN-up epilogue is VALU-bound, so that is time.
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }      // 0 for very negative x, inf for large x
__device__ __forceinline__ float fast_tanh(float x) {
    const float e = fast_exp2(2.8853900817779268f * x);        // exp(2x); inf for large x -> 1, 0 for very negative x -> -1
    return 1.0f - 2.0f * fast_rcp(e + 1.0f);
}
__device__ __forceinline__ float fast_erf(float x) {  // Abramowitz & Stegun 7.1.26, |err| <= 1.5e-7
    const float ax = fabsf(x);
    const float t = fast_rcp(1.0f + 0.3275911f * ax);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float r = 1.0f - poly * fast_exp2(-1.4426950408889634f * ax * ax);
    return copysignf(r, x);
}

__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
        case ACT_RELU: return fmaxf(v, 0.0f);
        case ACT_GELU_TANH: {  // pixelbert.py:326-328: v * 0.5 * (1 + tanh(c (v + 0.044715 v^3))) = v - v / (exp(2u) + 1)
            const float k1 = 2.0f * 0.7978845608028654f * 1.4426950408889634f, k2 = k1 * 0.044715f;     // exp(2u) = exp2(v (k1 + k2 v^2))
            // the fused multiply-adds are WRITTEN, operation for operation as in gelu_tanh2() below: the small-call routes (k_splitk_reduce: this scalar form) and the
            // tile engines (packed form) are bit-identical by construction, not by what -ffp-contract happens to fuse
            const float t = v * k2;
            const float u = __builtin_fmaf(v, t, k1);
            const float r = fast_rcp(fast_exp2(v * u) + 1.0f);
            return __builtin_fmaf(-v, r, v);
        }
        case ACT_GELU_ERF:     // lxrt/modeling.py:119
            return v * 0.5f * (1.0f + fast_erf(v * 0.70710678118654752f));
        case ACT_TANH: return fast_tanh(v);
        default: return v;
    }
}

// Four accumulator values through an activation.  tanh-GELU on PAIRS: the polynomial argument, the + 1 and the final v - v r as packed fp32
// operations (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32: two values per issue slot) -- the same operations in the same order as apply_act(),
// bit-identical results, 9 instead of 13 VALU instructions per pair.  The FFN-up epilogue (128 values per lane and tile, two waves per SIMD)
// is VALU-bound with the matrix pipe idle: every instruction less is time and energy.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t gelu_tanh2(f32x2_t v) {
    const float k1 = 2.0f * 0.7978845608028654f * 1.4426950408889634f, k2 = k1 * 0.044715f;
    const f32x2_t t = v * f32x2_t{k2, k2};
    const f32x2_t u = __builtin_elementwise_fma(v, t, f32x2_t{k1, k1});
    const f32x2_t w = v * u;
    f32x2_t e = {fast_exp2(w.x), fast_exp2(w.y)};
    e = e + f32x2_t{1.0f, 1.0f};
    const f32x2_t r = {fast_rcp(e.x), fast_rcp(e.y)};
    return __builtin_elementwise_fma(-v, r, v);
}
template <int ACT> __device__ __forceinline__ f32x4 apply_act4(f32x4 v) {
    if constexpr (ACT == ACT_GELU_TANH) {     // (erf-GELU: the compiler already packs its polynomial; an explicit pair form came out 0.3 % longer)
        const f32x2_t a = gelu_tanh2(f32x2_t{v[0], v[1]}), b = gelu_tanh2(f32x2_t{v[2], v[3]});
        return f32x4{a.x, a.y, b.x, b.y};
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], ACT);
        return v;
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Reductions over the four 16-lane rows of a wave (lanes l, l ^ 16, l ^ 32, l ^ 48; every lane gets the result) on v_permlane16_swap /
// v_permlane32_swap: VALU, where __shfl_xor(v, 16 / 32) is a ds_bpermute round trip through the LDS crossbar.  Same operations in the same
// order as  v = op(v, shfl_xor(v, 16)); v = op(v, shfl_xor(v, 32))  (op commutative), so results are bit-identical to the shuffle form.
__device__ __forceinline__ float rows4_sum(float v) {
    unsigned u = __float_as_uint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);      // {rows 0 0 2 2, rows 1 1 3 3}
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    u = __float_as_uint(v);
    const auto b = __builtin_amdgcn_permlane32_swap(u, u, false, false);      // {low low, high high}
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ float rows4_max(float v) {
    unsigned u = __float_as_uint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    u = __float_as_uint(v);
    const auto b = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

// row remap: logical row r -> physical row (r / grp) * stride + off + r % grp   (grp == 0: identity)
struct RowMap {
    int grp, stride, off;
    __device__ __forceinline__ long long operator()(int r) const {
        return grp ? (long long)(r / grp) * stride + off + (r % grp) : (long long)r;
    }
};
