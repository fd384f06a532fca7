"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  OCP e4m3fn arithmetic of precision mode 4 (BASELINE.json config 5: "fp8 MFMA
weights"), restated in numpy so the fp8 GEMM kernel can be checked bit-for-bit on its operands.

There is no reference code for this mode (the reference is fp32 throughout); what is restated here is the number format
(OCP 8-bit floating point, E4M3: 1 sign, 4 exponent bits with bias 7, 3 mantissa bits, no infinities, max finite 448,
subnormal step 2^-9) and this build's own quantisation rules (csrc/rowops.hip k_quant_rows_f8 / pack4_f8):

* activations: round-to-nearest-even to e4m3, saturating at +-448;
* weights: per output channel, scale = the smallest power of two with max|w| / scale <= 448; bytes = e4m3(w / scale).
"""
import numpy as np


def e4m3_round(x):
    """fp32/fp64 array -> the e4m3-representable value nearest to it (ties to even), saturating at +-448; returned as float64."""
    x = np.asarray(x, np.float64)
    s = np.sign(x)
    a = np.minimum(np.abs(x), 448.0)
    # exponent of the representable grid: normals 2^(e-3) for a in [2^e, 2^(e+1)), e >= -6; subnormals share the 2^-9 grid
    with np.errstate(divide="ignore"):
        e = np.floor(np.log2(np.where(a > 0, a, 1.0)))
    e = np.maximum(e, -6.0)
    step = np.exp2(e - 3.0)
    q = np.rint(a / step) * step          # np.rint rounds half to even; a / step is exact (power-of-two step)
    return s * np.minimum(q, 448.0)


def quant_weight_rows(w):
    """[N,K] -> (dequantised weights actually multiplied, i.e. e4m3(w / scale) * scale, scale[N])."""
    w = np.asarray(w, np.float64)
    m = np.abs(w).max(1)
    fr, ex = np.frexp(m)                   # m = fr * 2^ex, fr in [0.5, 1)
    e = np.where(m > 0, np.where(fr <= 0.875, ex - 9, ex - 8), 0)
    scale = np.exp2(e.astype(np.float64))
    return e4m3_round(w / scale[:, None]) * scale[:, None], scale


# ---- precision mode 5 (csrc/gemm_mx.hip): the h3 operand format and its weight copies, restated for the kernel tests ----
H3_SA = 13          # csrc/common.h MMS_H3_SA


def h3_split(x):
    """fp32 array -> (hi, lo): hi = fp16(x) (RNE), lo = e4m3((x - hi) * 2^SA) * 2^-SA, both as float64 -- what the two MFMA passes of
    precision mode 5 multiply (common.h split_h3: the subtraction and the scaling are exact in fp32)."""
    x = np.asarray(x, np.float32)
    hi = x.astype(np.float16).astype(np.float64)
    r = (x.astype(np.float64) - hi).astype(np.float32).astype(np.float64)      # exact in fp32 (|r| <= half an fp16 ulp of x)
    return hi, e4m3_round(r * 2.0 ** H3_SA) * 2.0 ** -H3_SA


def mx_weight_rows(w):
    """[N,K] bf16-exact weights -> (w16, w8): the fp16 copy (exact up to fp16's range below 2^-28 of the row maximum) and the e4m3 copy
    (quant_weight_rows), both dequantised to float64 (rowops.hip k_prep_w_mx)."""
    w = np.asarray(w, np.float64)
    m = np.abs(w).max(1)
    _, ex = np.frexp(m)
    e16 = np.where(m > 0, 14 - ex, 0).astype(np.float64)
    w16 = (w * np.exp2(e16)[:, None]).astype(np.float16).astype(np.float64) * np.exp2(-e16)[:, None]
    return w16, quant_weight_rows(w)[0]


def gemm_mx_ref(a, w):
    """what precision mode 5 computes for a @ w.T (fp64 accumulation): high pass on (fp16 a, fp16 w) + low pass on the e4m3 residual of a
    and the e4m3 copy of w."""
    ah, al = h3_split(a)
    w16, w8 = mx_weight_rows(w)
    return ah @ w16.T + al @ w8.T
