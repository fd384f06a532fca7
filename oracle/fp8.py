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
