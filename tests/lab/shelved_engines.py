"""Kernel tests of the measured-and-shelved engines that live in the LAB build only (libmmscore_lab.so: gemm_dw.hip = variant 28, the
fp16 + MX-scaled-e4m3 "1.5 pass" kernel of gemm_mx.hip).  Not collected by name: tests/test_lab_engines_gpu.py runs this file in a
process that loaded the lab library (tools/pytest_lab.py)."""
import numpy as np
import pytest
import torch

from kddcup_2020_multimodalitiesrecall_2nd_place_amd import lib, weights

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", [(300, 768, 2304, 0, False, False), (257, 768, 3072, 2, False, True), (64, 3072, 768, 0, True, False),
                                  (17000, 768, 768, 0, True, False), (700, 128, 512, 3, False, True)])
def test_gemm_dw_matches_fp64(case):
    """gemm_dw.hip (128 x 256 tiles, two 4-wave workgroups per CU) through mms_dbg_gemm's per-call engine choice (variant 28)."""
    from helpers import act_ref
    M, K, N, act, resid, planes = case
    l = lib.load()
    a = weights.normal("dw/a/%d/%d" % (M, K), (M, K), 1)
    w = weights.round_to_bf16(weights.normal("dw/w/%d/%d" % (N, K), (N, K), 1, 1.0 / np.sqrt(K)))
    bias = weights.normal("dw/b/%d" % N, (N,), 1, 0.1)
    r = weights.normal("dw/r/%d/%d" % (M, N), (M, N), 1) if resid else None
    dev = lambda x: torch.as_tensor(np.ascontiguousarray(x)).cuda()
    da, dw, db = dev(a), dev(w), dev(bias)
    dr = dev(r) if resid else None
    out = torch.empty((M, N), device="cuda", dtype=torch.float32)
    rc = l.mms_dbg_gemm(da.data_ptr(), M, K, K, dw.data_ptr(), N, db.data_ptr(), dr.data_ptr() if resid else None, act, 2, int(planes), 28,
                        out.data_ptr(), None)
    assert rc == 0, l.mms_global_error()
    ref = a.astype(np.float64) @ w.astype(np.float64).T + bias
    if resid:
        ref = ref + r
    ref = act_ref(ref, act)
    err = np.abs(out.cpu().numpy() - ref).max() / np.abs(ref).max()
    assert err < 3e-5, (case, err)


MX_CASES = [(300, 768, 768, 0, False), (16640 + 17, 768, 2304, 0, False), (2048, 3072, 768, 0, False), (4500, 768, 3072, 2, True),
            (1000, 1024, 512, 3, True), (256, 256, 256, 0, False)]


@pytest.mark.parametrize("case", MX_CASES)
def test_gemm_mx_matches_numpy_on_its_operands_and_the_true_product(case):
    """gemm_mx.hip (precision mode 5: fp16 high pass + MX-scaled e4m3 low pass with per-channel hardware scales) against (i) numpy on
    exactly the operands it multiplies (oracle/fp8.py gemm_mx_ref) -- agreement to fp32 accumulation noise -- and (ii) the fp64 product
    of the unsplit operands, where it must sit in the 2^-15 class (a single fp16 pass alone: 2^-11)."""
    from helpers import act_ref
    from kddcup_2020_multimodalitiesrecall_2nd_place_amd import lib
    from oracle import fp8 as F8
    M, K, N, act, out_h3 = case
    l = lib.load()
    a = weights.normal("mx/a/%d/%d" % (M, K), (M, K), 1)
    a[0, :6] = [70000.0, -3e-6, 100.0, 0.0, 2 ** -20, -65504.0]                 # fp16 overflow / underflow corners in one row
    a[0, :6] = np.clip(a[0, :6], -60000, 60000)
    w = weights.round_to_bf16(weights.normal("mx/w/%d/%d" % (N, K), (N, K), 1, 1.0 / np.sqrt(K)))
    w[3] *= 2.0 ** -9                                                            # a row with a much smaller scale
    w[5, ::7] = 0.0
    bias = weights.normal("mx/b/%d" % N, (N,), 1, 0.1)
    out = torch.empty((M, N), device="cuda", dtype=torch.float32)
    da, dw, db = (torch.as_tensor(np.ascontiguousarray(x)).cuda() for x in (a, w, bias))
    rc = l.mms_dbg_gemm_mx(da.data_ptr(), M, K, dw.data_ptr(), N, db.data_ptr(), act, int(out_h3), out.data_ptr(), None)
    assert rc == 0, l.mms_global_error()
    got = out.cpu().numpy().astype(np.float64)
    pre = F8.gemm_mx_ref(a, w) + bias
    ref = act_ref(pre, act)
    true = act_ref(a.astype(np.float64) @ w.astype(np.float64).T + bias, act)
    hi_only = act_ref(a.astype(np.float16).astype(np.float64) @ w.astype(np.float64).T + bias, act)
    scale = np.abs(true[1:]).max()
    e_ops, e_true, e_hi = (np.abs(got - ref)[1:].max() / scale, np.abs(got - true)[1:].max() / scale, np.abs(hi_only - true)[1:].max() / scale)
    print("\n[gemm_mx %s] vs its own operands %.2e   vs the true product %.2e   (fp16 pass alone %.2e)" % (case, e_ops, e_true, e_hi))
    tol_ops = 3e-5 if out_h3 else 2e-5           # out_h3: plus the h3 rounding of the output itself (2^-15 relative)
    assert e_ops < tol_ops, e_ops
    assert e_true < 6e-5 and e_true < e_hi / 3
    assert np.isfinite(got[0]).all()
