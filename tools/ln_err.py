"""Lab book: error of the LayerNorm-fused GEMM (mms_dbg_gemm_ln) and of the two-kernel route against an fp64 LayerNorm (GPU box)."""
import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kddcup_2020_multimodalitiesrecall_2nd_place_amd import lib, weights
l = lib.load()
def run(M, K, rscale, rmean, seed=1):
    a = weights.normal("e/a%d" % seed, (M, K), 1); w = weights.round_to_bf16(weights.normal("e/w%d" % K, (768, K), 1, 1/np.sqrt(K)))
    bias = weights.normal("e/b", (768,), 1, 0.1); r = weights.normal("e/r", (M, 768), 1) * rscale + rmean
    gam = weights.normal("e/g", (768,), 1, 0.1, 1.0); bet = weights.normal("e/be", (768,), 1, 0.1)
    d = lambda x: torch.as_tensor(np.ascontiguousarray(x)).cuda()
    da, dw, db, dr, dg, dbe = d(a), d(w), d(bias), d(r), d(gam), d(bet)
    out = torch.empty((M, 768), device="cuda"); mode = C.c_int32(0)
    assert l.mms_dbg_gemm_ln(da.data_ptr(), M, K, dw.data_ptr(), db.data_ptr(), dr.data_ptr(), dg.data_ptr(), dbe.data_ptr(), 0, out.data_ptr(), C.byref(mode), None) == 0
    v = a.astype(np.float64) @ w.astype(np.float64).T + bias + r
    mu = v.mean(1, keepdims=True); ref = (v - mu) / np.sqrt(((v - mu) ** 2).mean(1, keepdims=True) + 1e-12) * gam + bet
    got = out.cpu().numpy().astype(np.float64)
    # unfused route: gemm (fp32 out, residual added by LN kernel path is internal) -> emulate: gemm + resid via dbg_gemm with resid, then dbg_layernorm
    t = torch.empty((M, 768), device="cuda")
    assert l.mms_dbg_gemm(da.data_ptr(), M, K, K, dw.data_ptr(), 768, db.data_ptr(), dr.data_ptr(), 0, 2, 0, t.data_ptr(), None) == 0
    o2 = torch.empty((M, 768), device="cuda")
    assert l.mms_dbg_layernorm(t.data_ptr(), dg.data_ptr(), dbe.data_ptr(), M, o2.data_ptr(), None) == 0
    got2 = o2.cpu().numpy().astype(np.float64)
    e1 = np.abs(got - ref); e2 = np.abs(got2 - ref)
    print("M %d K %d resid scale %g mean %g mode %d | fused max %.2e rms %.2e | unfused max %.2e rms %.2e | worst row %d" % (M, K, rscale, rmean, mode.value, e1.max(), np.sqrt((e1**2).mean()), e2.max(), np.sqrt((e2**2).mean()), np.unravel_index(e1.argmax(), e1.shape)[0]))
run(20000, 768, 1.0, 0.3); run(20000, 768, 5.0, 0.0); run(20000, 768, 1.0, 5.0); run(20000, 3072, 1.0, 0.3); run(16400, 768, 0.2, 0.0)
