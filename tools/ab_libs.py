"""GPU box: A/B of two lab builds of libmmscore on one GEMM shape -- bitwise comparison of the results of mms_dbg_gemm on the same random
operands, then alternating timings of mms_dbg_gemm_bench.
python tools/ab_libs.py <lib A> <lib B> [N K act planes] [M] [variant]      (defaults: FFN-up of the bench batch, 3072 768 2 1, M = 454656, 26)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from kddcup_2020_multimodalitiesrecall_2nd_place_amd import lib

pa, pb = sys.argv[1], sys.argv[2]
N, K, act, planes = (int(x) for x in sys.argv[3:7]) if len(sys.argv) > 6 else (3072, 768, 2, 1)
M = int(sys.argv[7]) if len(sys.argv) > 7 else 454656
variant = int(sys.argv[8]) if len(sys.argv) > 8 else 26
libs = [lib.load(pa), lib.load(pb)]

# --- same operands through both builds: the outputs must be the same bits ---
Ms = 1024
g = torch.Generator(device="cuda").manual_seed(5)
a = torch.randn((Ms, K), device="cuda", generator=g)
w = torch.randn((N, K), device="cuda", generator=g) * 0.05
bias = torch.randn((N,), device="cuda", generator=g)
outs = []
for l in libs:
    c = torch.empty((Ms, N), device="cuda")
    rc = l.mms_dbg_gemm(C.c_void_p(a.data_ptr()), Ms, K, K, C.c_void_p(w.data_ptr()), N, C.c_void_p(bias.data_ptr()), None, act, 2, planes, variant,
                        C.c_void_p(c.data_ptr()), None)
    assert rc == 0, l.mms_global_error()
    torch.cuda.synchronize()
    outs.append(c)
print("bitwise equal:", bool(torch.equal(outs[0], outs[1])), " max |diff| %.3g" % float((outs[0] - outs[1]).abs().max()))

for rnd in range(4):
    row = []
    for name, l in zip("AB", libs):
        ms = C.c_float(0)
        rc = l.mms_dbg_gemm_bench(M, N, K, 2, act, planes, 0, variant, 10, C.byref(ms))
        assert rc == 0, l.mms_global_error()
        row.append("%s %7.3f ms %6.0f TFLOP/s" % (name, ms.value, 2.0 * M * N * K / ms.value / 1e9))
    print(" | ".join(row), flush=True)
