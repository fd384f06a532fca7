"""Run ONE GEMM shape/variant a few times (for rocprofv3 --pmc).  args: N K act planes resid variant nsplit [iters]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kddcup_2020_multimodalitiesrecall_2nd_place_amd import lib  # noqa: E402

M = int(os.environ.get("GB_M", 122880))
N, K, act, planes, resid, variant, nsplit = [int(x) for x in sys.argv[1:8]]
iters = int(sys.argv[8]) if len(sys.argv) > 8 else 3
l = lib.load(lib.LAB_LIB_PATH)   # lab build: `make -C kddcup_2020_multimodalitiesrecall_2nd_place_amd/csrc lab`
ms = C.c_float(0)
rc = l.mms_dbg_gemm_bench(M, N, K, nsplit, act, planes, resid, variant, iters, C.byref(ms))
assert rc == 0, l.mms_global_error()
print("N=%d K=%d variant=%d nsplit=%d: %.3f ms %.0f TF" % (N, K, variant, nsplit, ms.value, 2.0 * M * N * K / ms.value / 1e9))
