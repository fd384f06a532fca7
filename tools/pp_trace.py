"""Per-workgroup timeline of one gemm_pp launch (diagnostic variant 232): prologue / main loop / epilogue durations and how
synchronised the workgroups' epilogues are.  usage (GPU box): python tools/pp_trace.py [N K]"""
import ctypes as C
import os
import sys

os.environ.setdefault("MMS_GEMM_DIAG", "1")   # this tool may run the timing-only kernel variants

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kddcup_2020_multimodalitiesrecall_2nd_place_amd import lib  # noqa: E402

N, K = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2304, 768)
VARIANT = 232
M = 122880
l = lib.load(lib.LAB_LIB_PATH)   # lab build: `make -C kddcup_2020_multimodalitiesrecall_2nd_place_amd/csrc lab`
ms = C.c_float(0)
assert l.mms_dbg_gemm_bench(M, N, K, 2, 0, 0, 0, VARIANT, 1, C.byref(ms)) == 0, l.mms_global_error()
t = np.fromfile("/tmp/pp_trace.bin", np.uint64).reshape(-1, 5).astype(np.int64)
t = t[t[:, 3] > 0]
t0 = t[:, 0].min()
us = (t[:, :4] - t0) / 100.0                     # 100 MHz ticks -> us
pro, main, epi = us[:, 1] - us[:, 0], us[:, 2] - us[:, 1], us[:, 3] - us[:, 2]
print("N=%d K=%d: %d workgroups, kernel span %.0f us" % (N, K, len(t), us[:, 3].max()))
for name, x in (("prologue", pro), ("main loop", main), ("epilogue (until stores acknowledged)", epi)):
    print("  %-38s median %6.1f us   p10 %6.1f   p90 %6.1f" % (name, np.median(x), np.percentile(x, 10), np.percentile(x, 90)))
# how bunched are the epilogues?  histogram of main-loop end times modulo the median tile period
period = np.median(us[:, 3] - us[:, 0])
order = np.argsort(us[:, 2])
ends = us[order, 2]
for r in range(min(6, int(ends.max() / period))):
    sel = ends[(ends > r * period) & (ends <= (r + 1) * period)]
    if len(sel):
        print("  main-loop ends in [%5.0f, %5.0f) us: %4d workgroups, spread p5..p95 = %5.1f us" % (r * period, (r + 1) * period, len(sel), np.percentile(sel, 95) - np.percentile(sel, 5)))
