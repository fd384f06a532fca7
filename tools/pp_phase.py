"""Per-phase barrier stamps of the persistent ping-pong GEMM (lab variant 264): how long each wave row spends between two barriers doing
its own work (load section / MFMA section) and how long it waits at the barrier for the other row.  usage (GPU box): python tools/pp_phase.py [N K]"""
import ctypes as C
import os
import sys

os.environ.setdefault("MMS_GEMM_DIAG", "1")
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kddcup_2020_multimodalitiesrecall_2nd_place_amd import lib  # noqa: E402

N, K = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2304, 768)
l = lib.load(os.environ.get("MMS_LAB_LIB", lib.LAB_LIB_PATH))
ms = C.c_float(0)
assert l.mms_dbg_gemm_bench(122880, N, K, 2, 0, 0, 0, 264, 1, C.byref(ms)) == 0, l.mms_global_error()
t = np.fromfile("/tmp/pp_phase.bin", np.uint64).reshape(2, 512).astype(np.int64)
for w, name in ((0, "wave 0 (row 0)"), (1, "wave 4 (row 1)")):
    s = t[w][t[w] > 0]
    s = s - s[0]
    arrive, leave = s[0::2], s[1::2]            # every barrier: stamp before and after
    n = min(len(arrive), len(leave))
    arrive, leave = arrive[:n], leave[:n]
    wait = leave - arrive                        # time parked at the barrier
    work = arrive[1:] - leave[:-1]               # own work between two barriers
    # sections alternate: after an odd barrier of a phase comes the MFMA section, after the even one the next phase's load section
    k0 = 4                                       # skip the prologue barriers
    wk, wt = work[k0:k0 + 8 * 20], wait[k0 + 1:k0 + 1 + 8 * 20]
    a, b = wk[0::2], wk[1::2]
    print("%s: %d barriers; per K = %d tile %d cycles" % (name, n, K, s[-1]))
    print("   work between barriers, alternating sections: %.0f / %.0f cycles (median);  wait at the barrier after them: %.0f / %.0f" %
          (np.median(a), np.median(b), np.median(wt[0::2]), np.median(wt[1::2])))
    ph = (wk[:160].reshape(-1, 8), wt[:160].reshape(-1, 8))
    print("   per stage position (8 sections of a stage), median work :", " ".join("%4.0f" % x for x in np.median(ph[0], 0)))
    print("   per stage position,                         median wait :", " ".join("%4.0f" % x for x in np.median(ph[1], 0)))
