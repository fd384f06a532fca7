"""Where does a tile of the LayerNorm-fused GEMM (gemm_pp_ln.h) spend its time?  Lab build only: per-tile wall_clock64 stamps (100 MHz)
at loop start / epilogue entry / row sums in LDS / partner granules seen / statistics ready / stores acknowledged.
usage (GPU box): python tools/ln_trace.py [M] [K]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from kddcup_2020_multimodalitiesrecall_2nd_place_amd import lib

l = lib.load(lib.LAB_LIB_PATH)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 454656
K = int(sys.argv[2]) if len(sys.argv) > 2 else 768
g = torch.Generator(device="cuda"); g.manual_seed(1)
a = torch.randn((M, K), device="cuda", generator=g)
w = (torch.randn((768, K), device="cuda", generator=g) / K ** 0.5).bfloat16().float()
bias = torch.randn(768, device="cuda", generator=g) * 0.1
r = torch.randn((M, 768), device="cuda", generator=g)
gam = torch.ones(768, device="cuda"); bet = torch.zeros(768, device="cuda")
out = torch.empty((M, 768), device="cuda")
nbm = (M + 255) // 256
nvirt = 3 * nbm            # one stamp row per tile (row panel bm, column tile bn): index bm * 3 + bn
buf = torch.zeros((nvirt, 6), dtype=torch.int64, device="cuda")
l.mms_lab_ln_trace.argtypes = [C.c_void_p]
mode = C.c_int32(0)
for it in range(2):
    buf.zero_()
    l.mms_lab_ln_trace(buf.data_ptr())
    rc = l.mms_dbg_gemm_ln(a.data_ptr(), M, K, w.data_ptr(), bias.data_ptr(), r.data_ptr(), gam.data_ptr(), bet.data_ptr(), 0, out.data_ptr(), C.byref(mode), None)
    assert rc == 0
t = buf.cpu().numpy().astype(np.float64)
t = t[t[:, 5] > 0]
us = lambda x: x / 100.0
print("mode", mode.value, "tiles", len(t), "launch span %.1f us" % us(t[:, 5].max() - t[:, 0].min()))
names = ["main loop (start -> epilogue entry)", "bias + row sums -> LDS (1st barrier)", "publish + wait for partners", "read granules + statistics", "normalise + stores acknowledged"]
for k in range(5):
    d = us(t[:, k + 1] - t[:, k])
    print("%-42s mean %7.2f us  p50 %7.2f  p95 %7.2f  max %7.2f" % (names[k], d.mean(), np.median(d), np.percentile(d, 95), d.max()))
print("tile total mean %.2f us" % us(t[:, 5] - t[:, 0]).mean())
PR = 85                      # panels per persistent round on 256 CUs (gemm_pp.hip: 80 triples inside the XCDs + 5 across)
rounds = (np.arange(nvirt)[buf.cpu().numpy()[:, 5] > 0] // 3) // PR
tot = us(t[:, 5] - t[:, 0])
print("per persistent round: tiles, mean tile us, max tile us, mean main-loop us, mean wait us")
for rr in range(int(rounds.max()) + 1):
    m = rounds == rr
    print("  round %2d  %4d  %7.1f  %7.1f  %7.1f  %7.1f   start spread %.1f us" % (rr, m.sum(), tot[m].mean(), tot[m].max(), us(t[m, 1] - t[m, 0]).mean(),
          us(t[m, 3] - t[m, 2]).mean(), us(t[m, 0].max() - t[m, 0].min())))
