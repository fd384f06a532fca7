"""Per-call latency of one call surface at one batch size, with the host-side share split out (GPU box):
  python tools/small_batch.py zk 256 [calls]         wall time per call, time inside prepare() (host conversions), time inside the C call
Run under `rocprofv3 --kernel-trace --stats` to get the sum of kernel durations next to it (launch-bound vs execution-bound)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from kddcup_2020_multimodalitiesrecall_2nd_place_amd import scorers, synth, weights

name, B = sys.argv[1], int(sys.argv[2])
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 100
dev = torch.device("cuda", 0)
cfg = bench.CFGS[name]()
w = weights.make_weights(cfg)
s = scorers.make_scorer(cfg, w, precision=2)
ps = synth.make_pairs((B + 29) // 30, 30, tag="/bench0", with_feats=False).take(slice(0, B))
feats = bench.device_feats(ps, dev, 20200823)
fd = bench.device_feed(name, {name: cfg}, ps, feats, dev)
for _ in range(3):
    s.score_prepared(bench.prepare(s, name, fd))
torch.cuda.synchronize()
tp = tc = 0.0
t0 = time.perf_counter()
for _ in range(calls):
    a = time.perf_counter()
    p = bench.prepare(s, name, fd)
    b = time.perf_counter()
    s.score_prepared(p)
    c = time.perf_counter()
    torch.cuda.synchronize()
    tp += b - a; tc += c - b
dt = time.perf_counter() - t0
print("%s B=%d: %.3f ms per call (prepare %.3f ms, C call returns after %.3f ms, rest = waiting for the GPU) -> %.0f pairs/s"
      % (name, B, dt / calls * 1e3, tp / calls * 1e3, tc / calls * 1e3, B * calls / dt))
# the same calls enqueued back to back, ONE synchronisation at the end (a caller that does not read each result before the next call)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(calls):
    s.score_prepared(bench.prepare(s, name, fd))
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("%s B=%d, %d calls enqueued back to back: %.3f ms per call -> %.0f pairs/s" % (name, B, calls, dt / calls * 1e3, B * calls / dt))
