"""Experiment: two scorer handles on two HIP streams of ONE process, each on half of the pairs -- do the HBM-bound kernels (LayerNorm,
attention, tile-boundary stores) of one stream hide under the MFMA-bound GEMMs of the other?  With MMS_PP_GRID=128 (lab build) the
persistent GEMMs of each stream take half of the CUs, so both streams are resident at once.
usage (GPU box): [MMS_PP_GRID=128] python tools/dual_stream.py [model] [streams]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kddcup_2020_multimodalitiesrecall_2nd_place_amd import lib
lib.load(lib.LAB_LIB_PATH)
import bench  # noqa: E402
from kddcup_2020_multimodalitiesrecall_2nd_place_amd import scorers, synth, weights  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "zk"
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 2
TOTAL = int(sys.argv[3]) if len(sys.argv) > 3 else 30000      # pairs per step over all streams (mid-size calls: e.g. 256)
cfg = bench.CFGS[name]()
dev = torch.device("cuda", 0)
w = weights.make_weights(cfg)
sc, feed, streams = [], [], []
for i in range(NS):
    s = scorers.make_scorer(cfg, w, device=0)
    per = TOTAL // NS
    ps = synth.make_pairs((per + 29) // 30, 30, tag="/ds%d" % i, with_feats=False).take(slice(0, per))
    feats = bench.device_feats(ps, dev, 7 + i)
    sc.append(s); feed.append(bench.device_feed(name, {name: cfg}, ps, feats, dev)); streams.append(torch.cuda.Stream(dev))
total = (TOTAL // NS) * NS


def run(concurrent, steps=6 if TOTAL > 4000 else 100):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for i in range(NS):
            if concurrent:
                with torch.cuda.stream(streams[i]):
                    sc[i].score_prepared(bench.prepare(sc[i], name, feed[i]))
            else:
                sc[i].score_prepared(bench.prepare(sc[i], name, feed[i]))
    torch.cuda.synchronize()
    return steps * total / (time.perf_counter() - t0)


run(False, 3); run(True, 3)
for _ in range(2):
    print("%s grid %s: sequential %.0f pairs/s | %d streams %.0f pairs/s" % (name, os.environ.get("MMS_PP_GRID", "all"), run(False), NS, run(True)), flush=True)
