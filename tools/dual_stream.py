"""Experiment: do two scorer handles on two HIP streams of ONE process overlap usefully (GEMM epilogue bursts and the
HBM-bound LN / attention kernels of one stream under the MFMA-bound GEMMs of the other)?
usage (GPU box): python tools/dual_stream.py [model]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from kddcup_2020_multimodalitiesrecall_2nd_place_amd import scorers, synth, weights  # noqa: E402
from kddcup_2020_multimodalitiesrecall_2nd_place_amd.config import LdsConfig, LxmertConfig, ZkConfig  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "zk"
cfg = {"zk": ZkConfig(), "lds": LdsConfig(), "lxmert": LxmertConfig()}[name]
dev = torch.device("cuda", 0)
w = weights.make_weights(cfg)
NS = int(os.environ.get("DS_STREAMS", 2))
sc, prep, streams = [], [], []
for i in range(NS):
    s = scorers.make_scorer(cfg, w, device=0)
    ps = synth.make_pairs(1000 // NS, 30, tag="/ds%d" % i, with_feats=False)
    feats = bench.device_feats(ps, dev, 7 + i)
    sc.append(s); prep.append(bench.prepare(s, cfg, ps, feats)); streams.append(torch.cuda.Stream(dev))
n_pairs = sum(p_.n if hasattr(p_, "n") else 0 for p_ in prep) or (1000 // NS) * 30 * NS


def run(concurrent, steps=6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for i in range(NS):
            if concurrent:
                with torch.cuda.stream(streams[i]):
                    sc[i].score_prepared(prep[i])
            else:
                sc[i].score_prepared(prep[i])
    torch.cuda.synchronize()
    return steps * (1000 // NS) * 30 * NS / (time.perf_counter() - t0)


run(False, 2); run(True, 2)
for _ in range(2):
    print("%s: sequential %.0f pairs/s | %d streams %.0f pairs/s" % (name, run(False), NS, run(True)), flush=True)
