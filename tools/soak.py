"""GPU box: 400 back-to-back scoring calls on the headline workload -- throughput per block of 50 calls and free device memory
before / after (leak and drift check).  python tools/soak.py [calls]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from kddcup_2020_multimodalitiesrecall_2nd_place_amd import synth

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 400


class A:
    precision = 2; chunk = 0; fp32_weights = False; fuse_ln = False; dense = False; all_boxes = False


dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
scorer, members = bench.make_members("zk", A, 0)
cfgs = {n: m[0] for n, m in members.items()}
ps = synth.make_pairs(1000, 30, tag="/bench0", with_feats=False)
feats = bench.device_feats(ps, dev, 20200823)
feed = bench.device_feed("zk", cfgs, ps, feats, dev)
for _ in range(3):
    scorer.score_prepared(bench.prepare(scorer, "zk", feed))
torch.cuda.synchronize()
free0 = torch.cuda.mem_get_info()[0]
ref = None
t0 = time.perf_counter()
for i in range(calls):
    _, probs = scorer.score_prepared(bench.prepare(scorer, "zk", feed))
    if (i + 1) % 50 == 0:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        p = probs[:, 1].clone()
        same = True if ref is None else bool(torch.equal(p, ref))
        ref = p if ref is None else ref
        print("calls %4d  %8.0f pairs/s  free %.2f GB  bitwise-stable %s" % (i + 1, ps.n * 50 / (t1 - t0), torch.cuda.mem_get_info()[0] / 1e9, same), flush=True)
        t0 = time.perf_counter()
print("free memory before %.3f GB, after %.3f GB" % (free0 / 1e9, torch.cuda.mem_get_info()[0] / 1e9))
