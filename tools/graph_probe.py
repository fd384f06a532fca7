"""GPU box: does replaying a small scoring call as a captured HIP graph beat enqueueing its ~110 launches one by one?
python tools/graph_probe.py [zk|lds|lxmert] [pairs]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from kddcup_2020_multimodalitiesrecall_2nd_place_amd import scorers, synth, weights

name = sys.argv[1] if len(sys.argv) > 1 else "zk"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda", 0)
cfg = bench.CFGS[name]()
w = weights.make_weights(cfg)
s = scorers.make_scorer(cfg, w, precision=2)
ps = synth.make_pairs((B + 29) // 30, 30, tag="/bench0", with_feats=False).take(slice(0, B))
feats = bench.device_feats(ps, dev, 20200823)
fd = bench.device_feed(name, {name: cfg}, ps, feats, dev)
prep = bench.prepare(s, name, fd)
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    for _ in range(5):
        out = s.score_prepared(prep)
    side.synchronize()
    ref = out[1].clone()
    t0 = time.perf_counter()
    for _ in range(200):
        s.score_prepared(prep)
    side.synchronize()
    print("%s B=%d eager, 200 calls back to back: %.3f ms per call" % (name, B, (time.perf_counter() - t0) * 5))
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g, stream=side):
        out = s.score_prepared(prep)
except Exception as e:
    print("capture failed:", repr(e)[:300]); sys.exit(0)
g.replay(); torch.cuda.synchronize()
print("graph result equals eager:", bool(torch.equal(out[1], ref)))
t0 = time.perf_counter()
for _ in range(200):
    g.replay()
torch.cuda.synchronize()
print("%s B=%d graph replay, 200 back to back: %.3f ms per call" % (name, B, (time.perf_counter() - t0) * 5))
t0 = time.perf_counter()
for _ in range(200):
    g.replay(); torch.cuda.synchronize()
print("%s B=%d graph replay + sync each: %.3f ms per call" % (name, B, (time.perf_counter() - t0) * 5))
