"""GPU box: soak of the file-fed path -- one cached featurizer + one scorer through many passes over a TSV file (pipeline.stream_scores_tsv, whole file and
per-rank shards), then the fused three-model path (score_tsv_native): every pass must give the bits of the first, host RSS and free device memory must stay flat.
usage: python tools/soak_tsv.py [records] [passes]"""
import os
import resource
import sys
import time

import numpy as np
import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tools"))
from feat_bench import TABLE, VOCAB, write_tsv  # noqa: E402
from kddcup_2020_multimodalitiesrecall_2nd_place_amd import pipeline, scorers, weights  # noqa: E402
from kddcup_2020_multimodalitiesrecall_2nd_place_amd.config import LdsConfig, LxmertConfig, ZkConfig  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 30
path = "/tmp/soak_%d.tsv" % n
if not os.path.exists(path):
    write_tsv(path, n)
rss = lambda: resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6
free = lambda: (torch.cuda.synchronize(), torch.cuda.mem_get_info()[0] / 1e9)[1]
cfg = ZkConfig()
sc = scorers.make_scorer(cfg, weights.make_weights(cfg))
ref, t0 = None, time.time()
for i in range(passes):
    q, p, s = pipeline.stream_scores_tsv(sc, path, VOCAB, TABLE, batch_pairs=8192)
    if ref is None:
        ref = s
    if i == 1:                                   # (after the second pass: every buffer of the steady state exists)
        rss0, free0 = rss(), free()
    assert len(s) == n and np.array_equal(s, ref), "pass %d differs from the first" % i
print("zk: %d passes over %d records in %.1f s (%.0f pairs/s incl. everything), bit-identical; host max RSS %.2f -> %.2f GB, free device memory %.2f -> %.2f GB"
      % (passes, n, time.time() - t0, passes * n / (time.time() - t0), rss0, rss(), free0, free()), flush=True)
parts = []
for r in range(4):
    q, p, s, counts = pipeline.stream_scores_tsv(sc, path, VOCAB, TABLE, batch_pairs=8192, shard=(r, 4))
    parts.append(s)
cat = np.concatenate(parts)
print("4 byte shards through the same featurizer: %d records, max |score - whole-file pass| %.2e (shards place pairs differently in their sub-tiles: <= 1e-4)"
      % (len(cat), float(np.abs(cat - ref).max())), flush=True)
assert len(cat) == n and np.abs(cat - ref).max() < 1e-4
members = {"zk": sc, "lds": scorers.make_scorer(LdsConfig(), weights.make_weights(LdsConfig())), "lxmert": scorers.make_scorer(LxmertConfig(), weights.make_weights(LxmertConfig()))}
ens = pipeline.EnsembleScorer(members["zk"], members["lds"], members["lxmert"])
ref, t0 = None, time.time()
for i in range(max(3, passes // 6)):
    q, p, m, parts4 = ens.score_tsv_native(path, VOCAB, TABLE, batch_pairs=16384)
    if ref is None:
        ref = m
    if i == 1:
        rss0, free0 = rss(), free()
    assert len(m) == n and np.array_equal(m, ref), "ensemble pass %d differs" % i
print("ensemble: %d passes, bit-identical; host max RSS %.2f -> %.2f GB, free device memory %.2f -> %.2f GB (%.0f pairs/s)"
      % (max(3, passes // 6), rss0, rss(), free0, free(), max(3, passes // 6) * n / (time.time() - t0)), flush=True)
assert True
os.remove(path)
