"""End-to-end TSV -> scores throughput (SURVEY.md section 8(f) row 2 + the hot path): synthetic valid/testB-like TSV file,
native featurizer threads -> pinned buffers -> H2D on a copy stream -> scorer, all overlapped (pipeline.stream_scores_tsv).
usage (GPU box): python tools/e2e_tsv_bench.py [records] [model] [decode threads, comma list (0 = library default)]"""
import os
import sys
import time

import numpy as np
import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from kddcup_2020_multimodalitiesrecall_2nd_place_amd import featurizer as F, pipeline, scorers, weights  # noqa: E402
from kddcup_2020_multimodalitiesrecall_2nd_place_amd.config import LdsConfig, LxmertConfig, ZkConfig  # noqa: E402
from kddcup_2020_multimodalitiesrecall_2nd_place_amd.featurizer_native import NativeFeaturizer  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
name = sys.argv[2] if len(sys.argv) > 2 else "zk"
threads_list = [int(t) for t in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0]
D = os.path.join(R, "tests", "golden", "featurizer")
VOCAB, TABLE = os.path.join(D, "vocab_small.txt"), F.load_label_table(os.path.join(D, "labels.txt"))
path = "/tmp/e2e_%d.tsv" % n
rng = np.random.default_rng(1)
words = [w for w in open(VOCAB, encoding="utf-8").read().split() if not w.startswith("[") and w.isascii()]
classes = [int(k) for k in TABLE]
t0 = time.time()
with open(path, "w") as f:
    f.write("product_id\timage_h\timage_w\tnum_boxes\tboxes\tfeatures\tclass_labels\tquery\tquery_id\n")
    feats_pool = np.maximum(rng.standard_normal((64, 10, 2048)), 0).astype(np.float32)
    for i in range(n):
        nb = int(np.clip(round(rng.lognormal(1.2, 0.5)), 1, 10))       # mean ~3.8 boxes like the shipped files
        h, w = int(rng.integers(200, 1000)), int(rng.integers(200, 1000))
        boxes = np.sort(rng.uniform(0, 1, (nb, 4)), axis=1)[:, [0, 1, 2, 3]] * np.array([h, w, h, w])
        f.write(F.encode_record(i, h, w, boxes, feats_pool[i % 64, :nb], rng.choice(classes, nb), " ".join(rng.choice(words, int(rng.integers(2, 9)))), i // 30) + "\n")
print("wrote %s: %.2f GB in %.0f s" % (path, os.path.getsize(path) / 1e9, time.time() - t0), flush=True)

if name == "ensemble":      # BASELINE.json config 5: TSV -> four score tables -> merge -> uniqueness filter -> top-5 submission
    from collections import OrderedDict
    from kddcup_2020_multimodalitiesrecall_2nd_place_amd import ensemble as E
    sc = {m: scorers.make_scorer(c, weights.make_weights(c), device=0) for m, c in (("zk", ZkConfig()), ("lds", LdsConfig()), ("lxmert", LxmertConfig()))}
    ens = pipeline.EnsembleScorer(sc["zk"], sc["lds"], sc["lxmert"])
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        qid, pid, merged, parts = ens.score_tsv_native(path, VOCAB, TABLE, batch_pairs=16384)
        t1 = time.time()
        rows = E.submission_rows(qid, pid, merged)          # uniqueness filter + top-5 (main.py:64-104) on the three arrays
        E.write_submission("/tmp/e2e_submission.csv", rows)
        dt = time.time() - t0
    print("TSV -> 4 score tables -> submission.csv: %.0f pairs/s end to end (%d pairs, %d queries; scoring %.2f s, post-process %.2f s)"
          % (len(merged) / dt, len(merged), len(rows), t1 - t0, dt - (t1 - t0)), flush=True)
    os.remove(path)
    sys.exit(0)
cfg = {"zk": ZkConfig(), "lds": LdsConfig(), "lxmert": LxmertConfig()}[name]
sc = scorers.make_scorer(cfg, weights.make_weights(cfg), device=0)
for th in threads_list:
    nf = NativeFeaturizer(VOCAB, TABLE, name, threads=th, pinned=True, reuse_buffers=True, pools=3)
    for _ in range(3):
        nf.stats.clear()
        t0 = time.time(); k = sum(len(b["query_id"]) for b in nf.iter_file(path, 8192)); dt = time.time() - t0
    print("featurizer alone (%s of %d host threads): %.0f records/s, %.2f GB/s of TSV  [ms: %s]" % (
        th or "default", os.cpu_count(), k / dt, os.path.getsize(path) / dt / 1e9, ", ".join("%s %.0f" % (a_, b_ * 1e3) for a_, b_ in nf.stats.items())), flush=True)
    nf.close()
    for bp in (8192, 16384):
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.time()
            stt = {}
            qid, pid, score = pipeline.stream_scores_tsv(sc, path, VOCAB, TABLE, batch_pairs=bp, threads=th, stats=stt)
            torch.cuda.synchronize(); dt = time.time() - t0
        print("TSV -> scores, batch %d, threads %s: %.0f pairs/s (%d pairs, %.2f s)  [host ms: waiting for a decoded batch %.0f, for H2D %.0f, enqueueing %.0f, final drain %.0f; %d batches]"
              % (bp, th or "default", len(score) / dt, len(score), dt, stt["wait_decode_s"] * 1e3, stt["wait_h2d_s"] * 1e3, stt["enqueue_s"] * 1e3, stt["drain_s"] * 1e3, stt["batches"]), flush=True)
# device-resident rate of the same records (the ceiling of the lines above): the batches of the file, already on the device
nf = NativeFeaturizer(VOCAB, TABLE, name, pinned=False)
dev_batches = []
for b in nf.iter_file(path, 8192):
    dev_batches.append({k: (torch.from_numpy(np.array(v)).cuda() if isinstance(v, np.ndarray) and v.dtype.kind in "fiu" else v) for k, v in b.items() if k not in ("query_id", "product_id", "keep")})
    if len(dev_batches) == 4:
        break
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    for r in range(4):
        for d in dev_batches:
            scorers.score_batch(sc, d)
    torch.cuda.synchronize(); dt = time.time() - t0
print("device-resident, same records, batches of 8192: %.0f pairs/s" % (16 * 8192 / dt), flush=True)
os.remove(path)
