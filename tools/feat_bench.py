"""Host featurizer alone (no GPU): records/s of libmmfeat over a synthetic valid/testB-like TSV file, swept over thread counts, base64 tiers
and -- with --lib -- another build of the library (A/B against tools/ab/libmmfeat_r5.so).
usage: python tools/feat_bench.py [--records 60000] [--threads 8,16,32,64,128,256] [--lib path] [--tiers 0,1,2] [--pinned] [--affinity node0]"""
import argparse
import os
import sys
import time

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from kddcup_2020_multimodalitiesrecall_2nd_place_amd import featurizer as F  # noqa: E402
from kddcup_2020_multimodalitiesrecall_2nd_place_amd import featurizer_native as N  # noqa: E402

D = os.path.join(R, "tests", "golden", "featurizer")
VOCAB, TABLE = os.path.join(D, "vocab_small.txt"), F.load_label_table(os.path.join(D, "labels.txt"))


def write_tsv(path, n, seed=1):
    rng = np.random.default_rng(seed)
    words = [w for w in open(VOCAB, encoding="utf-8").read().split() if not w.startswith("[") and w.isascii()]
    classes = [int(k) for k in TABLE]
    with open(path, "w") as f:
        f.write("product_id\timage_h\timage_w\tnum_boxes\tboxes\tfeatures\tclass_labels\tquery\tquery_id\n")
        feats_pool = np.maximum(rng.standard_normal((64, 10, 2048)), 0).astype(np.float32)
        for i in range(n):
            nb = int(np.clip(round(rng.lognormal(1.2, 0.5)), 1, 10))       # mean ~3.8 boxes like the shipped files
            h, w = int(rng.integers(200, 1000)), int(rng.integers(200, 1000))
            boxes = np.sort(rng.uniform(0, 1, (nb, 4)), axis=1) * np.array([h, w, h, w])
            f.write(F.encode_record(i, h, w, boxes, feats_pool[i % 64, :nb], rng.choice(classes, nb),
                                    " ".join(rng.choice(words, int(rng.integers(2, 9)))), i // 30) + "\n")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--records", type=int, default=60000)
    ap.add_argument("--threads", default="0")
    ap.add_argument("--lib", default=None)
    ap.add_argument("--tiers", default="")
    ap.add_argument("--batch", type=int, default=8192)
    ap.add_argument("--pinned", action="store_true")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--no-prefault", action="store_true")
    a = ap.parse_args()
    path = "/tmp/featbench_%d.tsv" % a.records
    if not os.path.exists(path):
        t0 = time.time()
        write_tsv(path, a.records)
        print("wrote %s: %.2f GB in %.0f s" % (path, os.path.getsize(path) / 1e9, time.time() - t0), flush=True)
    gb = os.path.getsize(path) / 1e9
    lib = N.load(os.path.abspath(a.lib) if a.lib else None)
    libname = os.path.basename(a.lib or N.LIB_PATH)
    has_tier = hasattr(lib, "mmf_b64_tier")
    tiers = [int(t) for t in a.tiers.split(",") if t] if has_tier else []
    for tier in (tiers or [None]):
        if tier is not None and lib.mmf_b64_tier(tier) != tier:
            print("tier %d not available on this CPU" % tier)
            continue
        for th in [int(t) for t in a.threads.split(",")]:
            nf = N.NativeFeaturizer(VOCAB, TABLE, "zk", threads=th, pinned=a.pinned, reuse_buffers=True, pools=3)
            nf.prefault = not a.no_prefault
            nf.stats = {}
            best, parts = 0.0, {}
            for _ in range(a.reps):
                nf.stats.clear()
                t0 = time.time()
                k = sum(len(b["query_id"]) for b in nf.iter_file(path, a.batch, layout=False))
                dt = time.time() - t0
                if k / dt > best:
                    best, parts = k / dt, dict(nf.stats, total=dt)
            print("%s tier %s threads %3d batch %d%s%s: %8.0f records/s  %.2f GB/s of TSV   [ms: %s]" % (
                libname, "-" if tier is None else tier, th or os.cpu_count(), a.batch, " pinned" if a.pinned else "",
                " no-prefault" if a.no_prefault else "", best, best / a.records * gb,
                ", ".join("%s %.0f" % (k_, v * 1e3) for k_, v in parts.items())), flush=True)
            nf.close()


if __name__ == "__main__":
    main()
