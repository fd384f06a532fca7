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
    ap.add_argument("--ranks", type=int, default=0, help="N processes, each decoding ITS query block of the one file with pipeline.decode_threads_for(N) threads (the host side of an N-GPU job)")
    ap.add_argument("--rank", type=int, default=-1, help=argparse.SUPPRESS)
    ap.add_argument("--shard-by", default="bytes", choices=["bytes", "queries"])
    ap.add_argument("--rank-threads", type=int, default=0, help="decode threads per rank (default: pipeline.decode_threads_for(ranks))")
    a = ap.parse_args()
    path = "/tmp/featbench_%d.tsv" % a.records
    if not os.path.exists(path):
        t0 = time.time()
        write_tsv(path, a.records)
        print("wrote %s: %.2f GB in %.0f s" % (path, os.path.getsize(path) / 1e9, time.time() - t0), flush=True)
    gb = os.path.getsize(path) / 1e9
    if a.ranks and a.rank < 0:                      # parent: start the ranks together, wait, report the aggregate
        import subprocess
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--records", str(a.records), "--ranks", str(a.ranks), "--rank", str(r),
                                   "--batch", str(a.batch), "--reps", str(a.reps), "--shard-by", a.shard_by, "--rank-threads", str(a.rank_threads)] + (["--pinned"] if a.pinned else []) + (["--no-prefault"] if a.no_prefault else []),
                                  stdout=subprocess.PIPE, text=True)
                 for r in range(a.ranks)]
        outs = [p_.communicate()[0] for p_ in procs]
        rows = [l.split() for o in outs for l in o.splitlines() if l.startswith("RANK")]
        t0, t1 = min(float(r[2]) for r in rows), max(float(r[3]) for r in rows)
        n = sum(int(r[4]) for r in rows)
        print("%d ranks x %s threads, one %d-record file, each rank its share (cut by %s)%s%s: %d records in %.3f s wall = %.0f records/s aggregate (per rank %s)" % (
            a.ranks, rows[0][5], a.records, a.shard_by, " (pinned)" if a.pinned else "", " no-prefault" if a.no_prefault else "", n, t1 - t0, n / (t1 - t0), " ".join("%.0fk" % (int(r[4]) / (float(r[3]) - float(r[2])) / 1e3) for r in rows)))
        return
    if a.ranks:
        from kddcup_2020_multimodalitiesrecall_2nd_place_amd import pipeline
        th = a.rank_threads or pipeline.decode_threads_for(a.ranks)
        nf = N.NativeFeaturizer(VOCAB, TABLE, "zk", threads=th, pinned=a.pinned, reuse_buffers=True, pools=3)
        nf.prefault = not a.no_prefault

        def shard():
            if a.shard_by == "queries":
                return dict(records=pipeline.tsv_shard(nf, path, a.rank, a.ranks)[0])
            return dict(byte_range=nf.byte_shard(path, a.rank, a.ranks))
        for _ in range(a.reps - 1):                  # warm: buffers, page cache, helper threads
            sum(len(b["query_id"]) for b in nf.iter_file(path, a.batch, layout=False, ramp=1024, **shard()))
        # all ranks start their timed pass at the same wall-clock second
        start = (int(time.time()) // 2 + 2) * 2.0
        while time.time() < start:
            time.sleep(0.001)
        t0 = time.time()
        k = sum(len(b["query_id"]) for b in nf.iter_file(path, a.batch, layout=False, ramp=1024, **shard()))      # finding the shard is part of a rank's job
        print("RANK %d %.6f %.6f %d %d" % (a.rank, t0, time.time(), k, th), flush=True)
        return
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
