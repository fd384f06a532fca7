"""Per-tile timeline of the fused QKV + attention kernel (lab build, MMS_QA_TRACE=1 -> /tmp/qa_trace.bin; csrc/qkv_attn.hip).
usage (GPU box): MMS_QA_TRACE=1 python tools/qa_trace.py   -- runs one zk bench-size score call through the lab library and prints the medians"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def report(path="/tmp/qa_trace.bin"):
    t = np.fromfile(path, dtype=np.uint64).reshape(-1, 8, 16).astype(np.int64)
    names = ["main loop", "dump 0", "attention 0", "dump 1", "attention 1", "next prologue"]
    for tile in range(1, 6):
        x = t[:, tile, :7]
        ok = x[:, 0] > 0
        d = np.diff(x[ok], axis=1)
        print("tile %d (%d workgroups): " % (tile, ok.sum()) + "  ".join("%s %d" % (n, np.median(d[:, i])) for i, n in enumerate(names))
              + "   | total %d ticks" % np.median(x[ok, 6] - x[ok, 0]))
        y = t[:, tile, :]
        sub = y[ok][:, 8] > 0
        if sub.any():      # wave 0's own progress (split-bf16 route): inside dump 0 and inside attention 1
            z = y[ok][sub]
            print("        wave 0: dump body %d  stores landed +%d  barrier +%d   || attention 1: range + Q %d  first step +%d  steps done +%d  stores issued +%d  barrier +%d" % (
                np.median(z[:, 8] - z[:, 1]), np.median(z[:, 9] - z[:, 8]), np.median(z[:, 2] - z[:, 9]),
                np.median(z[:, 10] - z[:, 4]), np.median(z[:, 11] - z[:, 10]), np.median(z[:, 12] - z[:, 11]), np.median(z[:, 13] - z[:, 12]), np.median(z[:, 5] - z[:, 13])))


if __name__ == "__main__":
    os.environ["MMS_QA_TRACE"] = "1"
    import torch
    from kddcup_2020_multimodalitiesrecall_2nd_place_amd import lib, scorers, synth, weights
    from kddcup_2020_multimodalitiesrecall_2nd_place_amd.config import ZkConfig
    lib.load(os.environ.get("MMS_LAB_LIB", lib.LAB_LIB_PATH))
    cfg = ZkConfig(layers=2)
    w = weights.make_weights(cfg)
    ps = synth.make_pairs(1000, 30, with_feats=False)
    dev = torch.device("cuda")
    feats = torch.randn((ps.n, 10, 2048), device=dev).clamp_(min=0)
    feats *= (torch.arange(10, device=dev)[None, :] < torch.as_tensor(ps.num_boxes, device=dev)[:, None])[:, :, None]
    ps.feats = feats
    b = synth.batch_for(cfg, ps)
    s = scorers.make_scorer(cfg, w, precision=2, fuse_attention=int(os.environ.get("QA_FUSE", 1)))
    scorers.score_batch(s, b)
    torch.cuda.synchronize()
    s.close()
    report()
