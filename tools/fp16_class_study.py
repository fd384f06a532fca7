"""Could a GEMM class run ONE MFMA pass on an fp16 high plane (2^-12 relative operand error) instead of the two bf16 passes?
VERDICT r4 item 3.  CPU study on the fp64 oracle (no GPU): the activations entering the matrices of one class are rounded to
fp16 (the weights are bf16-representable, hence fp16-exact in their range), everything else stays fp64; the logits are compared
with the unrounded fp64 oracle at FULL depth.  `--fmt bf16` rounds to bf16 instead: that reproduces tools/x1_study.py (measured
on the GPU, profiles/r03z_x1_study.txt) and so checks this emulation.

usage: python tools/fp16_class_study.py [--queries 100] [--cands 30] [--models zk,lds,lxmert] [--fmt fp16]
Bar (VERDICT): a class qualifies if max vec-rel stays <= 2.5e-4 on all three models.
"""
import argparse
import os
import re
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from kddcup_2020_multimodalitiesrecall_2nd_place_amd import synth, weights
from kddcup_2020_multimodalitiesrecall_2nd_place_amd.config import LdsConfig, LxmertConfig, ZkConfig
from oracle import np_models as O

CLASSES = {
    "qkv": re.compile(r"(/attention/self/(query|key|value)/kernel$)|(\.(query|key|value)\.weight$)"),
    "att_out": re.compile(r"(/attention/output/dense/kernel$)|((attention|att|_att)\.output\.dense\.weight$)"),
    "ffn_up": re.compile(r"(/intermediate/dense/kernel$)|((intermediate|_inter)\.dense\.weight$)"),
    "ffn_down": re.compile(r"((?<!attention)/output/dense/kernel$)|((layer\.\d+|r_layers\.\d+|lang|visn)[._]output\.dense\.weight$)"),
}


def round_to(x, fmt):
    if fmt == "fp16":
        return x.astype(np.float16).astype(x.dtype)
    u = x.astype(np.float32).view(np.uint32)          # bf16, round to nearest even
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.view(np.float32).astype(x.dtype)


class RoundedLeft(np.ndarray):
    """A weight matrix whose LEFT operand (the activation) is rounded before the product: x @ W -> round(x) @ W."""
    fmt = "fp16"

    def __rmatmul__(self, x):
        return np.matmul(round_to(np.asarray(x), self.fmt), np.asarray(self))


def run(cfg, w, batch, cls, fmt):
    pat = [CLASSES[c] for c in cls]
    cast = O._cast

    def cast_wrap(ws, dtype):
        out = cast(ws, dtype)
        n = 0
        for k in list(out):
            if any(p.search(k) for p in pat):
                v = out[k].view(RoundedLeft)
                v.fmt = fmt
                out[k] = v
                n += 1
        cast_wrap.n = n
        return out

    O._cast = cast_wrap
    try:
        logits, _ = O.forward(cfg, w, batch, np.float64)
    finally:
        O._cast = cast
    return np.asarray(logits), getattr(cast_wrap, "n", 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--queries", type=int, default=100)
    ap.add_argument("--cands", type=int, default=30)
    ap.add_argument("--chunk", type=int, default=300)
    ap.add_argument("--models", default="zk,lds,lxmert")
    ap.add_argument("--fmt", default="fp16")
    a = ap.parse_args()
    cfgs = {"zk": ZkConfig(), "lds": LdsConfig(), "lxmert": LxmertConfig()}
    variants = [("qkv",), ("att_out",), ("ffn_up",), ("ffn_down",), ("qkv", "att_out", "ffn_up", "ffn_down")]
    print("# %s high plane, one pass, per GEMM class; %d queries x %d candidates, full depth, vs the fp64 oracle" % (a.fmt, a.queries, a.cands))
    for name in a.models.split(","):
        cfg = cfgs[name]
        w = weights.make_weights(cfg)
        ps = synth.make_pairs(a.queries, a.cands, tag="/fp16study")
        b = synth.batch_for(cfg, ps)
        n = ps.n
        errs = {v: [] for v in variants}
        t0 = time.time()
        for c0 in range(0, n, a.chunk):
            sl = slice(c0, min(n, c0 + a.chunk))
            sub = {k: (v[sl] if hasattr(v, "__len__") and len(v) == n else v) for k, v in b.items()}
            ref, _ = O.forward(cfg, w, sub, np.float64)
            for v in variants:
                got, nw = run(cfg, w, sub, v, a.fmt)
                assert nw > 0, (name, v)
                errs[v].append(np.linalg.norm(got - ref, axis=1) / np.linalg.norm(ref, axis=1))
        for v in variants:
            e = np.concatenate(errs[v])
            print("%-7s %-32s pairs %5d  max %.2e  p99 %.2e  median %.2e  %s" % (
                name, "+".join(v), len(e), e.max(), np.percentile(e, 99), np.median(e), "PASS" if e.max() <= 2.5e-4 else "fail"), flush=True)
        print("# %s: %.0f s" % (name, time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
