"""Full-depth deviation of mms_config.fuse_attention = 1 / 2 from the two-kernel route and from the fp64 oracle (GPU box).
usage: python tools/fuse_attn_parity.py [n_queries]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kddcup_2020_multimodalitiesrecall_2nd_place_amd import scorers, synth, weights  # noqa: E402
from kddcup_2020_multimodalitiesrecall_2nd_place_amd.config import LdsConfig, LxmertConfig, ZkConfig  # noqa: E402
from oracle import np_models as O  # noqa: E402  (checker only)

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda")
for name, cfg in (("zk", ZkConfig()), ("lds", LdsConfig()), ("lxmert", LxmertConfig())):
    w = weights.make_weights(cfg)
    ps = synth.make_pairs(nq, 30, tag="/fap", with_feats=False)
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    feats = torch.randn((ps.n, 10, 2048), device=dev, generator=g).clamp_(min=0)
    feats *= (torch.arange(10, device=dev)[None, :] < torch.as_tensor(ps.num_boxes, device=dev)[:, None])[:, :, None]
    ps.feats = feats
    b = synth.batch_for(cfg, ps)
    out = {}
    for f in (0, 1, 2):
        s = scorers.make_scorer(cfg, w, precision=2, fuse_attention=f)
        out[f] = scorers.score_batch(s, b)[0].double().cpu().numpy()
        n = s.handle.counter(0)
        s.close()
        assert (n > 0) == (f > 0)
    def vr0(a, c):
        return np.linalg.norm(a - c, axis=1) / np.maximum(np.linalg.norm(c, axis=1), 1e-30)
    worst = np.argsort(-vr0(out[2], out[0]))[:6]        # ... and the pairs that moved most: how far are THEY from the oracle?
    idx = np.sort(np.unique(np.concatenate([np.random.RandomState(5).choice(ps.n, 12, replace=False), worst])))
    ti = torch.as_tensor(idx, device=dev)
    sub = {k: (v[ti].cpu().numpy() if torch.is_tensor(v) else (v[idx] if hasattr(v, "__len__") and len(v) == ps.n else v)) for k, v in b.items()}
    ref, _ = O.forward(cfg, w, sub, np.float64)

    def vr(a, c):
        return np.linalg.norm(a - c, axis=1) / np.maximum(np.linalg.norm(c, axis=1), 1e-30)
    d1, d2 = vr(out[1], out[0]), vr(out[2], out[0])
    print("%-6s %d pairs, full depth | fuse 1 vs two-kernel: max %.1e | fuse 2 vs two-kernel: median %.1e max %.1e | vs fp64 oracle (12 random pairs + the 6 that moved most): "
          "two-kernel max %.1e, fuse 2 max %.1e" % (name, ps.n, d1.max(), np.median(d2), d2.max(), vr(out[0][idx], ref).max(), vr(out[2][idx], ref).max()), flush=True)
    pos = np.searchsorted(idx, worst)
    f = lambda a: " ".join("%.1e" % x for x in a)
    print("       the 6 pairs that moved most: |logit| %s | fuse 2 vs two-kernel %s | two-kernel vs oracle %s | fuse 2 vs oracle %s" % (
        " ".join("%.3f" % x for x in np.linalg.norm(out[0][worst], axis=1)), f(vr(out[2][worst], out[0][worst])), f(vr(out[0][worst], ref[pos])), f(vr(out[2][worst], ref[pos]))), flush=True)
