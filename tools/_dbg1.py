import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from kddcup_2020_multimodalitiesrecall_2nd_place_amd import lib, scorers, synth, weights
from kddcup_2020_multimodalitiesrecall_2nd_place_amd.config import LxmertConfig, ZkConfig

def feed(cfg, nq, nc, tag):
    ps = synth.make_pairs(nq, nc, tag=tag, with_feats=False)
    dev = torch.device("cuda")
    g = torch.Generator(device=dev); g.manual_seed(321)
    feats = torch.randn((ps.n, 10, 2048), device=dev, generator=g).clamp_(min=0)
    feats *= (torch.arange(10, device=dev)[None, :] < torch.as_tensor(ps.num_boxes, device=dev)[:, None])[:, :, None]
    ps.feats = feats
    return ps, synth.batch_for(cfg, ps)

cfg = LxmertConfig(l_layers=1, r_layers=0, x_layers=1)
w = weights.make_weights(cfg)
ps, b = feed(cfg, 1000, 30, "/full")
s = scorers.make_scorer(cfg, w, precision=2, fuse_attention=2)
outs = [scorers.score_batch(s, b)[0].cpu().numpy() for _ in range(6)]
s.close()
mask = np.asarray(b["input_mask"].cpu() if torch.is_tensor(b["input_mask"]) else b["input_mask"])
ntok = mask.sum(1)
for k in range(1, 6):
    bad = (outs[0] != outs[k]).any(1)
    qbad = bad.reshape(1000, 30)
    nq_all = int(qbad.all(1).sum()); nq_any = int(qbad.any(1).sum())
    idx = np.nonzero(qbad.any(1))[0]
    print("run", k, "pairs differing", int(bad.sum()), "queries with all 30 differing", nq_all, "any", nq_any, "first bad queries", idx[:20], "tokens of bad queries", np.bincount(ntok.reshape(1000, 30)[idx, 0], minlength=24), flush=True)
print("token histogram all", np.bincount(ntok.reshape(1000,30)[:,0], minlength=24))
