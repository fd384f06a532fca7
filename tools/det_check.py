"""Lab book: are repeated runs bit-identical in every precision / packing / LayerNorm-fusion combination?  (lds, GPU box)"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kddcup_2020_multimodalitiesrecall_2nd_place_amd import scorers, synth, weights
from kddcup_2020_multimodalitiesrecall_2nd_place_amd.config import LdsConfig
cfg = LdsConfig(layers=2)
w = weights.make_weights(cfg)
ps = synth.make_pairs(150, 30, tag="/fuseln", with_feats=False)
dev = torch.device("cuda"); g = torch.Generator(device=dev); g.manual_seed(77)
feats = torch.randn((ps.n, 10, 2048), device=dev, generator=g).clamp_(min=0)
feats *= (torch.arange(10, device=dev)[None, :] < torch.as_tensor(ps.num_boxes, device=dev)[:, None])[:, :, None]
ps.feats = feats
b = synth.batch_for(cfg, ps)
for prec in (2, 4):
  for pack in (False, True):
    for fuse in (False, True):
        s = scorers.make_scorer(cfg, w, precision=prec, pack_tokens=pack, fuse_layernorm=fuse)
        outs = [scorers.score_batch(s, b)[0].cpu().numpy() for _ in range(4)]
        s.close()
        print("prec", prec, "pack", pack, "fuse", fuse, "identical runs:", [bool(np.array_equal(outs[0], o)) for o in outs[1:]], "max diff %.2e" % max(np.abs(outs[0]-o).max() for o in outs[1:]))
