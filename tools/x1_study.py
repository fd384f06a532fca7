"""Which GEMM classes could drop the lo pass?  Full-depth zk / lxmert, 24 pairs, vec-rel logit error vs the fp64 oracle
for every MMS_X1_MASK (1 qkv, 2 att-out, 4 ffn-up, 8 ffn-down).  Experiment only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from kddcup_2020_multimodalitiesrecall_2nd_place_amd import lib, scorers, synth, weights
lib.load(lib.LAB_LIB_PATH)   # MMS_X1_MASK is honoured by the lab build only (`make -C .../csrc lab`)
from kddcup_2020_multimodalitiesrecall_2nd_place_amd.config import ZkConfig, LxmertConfig
from oracle import np_models as O
for cfg in (ZkConfig(), LxmertConfig()):
    w = weights.make_weights(cfg)
    ps = synth.make_pairs(2, 12, tag="/full")
    b = synth.batch_for(cfg, ps)
    ref, _ = O.forward(cfg, w, b, np.float64)
    for mask in (0, 1, 2, 4, 8, 3, 15):
        os.environ["MMS_X1_MASK"] = str(mask)
        s = scorers.make_scorer(cfg, w)
        l, _ = scorers.score_batch(s, b); torch.cuda.synchronize()
        e = np.linalg.norm(l.cpu().numpy() - ref, axis=1) / np.linalg.norm(ref, axis=1)
        s.close()
        print(cfg.name, "x1_mask", mask, "max %.2e median %.2e" % (e.max(), np.median(e)), flush=True)
