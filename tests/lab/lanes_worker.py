"""Lab-library worker of tests/test_lanes_gpu.py: lxmert logits of a few call sizes, written to argv[1].  Run once with MMS_LANE_ROWS=0 (one launch lane) and once
with the shipped setting; the two files must hold the same bits wherever the lanes leave the arithmetic alone."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from kddcup_2020_multimodalitiesrecall_2nd_place_amd import lib  # noqa: E402
lib.load(lib.LAB_LIB_PATH)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from kddcup_2020_multimodalitiesrecall_2nd_place_amd import scorers, synth, weights  # noqa: E402
from kddcup_2020_multimodalitiesrecall_2nd_place_amd.config import LxmertConfig  # noqa: E402

cfg = LxmertConfig(l_layers=2, r_layers=2, x_layers=3)
w = weights.make_weights(cfg)
s = scorers.make_scorer(cfg, w, precision=2)
out = {}
for tag, nq, cand in (("B7", 1, 7), ("B60", 2, 30), ("B300", 10, 30), ("B700", 25, 28), ("B3000", 100, 30), ("distinct", 40, 1)):
    ps = synth.make_pairs(nq, cand, vocab=cfg.vocab, tag="/lanes" + tag)
    b = synth.batch_for(cfg, ps)
    l1 = scorers.score_batch(s, b)[0].cpu().numpy().copy()
    l2 = scorers.score_batch(s, b)[0].cpu().numpy().copy()
    assert np.array_equal(l1, l2), "identical calls differ at " + tag
    out[tag] = l1
s.close()
np.savez(sys.argv[1], **out)
