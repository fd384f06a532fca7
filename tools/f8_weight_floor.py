"""What does the fp8 WEIGHT format alone cost?  (CPU, fp64 oracle; no GPU needed)

BASELINE.json config 5 asks for "fp8 MFMA weights".  This script quantises every encoder GEMM matrix of the full-depth zk / lxmert models
to e4m3 with one power-of-two scale per output channel (exactly what csrc/rowops.hip k_quant_rows_f8 stores), leaves EVERYTHING else
in fp64 (activations unquantised), and reports the final-logit error against the unquantised model.  That error is the floor of any
fp8-weight mode on these weights, whatever is done to the activations (block scales, second planes): a 3-bit mantissa is a 3-bit
mantissa.  Output recorded in profiles/r03e_f8_weight_floor.txt."""
import os
import sys

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
from helpers import vecrel  # noqa: E402
from kddcup_2020_multimodalitiesrecall_2nd_place_amd import synth, weights  # noqa: E402
from kddcup_2020_multimodalitiesrecall_2nd_place_amd.config import LdsConfig, LxmertConfig, ZkConfig  # noqa: E402
from oracle import fp8 as F8, np_models as O  # noqa: E402

for cfg in (ZkConfig(), LdsConfig(), LxmertConfig()):
    w = weights.make_weights(cfg)
    ps = synth.make_pairs(3, 8, tag="/f8w")
    b = synth.batch_for(cfg, ps)
    ref, _ = O.forward(cfg, w, b, np.float64)
    wq, n = dict(w), 0
    for k, v in w.items():
        if v.ndim == 2 and min(v.shape) >= 768 and "embeddings" not in k and any(t in k for t in ("attention", "intermediate", "output", "query", "key", "value")):
            tf = k.endswith("kernel")                      # TF kernels are [in, out], torch weights [out, in]: scale per OUTPUT channel
            q, _ = F8.quant_weight_rows(v.T if tf else v)
            wq[k] = (q.T if tf else q).astype(np.float32)
            n += 1
    got, _ = O.forward(cfg, wq, b, np.float64)
    e = vecrel(got, ref)
    print("%-7s e4m3 WEIGHTS only (%3d matrices, activations fp64): logit vec-rel median %.3e  max %.3e" % (cfg.name, n, np.median(e), e.max()), flush=True)
