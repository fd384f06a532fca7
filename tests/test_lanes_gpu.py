"""lxmert's two launch lanes (api.hip mms_handle::side; lxrt/modeling.py:444-493: the language and the vision stream's sub-layers between two cross attentions are
independent, and so are the language layers and the box stream's layers in front of the first cross attention): the same kernels on the same operands, so a call on two
lanes gives the bits of the call on one lane wherever no launch reaches the fused-LayerNorm regime (>= 98 304 rows), and fp32 round-off of the other LayerNorm route above."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_two_lanes_give_the_bits_of_one_lane(tmp_path):
    from kddcup_2020_multimodalitiesrecall_2nd_place_amd import lib
    if not os.path.exists(lib.LAB_LIB_PATH):
        pytest.skip("libmmscore_lab.so not built (make -C .../csrc lab)")
    res = {}
    for tag, env in (("one", {"MMS_LANE_ROWS": "0"}), ("two", {}), ("query", {"MMS_LANE_ROWS": "1"})):
        f = str(tmp_path / (tag + ".npz"))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "lab", "lanes_worker.py"), f], env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
        res[tag] = np.load(f)
    for k in res["one"].files:
        a, b, q = res["one"][k], res["two"][k], res["query"][k]
        rows = {"B7": 7, "B60": 60, "B300": 300, "B700": 700, "B3000": 3000, "distinct": 40}[k] * 23
        if rows < 98304:      # (language rows = pairs x text_len: the longest launch of the call)
            assert np.array_equal(a, b) and np.array_equal(a, q), (k, float(np.abs(a - b).max()), float(np.abs(a - q).max()))
        else:                 # one lane: LayerNorm in the GEMM epilogue; two lanes: the LayerNorm kernel
            assert np.abs(a - b).max() < 2e-4 and np.abs(a - q).max() < 2e-4, (k, float(np.abs(a - b).max()))


def test_product_build_takes_the_lanes_where_documented():
    """mms_dbg_counter 4 = fork / join pairs: a 256-pair lxmert call (the reference's batch size, param.py:46) forks for its distinct-query stage and once per full X layer;
    a call above LANE_ROWS (400 000 token rows) only for the query stage; zk never; the fused three-model call forks its members once per wave below 5000 pairs."""
    import torch
    from helpers import small_cfg
    from kddcup_2020_multimodalitiesrecall_2nd_place_amd import pipeline, scorers, synth, weights
    cfg = small_cfg("lxmert", x_layers=3)
    s = scorers.make_scorer(cfg, weights.make_weights(cfg))
    ps = synth.make_pairs(9, (28, 30), vocab=cfg.vocab, tag="/lanes_ctr")
    b = synth.batch_for(cfg, ps)
    scorers.score_batch(s, b)
    n1 = s.handle.counter(4)
    assert n1 == 1 + (cfg.x_layers - 1), n1          # query stage + the X layers in front of the trimmed last one
    big = synth.make_pairs(450, 30, vocab=cfg.vocab, tag="/lanes_ctr_big")      # 13 500 pairs x 33 rows > 400 000
    scorers.score_batch(s, synth.batch_for(cfg, big))
    assert s.handle.counter(4) == n1 + 1
    lone = synth.make_pairs(40, 1, vocab=cfg.vocab, tag="/lanes_ctr_lone")      # no query shared: language layers beside the box stream's layers inside the chunk
    scorers.score_batch(s, synth.batch_for(cfg, lone))
    assert s.handle.counter(4) == n1 + 1 + 1 + (cfg.x_layers - 1)
    s.close()
    cfgs = {n: small_cfg(n) for n in ("zk", "lds", "lxmert")}
    sc = {n: scorers.make_scorer(c, weights.make_weights(c)) for n, c in cfgs.items()}
    ens = scorers.EnsembleScorer(sc["zk"], sc["lds"], sc["lxmert"])
    zb = synth.zk_batch(ps, cfgs["zk"].text_len)
    zb2 = synth.zk_batch(synth.sen2forest_variant(ps), cfgs["zk"].text_len)
    xb = synth.lxmert_batch(ps, cfgs["lxmert"].text_len)
    ens(pipeline.ensemble_feed(zb, zb2, xb))
    assert sc["zk"].handle.counter(4) == 1 and sc["lds"].handle.counter(4) == 0 and sc["lxmert"].handle.counter(4) >= 1
    scorers.score_batch(sc["zk"], zb)
    assert sc["zk"].handle.counter(4) == 1
    ens.close()


def test_per_launch_timing_does_not_change_the_logits():
    """with mms_gemm_timing on, a call that would run on two lanes runs on one -- and must still take the LayerNorm-kernel route of the
    untimed call (no fused epilogue), so that instrumentation never changes the bits.  3 500 pairs: 115 500 token rows, above the fused-LayerNorm
    bound (98 304) and below LANE_ROWS; the fused three-model call likewise (its members' lanes are off under timing)."""
    import torch
    from helpers import small_cfg
    from kddcup_2020_multimodalitiesrecall_2nd_place_amd import pipeline, scorers, synth, weights
    cfg = small_cfg("lxmert")
    ps = synth.make_pairs(120, (28, 30), vocab=cfg.vocab, tag="/lanes_timing")
    b = synth.batch_for(cfg, ps)
    assert ps.n * 33 >= 98304
    s = scorers.make_scorer(cfg, weights.make_weights(cfg))
    plain, _ = scorers.score_batch(s, b)
    forks = s.handle.counter(4)
    assert forks > 0 and s.handle.counter(1) == 0                # two lanes, no LayerNorm-fused GEMM launch
    s.handle.gemm_timing(True, True)
    timed, _ = scorers.score_batch(s, b)
    s.handle.gemm_timing(False, True)
    assert s.handle.counter(1) == 0                              # still none: the route of the untimed call
    assert torch.equal(plain, timed)
    again, _ = scorers.score_batch(s, b)
    assert torch.equal(plain, again)
    s.close()
    cfgs = {n: small_cfg(n) for n in ("zk", "lds", "lxmert")}
    sc = {n: scorers.make_scorer(c, weights.make_weights(c)) for n, c in cfgs.items()}
    ens = scorers.EnsembleScorer(sc["zk"], sc["lds"], sc["lxmert"])
    zb = synth.zk_batch(ps, cfgs["zk"].text_len)
    zb2 = synth.zk_batch(synth.sen2forest_variant(ps), cfgs["zk"].text_len)
    xb = synth.lxmert_batch(ps, cfgs["lxmert"].text_len)
    feed = pipeline.ensemble_feed(zb, zb2, xb)
    m0, mem0 = ens(feed)
    for h in (sc["zk"].handle, sc["lds"].handle, sc["lxmert"].handle):
        h.gemm_timing(True, True)
    m1, mem1 = ens(feed)
    for h in (sc["zk"].handle, sc["lds"].handle, sc["lxmert"].handle):
        h.gemm_timing(False, True)
    assert torch.equal(m0, m1) and torch.equal(mem0, mem1)
    assert all(sc[n].handle.counter(1) == 0 for n in sc)         # no member took the fused epilogue beside the others' lanes, timed or not
    ens.close()
