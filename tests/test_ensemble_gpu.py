"""GPU suite: the fused three-model entry point ``mms_score_ensemble`` (BASELINE.json config 5; code/main.py:41-59) -- against the oracle's merged scores at full
model size, against four separate calls, members side by side on their lanes, chunking, the second zk member with no / all queries rewritten, full-size properties."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from helpers import TOL_P2, act_ref, fp32ckpt_case, small_cfg, vecrel
from kddcup_2020_multimodalitiesrecall_2nd_place_amd import lib, pipeline, scorers, synth, weights
from kddcup_2020_multimodalitiesrecall_2nd_place_amd.config import LdsConfig, LxmertConfig, ZkConfig
from oracle import fp8 as F8
from oracle import np_models as O

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


def _members(cfgs, **kw):
    ws = {n: weights.make_weights(c) for n, c in cfgs.items()}
    sc = {n: scorers.make_scorer(cfgs[n], ws[n], **kw) for n in cfgs}
    return ws, sc


def _feeds(cfgs, ps, feats=None):
    if feats is not None:
        ps.feats = feats
    zb = synth.zk_batch(ps, cfgs["zk"].text_len)
    zb2 = synth.zk_batch(synth.sen2forest_variant(ps), cfgs["zk"].text_len)
    lb = synth.lds_batch(ps, cfgs["lds"].text_len)
    xb = synth.lxmert_batch(ps, cfgs["lxmert"].text_len)
    return zb, zb2, lb, xb




# ---------------------------------------------------------------------------------------------------------------------
# config 5: fused ensemble
# ---------------------------------------------------------------------------------------------------------------------
def test_fused_ensemble_full_model_size_matches_oracle_merged_scores():
    """BASELINE.json config 5 at full model size (12-layer zk / lds, 9-5-5 lxmert): merged score of mms_score_ensemble against
    0.2*zk + 0.2*zk(sen2forest) + 0.3*lds + 0.3*lxmert of the fp64 oracle's four forwards (code/main.py:59)."""
    cfgs = {"zk": ZkConfig(), "lds": LdsConfig(), "lxmert": LxmertConfig()}
    ws, sc = _members(cfgs)
    ps = synth.make_pairs(3, 4, tag="/ens_full")                      # query 10002 (% 3 == 0) gets the rewritten variant
    zb, zb2, lb, xb = _feeds(cfgs, ps)
    ens = scorers.EnsembleScorer(sc["zk"], sc["lds"], sc["lxmert"])
    merged, mem = ens(pipeline.ensemble_feed(zb, zb2, xb))
    torch.cuda.synchronize()
    merged, mem = merged.cpu().numpy(), mem.cpu().numpy()
    ens.close()
    r = [O.forward(cfgs["zk"], ws["zk"], zb, np.float64)[1][:, 1], O.forward(cfgs["zk"], ws["zk"], zb2, np.float64)[1][:, 1],
         O.forward(cfgs["lds"], ws["lds"], lb, np.float64)[1][:, 1], O.forward(cfgs["lxmert"], ws["lxmert"], xb, np.float64)[1][:, 1]]
    for k in range(4):
        assert np.abs(mem[k] - r[k]).max() < 1e-3, (k, np.abs(mem[k] - r[k]).max())
    ref = 0.2 * r[0] + 0.2 * r[1] + 0.3 * r[2] + 0.3 * r[3]
    print("\n[ensemble, full size] max |merged - oracle| %.2e" % np.abs(merged - ref).max())
    assert np.abs(merged - ref).max() < 1e-3
    assert not np.allclose(mem[0], mem[1])                            # the rewrite changed at least one query


def test_fused_ensemble_members_side_by_side_equal_separate_calls():
    """Waves of up to 409 pairs run the three members on three streams (api.hip mms_score_ensemble, ens_lanes; lxmert on its own two lanes inside): ~290 pairs here, every
    stream in the fused-attention regime (>= 1024 token rows), two identical calls bit-identical, members 0 / 2 / 3 bit-identical to the single-model calls on the same pairs."""
    cfgs = {n: small_cfg(n) for n in ("zk", "lds", "lxmert")}
    ws, sc = _members(cfgs)
    ps = synth.make_pairs(10, (28, 30), vocab=cfgs["zk"].vocab, tag="/ens_lanes")
    zb, zb2, lb, xb = _feeds(cfgs, ps)
    sep = [scorers.score_batch(sc["zk"], zb)[1][:, 1], None, scorers.score_batch(sc["lds"], lb)[1][:, 1], scorers.score_batch(sc["lxmert"], xb)[1][:, 1]]
    ens = scorers.EnsembleScorer(sc["zk"], sc["lds"], sc["lxmert"])
    feed = pipeline.ensemble_feed(zb, zb2, xb)
    merged, mem = ens(feed)
    merged_b, mem_b = ens(feed)
    assert torch.equal(merged, merged_b) and torch.equal(mem, mem_b)
    for k in (0, 2, 3):
        assert torch.equal(mem[k], sep[k]), (k, float((mem[k] - sep[k]).abs().max()))
    ens.close()


def test_fused_ensemble_equals_four_separate_calls_and_chunks():
    """The fused call shares the feature split, the label de-duplication and zk's image-token stage; none of that may change a
    score: bit-identical to the four single-model calls, also when the batch is cut into ragged launch waves."""
    cfgs = {n: small_cfg(n) for n in ("zk", "lds", "lxmert")}
    ws, sc = _members(cfgs)
    ps = synth.make_pairs(7, (3, 9), vocab=cfgs["zk"].vocab, tag="/ens_eq")
    zb, zb2, lb, xb = _feeds(cfgs, ps)
    sep = [scorers.score_batch(sc["zk"], zb)[1][:, 1], scorers.score_batch(sc["zk"], zb2)[1][:, 1],
           scorers.score_batch(sc["lds"], lb)[1][:, 1], scorers.score_batch(sc["lxmert"], xb)[1][:, 1]]
    ens = scorers.EnsembleScorer(sc["zk"], sc["lds"], sc["lxmert"])
    merged, mem = ens(pipeline.ensemble_feed(zb, zb2, xb))
    for k in (0, 2, 3):
        assert torch.equal(mem[k], sep[k]), k
    # member 1 (zk on the rewritten query): the fused call re-encodes only the pairs whose query changed -- here a launch of < 256 token rows,
    # i.e. the tiny-launch route (api.hip TINY_ROWS), where the separate call scores all pairs in one launch above it: fp32 summation order
    assert (mem[1] - sep[1]).abs().max() < 2e-5, (mem[1] - sep[1]).abs().max()
    unchanged = torch.as_tensor(np.array([int(q) % 3 != 0 for q in ps.query_id]), device=mem.device)
    assert torch.equal(mem[1][unchanged], mem[0][unchanged])          # untouched queries reuse member 0's score, bit for bit
    w = ens.WEIGHTS
    exp = ((w[0] * mem[0] + w[1] * mem[1]) + w[2] * mem[2]) + w[3] * mem[3]
    assert torch.equal(merged, exp)
    ens.close()
    _, sc2 = _members(cfgs, chunk_pairs=5)
    ens2 = scorers.EnsembleScorer(sc2["zk"], sc2["lds"], sc2["lxmert"])
    merged2, _ = ens2(pipeline.ensemble_feed(zb, zb2, xb))
    assert (merged2 - merged).abs().max() < 2e-5          # waves of 5 pairs: the tiny-launch regime (fp32 summation order)
    m0, mem0 = ens2({k: v[:0] for k, v in pipeline.ensemble_feed(zb, zb2, xb).items()})
    assert m0.shape == (0,) and mem0.shape == (4, 0)
    ens2.close()


def test_full_size_ensemble_properties():
    """config 5 at the headline size (1000 x 30 pairs, full models): a subset against the oracle, permutation equivariance."""
    cfgs = {"zk": ZkConfig(), "lds": LdsConfig(), "lxmert": LxmertConfig()}
    ws, sc = _members(cfgs)
    ps = synth.make_pairs(1000, 30, tag="/ens_size", with_feats=False)
    dev = torch.device("cuda")
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    feats = torch.randn((ps.n, 10, 2048), device=dev, generator=g).clamp_(min=0)
    feats *= (torch.arange(10, device=dev)[None, :] < torch.as_tensor(ps.num_boxes, device=dev)[:, None])[:, :, None]
    zb, zb2, lb, xb = _feeds(cfgs, ps, feats)
    feed = pipeline.ensemble_feed(zb, zb2, xb)
    ens = scorers.EnsembleScorer(sc["zk"], sc["lds"], sc["lxmert"])
    m1, mem1 = ens(feed)
    m1b, _ = ens(feed)
    assert torch.equal(m1, m1b) and torch.isfinite(m1).all() and (m1 >= 0).all() and (m1 <= 1.0 + 1e-6).all()
    idx = np.sort(np.random.RandomState(0).choice(ps.n, 6, replace=False))
    ti = torch.as_tensor(idx, device=dev)
    cut = lambda b: {k: (v[ti].cpu().numpy() if torch.is_tensor(v) else v[idx]) for k, v in b.items()}
    r = (0.2 * O.forward(cfgs["zk"], ws["zk"], cut(zb), np.float64)[1][:, 1] + 0.2 * O.forward(cfgs["zk"], ws["zk"], cut(zb2), np.float64)[1][:, 1]
         + 0.3 * O.forward(cfgs["lds"], ws["lds"], cut(lb), np.float64)[1][:, 1] + 0.3 * O.forward(cfgs["lxmert"], ws["lxmert"], cut(xb), np.float64)[1][:, 1])
    assert np.abs(m1.cpu().numpy()[idx] - r).max() < 1e-3
    perm = np.random.RandomState(1).permutation(ps.n)
    tp = torch.as_tensor(perm, device=dev)
    fp = {k: (v[tp] if torch.is_tensor(v) else v[perm]) for k, v in feed.items()}
    mp_, _ = ens(fp)
    assert (mp_ - m1[tp]).abs().max() < 2e-5
    ens.close()


def test_ensemble_second_zk_member_with_no_and_all_queries_changed():
    cfgs = {n: small_cfg(n) for n in ("zk", "lds", "lxmert")}
    ws, sc = _members(cfgs, chunk_pairs=6)
    ps = synth.make_pairs(4, (3, 5), vocab=cfgs["zk"].vocab, tag="/ens_s2f")
    zb, zb2, lb, xb = _feeds(cfgs, ps)
    ens = scorers.EnsembleScorer(sc["zk"], sc["lds"], sc["lxmert"])
    _, m_same = ens(pipeline.ensemble_feed(zb, zb, xb))                       # rewrite changed nothing: member 1 == member 0
    assert torch.equal(m_same[0], m_same[1])
    allch = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in zb.items()}
    allch["np_idx_query_"][:, 1] = 2000                                       # every query rewritten
    _, m_all = ens(pipeline.ensemble_feed(zb, allch, xb))
    sep = scorers.score_batch(sc["zk"], allch)[1][:, 1]
    assert torch.equal(m_all[1], sep) and torch.equal(m_all[0], m_same[0])
    _, m_mix = ens(pipeline.ensemble_feed(zb, zb2, xb))
    assert torch.equal(m_mix[1], scorers.score_batch(sc["zk"], zb2)[1][:, 1])
    ens.close()
