"""GPU suite, model level: the HIP scorers (through the C ABI) against the CPU oracle and against
the reference-generated goldens.  Tolerance: per-pair ||dlogit||/||logit|| <= 1e-3 in precision
mode 2 (the north-star bound); precision mode 1 is reported against its own looser bound."""
import numpy as np
import pytest
import torch

from helpers import TOL_P1, TOL_P2, lxmert_case_from_meta, load_golden, small_cfg, vecrel
from kddcup_2020_multimodalitiesrecall_2nd_place_amd import scorers, synth, weights
from kddcup_2020_multimodalitiesrecall_2nd_place_amd.config import LdsConfig, LxmertConfig, ZkConfig
from oracle import np_models as O

pytestmark = pytest.mark.gpu


def _batch(cfg, ps, **kw):
    return synth.batch_for(cfg, ps, **kw)


def _hip_logits(cfg, w, b, **kw):
    s = scorers.make_scorer(cfg, w, **kw)
    logits, probs = scorers.score_batch(s, b)
    torch.cuda.synchronize()
    out = logits.cpu().numpy(), probs.cpu().numpy()
    s.close()
    return out


@pytest.mark.parametrize("name", ["zk", "lds", "lxmert"])
def test_stagewise_hidden_state_matches_oracle(name):
    """Localises any mismatch: embedding output, then each encoder step (stop_after debug knob)."""
    cfg = small_cfg(name)
    w = weights.make_weights(cfg)
    ps = synth.make_pairs(3, (3, 6), vocab=cfg.vocab, tag="/stage")
    b = _batch(cfg, ps)
    inter = {}
    O.forward(cfg, w, b, np.float64, inter)
    if name == "lxmert":
        T = cfg.text_len
        steps = [(0, np.concatenate([inter["lang_emb"].reshape(-1, 768), inter["visn_emb"].reshape(-1, 768)]))]
        steps.append((cfg.l_layers + cfg.r_layers,
                      np.concatenate([inter["lang_l"].reshape(-1, 768), inter["visn_r"].reshape(-1, 768)])))
        for i in range(cfg.x_layers):
            steps.append((cfg.l_layers + cfg.r_layers + i + 1,
                          np.concatenate([inter["lang_x%d" % i].reshape(-1, 768), inter["visn_x%d" % i].reshape(-1, 768)])))
    else:
        steps = [(0, inter["embedding_output"].reshape(-1, 768))]
        steps += [(i + 1, inter["layer_%d" % i].reshape(-1, 768)) for i in range(cfg.layers)]
    for stop, ref in steps:
        s = scorers.make_scorer(cfg, w, stop_after=stop, pack_tokens=False)  # dense rows: directly comparable
        scorers.score_batch(s, b)
        got = s.read_hidden(ref.shape[0]).cpu().numpy()
        s.close()
        err = np.abs(got - ref).max()
        assert err < 2e-4 * max(1.0, np.abs(ref).max()), (name, "stop_after", stop, err)


@pytest.mark.parametrize("name", ["zk", "lds", "lxmert"])
@pytest.mark.parametrize("dedup", [True, False])
@pytest.mark.parametrize("pack", [True, False])
def test_shallow_logits_match_oracle(name, dedup, pack):
    cfg = small_cfg(name)
    w = weights.make_weights(cfg)
    ps = synth.make_pairs(5, (2, 7), vocab=cfg.vocab, tag="/shallow")
    b = _batch(cfg, ps, labels="valid") if name == "zk" else _batch(cfg, ps)
    ref, ref_p = O.forward(cfg, w, b, np.float64)
    got, got_p = _hip_logits(cfg, w, b, dedup_labels=dedup, pack_tokens=pack)
    assert vecrel(got, ref).max() < TOL_P2, vecrel(got, ref)
    assert np.abs(got_p - ref_p).max() < 1e-3


@pytest.mark.parametrize("name", ["zk", "lxmert"])
def test_packed_equals_dense_and_mask_edge_cases(name):
    """Dropping padded tokens must not change a single logit: a masked key's softmax weight is exactly 0.
    Edge cases: every key masked (pair kept whole, uniform softmax), CLS masked as a key but still a query
    row, non-prefix masks (lxmert), no live box."""
    cfg = small_cfg(name)
    w = weights.make_weights(cfg)
    ps = synth.make_pairs(4, (3, 5), vocab=cfg.vocab, tag="/packedge")
    b = _batch(cfg, ps)
    if name == "zk":
        b["len_query_"][0] = 0; b["num_boxes"][0] = 0          # nothing live: reference attends uniformly to all 30
        b["len_query_"][1] = 0                                  # CLS masked as key, boxes live
        b["num_boxes"][2] = 0                                   # text only
        b["len_query_"][3] = 20; b["num_boxes"][3] = 13         # everything live, num_boxes > 10
    else:
        b["input_mask"][0] = 0; b["visual_attention_mask"][0] = 0
        b["input_mask"][1] = 0; b["input_mask"][1, [2, 5, 9]] = 1     # non-prefix mask, CLS masked
        b["visual_attention_mask"][2] = 0; b["visual_attention_mask"][2, [1, 7]] = 1
        b["input_mask"][3] = 1; b["visual_attention_mask"][3] = 1
    ref, _ = O.forward(cfg, w, b, np.float64)
    dense, _ = _hip_logits(cfg, w, b, pack_tokens=False)
    packed, _ = _hip_logits(cfg, w, b, pack_tokens=True)
    assert vecrel(dense, ref).max() < TOL_P2, vecrel(dense, ref)
    assert vecrel(packed, ref).max() < TOL_P2, vecrel(packed, ref)
    # not bitwise: dropping masked keys changes which lanes hold the live keys, i.e. the fp32 summation
    # ORDER of softmax / P.V (the dropped terms themselves are exact zeros) -> roundoff-level differences
    assert np.abs(packed - dense).max() < 1e-4, np.abs(packed - dense).max(0)
    # chunks of 5 pairs are launches of < 256 token rows: the tiny-launch route (wide projections split over K, api.hip TINY_ROWS) sums K in
    # another order than the 16-pair launch above -> fp32 round-off; chunks inside one launch-size regime are bit-identical
    # (test_reference_call_sizes_of_one_and_five_pairs, test_testB_like_set_single_gpu_matches_shardwise_scoring)
    packed_chunked, _ = _hip_logits(cfg, w, b, pack_tokens=True, chunk_pairs=5)
    assert np.abs(packed_chunked - packed).max() < 1e-4


@pytest.mark.parametrize("name", ["zk", "lds", "lxmert"])
def test_full_depth_logits_match_oracle(name):
    """BASELINE.json config 1 shape class (full 12 / 9-5-5 depth, full width), 2 queries x 12."""
    cfg = {"zk": ZkConfig(), "lds": LdsConfig(), "lxmert": LxmertConfig()}[name]
    w = weights.make_weights(cfg)
    ps = synth.make_pairs(2, 12, tag="/full")
    b = _batch(cfg, ps)
    ref, ref_p = O.forward(cfg, w, b, np.float64)
    got, got_p = _hip_logits(cfg, w, b)
    e2 = vecrel(got, ref)
    assert e2.max() < TOL_P2, e2
    assert np.abs(got_p - ref_p).max() < 1e-3
    got1, _ = _hip_logits(cfg, w, b, precision=1)
    e1 = vecrel(got1, ref)
    print("\n[%s] vec-rel logit error: precision2 max %.2e  precision1 max %.2e" % (name, e2.max(), e1.max()))
    assert e1.max() < TOL_P1, e1


@pytest.mark.parametrize("name", ["zk", "lds", "lxmert"])
def test_ndcg5_parity_with_oracle_scores(name):
    """nDCG@5 computed from the HIP scores equals nDCG@5 from the oracle's scores within 1e-3 on a
    valid-like synthetic set (ragged candidate lists, ~6 relevant per query), and the per-query top-5 product lists agree
    wherever the oracle's 5th / 6th scores are not a near tie.  Reduced depth keeps the fp64 oracle in seconds."""
    from kddcup_2020_multimodalitiesrecall_2nd_place_amd import ndcg
    cfg = {"zk": ZkConfig(layers=4), "lds": LdsConfig(layers=4), "lxmert": LxmertConfig(l_layers=3, r_layers=2, x_layers=2)}[name]
    w = weights.make_weights(cfg)
    ps = synth.make_pairs(16, (8, 30), tag="/ndcg")
    b = _batch(cfg, ps)
    if name == "zk":
        b["labels"] = ps.relevance.astype(np.int64)          # valid convention: the AM-softmax head sees the ground truth
    _, ref_p = O.forward(cfg, w, b, np.float64)
    _, got_p = _hip_logits(cfg, w, b)
    truth = {}
    for q, p_, r in zip(ps.query_id, ps.product_id, ps.relevance):
        truth.setdefault(str(int(q)), [])
        if r:
            truth[str(int(q))].append(str(int(p_)))
    truth = {q: v for q, v in truth.items() if v}
    n_ref = ndcg.ndcg_from_arrays(ps.query_id, ps.product_id, ref_p[:, 1], truth)
    n_got = ndcg.ndcg_from_arrays(ps.query_id, ps.product_id, got_p[:, 1], truth)
    print("\n[%s] nDCG@5 oracle %.6f  HIP %.6f" % (name, n_ref, n_got))
    assert abs(n_ref - n_got) <= 1e-3
    assert 0.0 < n_ref <= 1.0
    for q in np.unique(ps.query_id):
        m = ps.query_id == q
        r, g, pid = ref_p[m, 1], got_p[m, 1], ps.product_id[m]
        order = np.argsort(-r, kind="stable")
        if len(order) > 5 and abs(r[order[4]] - r[order[5]]) < 1e-4:
            continue
        assert set(pid[order[:5]]) == set(pid[np.argsort(-g, kind="stable")[:5]])


@pytest.mark.parametrize("gname", ["lxmert_shallow.npz", "lxmert_full.npz"])
def test_lxmert_matches_reference_golden(gname):
    """HIP path against the REFERENCE's own fp32 outputs (tests/golden/make_lxmert_golden.py)."""
    g, meta = load_golden(gname)
    cfg, w, b = lxmert_case_from_meta(meta)
    got, _ = _hip_logits(cfg, w, b)
    assert vecrel(got, g["logit"]).max() < TOL_P2


def test_zk_margin_branch_on_gpu():
    cfg = small_cfg("zk", layers=1)
    w = weights.make_weights(cfg)
    ps = synth.make_pairs(2, 4, vocab=cfg.vocab, tag="/margin")
    b = synth.zk_batch(ps, cfg.text_len)
    inter = {}
    O.forward(cfg, w, b, np.float64, inter)
    k = w["cls/seq_relationship/am_kernel"].copy()
    k[:, 1] = inter["pooled"][0] / np.linalg.norm(inter["pooled"][0])
    w2 = dict(w)
    w2["cls/seq_relationship/am_kernel"] = k.astype(np.float32)
    for lab in (0, 1):
        bb = dict(b, labels=np.full(ps.n, lab, np.int64))
        ref, _ = O.forward(cfg, w2, bb, np.float64)
        got, _ = _hip_logits(cfg, w2, bb)
        assert np.abs(got - ref).max() < 3e-2 and vecrel(got, ref).max() < TOL_P2, (lab, got, ref)


def test_edge_cases_chunking_ragged_and_truncation():
    """chunk_pairs < B (ragged last chunk), B == 1, num_boxes > 10 (clamped by sequence_mask),
    over-long query (SEP truncated, load_data_v4.py:83-85), all-10-box pairs."""
    cfg = small_cfg("zk", layers=1)
    w = weights.make_weights(cfg)
    ps = synth.make_pairs(3, (3, 5), vocab=cfg.vocab, tag="/edge", max_query_body=25, all_boxes=True)
    b = synth.zk_batch(ps, cfg.text_len)
    b["num_boxes"] = b["num_boxes"] + 5
    ref, _ = O.forward(cfg, w, b, np.float64)
    got, _ = _hip_logits(cfg, w, b, chunk_pairs=4)
    assert vecrel(got, ref).max() < TOL_P2
    one = {k: v[:1] for k, v in b.items()}
    got1, _ = _hip_logits(cfg, w, one)
    assert vecrel(got1, ref[:1]).max() < TOL_P2


def test_call_surfaces_numpy_in_numpy_out():
    cfg = small_cfg("lds", layers=1)
    w = weights.make_weights(cfg)
    ps = synth.make_pairs(2, 3, vocab=cfg.vocab, tag="/surf")
    b = synth.lds_batch(ps, cfg.text_len)
    s = scorers.LdsScorer(cfg, w)
    probs = s(b)
    assert isinstance(probs, np.ndarray) and probs.shape == (ps.n, 2)
    _, ref_p = O.forward(cfg, w, b, np.float64)
    assert np.abs(probs - ref_p).max() < 1e-3
    s.close()
    cfg = small_cfg("lxmert", l_layers=1, r_layers=1, x_layers=1)
    w = weights.make_weights(cfg)
    b = synth.lxmert_batch(ps, cfg.text_len)
    s = scorers.LxmertScorer(cfg, w)
    x_norm, mlm, logit = s(b["input_ids"], b["boxes_label_input_ids"], None, b["input_mask"], None,
                           b["boxes_label_input_mask"], b["feats"], b["boxes"], b["visual_attention_mask"])
    assert mlm is None and logit.shape == (ps.n, 2)
    ref, _ = O.forward(cfg, w, b, np.float64)
    assert vecrel(logit, ref).max() < TOL_P2
    s.close()
    cfg = small_cfg("zk", layers=1)
    w = weights.make_weights(cfg)
    b = synth.zk_batch(ps, cfg.text_len)
    s = scorers.ZkScorer(cfg, w)
    loss, probs, loss_list = s(b["num_boxes"], b["np_boxes_5"], b["np_images_features"], b["np_idx_class_labels"], None,
                               b["np_idx_query_"], b["len_query_"], b["labels"], b["segment_ids"], None, None, False)
    _, ref_p = O.forward(cfg, w, b, np.float64)
    assert np.abs(probs - ref_p).max() < 1e-3 and len(loss_list) == 1
    s.close()


def test_tsv_to_score_file_pipeline(tmp_path):
    """featurizer -> HIP scorer -> score file -> reader, against the oracle on the same featurised batch."""
    import os
    from helpers import GOLDEN
    from kddcup_2020_multimodalitiesrecall_2nd_place_amd import featurizer as F, pipeline, scorefile
    d = os.path.join(GOLDEN, "featurizer")
    tok = F.WordPieceTokenizer(os.path.join(d, "vocab_small.txt"))
    table = F.load_label_table(os.path.join(d, "labels.txt"))
    lines = open(os.path.join(d, "records.tsv")).read().splitlines()
    for name in ("zk", "lxmert"):
        cfg = small_cfg(name)
        w = weights.make_weights(cfg)
        s = scorers.make_scorer(cfg, w)
        out = tmp_path / ("scores_%s" % name + (".csv" if name == "lxmert" else ".txt"))
        qid, pid, score = pipeline.predict_tsv(s, ["product_id\tfoo"] + lines, table, tok, str(out), batch_pairs=3)
        if name == "lxmert":     # the KDD.predict-shaped driver: same scores, grouped by query, CSV in the reference's dict order
            mp, ml, rsp = pipeline.kdd_predict(s, ["product_id\tfoo"] + lines, table, tok, str(tmp_path / "kdd.csv"), batch_pairs=4)
            assert ml == [1] * len(lines) and mp == [int(x > 0.5) for x in score]
            assert sum(len(v) for v in rsp.values()) == len(lines)
            for q, p_, sc in zip(qid, pid, score):
                assert abs(dict(rsp[int(q)])[int(p_)] - sc) < 1e-6
            a_, b_ = scorefile.read_scores(str(tmp_path / "kdd.csv")), scorefile.read_scores(str(out))
            assert {q: set(v) for q, v in a_.items()} == {q: set(v) for q, v in b_.items()}
            assert all(abs(a_[q][p_] - b_[q][p_]) < 1e-6 for q in a_ for p_ in a_[q])
        s.close()
        recs = [F.read_line(l, table, tok) for l in lines]
        b = (F.zk_batch if name == "zk" else F.lxmert_batch)(recs, cfg.text_len)
        _, ref_p = O.forward(cfg, w, b, np.float64)
        assert np.abs(score - ref_p[:, 1]).max() < 1e-3
        back = scorefile.read_scores(str(out))
        for q, p_, sc in zip(qid, pid, score):
            assert abs(back[str(q)][str(p_)] - sc) < 1e-6
        # the native featurizer route (libmmfeat -> pinned buffers -> same scorer) writes the same file
        tsv = tmp_path / ("valid_%s.tsv" % name)
        tsv.write_text("product_id\tfoo\n" + "\n".join(lines) + "\n", encoding="utf-8")
        s2 = scorers.make_scorer(cfg, w)
        out2 = tmp_path / ("native_" + out.name)
        vocab = os.path.join(d, "vocab_small.txt")
        if name == "lxmert":                                  # predict_tsv above used the TF-flavour tokenizer for both models
            from kddcup_2020_multimodalitiesrecall_2nd_place_amd.featurizer_native import NativeFeaturizer
            nb = NativeFeaturizer(vocab, table, "lxmert").batch(lines)
            tok_hf = F.WordPieceTokenizer(vocab, max_input_chars_per_word=100, never_split=F.SPECIALS)
            assert np.array_equal(nb["input_ids"], F.lxmert_batch([F.read_line(l, table, tok_hf) for l in lines], cfg.text_len)["input_ids"])
        q2, p2, sc2 = pipeline.predict_tsv_native(s2, str(tsv), vocab, table, str(out2), batch_pairs=3)
        s2.close()
        assert np.array_equal(q2, qid) and np.array_equal(p2, pid)
        assert np.abs(sc2 - score).max() < 1e-5


@pytest.mark.parametrize("name", ["zk", "lxmert"])
def test_precision3_follows_unrounded_fp32_checkpoint(name):
    """Real checkpoints are not bf16-representable.  Mode 3 (activations and weights split, 3 MFMA passes) must
    stay inside 1e-3 against the fp64 oracle on the UNROUNDED fp32 weights; mode 2 (weights stored as bf16) is
    reported next to it (~5e-3 from the weight rounding alone)."""
    cfg = {"zk": ZkConfig(layers=4), "lxmert": LxmertConfig(l_layers=3, r_layers=2, x_layers=2)}[name]
    w = weights.make_weights(cfg, bf16_matrices=False)
    ps = synth.make_pairs(2, 8, tag="/p3")
    b = _batch(cfg, ps)
    ref, _ = O.forward(cfg, w, b, np.float64)
    got3, _ = _hip_logits(cfg, w, b, precision=3)
    got2, _ = _hip_logits(cfg, w, b, precision=2)
    e3, e2 = vecrel(got3, ref).max(), vecrel(got2, ref).max()
    print("\n[%s, fp32 weights] vec-rel logit error: precision3 %.2e  precision2 %.2e" % (name, e3, e2))
    assert e3 < TOL_P2, e3
    got_auto, _ = _hip_logits(cfg, w, b)                       # default precision="auto" picks mode 3 for such a checkpoint
    assert np.array_equal(got_auto, got3)
    assert e2 < TOL_P1


def test_full_size_workload_properties():
    """BASELINE.json's headline size (1000 queries x 30 candidates, full 12-layer zk): size-independent properties
    -- a random 24-pair subset against the oracle, permutation equivariance over pairs, independence from the
    internal chunk size, idempotence, duplicated pairs scoring identically."""
    cfg = ZkConfig()
    w = weights.make_weights(cfg)
    ps = synth.make_pairs(1000, 30, tag="/fullsize", with_feats=False)
    dev = torch.device("cuda")
    g = torch.Generator(device=dev)
    g.manual_seed(123)
    feats = torch.randn((ps.n, 10, 2048), device=dev, generator=g).clamp_(min=0)
    feats *= (torch.arange(10, device=dev)[None, :] < torch.as_tensor(ps.num_boxes, device=dev)[:, None])[:, :, None]
    ps.feats = feats
    b = synth.zk_batch(ps, cfg.text_len)
    # duplicate pair 7 into slot 29999
    for k in b:
        if torch.is_tensor(b[k]):
            b[k][-1] = b[k][7]
        else:
            b[k][-1] = b[k][7]
    s = scorers.ZkScorer(cfg, w, chunk_pairs=8192)
    l1, p1 = scorers.score_batch(s, b)
    l1b, _ = scorers.score_batch(s, b)
    torch.cuda.synchronize()
    assert torch.equal(l1, l1b)                                   # idempotent / deterministic
    # identical inputs in two different slots of the batch: the library default (fuse_attention = 2) attends 16-query tiles of a sub-tile block-diagonally, so the
    # order in which a pair's keys are summed depends on where the pair sits in its sub-tile: equal to fp32 summation order (far inside the 1e-3 contract) ...
    assert vecrel(l1[-1:].cpu().numpy(), l1[7:8].cpu().numpy()).max() < 1e-4
    # ... and bit-identical with the exact-fp32 attention arithmetic (fuse_attention = 1: one work item per pair, position-independent)
    se = scorers.ZkScorer(cfg, w, chunk_pairs=8192, fuse_attention=1)
    le, _ = scorers.score_batch(se, b)
    assert se.handle.counter(0) > 0 and torch.equal(le[-1], le[7])
    se.close()
    assert torch.isfinite(l1).all() and (p1.sum(1) - 1).abs().max() < 1e-6
    l1 = l1.cpu().numpy()
    # oracle on a random subset
    idx = np.sort(np.random.RandomState(0).choice(ps.n, 24, replace=False))
    sub = {k: (v[torch.as_tensor(idx, device=dev)].cpu().numpy() if torch.is_tensor(v) else v[idx]) for k, v in b.items()}
    ref, _ = O.forward(cfg, w, sub, np.float64)
    assert vecrel(l1[idx], ref).max() < TOL_P2
    # permutation equivariance (the packed layout, row tiles and attention offsets all change)
    perm = np.random.RandomState(1).permutation(ps.n)
    bp = {k: (v[torch.as_tensor(perm, device=dev)] if torch.is_tensor(v) else v[perm]) for k, v in b.items()}
    lp, _ = scorers.score_batch(s, bp)
    assert np.abs(lp.cpu().numpy() - l1[perm]).max() < 2e-4
    s.close()
    # chunk-size independence
    s2 = scorers.ZkScorer(cfg, w, chunk_pairs=3001)
    l2, _ = scorers.score_batch(s2, b)
    assert np.abs(l2.cpu().numpy() - l1).max() < 2e-4
    s2.close()


def test_empty_and_single_pair_batches():
    """B == 0 is a no-op returning empty outputs; B == 1 works (the reference's zk driver feeds one pair per call)."""
    for name in ("zk", "lds", "lxmert"):
        cfg = small_cfg(name, **({"layers": 1} if name != "lxmert" else {"l_layers": 1, "r_layers": 1, "x_layers": 1}))
        w = weights.make_weights(cfg)
        ps = synth.make_pairs(1, 2, vocab=cfg.vocab, tag="/one")
        b = _batch(cfg, ps)
        s = scorers.make_scorer(cfg, w)
        one = {k: v[:1] for k, v in b.items()}
        l1, p1 = scorers.score_batch(s, one)
        ref, _ = O.forward(cfg, w, one, np.float64)
        assert vecrel(l1.cpu().numpy(), ref).max() < TOL_P2
        zero = {k: v[:0] for k, v in b.items()}
        l0, p0 = scorers.score_batch(s, zero)
        assert l0.shape == (0, 2) and p0.shape == (0, 2)
        s.close()


@pytest.mark.parametrize("name", ["zk", "lds", "lxmert"])
@pytest.mark.parametrize("precision", [2, 3])
def test_reference_call_sizes_of_one_and_five_pairs(name, precision):
    """The reference drivers call with 1 (zk, evaluate_normal.py:15,216) and 5 (lds, run_pretraining_predict_score.py:523) pairs: launches
    of < 256 token rows run the tiny-launch route (api.hip gemm(): wide projections split over K + k_splitk_reduce; N = 768 projections split
    with their partials summed in the LayerNorm kernel).  Against the fp64 oracle, and five pairs in one call == five calls of one, bit for bit."""
    cfg = small_cfg(name, **({"layers": 3} if name != "lxmert" else {}))
    w = weights.make_weights(cfg, bf16_matrices=(precision != 3))
    ps = synth.make_pairs(1, 5, vocab=cfg.vocab, tag="/callsize")
    b = _batch(cfg, ps)
    s = scorers.make_scorer(cfg, w, precision=precision)
    five = scorers.score_batch(s, b)[0].cpu().numpy()
    ones = np.concatenate([scorers.score_batch(s, {k: v[i:i + 1] for k, v in b.items()})[0].cpu().numpy() for i in range(5)])
    s.close()
    ref, _ = O.forward(cfg, w, b, np.float64)
    assert vecrel(five, ref).max() < TOL_P2, vecrel(five, ref).max()
    assert np.array_equal(five, ones), np.abs(five - ones).max()


def test_three_model_ensemble_on_one_gpu():
    """Config 5 host side: zk (x2 query variants) + lds + lxmert on the same shard, merged 0.2/0.2/0.3/0.3 (main.py:59)."""
    import os
    from helpers import GOLDEN
    from kddcup_2020_multimodalitiesrecall_2nd_place_amd import featurizer as F, pipeline
    d = os.path.join(GOLDEN, "featurizer")
    tok_tf = F.WordPieceTokenizer(os.path.join(d, "vocab_small.txt"))
    tok_hf = F.WordPieceTokenizer(os.path.join(d, "vocab_small.txt"), max_input_chars_per_word=100, never_split=F.SPECIALS)
    table = F.load_label_table(os.path.join(d, "labels.txt"))
    lines = open(os.path.join(d, "records.tsv")).read().splitlines()
    cfgs = {n: small_cfg(n) for n in ("zk", "lds", "lxmert")}
    ws = {n: weights.make_weights(c) for n, c in cfgs.items()}
    sc = {n: scorers.make_scorer(cfgs[n], ws[n]) for n in cfgs}
    ens = pipeline.EnsembleScorer(sc["zk"], sc["lds"], sc["lxmert"])
    qid, pid, merged, parts = ens.score_lines(lines, table, tok_tf, tok_hf, batch_pairs=4)
    # the native-featurizer route over the same records as a TSV file (header + blank line): same merged scores
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".tsv", delete=False, encoding="utf-8") as f:
        f.write("product_id\tx\n" + "\n\n".join(lines) + "\n")
    q2, p2, merged2, parts2 = ens.score_tsv_native(f.name, os.path.join(d, "vocab_small.txt"), table, batch_pairs=2)
    os.remove(f.name)
    assert np.array_equal(q2, qid) and np.array_equal(p2, pid)
    assert np.abs(merged2 - merged).max() < 1e-5 and all(np.abs(x - y).max() < 1e-5 for x, y in zip(parts, parts2))
    for s_ in sc.values():
        s_.close()
    rec = [F.read_line(l, table, tok_tf) for l in lines]
    rec2 = [F.read_line(l, table, tok_tf, sen2forest=True) for l in lines]
    rech = [F.read_line(l, table, tok_hf) for l in lines]
    r1 = O.forward(cfgs["zk"], ws["zk"], F.zk_batch(rec), np.float64)[1][:, 1]
    r2 = O.forward(cfgs["zk"], ws["zk"], F.zk_batch(rec2), np.float64)[1][:, 1]
    r3 = O.forward(cfgs["lds"], ws["lds"], F.lds_batch(rec), np.float64)[1][:, 1]
    r4 = O.forward(cfgs["lxmert"], ws["lxmert"], F.lxmert_batch(rech), np.float64)[1][:, 1]
    assert np.abs(merged - (0.2 * r1 + 0.2 * r2 + 0.3 * r3 + 0.3 * r4)).max() < 1e-3
    assert not np.allclose(parts[0], parts[1])      # the sen2forest rewrite changes record 1's query
