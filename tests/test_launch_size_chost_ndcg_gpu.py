"""GPU suite: launch-size independence in every precision mode (the big-M engines against the small-tile engines on
the SAME pairs), head-major stores of the three-pass engine, fp8 x fused ensemble, full-depth nDCG@5 parity."""
import numpy as np
import pytest
import torch

from helpers import TOL_P2, small_cfg, vecrel
from kddcup_2020_multimodalitiesrecall_2nd_place_amd import scorers, synth, weights
from kddcup_2020_multimodalitiesrecall_2nd_place_amd.config import LdsConfig, LxmertConfig, ZkConfig
from oracle import np_models as O

pytestmark = pytest.mark.gpu


def _hip_logits(cfg, w, b, **kw):
    s = scorers.make_scorer(cfg, w, **kw)
    logits, probs = scorers.score_batch(s, b)
    torch.cuda.synchronize()
    out = logits.cpu().numpy(), probs.cpu().numpy()
    s.close()
    return out


@pytest.mark.parametrize("name", ["zk", "lxmert", "lds"])
@pytest.mark.parametrize("precision", [2, 3])
@pytest.mark.parametrize("pack,exact", [(True, False), (False, False), (True, True)])
def test_large_launch_equals_small_chunks(name, precision, pack, exact):
    """A launch of >= 16384 token rows runs the persistent ping-pong engines (gemm_pp.hip, and gemm_ppw.hip in mode 3, whose
    32-column wave tiles store the head-major Q/K/V blocks from two waves per head), a 48-pair chunk the register-staged tiles.
    Same pairs, same weights: the two routes may differ by fp32 summation order only.  (the mode-3 big-M route had
    never been compared with anything.)  `exact`: the big launch on the exact-fp32 attention MFMAs and the two-kernel LayerNorm -- then the
    two sides differ by summation order alone (3e-4 on the worst of ~1300 pairs of these shallow random models); the DEFAULT big-launch route (round 4: split-bf16 attention MFMAs, LayerNorm with a
    one-pass variance in the GEMM epilogue) adds its own 2^-16-class terms, bounded at 5e-4 here and against the oracle below."""
    cfg = small_cfg(name)
    w = weights.make_weights(cfg, bf16_matrices=(precision != 3))
    ps = synth.make_pairs(48, (24, 30), vocab=cfg.vocab, tag="/big%d" % precision)     # ~1080 pairs: >= 16384 rows even when packed
    b = synth.batch_for(cfg, ps)
    routes = dict(fuse_attention=0 if name == "lxmert" else 1, fuse_layernorm=0) if exact else {}
    big, _ = _hip_logits(cfg, w, b, precision=precision, pack_tokens=pack, **routes)
    small, _ = _hip_logits(cfg, w, b, precision=precision, pack_tokens=pack, chunk_pairs=48, **routes)
    # the metric of SURVEY.md section 8(d) with its absolute floor: a few of the ~1300 pairs of these shallow random models have logit vectors that
    # nearly vanish, where a purely relative difference of two fp32-faithful routes diverges
    e = np.linalg.norm(big - small, axis=1) / np.maximum(np.linalg.norm(small, axis=1), 0.1)
    print("\n[%s p%d pack=%d] %d pairs: one launch vs 48-pair chunks, vec-rel max %.2e (median %.2e)" % (name, precision, pack, len(big), e.max(), np.median(e)))
    assert e.max() < (5e-4 if not exact else 3e-4) and np.median(e) < 3e-5, (e.max(), np.median(e))       # exact: summation order only (big-M engines vs split-K small tiles)
    # and the pairs that moved most + a sample of the rest against the fp64 oracle: BOTH routes inside the contract
    sel = np.unique(np.concatenate([np.argsort(-e)[:6], np.arange(0, len(big), max(1, len(big) // 18))[:18]]))
    sub = {k: (v[sel] if hasattr(v, "shape") and v.shape[:1] == (len(big),) else v) for k, v in b.items()}
    ref, _ = O.forward(cfg, w, sub, np.float64)
    floor = np.maximum(np.linalg.norm(ref, axis=1), 0.1)
    assert (np.linalg.norm(big[sel] - ref, axis=1) / floor).max() < TOL_P2 and (np.linalg.norm(small[sel] - ref, axis=1) / floor).max() < TOL_P2


def test_plain_c_host_scores_on_the_gpu_and_matches_ctypes(tmp_path):
    """INTEGRATION.md's C host for real -- tests/chost/lds_host.c, compiled with gcc against include/mmscore.h,
    loads seeded weights from a flat file, scores 8 pairs on the GPU through mms_create / mms_load_weight / mms_finalize /
    mms_score_lds with buffers it allocated itself (HIP runtime C API), and its logits equal the ctypes route's bit for bit."""
    import os
    import shutil
    import struct
    import subprocess
    from kddcup_2020_multimodalitiesrecall_2nd_place_amd import lib
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = small_cfg("lds")
    w = weights.make_weights(cfg)
    ps = synth.make_pairs(1, 8, vocab=cfg.vocab, tag="/chost")
    b = synth.batch_for(cfg, ps)
    wf, ff, of = tmp_path / "w.bin", tmp_path / "feed.bin", tmp_path / "logits.bin"
    with open(wf, "wb") as f:
        f.write(struct.pack("<i", len(w)))
        for k, v in w.items():
            a = np.ascontiguousarray(v, np.float32)
            f.write(struct.pack("<i", len(k)) + k.encode() + struct.pack("<i", a.ndim) + struct.pack("<%dq" % a.ndim, *a.shape) + a.tobytes())
    with open(ff, "wb") as f:
        f.write(struct.pack("<qq", ps.n, cfg.text_len))
        for key, dt in (("input_ids", np.int64), ("segment_ids", np.int64), ("features", np.float32), ("labelfeat", np.int64)):
            f.write(np.ascontiguousarray(b[key], dt).tobytes())
    csrc = os.path.dirname(lib.LIB_PATH)
    exe = tmp_path / "lds_host"
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(root, "include"),
                           "-I", "/opt/rocm/include", os.path.join(root, "tests", "chost", "lds_host.c"), "-o", str(exe), "-L", csrc,
                           "-lmmscore", "-L", "/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + csrc, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([str(exe), str(cfg.layers), str(cfg.vocab), str(cfg.inter), str(wf), str(ff), str(of)], capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr + out.stdout
    got = np.fromfile(of, np.float32).reshape(ps.n, 2)
    ref, _ = _hip_logits(cfg, w, b, precision=2)
    assert np.array_equal(got, ref), np.abs(got - ref).max()
    oracle, _ = O.forward(cfg, w, b, np.float64)
    assert vecrel(got, oracle).max() < TOL_P2


def test_fp8_fused_ensemble_full_model_size():
    """BASELINE.json config 5 as it is worded -- the 3-model ensemble through the fused entry point WITH fp8 MFMA weights
    (precision 4), full model size.  The fused call must equal the four separate precision-4 calls bit for bit; against the fp64
    oracle the merged score deviates by what the fp8 format costs (reported; outside the 1e-3 contract)."""
    from kddcup_2020_multimodalitiesrecall_2nd_place_amd import pipeline
    cfgs = {"zk": ZkConfig(), "lds": LdsConfig(), "lxmert": LxmertConfig()}
    ws = {n: weights.make_weights(c) for n, c in cfgs.items()}
    sc = {n: scorers.make_scorer(cfgs[n], ws[n], precision=4) for n in cfgs}
    ps = synth.make_pairs(6, (8, 14), tag="/ens_f8")
    zb = synth.zk_batch(ps, cfgs["zk"].text_len)
    zb2 = synth.zk_batch(synth.sen2forest_variant(ps), cfgs["zk"].text_len)
    lb = synth.lds_batch(ps, cfgs["lds"].text_len)
    xb = synth.lxmert_batch(ps, cfgs["lxmert"].text_len)
    sep = [scorers.score_batch(sc["zk"], zb)[1][:, 1], scorers.score_batch(sc["zk"], zb2)[1][:, 1],
           scorers.score_batch(sc["lds"], lb)[1][:, 1], scorers.score_batch(sc["lxmert"], xb)[1][:, 1]]
    ens = scorers.EnsembleScorer(sc["zk"], sc["lds"], sc["lxmert"])
    merged, mem = ens(pipeline.ensemble_feed(zb, zb2, xb))
    torch.cuda.synchronize()
    for k in range(4):
        assert torch.equal(mem[k], sep[k]), k
    w = ens.WEIGHTS
    assert torch.equal(merged, ((w[0] * sep[0] + w[1] * sep[1]) + w[2] * sep[2]) + w[3] * sep[3])
    merged = merged.cpu().numpy()
    ens.close()
    r = [O.forward(cfgs["zk"], ws["zk"], zb, np.float64)[1][:, 1], O.forward(cfgs["zk"], ws["zk"], zb2, np.float64)[1][:, 1],
         O.forward(cfgs["lds"], ws["lds"], lb, np.float64)[1][:, 1], O.forward(cfgs["lxmert"], ws["lxmert"], xb, np.float64)[1][:, 1]]
    ref = 0.2 * r[0] + 0.2 * r[1] + 0.3 * r[2] + 0.3 * r[3]
    d = np.abs(merged - ref)
    print("\n[ensemble, precision 4, full size, %d pairs] |merged - oracle|: median %.3f max %.3f; corr %.3f"
          % (len(ref), np.median(d), d.max(), np.corrcoef(merged, ref)[0, 1]))
    assert np.isfinite(merged).all() and d.max() < 0.15 and np.corrcoef(merged, ref)[0, 1] > 0.8


@pytest.mark.parametrize("name,n_queries", [("zk", 96), ("lds", 56), ("lxmert", 24)])
def test_full_depth_ndcg5_on_a_valid_like_set(name, n_queries):
    """north_star's "nDCG@5 on valid within 1e-3" at the depth the models ship with (12 / 9-5-5 layers) on a
    valid-like set: queries x 9..30 candidates with ground-truth relevance (evaluation.py:4-38), the zk head fed those labels
    (load_data_v4.py:259-263).  Checker = the oracle's fp32 CPU port (torch restatement for zk / lds, numpy fp32 for lxmert; the fp64
    numpy oracle would take an hour; they agree to 1e-5 on the CPU suite).  valid.tsv has 496 queries; the CPU port manages ~55 (zk),
    ~40 (lds), ~12 (lxmert) pairs/s on the GPU box's host cores, so the sets are a fifth / a ninth / a twentieth of that size (round 4: the
    CPU port's time is what this test costs -- 215 s of an 800 s suite with 248 / 124 / 48 queries; the driver stops the suite at 1200 s)."""
    from kddcup_2020_multimodalitiesrecall_2nd_place_amd import ndcg
    from oracle import torch_models
    cfg = {"zk": ZkConfig(), "lds": LdsConfig(), "lxmert": LxmertConfig()}[name]
    w = weights.make_weights(cfg)
    ps = synth.make_pairs(n_queries, (9, 30), tag="/valid_full")
    b = synth.batch_for(cfg, ps)
    if name == "zk":
        b["labels"] = ps.relevance.astype(np.int64)
    _, got_p = _hip_logits(cfg, w, b)
    ref_p = []
    for lo in range(0, ps.n, 512):           # the CPU port in batches (memory)
        sub = {k: (v[lo:lo + 512] if hasattr(v, "shape") and v.shape[:1] == (ps.n,) else v) for k, v in b.items()}
        out = torch_models.forward(cfg, w, sub) if name != "lxmert" else O.forward(cfg, w, sub, np.float32)
        ref_p.append(np.asarray(out[1], np.float64))
    ref_p = np.concatenate(ref_p)
    truth = {}
    for q, p_, r in zip(ps.query_id, ps.product_id, ps.relevance):
        if r:
            truth.setdefault(str(int(q)), []).append(str(int(p_)))
    n_ref = ndcg.ndcg_from_arrays(ps.query_id, ps.product_id, ref_p[:, 1], truth)
    n_got = ndcg.ndcg_from_arrays(ps.query_id, ps.product_id, got_p[:, 1], truth)
    print("\n[%s, full depth, %d queries / %d pairs] nDCG@5: CPU port %.6f  HIP %.6f   max |score diff| %.2e"
          % (name, len(truth), ps.n, n_ref, n_got, np.abs(got_p[:, 1] - ref_p[:, 1]).max()))
    assert abs(n_ref - n_got) <= 1e-3
    agree = total = 0
    for q in np.unique(ps.query_id):
        m = ps.query_id == q
        r, g, pid = ref_p[m, 1], got_p[m, 1], ps.product_id[m]
        order = np.argsort(-r, kind="stable")
        if len(order) > 5 and abs(r[order[4]] - r[order[5]]) < 1e-4:
            continue                          # a near tie at the cut: either top-5 set is right
        total += 1
        agree += set(pid[order[:5]]) == set(pid[np.argsort(-g, kind="stable")[:5]])
    assert agree == total, (agree, total)
