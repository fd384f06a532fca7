"""Kernel-level parity of the fused QKV projection + attention launch (csrc/qkv_attn.hip) through mms_dbg_qkv_attn: ONE launch on synthetic ragged
token streams against an fp64 numpy restatement of  [Q | K | V] = x W^T + b,  softmax(Q K^T / 8 + key mask) V  per head and pair
(pixelbert.py:767-850, lxrt/modeling.py:300-352) -- self-attention of one stream and the CROSS mode of lxmert's X layers (modeling.py:460-464: queries of either
stream attend the keys of the OTHER stream of their pair), for the pair-length extremes the sub-tile planner and the block-diagonal attention have to get right."""
import ctypes as C

import numpy as np
import pytest
import torch

from kddcup_2020_multimodalitiesrecall_2nd_place_amd import lib, weights

pytestmark = pytest.mark.gpu
H, HEADS = 768, 12


def _ref(x, w, b, segs_q, segs_k, kadd):
    """segs_q / segs_k: per pair (first row, rows) of its queries / keys in x; kadd: additive key mask by row."""
    qkv = x.astype(np.float64) @ w.astype(np.float64).T + b
    q, k, v = qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]
    out = np.zeros((x.shape[0], H))
    for (q0, nq), (k0, nk) in zip(segs_q, segs_k):
        for h in range(HEADS):
            c = slice(64 * h, 64 * h + 64)
            s = q[q0:q0 + nq, c] @ k[k0:k0 + nk, c].T / 8.0 + kadd[k0:k0 + nk][None, :]
            s -= s.max(1, keepdims=True)
            p = np.exp(s)
            out[q0:q0 + nq, c] = (p / p.sum(1, keepdims=True)) @ v[k0:k0 + nk, c]
    return out


def _stream(rs, n, lo, hi, mode):
    """token counts of n pairs: uniform in [lo, hi], or the extremes"""
    if mode == "uniform":
        return rs.randint(lo, hi + 1, size=n)
    if mode == "ones":
        return np.ones(n, np.int64)
    if mode == "extremes":
        return rs.choice([lo, lo, hi], size=n)
    raise ValueError(mode)


def _run(cnt1, cnt2, S1, S2, mode, dense=False, seed=0):
    rs = np.random.RandomState(seed)
    n = len(cnt1)
    cross = cnt2 is not None
    if dense:
        cnt1 = np.full(n, S1)
        cnt2 = np.full(n, S2) if cross else None
    off1 = np.concatenate([[0], np.cumsum(cnt1)[:-1]])
    rows1 = int(cnt1.sum())
    off2 = np.concatenate([[0], np.cumsum(cnt2)[:-1]]) if cross else None
    rows2 = int(cnt2.sum()) if cross else 0
    rows = rows1 + rows2
    x = rs.randn(rows, H).astype(np.float32)
    w = weights.round_to_bf16((rs.randn(3 * H, H) / np.sqrt(H)).astype(np.float32))
    b = (0.1 * rs.randn(3 * H)).astype(np.float32)
    kadd = np.where(rs.rand(rows) < 0.25, -10000.0, 0.0).astype(np.float32)
    for o, c in zip(off1, cnt1):                                   # every pair keeps one unmasked key per stream
        kadd[o] = 0.0
    if cross:
        for o, c in zip(off2, cnt2):
            kadd[rows1 + o] = 0.0
    l = lib.load()
    dev = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a).astype(dt)).cuda()
    dx, dw, db, dk = dev(x, np.float32), dev(w, np.float32), dev(b, np.float32), dev(kadd, np.float32)
    d1 = (dev(off1, np.int32), dev(cnt1, np.int32)) if not dense else (None, None)
    d2 = (dev(off2, np.int32), dev(cnt2, np.int32)) if cross and not dense else (None, None)
    out = torch.empty((rows, H), device="cuda", dtype=torch.float32)
    nsub = C.c_int32(0)
    p = lambda t: t.data_ptr() if t is not None else None
    rc = l.mms_dbg_qkv_attn(dx.data_ptr(), rows1, rows2, p(d1[0]), p(d1[1]), p(d2[0]), p(d2[1]), n, S1, S2 if cross else 0, dw.data_ptr(), db.data_ptr(),
                            dk.data_ptr(), dk.data_ptr() + 4 * rows1 if cross else None, mode, out.data_ptr(), C.byref(nsub), None)
    assert rc == 0, l.mms_global_error()
    s1 = list(zip(off1, cnt1))
    if cross:
        s2 = [(rows1 + o, c) for o, c in zip(off2, cnt2)]
        ref = _ref(x, w, b, s1 + s2, s2 + s1, kadd)
    else:
        ref = _ref(x, w, b, s1, s1, kadd)
    got = out.cpu().numpy().astype(np.float64)
    err = np.abs(got - ref).max() / np.abs(ref).max()
    return err, nsub.value, rows


@pytest.mark.parametrize("case", [
    ("zk-like", 600, 2, 30, "uniform"), ("box stream", 2500, 1, 10, "uniform"), ("one token per pair", 3000, 1, 1, "ones"),
    ("2 or 48", 400, 2, 48, "extremes"), ("a pair fills a sub-tile", 60, 100, 128, "uniform"), ("one pair", 1, 7, 7, "uniform")])
def test_split_bf16_self_attention_of_one_launch(case):
    name, n, lo, hi, dist = case
    rs = np.random.RandomState(len(name))
    cnt = _stream(rs, n, lo, hi, dist)
    err, nsub, rows = _run(cnt, None, hi, 0, 2, seed=n)
    print("%s: %d pairs, %d rows, %d sub-tiles, max error %.2e" % (name, n, rows, nsub, err))
    assert err < 3e-5, (name, err)
    assert nsub >= (rows + 127) // 128                                    # the greedy plan: no sub-tile beyond 128 rows / 96 pairs
    want, r, c = 1, 0, 0                                                  # ... and exactly the greedy count: consecutive pairs, a sub-tile closes when the next pair would overflow it
    for t in cnt:
        if r + t > 128 or c == 96:
            want, r, c = want + 1, 0, 0
        r, c = r + t, c + 1
    assert nsub == want, (nsub, want)


@pytest.mark.parametrize("case", [("lxmert-like", 1500, (2, 23), (1, 10)), ("short both", 2000, (1, 3), (1, 2)), ("long both", 300, (20, 60), (30, 60)), ("one pair", 1, (5, 5), (3, 3))])
def test_split_bf16_cross_attention_of_one_launch(case):
    name, n, r1, r2 = case
    rs = np.random.RandomState(7 + n)
    c1, c2 = rs.randint(r1[0], r1[1] + 1, size=n), rs.randint(r2[0], r2[1] + 1, size=n)
    err, nsub, rows = _run(c1, c2, r1[1], r2[1], 2, seed=n)
    print("%s: %d pairs, %d rows, %d sub-tiles, max error %.2e" % (name, n, rows, nsub, err))
    assert err < 3e-5, (name, err)


def test_dense_streams_and_the_exact_route():
    """dense layout (off == NULL: S tokens per pair) on the split-bf16 route, self and cross; the exact-fp32 attention route (mode 1) on ragged pairs of 16 .. 48 tokens"""
    n = 700
    e1, _, _ = _run(np.zeros(n, np.int64), None, 30, 0, 2, dense=True, seed=1)
    e2, _, _ = _run(np.zeros(n, np.int64), np.zeros(n, np.int64), 23, 10, 2, dense=True, seed=2)
    rs = np.random.RandomState(3)
    e3, _, _ = _run(rs.randint(16, 49, size=500), None, 48, 0, 1, seed=3)
    e4, _, _ = _run(rs.randint(1, 31, size=900), None, 30, 0, 1, seed=4)
    print("dense self %.2e, dense cross %.2e, exact route %.2e / %.2e" % (e1, e2, e3, e4))
    assert max(e1, e2) < 3e-5 and max(e3, e4) < 2e-5


def test_shapes_a_route_does_not_take_are_refused():
    l = lib.load()
    z = torch.zeros((256, H), device="cuda")
    w, b = torch.zeros((3 * H, H), device="cuda"), torch.zeros(3 * H, device="cuda")
    out = torch.empty((256, H), device="cuda")
    # MMS_ERR_ARG (1): a pair longer than a sub-tile; the exact route has no cross mode
    assert l.mms_dbg_qkv_attn(z.data_ptr(), 256, 0, None, None, None, None, 1, 256, 0, w.data_ptr(), b.data_ptr(), None, None, 2, out.data_ptr(), None, None) == 1
    assert l.mms_dbg_qkv_attn(z.data_ptr(), 128, 128, None, None, None, None, 8, 16, 16, w.data_ptr(), b.data_ptr(), None, None, 1, out.data_ptr(), None, None) == 1
