"""CPU suite: libmmfeat (include/mmfeat.h, csrc/featurizer.cpp) against the Python featurizer -- which is itself pinned to the
reference's tokenizer / read_line outputs by tests/test_featurizer.py -- on the golden records and on randomised records."""
import ctypes
import json
import os
import re

import numpy as np
import pytest

from helpers import GOLDEN

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from kddcup_2020_multimodalitiesrecall_2nd_place_amd import featurizer as F
from kddcup_2020_multimodalitiesrecall_2nd_place_amd import featurizer_native as N

D = os.path.join(GOLDEN, "featurizer")
VOCAB = os.path.join(D, "vocab_small.txt")
TABLE = F.load_label_table(os.path.join(D, "labels.txt"))
PY_BATCH = {"zk": F.zk_batch, "lds": F.lds_batch, "lxmert": F.lxmert_batch}


def _py_tok(model):
    hf = model == "lxmert"
    return F.WordPieceTokenizer(VOCAB, max_input_chars_per_word=100 if hf else 200, never_split=F.SPECIALS if hf else ())


def _same(a: dict, b: dict, keys=None):
    for k in (keys or b.keys()):
        x, y = np.asarray(a[k]), np.asarray(b[k])
        if y.dtype.kind in "US":
            assert [str(v) for v in x] == [str(v) for v in y], k
        else:
            assert x.shape == y.shape and x.dtype == y.dtype, (k, x.dtype, y.dtype, x.shape, y.shape)
            assert np.array_equal(x, y), k                      # bit-exact, floats included


def test_header_and_exports():
    src = open(os.path.join(ROOT, "include", "mmfeat.h")).read()
    declared = sorted(set(re.findall(r"\b(mmf_[a-z0-9_]+)\s*\(", src)))
    assert declared == sorted(N.EXPORTS)
    lib = ctypes.CDLL(N.LIB_PATH)
    for s in declared:
        assert hasattr(lib, s), s


@pytest.mark.parametrize("model", ["zk", "lds", "lxmert"])
@pytest.mark.parametrize("s2f", [False, True])
def test_golden_records_match_python_featurizer(model, s2f):
    lines = open(os.path.join(D, "records.tsv")).read().splitlines()
    nf = N.NativeFeaturizer(VOCAB, TABLE, model, threads=3)
    got = nf.batch(lines, sen2forest=s2f)
    recs = [F.read_line(l, TABLE, _py_tok(model), sen2forest=s2f) for l in lines]
    _same(got, PY_BATCH[model](recs))
    assert got["query_id"].tolist() == [r.query_id for r in recs] and got["product_id"].tolist() == [r.product_id for r in recs]


def test_ascii_tokenizer_matches_reference_cases():
    gold = json.load(open(os.path.join(D, "tokenizer_golden.json")))["cases"]
    nf = N.NativeFeaturizer(VOCAB, TABLE, "lxmert")
    n_ascii = 0
    for c in gold:
        ids = nf.tokenize_ascii(c["text"])
        if all(ord(ch) < 128 for ch in c["text"]):
            n_ascii += 1
            assert ids == c["ids"], (c["text"], ids, c["ids"])
        else:
            assert ids is None                                # deferred to the Unicode tokenizer
    assert n_ascii >= 10
    tf = N.NativeFeaturizer(VOCAB, TABLE, "zk")
    for text in ["a" * 150, "[CLS] x [SEP]", "Women's  DRESS,red\t(new)", "", "   ", "x\x00y\x07z", "~!@#"]:
        assert tf.tokenize_ascii(text) == tf.py_tok.convert_tokens_to_ids(tf.py_tok.tokenize(text)), text
        assert nf.tokenize_ascii(text) == nf.py_tok.convert_tokens_to_ids(nf.py_tok.tokenize(text)), text


def _random_lines(n, seed):
    rng = np.random.default_rng(seed)
    words = [w for w in open(VOCAB, encoding="utf-8").read().split() if not w.startswith("[")]
    extra = ["Women's", "T-shirt", "sen department of", "XXL", "zzzqqq", "über", "naïve", "连衣裙", "50%", "a" * 120, "(red)", "[SEP]", "x\x07y"]
    classes = [int(k) for k in TABLE]
    out = []
    for i in range(n):
        nb = int(rng.integers(0, 15))
        h, w = int(rng.integers(50, 1200)), int(rng.integers(50, 1200))
        boxes = rng.uniform(0, 1, (nb, 4)).astype(np.float32) * np.array([h, w, h, w], np.float32)
        feats = rng.standard_normal((nb, 2048)).astype(np.float32)
        q = " ".join(str(rng.choice(words if rng.random() < 0.8 else extra)) for _ in range(int(rng.integers(0, 30))))
        out.append(F.encode_record(int(rng.integers(1, 1 << 40)), h, w, boxes, feats, rng.choice(classes, nb), q, i // 7))
    return out


@pytest.mark.parametrize("model", ["zk", "lds", "lxmert"])
def test_random_records_match_python_featurizer(model):
    lines = _random_lines(160, 11)
    recs = [F.read_line(l, TABLE, _py_tok(model), sen2forest=True) for l in lines]
    want = PY_BATCH[model](recs)
    for threads, lines_in in ((1, lines), (0, [l.encode() + b"\n" for l in lines])):   # bytes with newline accepted too
        got = N.NativeFeaturizer(VOCAB, TABLE, model, threads=threads).batch(lines_in, sen2forest=True)
        _same(got, want)
    flagged = N.NativeFeaturizer(VOCAB, TABLE, model).featurize(lines)["needs_host_tokenizer"]
    assert 0 < flagged.sum() < len(lines)                     # both tokenizer routes were exercised


def test_empty_and_malformed_inputs():
    nf = N.NativeFeaturizer(VOCAB, TABLE, "zk")
    assert nf.batch([])["np_idx_query_"].shape == (0, 20)
    good = _random_lines(3, 5)
    with pytest.raises(ValueError, match="record 1: .*9 tab"):
        nf.batch([good[0], "1\t2\t3", good[2]])
    f = good[1].split("\t")
    f[3] = str(int(f[3]) + 1)                                   # num_boxes disagrees with the payloads
    with pytest.raises(ValueError, match="record 2: .*num_boxes"):
        nf.batch([good[0], good[2], "\t".join(f)])
    f = good[1].split("\t")
    f[5] = "!" + f[5][1:]
    if int(f[3]) > 0:
        with pytest.raises(ValueError, match="base64"):
            nf.batch(["\t".join(f)])
    rec = F.encode_record(1, 10, 10, [[1, 1, 2, 2]], np.zeros((1, 2048)), [987654], "x", 1)
    with pytest.raises(ValueError, match="class id 987654"):
        nf.batch([rec])
    h = ctypes.c_void_p()
    assert N.load().mmf_create(b"/nonexistent/vocab.txt", 200, 0, ctypes.byref(h)) == -2
    assert b"cannot open vocab" in N.load().mmf_last_error()


@pytest.mark.parametrize("model", ["zk", "lxmert"])
def test_file_streaming_matches_in_memory(tmp_path, model):
    lines = _random_lines(45, 3)
    body = "product_id\timage_h\timage_w\tnum_boxes\tboxes\tfeatures\tclass_labels\tquery\tquery_id\n"
    for i, l in enumerate(lines):
        body += l + ("\r\n" if i % 5 == 0 else "\n") + ("\n   \n" if i % 11 == 0 else "")
    body = body.rstrip("\n")                                     # last record without a newline
    p = tmp_path / "valid.tsv"
    p.write_bytes(body.encode("utf-8"))
    want = N.NativeFeaturizer(VOCAB, TABLE, model).batch(lines)
    nf = N.NativeFeaturizer(VOCAB, TABLE, model, threads=2, reuse_buffers=True)
    for bl in (7, 45, 1000):
        got = [{k: np.array(v) for k, v in b.items() if k != "keep"} for b in nf.iter_file(str(p), bl)]   # copies: buffers are reused
        assert [len(b["query_id"]) for b in got] == [min(bl, 45 - s) for s in range(0, 45, bl)]
        for k in want:
            if k != "keep":
                assert np.array_equal(np.concatenate([b[k] for b in got]), want[k]), k
    (tmp_path / "empty.tsv").write_bytes(b"")
    assert list(nf.iter_file(str(tmp_path / "empty.tsv"))) == []
    (tmp_path / "hdr.tsv").write_bytes(b"product_id\tx\n\n")
    assert list(nf.iter_file(str(tmp_path / "hdr.tsv"))) == []


def test_every_base64_tier_matches_the_scalar_decoder():
    """AVX2 / AVX-512 VBMI bulk decoders (whatever this CPU has) against the table decoder: same arrays bit for bit, same errors."""
    lib = N.load()
    best = lib.mmf_b64_tier(-1)
    assert best >= 0 and lib.mmf_b64_tier(best + 1) == -1 and lib.mmf_b64_tier(-1) == best
    lines = _random_lines(60, 21)
    bad = []
    for k, pos in enumerate((0, 31, 32, 63, 64, 65, 200, 4095, 4096, 10000)):       # a bad character at block seams of both vector widths
        f = _random_lines(1, 100 + k)[0].split("\t")
        while int(f[3]) < 2:
            f = _random_lines(1, 300 + k)[0].split("\t"); k += 50
        f[5] = f[5][:pos] + "\x80!*="[k % 4] + f[5][pos + 1:]
        bad.append("\t".join(f))
    try:
        ref = None
        for tier in range(best + 1):
            assert lib.mmf_b64_tier(tier) == tier
            nf = N.NativeFeaturizer(VOCAB, TABLE, "zk", threads=2)
            got = nf.featurize(lines)
            if ref is None:
                ref = {k: np.array(v) for k, v in got.items() if k != "keep"}
            else:
                for k in ref:
                    assert np.array_equal(got[k], ref[k]), (tier, k)
            for b in bad:
                with pytest.raises(ValueError, match="base64"):
                    nf.featurize([b])
    finally:
        lib.mmf_b64_tier(best)
    print("\n[base64 tiers tested: 0..%d]" % best)


def test_reused_buffers_rewrite_only_stale_padding_and_threads_may_change():
    """feat_rows_live: a buffer set that is reused keeps rows [nb, 10) zero without rewriting them -- whatever the previous records'
    box counts were, the batch equals one decoded into fresh buffers; the context's helper threads serve calls of any width."""
    a, b, c = _random_lines(50, 31), _random_lines(64, 32), _random_lines(37, 33)
    fresh = N.NativeFeaturizer(VOCAB, TABLE, "zk", threads=1)
    nf = N.NativeFeaturizer(VOCAB, TABLE, "zk", threads=3, reuse_buffers=True, pools=1)
    for rnd, lines in enumerate((a, b, c, a[::-1], b[:5], c + a)):
        nf.threads = (3, 1, 8, 2, 5, 4)[rnd]
        got = nf.featurize(lines)
        want = fresh.featurize(lines)
        for k in want:
            if k != "keep":
                assert np.array_equal(got[k], want[k]), (rnd, k)
        live = nf._pools[0]["feat_rows_live"][0]
        assert np.array_equal(live[:len(lines)], np.minimum(want["num_boxes"], 10))
    # a failed call leaves the rows it touched marked dirty, and the next good call is still right
    f = b[3].split("\t")
    while int(f[3]) < 1:
        f = b[4].split("\t")
    f[5] = "!" + f[5][1:]
    with pytest.raises(ValueError):
        nf.featurize(a[:3] + ["\t".join(f)])
    got, want = nf.featurize(c), fresh.featurize(c)
    assert all(np.array_equal(got[k], want[k]) for k in want if k != "keep")


def test_ramp_batches_double_up_to_the_batch_size(tmp_path):
    lines = _random_lines(45, 3)
    p = tmp_path / "v.tsv"
    p.write_bytes(("\n".join(lines) + "\n").encode("utf-8"))
    nf = N.NativeFeaturizer(VOCAB, TABLE, "zk", threads=2, reuse_buffers=True, pools=2)
    want = N.NativeFeaturizer(VOCAB, TABLE, "zk").batch(lines)
    got = [{k: np.array(v) for k, v in b.items() if k != "keep"} for b in nf.iter_file(str(p), 16, ramp=3)]
    assert [len(b["query_id"]) for b in got] == [3, 6, 12, 16, 8]
    for k in want:
        if k != "keep":
            assert np.array_equal(np.concatenate([b[k] for b in got]), want[k]), k


def test_rank_shards_of_a_shared_file_partition_it_by_query(tmp_path):
    """pipeline.tsv_shard + iter_file(records=...): every rank of an N-rank job decodes exactly its contiguous query block of the file, the blocks
    partition the file, and the per-rank counts every rank computes are the same list (no size exchange before the score gather)."""
    from kddcup_2020_multimodalitiesrecall_2nd_place_amd import pipeline
    lines = _random_lines(83, 41)                      # query id = i // 7: 12 queries, the last one short
    body = "product_id\timage_h\timage_w\tnum_boxes\tboxes\tfeatures\tclass_labels\tquery\tquery_id\n" + "\n\n".join(lines) + "\n"
    p = tmp_path / "testB.tsv"
    p.write_bytes(body.encode("utf-8"))
    nf = N.NativeFeaturizer(VOCAB, TABLE, "zk", threads=2, reuse_buffers=True, pools=2)
    whole = N.NativeFeaturizer(VOCAB, TABLE, "zk").batch(lines)
    assert np.array_equal(nf.query_ids(str(p)), whole["query_id"])
    for world in (1, 2, 3, 8, 16):
        got_q, got_p, all_counts = [], [], None
        for rank in range(world):
            (lo, hi), counts = pipeline.tsv_shard(nf, str(p), rank, world)
            all_counts = all_counts or counts
            assert counts == all_counts and hi - lo == counts[rank]
            batches = [{k: np.array(v) for k, v in b.items() if k != "keep"} for b in nf.iter_file(str(p), 16, ramp=4, records=(lo, hi))]
            n = sum(len(b["query_id"]) for b in batches)
            assert n == counts[rank]
            if n:
                q = np.concatenate([b["query_id"] for b in batches])
                assert np.array_equal(q, whole["query_id"][lo:hi])
                assert np.array_equal(np.concatenate([b["np_images_features"] for b in batches]), whole["np_images_features"][lo:hi])
                got_q.append(q)
                got_p.append(np.concatenate([b["product_id"] for b in batches]))
        assert sum(all_counts) == len(lines)
        assert np.array_equal(np.concatenate(got_q), whole["query_id"]) and np.array_equal(np.concatenate(got_p), whole["product_id"])
        if world <= 8:                                  # a query's candidates stay on one rank
            ends = np.cumsum(all_counts)[:-1]
            assert all(whole["query_id"][e - 1] != whole["query_id"][e] for e in ends if 0 < e < len(lines))
    assert pipeline.decode_threads_for(8) >= 4 and pipeline.decode_threads_for(1) <= 64
    # byte shards: cut at query boundaries near size * r / world, O(1) per rank; same partition property, no index pass
    for world in (1, 2, 3, 5, 8, 40):
        parts, cuts = [], []
        for rank in range(world):
            b0, b1 = nf.byte_shard(str(p), rank, world)
            cuts.append((b0, b1))
            got = [np.array(b["query_id"]) for b in nf.iter_file(str(p), 16, ramp=4, byte_range=(b0, b1))]
            parts.append(np.concatenate(got) if got else np.zeros(0, np.int64))
        assert cuts[0][0] == 0 and cuts[-1][1] == os.path.getsize(p) and all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
        assert np.array_equal(np.concatenate(parts), whole["query_id"])
        for a_, b_ in zip(parts[:-1], parts[1:]):       # no query on two ranks
            assert not (len(a_) and len(b_)) or a_[-1] != b_[0]
        nonempty = [x for x in parts if len(x)]
        assert all(set(x.tolist()).isdisjoint(y.tolist()) for i, x in enumerate(nonempty) for y in nonempty[i + 1:])


def test_second_pass_without_features_matches_the_full_pass_elsewhere():
    """want_feats=False (mmf_batch_out.feats == NULL): everything but the 2048-d features, which stay undecoded -- the fused three-model feed's second and third pass."""
    lines = _random_lines(40, 17)
    for model in ("zk", "lxmert"):
        full = N.NativeFeaturizer(VOCAB, TABLE, model, threads=2).batch(lines, sen2forest=True)
        lite = N.NativeFeaturizer(VOCAB, TABLE, model, threads=2, reuse_buffers=True, pools=2, want_feats=False)
        for _ in range(3):
            got = lite.batch(lines, sen2forest=True)
            fk = "np_images_features" if model == "zk" else "feats"
            assert got[fk] is None
            for k in full:
                if k not in (fk, "keep"):
                    assert np.array_equal(got[k], full[k]), k
    f = lines[3].split("\t")
    f[3] = str(int(f[3]) + 1)
    with pytest.raises(ValueError, match="num_boxes"):             # the payload lengths are still checked
        N.NativeFeaturizer(VOCAB, TABLE, "zk", want_feats=False).batch(["\t".join(f)])


def test_repeated_passes_with_one_featurizer_leave_no_stale_release_ranges(tmp_path):
    """Regression (round 6): the consumed-batch ranges queued by mmf_release_later pointed into the file mapping of the PREVIOUS pass when the same featurizer
    streamed a second file; the next decode dropped (MADV_DONTNEED = zeroed) whatever had been mapped there since -- heap pages: glibc aborted at exit.  iter_spans
    now forgets pending ranges before it unmaps.  Run in a child process: several passes with allocations in between, results equal, clean exit."""
    import subprocess
    import sys
    lines = _random_lines(400, 55)
    p = tmp_path / "big.tsv"
    p.write_bytes(("\n".join(lines) + "\n").encode("utf-8"))
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
from test_featurizer_native import N, VOCAB, TABLE
nf = N.NativeFeaturizer(VOCAB, TABLE, "zk", threads=4, reuse_buffers=True, pools=3)
ref, junk = None, []
for rep in range(5):
    got = np.concatenate([np.array(b["np_images_features"]).sum(axis=(1, 2)) for b in nf.iter_file(%r, 64, ramp=16)])
    junk.append(np.ones((1 << 22) + rep, np.uint8))            # fresh mappings where the old file mapping was
    assert ref is None or np.array_equal(got, ref)
    ref = got
    assert all(int(j.min()) == 1 for j in junk)                # nobody zeroed them
nf.close()
print("PASSES_OK", len(ref))
''' % (ROOT, os.path.join(ROOT, "tests"), str(p))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "PASSES_OK 400" in out.stdout, (out.returncode, out.stderr[-2000:])
    # the deterministic half: nothing is pending once a pass has ended -- completed, or abandoned half way
    nf = N.NativeFeaturizer(VOCAB, TABLE, "zk", threads=2, reuse_buffers=True, pools=2)
    assert sum(len(b["query_id"]) for b in nf.iter_file(str(p), 64)) == 400
    assert nf.lib.mmf_release_later(nf._h, None, 0) == 0
    it = nf.iter_file(str(p), 64)
    next(it); next(it)
    it.close()
    assert nf.lib.mmf_release_later(nf._h, None, 0) == 0
    spans = nf.iter_spans(str(p), 64)
    next(spans); next(spans)                                    # two batches handed out, the first one's range queued ...
    assert nf.lib.mmf_release_later(nf._h, None, 0) >= 1       # ... (this is what used to survive the unmap)
    spans.close()
