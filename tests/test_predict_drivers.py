"""CPU suite: the predict drivers' host logic (TSV lines -> featurizer -> batches -> score file / KDD.predict triple) with a STUB in
place of the HIP scorer -- record order, header skipping, batching, file protocol.  The scores themselves are tested on the GPU
(tests/test_parity_gpu.py::test_tsv_to_score_file_pipeline)."""
import os
import types

import numpy as np
import torch

from helpers import GOLDEN
from kddcup_2020_multimodalitiesrecall_2nd_place_amd import featurizer as F, pipeline, scorefile
from kddcup_2020_multimodalitiesrecall_2nd_place_amd.config import LxmertConfig, ZkConfig

D = os.path.join(GOLDEN, "featurizer")


class StubScorer:
    """prepare / score_prepared of the real scorers, scoring a pair by a hash of what the featurizer produced for it."""

    def __init__(self, cfg):
        self.cfg = cfg
        self.calls = []

    def prepare(self, *args, **kw):
        feats = args[3] if self.cfg.name == "lxmert" else args[2]          # forward(...) / model_attention_channel_e(...) order
        ids = args[0] if self.cfg.name == "lxmert" else args[4]
        self.calls.append(len(ids))
        return np.asarray(feats, np.float64).reshape(len(ids), -1).sum(1) + np.asarray(ids, np.float64).sum(1)

    def score_prepared(self, key):
        p1 = torch.as_tensor(0.5 + 0.5 * np.sin(key), dtype=torch.float32)
        probs = torch.stack([1 - p1, p1], 1)
        return torch.log(probs), probs


def _lines():
    return open(os.path.join(D, "records.tsv")).read().splitlines()


def test_predict_tsv_writes_the_reference_file_formats(tmp_path):
    tok = F.WordPieceTokenizer(os.path.join(D, "vocab_small.txt"))
    table = F.load_label_table(os.path.join(D, "labels.txt"))
    lines = _lines()
    for cfg, name in ((ZkConfig(), "scores.txt"), (LxmertConfig(), "scores.csv")):
        s = StubScorer(cfg)
        out = tmp_path / name
        qid, pid, score = pipeline.predict_tsv(s, ["product_id\timage_h\t..."] + lines + [""], table, tok, str(out), batch_pairs=4)
        assert len(qid) == len(lines) and s.calls == [4] * (len(lines) // 4) + ([len(lines) % 4] if len(lines) % 4 else [])
        recs = [F.read_line(l, table, tok) for l in lines]
        assert [r.query_id for r in recs] == list(qid) and [r.product_id for r in recs] == list(pid)
        text = open(out, newline="").read()
        if cfg.name == "lxmert":                    # csv.DictWriter: header + \r\n (kdd_model.py:114-128)
            assert text.startswith("query-id,product-id,score\r\n") and text.count("\r\n") == len(lines) + 1
        else:                                       # "%s\t%s\t%s\n" (evaluate_normal.py:242-243)
            assert text.count("\n") == len(lines) and text.count("\t") == 2 * len(lines)
        back = scorefile.read_scores(str(out))
        for q, p_, sc in zip(qid, pid, score):
            assert abs(back[str(q)][str(p_)] - sc) < 1e-6
        # one batch or many: same scores in the same order
        s2 = StubScorer(cfg)
        _, _, score2 = pipeline.predict_tsv(s2, lines, table, tok, str(tmp_path / ("b_" + name)), batch_pairs=10 ** 6)
        assert s2.calls == [len(lines)] and np.allclose(score, score2, atol=1e-7)


def test_kdd_predict_triple_and_csv(tmp_path):
    tok = F.WordPieceTokenizer(os.path.join(D, "vocab_small.txt"))
    table = F.load_label_table(os.path.join(D, "labels.txt"))
    lines = _lines()
    s = StubScorer(LxmertConfig())
    out = tmp_path / "testB_score_lxmert.csv"
    match_pred, match_label, rank = pipeline.kdd_predict(s, ["product_id\tfoo"] + lines, table, tok, str(out), batch_pairs=3)
    qid, pid, score = pipeline.score_records(StubScorer(LxmertConfig()), [F.read_line(l, table, tok) for l in lines], 3)
    assert match_label == [1] * len(lines)                                  # kdd_data.py:74
    assert match_pred == [int(x > 0.5) for x in score]                      # argmax of the two-class softmax (kdd_model.py:112)
    assert isinstance(rank, dict) and sum(len(v) for v in rank.values()) == len(lines)
    first_seen = list(dict.fromkeys(int(q) for q in qid))
    assert list(rank) == first_seen                                          # dict walk order of the reference's defaultdict
    for q, p_, sc in zip(qid, pid, score):
        assert abs(dict(rank[int(q)])[int(p_)] - sc) < 1e-7
    rows = open(out, newline="").read().split("\r\n")
    assert rows[0] == "query-id,product-id,score" and len(rows) == len(lines) + 2 and rows[-1] == ""
    assert [int(r.split(",")[0]) for r in rows[1:-1]] == [q for q in first_seen for _ in rank[q]]      # grouped by query
    # save=False: nothing written
    assert pipeline.kdd_predict(StubScorer(LxmertConfig()), lines, table, tok)[2].keys() == rank.keys()
