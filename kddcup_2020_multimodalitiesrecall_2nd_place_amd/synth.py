"""Synthetic (query, candidate-image) pair sets in the reference's padded batch layouts.

No competition data ships with the reference (README.md:18-41 only names the directories), so the
workloads of BASELINE.json are synthesised to the statistics the reference documents
(SURVEY.md section 8(d)): boxes/image mean 3.8 (report Table 1), query length 3..max, candidate sets
of 8-30 per query (prediction_result/*.txt), ResNet-like non-negative 2048-d box features, zero
padding to 10 boxes exactly as ``seq_padding_2`` does (code/imagebert_zk/load_data_v4.py:91-102),
``[CLS] .. [SEP]`` framing with ids 101/102 (load_data_v4.py:158), label text of <= 8 WordPieces
zero-padded (load_data_v4.py:157).

The three ``*_batch`` functions emit exactly the arrays (names, dtypes, shapes) the three reference
forwards are fed.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .config import CLS_ID, FEAT_DIM, LABEL_LEN, N_BOX, SEP_ID
from .weights import normal, uniform01

SEED = 20200823


@dataclass
class PairSet:
    """Model-independent description of B = sum(cands) pairs."""
    query_id: np.ndarray      # [B] int64
    product_id: np.ndarray    # [B] int64
    query_tokens: list        # per pair: list[int] body ids (no CLS/SEP), shared within a query
    num_boxes: np.ndarray     # [B] int32, 1..N_BOX (or > N_BOX for the truncation edge case)
    corners: np.ndarray       # [B, N_BOX, 4] float32, rows >= num_boxes are zero
    feats: np.ndarray         # [B, N_BOX, 2048] float32, rows >= num_boxes are zero
    class_id: np.ndarray      # [B, N_BOX] int64 (-1 on padded rows)
    class_table: np.ndarray   # [C, LABEL_LEN] int64 label-text ids, zero padded
    relevance: np.ndarray     # [B] int64 0/1 synthetic ground truth ("valid" convention)

    @property
    def n(self):
        return int(self.query_id.shape[0])

    def take(self, idx) -> "PairSet":
        """Sub-set of pairs (slice or index array): a rank's shard of a job, a test's sample."""
        ix = np.arange(self.n)[idx]
        return PairSet(self.query_id[ix], self.product_id[ix], [self.query_tokens[i] for i in ix], self.num_boxes[ix], self.corners[ix],
                       None if self.feats is None else self.feats[ix], self.class_id[ix], self.class_table, self.relevance[ix])


def make_class_table(n_classes: int = 33, vocab: int = 21128, seed: int = SEED) -> np.ndarray:
    u = uniform01("class_table/len", n_classes, seed)
    lens = 1 + np.floor(u * LABEL_LEN).astype(np.int64)
    ids = 106 + np.floor(uniform01("class_table/ids", n_classes * LABEL_LEN, seed) * (vocab - 106))
    ids = ids.astype(np.int64).reshape(n_classes, LABEL_LEN)
    ids[np.arange(LABEL_LEN)[None, :] >= lens[:, None]] = 0
    return ids


def make_pairs(n_queries: int, cands, *, seed: int = SEED, vocab: int = 21128, n_classes: int = 33,
               max_query_body: int = 18, all_boxes: bool = False, tag: str = "", with_feats: bool = True,
               query_offset: int = 0, box_mu: float = 1.1) -> PairSet:
    """``cands``: int (fixed candidates/query) or (lo, hi) for ragged candidate sets.
    ``with_feats=False`` leaves ``feats`` None (bench generates the 2.4 GB feature block on the GPU).
    ``query_offset`` shifts query ids (per-rank shards of one logical job).  ``box_mu``: location of the lognormal box count
    (1.1: the documented workload; bench.py sweeps it to show how the rate depends on the live-token fraction)."""
    t = "pairs%s/" % tag
    if isinstance(cands, int):
        per_q = np.full(n_queries, cands, dtype=np.int64)
    else:
        lo, hi = cands
        per_q = lo + np.floor(uniform01(t + "cands", n_queries, seed) * (hi - lo + 1)).astype(np.int64)
    B = int(per_q.sum())
    qidx = np.repeat(np.arange(n_queries), per_q)
    query_id = 10000 + query_offset + qidx.astype(np.int64)
    product_id = 500000 + (np.arange(B, dtype=np.int64) + 64 * query_offset) * 7

    qlen = 1 + np.floor(uniform01(t + "qlen", n_queries, seed) * max_query_body).astype(np.int64)
    qids = 106 + np.floor(uniform01(t + "qids", n_queries * max_query_body, seed) * (vocab - 106))
    qids = qids.astype(np.int64).reshape(n_queries, max_query_body)
    q_tokens = [qids[i, :qlen[i]].tolist() for i in range(n_queries)]
    query_tokens = [q_tokens[i] for i in qidx]

    if all_boxes:
        num_boxes = np.full(B, N_BOX, dtype=np.int32)
    else:
        z = normal(t + "nbox", (B,), seed).astype(np.float64)
        num_boxes = np.clip(np.rint(np.exp(box_mu + 0.6 * z)), 1, N_BOX).astype(np.int32)  # box_mu 1.1: mean ~3.5 .. 3.8
    live = np.arange(N_BOX)[None, :] < num_boxes[:, None]

    feats = None
    if with_feats:
        feats = np.maximum(normal(t + "feats", (B, N_BOX, FEAT_DIM), seed), 0.0).astype(np.float32)
        feats *= live[:, :, None]
    u = uniform01(t + "corners", B * N_BOX * 4, seed).reshape(B, N_BOX, 2, 2)
    u = np.sort(u, axis=2)  # (y0,x0) <= (y1,x1)
    corners = np.stack([u[:, :, 0, 0], u[:, :, 0, 1], u[:, :, 1, 0], u[:, :, 1, 1]], -1)
    corners = (corners * live[:, :, None]).astype(np.float32)
    class_id = np.floor(uniform01(t + "class", B * N_BOX, seed) * n_classes).astype(np.int64)
    class_id = np.where(live, class_id.reshape(B, N_BOX), -1)
    relevance = (uniform01(t + "rel", B, seed) < 0.2).astype(np.int64)
    return PairSet(query_id, product_id, query_tokens, num_boxes, corners, feats, class_id,
                   make_class_table(n_classes, vocab, seed), relevance)


def sen2forest_variant(ps: PairSet) -> PairSet:
    """The second zk member of the ensemble scores the same pairs on a REWRITTEN query (``sen department of`` -> ``forest
    style``, load_data_v4.py:153-154): three WordPieces become two for the queries that contain the phrase.  Synthetic
    stand-in: every third query (by id) with >= 3 body tokens has its first three tokens replaced by two fixed ids."""
    out = ps.take(slice(None))
    toks = []
    for q, body in zip(ps.query_id, ps.query_tokens):
        toks.append([2000, 2001] + list(body[3:]) if (int(q) % 3 == 0 and len(body) >= 3) else list(body))
    out.query_tokens = toks
    return out


def _label_ids(ps: PairSet) -> np.ndarray:
    tab = np.concatenate([ps.class_table, np.zeros((1, LABEL_LEN), np.int64)], 0)  # row -1 -> zeros
    return tab[ps.class_id]


def _query(ps: PairSet, maxlen: int):
    ids = np.zeros((ps.n, maxlen), np.int64)
    lens = np.zeros(ps.n, np.int32)
    for i, body in enumerate(ps.query_tokens):
        seq = ([CLS_ID] + list(body) + [SEP_ID])[:maxlen]  # seq_padding truncation, load_data_v4.py:83-85
        ids[i, :len(seq)] = seq
        lens[i] = len(seq)
    return ids, lens


def zk_batch(ps: PairSet, text_len: int = 20, labels: str = "testB") -> dict:
    """Feed of ``model_attention_channel_e`` (code/imagebert_zk/evaluate_normal.py:141-152,227-238)."""
    ids, lens = _query(ps, text_len)
    c = ps.corners
    area = (c[:, :, 2] - c[:, :, 0]) * (c[:, :, 3] - c[:, :, 1])  # load_data_v4.py:144-145 (normalised)
    boxes_5 = np.concatenate([c, area[:, :, None]], -1).astype(np.float32)
    return {
        "num_boxes": ps.num_boxes.astype(np.int32),
        "np_boxes_5": boxes_5,
        "np_images_features": ps.feats,
        "np_idx_class_labels": _label_ids(ps).astype(np.int32),
        "np_idx_query_": ids.astype(np.int32),
        "len_query_": lens,
        # testB feeds 1 for every row, valid feeds ground truth (load_data_v4.py:259-265)
        "labels": (np.ones(ps.n, np.int64) if labels == "testB" else ps.relevance.copy()),
        "segment_ids": np.tile(np.array([0] * text_len + [1] * N_BOX, np.int32), (ps.n, 1)),
    }


def lds_batch(ps: PairSet, text_len: int = 20) -> dict:
    """``features`` dict of ``bertmodel`` (code/imagebert_lds/src/run_pretraining_predict_score.py:526-548)."""
    ids, _ = _query(ps, text_len)
    c = ps.corners
    area = (c[:, :, 2] - c[:, :, 0]) * (c[:, :, 3] - c[:, :, 1])
    return {
        "input_ids": ids,
        "segment_ids": np.zeros((ps.n, text_len), np.int64),
        "boxes": np.concatenate([c, area[:, :, None]], -1).astype(np.float32),  # fed but unused (:303)
        "features": ps.feats,
        "labelfeat": _label_ids(ps),
        "next_sentence_labels": np.ones(ps.n, np.int64),
        "query_id": ps.query_id.copy(),
        "product_id": ps.product_id.copy(),
    }


def lxmert_batch(ps: PairSet, text_len: int = 23) -> dict:
    """Positional args of ``KDDModel.forward`` (code/lxmert/src/tasks/kdd_model.py:97-100,183-186)."""
    ids, lens = _query(ps, text_len)
    lab = _label_ids(ps)
    return {
        "input_ids": ids,
        "boxes_label_input_ids": lab,
        "input_mask": (np.arange(text_len)[None, :] < lens[:, None]).astype(np.int64),
        "boxes_label_input_mask": (lab != 0).astype(np.int64),
        "feats": ps.feats,
        "boxes": ps.corners.copy(),
        "visual_attention_mask": (np.arange(N_BOX)[None, :] < ps.num_boxes[:, None]).astype(np.float32),
    }


def batch_for(cfg, ps: PairSet, **kw) -> dict:
    return {"zk": zk_batch, "lds": lds_batch, "lxmert": lxmert_batch}[cfg.name](ps, cfg.text_len, **kw)
