"""nDCG@k metric of the reference (three identical copies in the reference:
code/imagebert_lds/src/evaluation.py:4-38, code/imagebert_zk/evaluate_function.py:5-45, code/lxmert/src/utils.py:159-171).

DCG variant used there: ``r[0] + sum_{i>=1} r[i] / log2(i + 2)``; ideal vector = ones(len(ground truth));
per-query gain is 1 when the ranked product is in the query's answer set; mean over the ANSWER KEY's queries.
"""
from __future__ import annotations

import numpy as np


def dcg_at_k(r, k):
    r = np.asarray(r, dtype=float)[:k]
    if r.size:
        return float(r[0] + np.sum(r[1:] / np.log2(np.arange(3, r.size + 2))))
    return 0.0


def get_ndcg(r, ref, k):
    dcg_max = dcg_at_k(ref, k)
    if not dcg_max:
        return 0.0
    return dcg_at_k(r, k) / dcg_max


def evaluate_ndcg(rank_score_pred: dict, rank_label: dict, k: int = 5) -> float:
    """rank_score_pred: {query_id(str): [(product_id(str), score), ...]}; rank_label: {query_id: [product ids]}."""
    total = 0.0
    for q, truth in rank_label.items():
        rlist = sorted(rank_score_pred[str(q)], key=lambda x: x[1], reverse=True)  # stable, like list.sort
        gt = {str(p) for p in truth}
        pred_vec = [1.0 if str(p) in gt else 0.0 for p, _ in rlist]
        total += get_ndcg(pred_vec, [1.0] * len(gt), k)
    return total / len(rank_label)


def ndcg_from_arrays(query_id, product_id, score, rank_label: dict, k: int = 5) -> float:
    pred: dict = {}
    for q, p, s in zip(query_id, product_id, score):
        pred.setdefault(str(int(q)), []).append((str(int(p)), float(s)))
    return evaluate_ndcg(pred, rank_label, k)
