"""Score-file protocol between the per-model predict drivers and the ensemble.

Writers are byte-compatible with the reference's so ``code/main.py`` can parse the files unchanged:
* tab files: ``"%s\\t%s\\t%s\\n" % (query_id, product_id, score)`` (code/imagebert_zk/evaluate_normal.py:242-243,
  code/imagebert_lds/src/run_pretraining_predict_score.py:585-589); the score is printed the way
  ``str(np.float32)`` prints it (shortest round-trip repr, <= 9 significant digits, scientific below 1e-4)
  -- checked against the shipped prediction_result/*.txt lines.
* lxmert csv: header ``query-id,product-id,score`` then one row per pair (code/lxmert/src/tasks/kdd_model.py:117-129).
The reader accepts both (tab split / comma split, header skipped by the ``"query" in line`` test of main.py:33-35).
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np


def _fmt(score) -> str:
    return str(np.float32(score))


def write_score_tsv(path, query_ids, product_ids, scores, append: bool = False):
    """evaluate_normal.py opens in append mode when the file exists (:112-121); default here is truncate."""
    with open(path, "a" if append else "w") as f:
        for q, p, s in zip(query_ids, product_ids, scores):
            f.write("%s\t%s\t%s\n" % (int(q), int(p), _fmt(s)))


def write_score_csv(path, query_ids, product_ids, scores):
    with open(path, "w") as f:
        f.write("query-id,product-id,score\r\n")  # csv.DictWriter default line terminator
        for q, p, s in zip(query_ids, product_ids, scores):
            f.write("%s,%s,%s\r\n" % (int(q), int(p), _fmt(s)))


def read_scores(path) -> "OrderedDict[str, OrderedDict[str, float]]":
    """{query_id: {product_id: score}} with ids kept as strings, exactly like main.py:11-39."""
    sep = "," if str(path).endswith(".csv") else "\t"
    out: OrderedDict = OrderedDict()
    with open(path) as f:
        for line in f:
            if "query" in line:
                continue
            arr = line.strip().split(sep)
            if len(arr) < 3:
                continue
            out.setdefault(arr[0], OrderedDict())[arr[1]] = float(arr[2])
    return out
