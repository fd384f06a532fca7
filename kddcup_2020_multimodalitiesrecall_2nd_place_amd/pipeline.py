"""Predict drivers: TSV records -> featurizer -> HIP scorer -> score file, i.e. what ``code/main.py`` launches per
sub-model (code/imagebert_zk/evaluate_normal.py:222-252, code/imagebert_lds/src/run_pretraining_predict_score.py:566-589,
code/lxmert/src/tasks/kdd_model.py:46-129), batched instead of 1 / 5 / 256 pairs per framework call.
The written files are what ``ensemble.ensemble`` (main.py's merge) consumes.
"""
from __future__ import annotations

import numpy as np

from . import featurizer as F
from . import scorefile
from .scorers import score_batch


def score_records(scorer, records, batch_pairs: int = 8192):
    """Returns (query_id[B], product_id[B], score[B]) with score = softmax(logit)[:, 1]
    (evaluate_normal.py:242-243 / run_pretraining_predict_score.py:573-575 / kdd_model.py:102-112)."""
    name = scorer.cfg.name
    make = {"zk": F.zk_batch, "lds": F.lds_batch, "lxmert": F.lxmert_batch}[name]
    scores = []
    for s in range(0, len(records), batch_pairs):
        chunk = records[s:s + batch_pairs]
        _, probs = score_batch(scorer, make(chunk, scorer.cfg.text_len))
        scores.append(probs[:, 1].float().cpu().numpy())
    qid = np.array([r.query_id for r in records], np.int64)
    pid = np.array([r.product_id for r in records], np.int64)
    return qid, pid, (np.concatenate(scores) if scores else np.zeros(0, np.float32))


def predict_tsv(scorer, tsv_lines, label_table, tokenizer, out_path, sen2forest: bool = False, batch_pairs: int = 8192):
    """Featurise TSV lines (header lines containing 'product_id' are skipped like kdd_data.py:70-71), score, write."""
    records = [F.read_line(l, label_table, tokenizer, sen2forest) for l in tsv_lines if l.strip() and "product_id" not in l]
    qid, pid, score = score_records(scorer, records, batch_pairs)
    if scorer.cfg.name == "lxmert":
        scorefile.write_score_csv(out_path, qid, pid, score)
    else:
        scorefile.write_score_tsv(out_path, qid, pid, score)
    return qid, pid, score
