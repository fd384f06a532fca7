"""Golden fixtures for SURVEY.md section 8(f) rows 1 and 3, produced by the REFERENCE's own code run in the
build container (nothing is copied: the reference scripts are exec'd/imported from /root/reference).

1. ensemble: a 120-query subset of the four shipped testB score tables (prediction_result/*) is written
   to tests/golden/ensemble/ and the reference's code/main.py (merge 0.2/0.2/0.3/0.3, product-uniqueness
   filter gap 0.92, top-5 writer) is executed on that subset -> expected_submission.json.
   The full-file result is also checked against the shipped submission.csv and its digest stored.
2. nDCG@5: a 200-query subset of code/imagebert_lds/src/validscore_imagebert.txt + valid_answer.json is
   scored by the reference's code/imagebert_lds/src/evaluation.py -> expected_ndcg.json (plus the
   full-file value, 0.709813 = report's "ImageBertA 0.7098").

Usage: python tests/golden/make_ensemble_golden.py
"""
import builtins
import csv
import hashlib
import io
import json
import os
import shutil
import sys
import tempfile

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
FILES = ["testB_result_match_keyword_valid_finetune_251.txt",
         "testB_result_match_keyword_valid_finetune_251_sen_to_forest.txt",
         "testBscore_imagebert.txt", "testB_score_lxmert.csv"]


def run_reference_main(pred_dir):
    """exec code/main.py with cwd=<tmp>/code so its '../prediction_result/...' paths hit pred_dir's copy."""
    tmp = tempfile.mkdtemp(prefix="mms_ens_")
    os.makedirs(os.path.join(tmp, "code"))
    shutil.copytree(pred_dir, os.path.join(tmp, "prediction_result"))
    src = open(os.path.join(REF, "code", "main.py")).read()
    cwd = os.getcwd()
    os.chdir(os.path.join(tmp, "code"))
    real_system, real_open, real_print = os.system, io.open, builtins.print
    os.system = lambda *_a, **_k: 0                       # do not launch the py2/TF sub-model script
    io.open = lambda path, mode="r", *a, **k: real_open(path, mode.replace("b", ""), *a, newline="", **k)  # py2 'wb' csv
    builtins.print = lambda *a, **k: None
    try:
        exec(compile(src, "main.py", "exec"), {"__name__": "__main__"})
    finally:
        os.system, io.open, builtins.print = real_system, real_open, real_print
        os.chdir(cwd)
    import gc
    gc.collect()  # the reference never closes its csv file
    rows = list(csv.reader(open(os.path.join(tmp, "prediction_result", "submission.csv"))))
    shutil.rmtree(tmp, ignore_errors=True)
    return rows


def main():
    pred = os.path.join(REF, "prediction_result")
    # ---- full files: must reproduce the shipped submission.csv ----
    full_dir = tempfile.mkdtemp(prefix="mms_full_")
    for f in FILES:
        shutil.copy(os.path.join(pred, f), full_dir)
    rows = run_reference_main(full_dir)
    shipped = list(csv.reader(open(os.path.join(pred, "submission.csv"))))
    got = {r[0]: r[1:] for r in rows[1:]}
    exp = {r[0]: r[1:] for r in shipped[1:]}
    assert got == exp and len(got) == 994, "reference main.py does not reproduce the shipped submission.csv"
    digest = hashlib.sha256(json.dumps(sorted(exp.items())).encode()).hexdigest()
    shutil.rmtree(full_dir, ignore_errors=True)
    # ---- subset ----
    qids = sorted({line.split("\t")[0] for line in open(os.path.join(pred, FILES[2]))}, key=int)[::8][:120]
    keep = set(qids)
    out_dir = os.path.join(HERE, "ensemble")
    for f in FILES:
        sep = "," if f.endswith(".csv") else "\t"
        with open(os.path.join(out_dir, f), "w") as o:
            for line in open(os.path.join(pred, f)):
                if "query" in line or line.split(sep)[0] in keep:
                    o.write(line)
    sub_rows = run_reference_main(out_dir)
    json.dump({"header": sub_rows[0], "rows": {r[0]: r[1:] for r in sub_rows[1:]}, "n_queries": len(keep),
               "full_submission_sha256": digest, "full_n_rows": 994,
               "source": "reference code/main.py exec'd on the subset tables in this directory"},
              open(os.path.join(out_dir, "expected_submission.json"), "w"), indent=0)
    print("ensemble subset:", len(keep), "queries,", len(sub_rows) - 1, "rows")

    # ---- nDCG ----
    sys.path.insert(0, os.path.join(REF, "code", "imagebert_lds", "src"))
    sys.dont_write_bytecode = True
    if not hasattr(np, "asfarray"):
        np.asfarray = lambda a: np.asarray(a, dtype=float)   # evaluation.py:27 predates NumPy 2
    import evaluation
    ans_path = os.path.join(REF, "code", "imagebert_lds", "valid_answer.json")
    score_path = os.path.join(REF, "code", "imagebert_lds", "src", "validscore_imagebert.txt")

    def load(path, keepq=None):
        d = {}
        for line in open(path):
            q, p, s = line.strip().split("\t")
            if keepq is None or q in keepq:
                d.setdefault(q, []).append([p, float(s)])
        return d
    full = evaluation.evaluate(None, None, load(score_path), "ndcg", ans_path, 5)
    ans = json.load(open(ans_path))
    vq = sorted(ans.keys(), key=int)[::2][:200]
    nd_dir = os.path.join(HERE, "ndcg")
    sub_ans = {q: ans[q] for q in vq}
    json.dump(sub_ans, open(os.path.join(nd_dir, "valid_answer_subset.json"), "w"))
    with open(os.path.join(nd_dir, "validscore_subset.txt"), "w") as o:
        for line in open(score_path):
            if line.split("\t")[0] in sub_ans:
                o.write(line)
    sub = evaluation.evaluate(None, None, load(os.path.join(nd_dir, "validscore_subset.txt")), "ndcg",
                              os.path.join(nd_dir, "valid_answer_subset.json"), 5)
    sub1 = evaluation.evaluate(None, None, load(os.path.join(nd_dir, "validscore_subset.txt")), "ndcg",
                               os.path.join(nd_dir, "valid_answer_subset.json"), 1)
    json.dump({"ndcg5_subset": sub, "ndcg1_subset": sub1, "ndcg5_full_shipped_files": full, "n_queries_subset": len(vq),
               "source": "reference code/imagebert_lds/src/evaluation.py"}, open(os.path.join(nd_dir, "expected_ndcg.json"), "w"))
    print("ndcg full %.6f subset %.6f" % (full, sub))


if __name__ == "__main__":
    main()
