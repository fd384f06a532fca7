"""Golden vectors for the featurizer row, produced by the REFERENCE's own (importable) lxmert code run in the
build container: ``lxrt.tokenization.BertTokenizer`` on a small synthetic vocabulary written by this script
(the reference's vocab.txt is not used or shipped) and ``utils.read_line`` / ``utils.seq_padding*`` on
synthetic TSV records.  Outputs: tests/golden/featurizer/{vocab_small.txt, tokenizer_golden.json,
records.tsv, labels.txt, read_line_golden.npz}.

Usage: python tests/golden/make_featurizer_golden.py
"""
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "featurizer")
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
REF = "/root/reference/code"


def write_vocab(path):
    toks = ["[PAD]"] + ["[unused%d]" % i for i in range(1, 100)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]", "<S>", "<T>"]
    toks += list("abcdefghijklmnopqrstuvwxyz0123456789") + ["##" + c for c in "abcdefghijklmnopqrstuvwxyz0123456789"]
    toks += list("!\"#$%&'()*+,-./:;<=>?@[\\]^_`{|}~") + ["中", "国", "服", "装", "·"]
    words = ("women dress men shirt long sleeve cotton summer new style forest sen department of bag hand shoes leather "
             "red blue black white baby kids pants skirt coat jacket hat cap watch phone case cover table lamp chandelier "
             "flower brooch swimsuit school student high waisted letters hooded top bottom face the and for with").split()
    toks += words + ["##s", "##es", "##ing", "##ed", "##er", "##ly", "##ress", "##irt", "##eve", "##on", "##suit", "##light",
                     "un", "##able", "swim", "sun", "hand", "##bag", "cheong", "##sam", "t", "##shirt"]
    seen, out = set(), []
    for t in toks:
        if t not in seen:
            seen.add(t)
            out.append(t)
    with open(path, "w", encoding="utf-8") as f:
        f.write("\n".join(out) + "\n")
    return out


TEXTS = [
    "women dress", "Sen Department Of long sleeve shirts", "forest style dresses, cotton (summer)!", "unable swimsuit",
    "handbag hand bag cheongsam", "new-style men's T-shirt 2020", "  multiple   spaces\tand\ttabs\n", "café naïve résumé",
    "中国服装 women", "xyzzyq unknownword", "a" * 101, "b" * 201, "dress.", "", "...", "baby high waisted pants",
    "letters hooded", "drop resistance cute cup", "[CLS] [SEP] literal", "UPPER lower MiXeD", "semi;colon:co-lon", "tab\x00le�lamp",
    "sunlight sun light", "shoes,leather;bag", "· middle dot", "1234567890 watch9", "the and for with", "kids' coats & jackets",
]


def main():
    os.makedirs(OUT, exist_ok=True)
    vocab_path = os.path.join(OUT, "vocab_small.txt")
    write_vocab(vocab_path)
    # --- reference tokenizer (HF copy under lxmert, importable without TensorFlow) ---
    tmp = tempfile.mkdtemp(prefix="mms_feat_")
    os.makedirs(os.path.join(tmp, "run"))
    os.chdir(os.path.join(tmp, "run"))
    sys.dont_write_bytecode = True
    sys.path.insert(0, os.path.join(REF, "lxmert", "src"))
    sys.argv = ["kdd.py"]
    from lxrt.tokenization import BertTokenizer
    tok = BertTokenizer(vocab_path, do_lower_case=True)
    gold = []
    for t in TEXTS:
        pieces = tok.tokenize(t)
        gold.append({"text": t, "tokens": pieces, "ids": tok.convert_tokens_to_ids(pieces)})
    json.dump({"source": "reference lxrt.tokenization.BertTokenizer(vocab_small.txt, do_lower_case=True)", "cases": gold},
              open(os.path.join(OUT, "tokenizer_golden.json"), "w"), ensure_ascii=True, indent=0)
    # --- reference read_line + padding helpers on synthetic records ---
    from kddcup_2020_multimodalitiesrecall_2nd_place_amd import featurizer
    from kddcup_2020_multimodalitiesrecall_2nd_place_amd.weights import normal, uniform01
    labels = {"0": "women dress", "1": "long sleeve shirt", "2": "hand bag (leather)", "3": "table lamp, chandelier.",
              "4": "baby high waisted pants skirt coat jacket hat cap", "5": "swimsuit"}
    with open(os.path.join(OUT, "labels.txt"), "w") as f:
        for k, v in labels.items():
            f.write("%s\t%s\n" % (k, v))
    lines = []
    queries = ["women dress", "sen department of long sleeve shirts", "new style men's cotton shirt summer", "swimsuit",
               "baby high waisted pants skirt coat jacket hat cap watch phone case cover table lamp chandelier flower brooch and more for kids"]
    for i, q in enumerate(queries):
        n = [3, 1, 12, 10, 2][i]
        h, w = 600 + 10 * i, 800 - 7 * i
        u = uniform01("feat/raw%d" % i, n * 4, 7).reshape(n, 2, 2)
        u.sort(axis=1)
        raw = np.stack([u[:, 0, 0] * h, u[:, 0, 1] * w, u[:, 1, 0] * h, u[:, 1, 1] * w], 1).astype(np.float32)
        feats = np.maximum(normal("feat/f%d" % i, (n, 2048), 7), 0)
        classes = (np.arange(n) * 5 + i) % 6
        lines.append(featurizer.encode_record(1000 + i, h, w, raw, feats, classes, q, 50 + i))
    with open(os.path.join(OUT, "records.tsv"), "w") as f:
        f.write("\n".join(lines) + "\n")
    import utils as ref_utils   # code/lxmert/src/utils.py (parses argv via param.py at import)
    clean = {k: featurizer.clean_label_text(v) for k, v in labels.items()}   # kdd_data.py applies the same cleanup
    save = {}
    for i, line in enumerate(lines):
        np.random.seed(0)
        import random
        random.seed(0)
        (product_id, boxes, feats, idx_lab, idx_lab_mask, idx_query, query_id, query, str_labels, _mq, _miq, _ml) = \
            ref_utils.read_line(line, clean, tok)
        save["boxes_%d" % i] = np.asarray(boxes, np.float64)
        save["feats_%d" % i] = np.asarray(feats, np.float32)
        save["labids_%d" % i] = np.asarray(idx_lab, np.int64)
        save["labmask_%d" % i] = np.asarray(idx_lab_mask, np.int64)
        save["query_%d" % i] = np.asarray(idx_query, np.int64)
        save["ids_%d" % i] = np.asarray([product_id, query_id], np.int64)
    # batch padding through the reference helpers
    recs = [ref_utils.read_line(l, clean, tok) for l in lines]
    pq, pqm = ref_utils.seq_padding([r[5] for r in recs], maxlen=23, padding_value=0)
    pf, pfm = ref_utils.seq_padding_2([r[2] for r in recs], maxlen=10, padding_value=0)
    pb, _ = ref_utils.seq_padding_2([r[1] for r in recs], maxlen=10, padding_value=0)
    pl, _ = ref_utils.seq_padding_2([r[3] for r in recs], maxlen=10, padding_value=0)
    plm, _ = ref_utils.seq_padding_2([r[4] for r in recs], maxlen=10, padding_value=0)
    save.update(batch_query=np.asarray(pq, np.int64), batch_query_mask=np.asarray(pqm, np.int64), batch_feats_mask=np.asarray(pfm, np.float32),
                batch_boxes=np.asarray(pb, np.float64), batch_labids=np.asarray(pl, np.int64), batch_labmask=np.asarray(plm, np.int64),
                batch_feats_sum=np.asarray(pf, np.float64).sum(-1))
    np.savez_compressed(os.path.join(OUT, "read_line_golden.npz"), **save)
    print("featurizer goldens written:", len(gold), "tokenizer cases,", len(lines), "records")


if __name__ == "__main__":
    main()
