"""TSV wire format -> padded model batches: the callers' side of the hot path.

Restates, without TensorFlow / file-system side effects:
* the record parser ``read_line`` (code/imagebert_zk/load_data_v4.py:133-163, code/imagebert_lds/src/load_data_pred.py:94-121,
  code/lxmert/src/utils.py:23-59): tab-separated ``product_id, image_h, image_w, num_boxes, boxes(b64 f32[n,4]),
  features(b64 f32[n,2048]), class_labels(b64 i64[n]), query, query_id``; boxes divided by ``[h, w, h, w]``;
  zk/lds append the area term ``(b2-b0)*(b3-b1)/(w*h)`` computed on the RAW boxes; label text per box tokenised
  and padded/truncated to 8 ids; query framed as ``[CLS] .. [SEP]``;
* the label-text table cleanup (load_data_v4.py:34-38: ``, . ( )`` -> space, strip);
* ``seq_padding`` / ``seq_padding_2`` truncation to 20 (23) tokens / 10 boxes (load_data_v4.py:78-102);
* the ``sen2forest`` query rewrite (load_data_v4.py:153-154);
* the stock BERT Basic + WordPiece tokenizer the three sub-projects ship (imagebert_*/tokenization.py:161-359,
  lxmert/src/lxrt/tokenization.py:72-349; they differ in ``max_input_chars_per_word`` 200 vs 100 and in the HF
  copy's ``never_split`` list).

Host-side string/byte work, as in the reference; it only prepares the arrays the scorers consume.
"""
from __future__ import annotations

import base64
import unicodedata
from dataclasses import dataclass

import numpy as np

from .config import FEAT_DIM, LABEL_LEN, N_BOX

SPECIALS = ("[UNK]", "[SEP]", "[PAD]", "[CLS]", "[MASK]")


def load_vocab(path) -> dict:
    vocab = {}
    with open(path, encoding="utf-8") as f:
        for i, line in enumerate(f):
            tok = line.rstrip("\n").strip()
            vocab[tok] = i
    return vocab


def _is_ws(ch):
    return ch in " \t\n\r" or unicodedata.category(ch) == "Zs"


def _is_ctrl(ch):
    if ch in "\t\n\r":
        return False
    return unicodedata.category(ch).startswith("C")


def _is_punct(ch):
    cp = ord(ch)
    if 33 <= cp <= 47 or 58 <= cp <= 64 or 91 <= cp <= 96 or 123 <= cp <= 126:
        return True
    return unicodedata.category(ch).startswith("P")


def _is_cjk(cp):
    return (0x4E00 <= cp <= 0x9FFF or 0x3400 <= cp <= 0x4DBF or 0x20000 <= cp <= 0x2A6DF or 0x2A700 <= cp <= 0x2B73F
            or 0x2B740 <= cp <= 0x2B81F or 0x2B820 <= cp <= 0x2CEAF or 0xF900 <= cp <= 0xFAFF or 0x2F800 <= cp <= 0x2FA1F)


class WordPieceTokenizer:
    """BERT ``FullTokenizer`` / ``BertTokenizer`` behaviour (do_lower_case=True in all three sub-projects)."""

    def __init__(self, vocab, do_lower_case: bool = True, max_input_chars_per_word: int = 200, never_split=()):
        self.vocab = load_vocab(vocab) if isinstance(vocab, str) else dict(vocab)
        self.lower = do_lower_case
        self.max_chars = max_input_chars_per_word
        self.never_split = tuple(never_split)
        self.unk = "[UNK]"

    # ---- basic tokenizer ----
    def _basic(self, text: str):
        out = []
        for ch in text:                                   # _clean_text
            cp = ord(ch)
            if cp == 0 or cp == 0xFFFD or _is_ctrl(ch):
                continue
            out.append(" " if _is_ws(ch) else ch)
        spaced = []
        for ch in out:                                    # _tokenize_chinese_chars
            if _is_cjk(ord(ch)):
                spaced.extend((" ", ch, " "))
            else:
                spaced.append(ch)
        pieces = []
        for tok in "".join(spaced).split():
            if self.lower and tok not in self.never_split:
                tok = "".join(c for c in unicodedata.normalize("NFD", tok.lower()) if unicodedata.category(c) != "Mn")
            if tok in self.never_split:
                pieces.append(tok)
                continue
            cur = ""
            for ch in tok:                                # _run_split_on_punc
                if _is_punct(ch):
                    if cur:
                        pieces.append(cur)
                        cur = ""
                    pieces.append(ch)
                else:
                    cur += ch
            if cur:
                pieces.append(cur)
        return " ".join(pieces).split()

    # ---- greedy longest-match-first wordpiece ----
    def _wordpiece(self, word: str):
        if len(word) > self.max_chars:
            return [self.unk]
        out, start = [], 0
        while start < len(word):
            end, cur = len(word), None
            while start < end:
                sub = word[start:end]
                if start > 0:
                    sub = "##" + sub
                if sub in self.vocab:
                    cur = sub
                    break
                end -= 1
            if cur is None:
                return [self.unk]
            out.append(cur)
            start = end
        return out

    def tokenize(self, text: str):
        toks = []
        for w in self._basic(text):
            toks.extend(self._wordpiece(w))
        return toks

    def convert_tokens_to_ids(self, tokens):
        return [self.vocab[t] for t in tokens]            # KeyError for a missing special token, like the reference

    def encode_query(self, query: str):
        return self.convert_tokens_to_ids(["[CLS]"] + self.tokenize(query) + ["[SEP]"])


def clean_label_text(raw: str) -> str:
    """load_data_v4.py:36-37."""
    return raw.replace(",", " ").replace(".", " ").replace("(", " ").replace(")", " ").strip()


def load_label_table(path) -> dict:
    """``multimodal_labels.txt``: ``<class id>\\t<label text>`` per line (load_data_v4.py:34-38)."""
    table = {}
    with open(path, encoding="utf-8") as f:
        for line in f:
            arr = line.strip().split("\t")
            if len(arr) >= 2:
                table[arr[0]] = clean_label_text(arr[1])
    return table


@dataclass
class Record:
    product_id: int
    query_id: int
    image_h: int
    image_w: int
    num_boxes: int
    boxes: np.ndarray          # [n,4] normalised corners
    area: np.ndarray           # [n]   (b2-b0)*(b3-b1)/(w*h) on the raw boxes
    feats: np.ndarray          # [n,2048]
    label_ids: np.ndarray      # [n,8] int64, zero padded / truncated
    label_lens: list
    query: str
    query_ids: list            # [CLS] .. [SEP], untruncated


def read_line(line: str, label_table: dict, tokenizer: WordPieceTokenizer, sen2forest: bool = False) -> Record:
    arr = line.strip().split("\t")
    product_id, image_h, image_w, n = int(arr[0]), int(arr[1]), int(arr[2]), int(arr[3])
    raw = np.frombuffer(base64.b64decode(arr[4]), dtype=np.float32).reshape(n, 4)
    # the reference divides a float32 array by a Python list -> float64 division, stored back as float32
    boxes = (raw / [image_h, image_w, image_h, image_w]).astype(np.float32)
    area = ((raw[:, 2] - raw[:, 0]) * (raw[:, 3] - raw[:, 1]) / (image_w * image_h)).astype(np.float32)
    feats = np.frombuffer(base64.b64decode(arr[5]), dtype=np.float32).reshape(n, FEAT_DIM)
    classes = np.frombuffer(base64.b64decode(arr[6]), dtype=np.int64).reshape(n)
    ids, lens = np.zeros((n, LABEL_LEN), np.int64), []
    for i, c in enumerate(classes):
        t = tokenizer.convert_tokens_to_ids(tokenizer.tokenize(label_table[str(int(c))]))
        lens.append(len(t))
        t = t[:LABEL_LEN]                                  # seq_padding(idx_class_labels, 8, 0)
        ids[i, :len(t)] = t
    query = arr[7]
    if sen2forest:
        query = query.replace("sen department of", "forest style")
    return Record(product_id, int(arr[8]), image_h, image_w, n, boxes, area, feats, ids, lens, query,
                  tokenizer.encode_query(query))


def _pad_rows(x: np.ndarray, maxlen: int) -> np.ndarray:
    """seq_padding_2: zero-pad / truncate the leading axis."""
    out = np.zeros((maxlen,) + x.shape[1:], x.dtype)
    k = min(maxlen, x.shape[0])
    out[:k] = x[:k]
    return out


def _pad_ids(ids, maxlen: int):
    out = np.zeros(maxlen, np.int64)
    k = min(maxlen, len(ids))
    out[:k] = ids[:k]
    return out, k


def zk_batch(records, text_len: int = 20, labels=None) -> dict:
    """Arrays of code/imagebert_zk/evaluate_normal.py:227-238 (testB convention: label 1 for every row)."""
    n = len(records)
    q = np.stack([_pad_ids(r.query_ids, text_len)[0] for r in records]).astype(np.int32)
    return {
        "num_boxes": np.array([r.num_boxes for r in records], np.int32),
        "np_boxes_5": np.stack([_pad_rows(np.concatenate([r.boxes, r.area[:, None]], 1), N_BOX) for r in records]),
        "np_images_features": np.stack([_pad_rows(r.feats, N_BOX) for r in records]),
        "np_idx_class_labels": np.stack([_pad_rows(r.label_ids, N_BOX) for r in records]).astype(np.int32),
        "np_idx_query_": q,
        "len_query_": np.array([len(r.query_ids) for r in records], np.int32),      # untruncated (load_data_v4.py:267)
        "labels": np.ones(n, np.int64) if labels is None else np.asarray(labels, np.int64),
        "segment_ids": np.tile(np.array([0] * text_len + [1] * N_BOX, np.int32), (n, 1)),
    }


def lds_batch(records, text_len: int = 20) -> dict:
    """``features`` dict of code/imagebert_lds/src/load_data_pred.py:221-243."""
    n = len(records)
    return {
        "input_ids": np.stack([_pad_ids(r.query_ids, text_len)[0] for r in records]),
        "segment_ids": np.zeros((n, text_len), np.int64),
        "boxes": np.stack([_pad_rows(np.concatenate([r.boxes, r.area[:, None]], 1), N_BOX) for r in records]),
        "features": np.stack([_pad_rows(r.feats, N_BOX) for r in records]),
        "labelfeat": np.stack([_pad_rows(r.label_ids, N_BOX) for r in records]),
        "next_sentence_labels": np.zeros(n, np.int64),
        "query_id": np.array([str(r.query_id) for r in records]),
        "product_id": np.array([str(r.product_id) for r in records]),
    }


def lxmert_batch(records, text_len: int = 23) -> dict:
    """Arrays of code/lxmert/src/tasks/kdd_data.py:88-109 (masks come from the padding helpers of utils.py)."""
    ids, lens = zip(*[_pad_ids(r.query_ids, text_len) for r in records])
    lab = np.stack([_pad_rows(r.label_ids, N_BOX) for r in records])
    nb = np.array([min(r.num_boxes, N_BOX) for r in records])
    labmask = np.zeros(lab.shape, np.int64)
    for i, r in enumerate(records):
        for j, l in enumerate(r.label_lens[:N_BOX]):
            labmask[i, j, :min(l, LABEL_LEN)] = 1
    return {
        "input_ids": np.stack(ids),
        "boxes_label_input_ids": lab,
        "input_mask": (np.arange(text_len)[None, :] < np.array(lens)[:, None]).astype(np.int64),
        "boxes_label_input_mask": labmask,
        "feats": np.stack([_pad_rows(r.feats, N_BOX) for r in records]),
        "boxes": np.stack([_pad_rows(r.boxes, N_BOX) for r in records]),
        "visual_attention_mask": (np.arange(N_BOX)[None, :] < nb[:, None]).astype(np.float32),
    }


def encode_record(product_id, image_h, image_w, raw_boxes, feats, classes, query, query_id) -> str:
    """Inverse of ``read_line`` (for tests / synthetic TSVs)."""
    b64 = lambda a: base64.b64encode(np.ascontiguousarray(a).tobytes()).decode()
    return "\t".join([str(product_id), str(image_h), str(image_w), str(len(raw_boxes)), b64(np.asarray(raw_boxes, np.float32)),
                      b64(np.asarray(feats, np.float32)), b64(np.asarray(classes, np.int64)), query, str(query_id)])
