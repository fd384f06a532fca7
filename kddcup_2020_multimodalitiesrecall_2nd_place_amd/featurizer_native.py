"""ctypes binding of libmmfeat (include/mmfeat.h): the multi-threaded native TSV featurizer.

Same outputs as ``featurizer.read_line`` + ``featurizer.{zk,lds,lxmert}_batch`` (tests/test_featurizer_native.py),
written straight into pinned host buffers when torch is available so each array needs one H2D copy.
Queries with non-ASCII bytes are tokenised by the Python ``WordPieceTokenizer`` (full Unicode rules).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import featurizer as F
from .config import FEAT_DIM, LABEL_LEN, N_BOX

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libmmfeat.so")
EXPORTS = ("mmf_create", "mmf_destroy", "mmf_last_error", "mmf_set_label", "mmf_tokenize_ascii", "mmf_featurize",
           "mmf_featurize_spans", "mmf_split_lines", "mmf_b64_tier", "mmf_prefault", "mmf_release_later", "mmf_query_ids")


class BatchOut(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("product_id", "query_id", "num_boxes", "boxes", "feats", "label_ids", "label_len",
                                          "query_ids", "query_len", "needs_host_tokenizer", "query_span", "feat_rows_live")]


_lib = None


def load(path=None):
    """dlopen libmmfeat.so (``path``: another build of it -- tools/feat_bench.py's A/B against an older revision)."""
    global _lib
    if _lib is None:
        path = path or LIB_PATH
        if not os.path.exists(path):
            raise RuntimeError("native featurizer missing: %s (run __graft_entry__.build())" % path)
        l = C.CDLL(path)
        l.mmf_last_error.restype = C.c_char_p
        l.mmf_create.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
        l.mmf_destroy.argtypes = [C.c_void_p]
        l.mmf_destroy.restype = None
        l.mmf_set_label.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32]
        l.mmf_tokenize_ascii.argtypes = [C.c_void_p, C.c_char_p, C.c_int64, C.c_void_p, C.c_int32]
        l.mmf_featurize.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                    C.POINTER(BatchOut)]
        l.mmf_featurize_spans.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                          C.c_int32, C.POINTER(BatchOut)]
        l.mmf_split_lines.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
        l.mmf_split_lines.restype = C.c_int64
        if path == LIB_PATH or hasattr(l, "mmf_b64_tier"):      # (a round-5 build handed to load() for an A/B has neither)
            l.mmf_b64_tier.argtypes = [C.c_int32]
            l.mmf_prefault.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32]
            l.mmf_release_later.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
            l.mmf_query_ids.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        _lib = l
    return _lib


def _host(shape, dtype, pinned):
    if pinned:
        import torch
        t = torch.empty(shape, dtype=getattr(torch, np.dtype(dtype).name), pin_memory=True)
        return t.numpy(), t
    a = np.empty(shape, dtype)
    return a, a


class NativeFeaturizer:
    """model: 'zk' | 'lds' | 'lxmert' (box_dim, text_len, tokenizer flavour follow the sub-project)."""

    def __init__(self, vocab_path: str, label_table: dict, model: str = "zk", threads: int = 0, pinned: bool = False,
                 reuse_buffers: bool = False, pools: int = 1, want_feats: bool = True):
        """reuse_buffers: keep ``pools`` (pinned) sets of output buffers, grown on demand and used round-robin -- the
        returned arrays are then views that the ``pools``-th following call overwrites (streaming use: featurize ->
        H2D copy -> featurize ...).  Page-faulting fresh 336 KB/row buffers costs more than the decode itself, so the
        streaming drivers turn this on.  want_feats=False: the 2048-d box features are neither decoded nor allocated (the batch dicts
        carry None for them) -- for a second pass over records whose features another featurizer already produced."""
        self.want_feats = want_feats
        self.lib = load()
        self.reuse, self._caps, self._pools, self._turn = reuse_buffers, [0] * pools, [None] * pools, 0
        self.model = model
        self.text_len = 23 if model == "lxmert" else 20
        self.box_dim = 4 if model == "lxmert" else 5
        hf = model == "lxmert"
        self.py_tok = F.WordPieceTokenizer(vocab_path, max_input_chars_per_word=100 if hf else 200,
                                           never_split=F.SPECIALS if hf else ())
        self.threads, self.pinned = threads, pinned
        self.stats = {}                                   # seconds per stage, accumulated by iter_spans / _run
        self.prefault = True                              # iter_spans maps a batch's pages in parallel before splitting it (mmf_prefault)
        self._h = C.c_void_p()
        if self.lib.mmf_create(vocab_path.encode(), 100 if hf else 200, int(hf), C.byref(self._h)) != 0:
            raise RuntimeError("mmf_create: " + self.lib.mmf_last_error().decode())
        for cls, text in label_table.items():             # ~30 classes: tokenised once with the full Unicode tokenizer
            ids = np.asarray(self.py_tok.convert_tokens_to_ids(self.py_tok.tokenize(text)), np.int32)
            if self.lib.mmf_set_label(self._h, int(cls), ids.ctypes.data if ids.size else None, int(ids.size)) != 0:
                raise RuntimeError("mmf_set_label: " + self.lib.mmf_last_error().decode())

    def tokenize_ascii(self, text: str):
        b = text.encode("utf-8")
        ids = np.empty(max(4 * len(b) + 4, 8), np.int32)
        n = self.lib.mmf_tokenize_ascii(self._h, b, len(b), ids.ctypes.data, ids.size)
        return None if n == -2 else ids[:n].tolist()

    def _spec(self, n):
        T = self.text_len
        return dict(product_id=((n,), np.int64), query_id=((n,), np.int64), num_boxes=((n,), np.int32),
                    boxes=((n, N_BOX, self.box_dim), np.float32), feats=((n if self.want_feats else 0, N_BOX, FEAT_DIM), np.float32),
                    label_ids=((n, N_BOX, LABEL_LEN), np.int32), label_len=((n, N_BOX), np.int32),
                    query_ids=((n, T), np.int32), query_len=((n,), np.int32), needs_host_tokenizer=((n,), np.uint8),
                    query_span=((n, 2), np.int64))

    def _buffers(self, n):
        if not self.reuse:
            pairs = {k: _host(shape, dt, self.pinned and n > 0) for k, (shape, dt) in self._spec(n).items()}
            return {k: v[0] for k, v in pairs.items()}, {k: v[1] for k, v in pairs.items()}
        t = self._turn
        self._turn = (t + 1) % len(self._pools)
        if n > self._caps[t]:
            self._caps[t] = max(n, 2 * self._caps[t])
            self._pools[t] = {k: _host(shape, dt, self.pinned) for k, (shape, dt) in self._spec(self._caps[t]).items()}
            # mmf_batch_out.feat_rows_live: a fresh buffer holds anything -> every box row counts as dirty; from then on the library keeps
            # the count per row, and only the part of the zero padding that a previous record wrote over is written again
            live = np.full(self._caps[t], N_BOX, np.int32)
            self._pools[t]["feat_rows_live"] = (live, live)
        return {k: (v[0][:n] if (k != "feats" or self.want_feats) else v[0]) for k, v in self._pools[t].items()}, {k: v[1] for k, v in self._pools[t].items()}

    def featurize(self, lines, sen2forest: bool = False) -> dict:
        """lines: iterable of TSV records (str or bytes).  Returns the raw padded arrays (+ ``keep`` = pinned owners)."""
        enc = [(l if isinstance(l, bytes) else l.encode("utf-8")) for l in lines]
        offsets = np.zeros(len(enc) + 1, np.int64)
        np.cumsum([len(e) for e in enc], out=offsets[1:])
        return self.featurize_bytes(b"".join(enc), offsets, sen2forest)

    def featurize_bytes(self, data, offsets: np.ndarray, sen2forest: bool = False) -> dict:
        """record i = data[offsets[i]:offsets[i+1]] (``data``: bytes)."""
        offsets = np.ascontiguousarray(offsets, np.int64)
        keep = np.frombuffer(data, np.uint8)
        return self._run(keep.ctypes.data if keep.size else None, lambda a, b: bytes(data[a:b]), offsets[:-1], offsets[1:], sen2forest)

    def _run(self, base, getbytes, starts, ends, sen2forest):
        """base: address of the byte buffer; getbytes(a, b): bytes [a, b) of it (for the host-tokenizer fallback)."""
        starts, ends = np.ascontiguousarray(starts, np.int64), np.ascontiguousarray(ends, np.int64)
        n, T = len(starts), self.text_len
        arr, keep = self._buffers(n)
        live = arr.pop("feat_rows_live", None)                    # reused buffer sets only
        out = BatchOut(*([(arr[k].ctypes.data if (k != "feats" or self.want_feats) else None)
                          for k in ("product_id", "query_id", "num_boxes", "boxes", "feats", "label_ids",
                                    "label_len", "query_ids", "query_len", "needs_host_tokenizer", "query_span")]
                         + [live.ctypes.data if live is not None and n else None]))
        if n:
            import time
            t0 = time.perf_counter()
            rc = self.lib.mmf_featurize_spans(self._h, base, starts.ctypes.data, ends.ctypes.data, n, T, self.box_dim, int(sen2forest),
                                              self.threads, C.byref(out))
            self.stats["decode"] = self.stats.get("decode", 0.0) + time.perf_counter() - t0
            if rc != 0:
                raise ValueError("mmf_featurize failed (%d): %s" % (rc, self.lib.mmf_last_error().decode()))
        for i in np.nonzero(arr["needs_host_tokenizer"])[0]:      # non-ASCII queries: full Unicode tokenizer
            q = getbytes(int(arr["query_span"][i, 0]), int(arr["query_span"][i, 1])).decode("utf-8")
            if sen2forest:
                q = q.replace("sen department of", "forest style")
            ids = self.py_tok.encode_query(q)
            arr["query_len"][i] = len(ids)
            arr["query_ids"][i] = 0
            arr["query_ids"][i, :min(T, len(ids))] = ids[:T]
        arr["keep"] = keep
        if not self.want_feats:
            arr["feats"] = None
        return arr

    def query_ids(self, path: str) -> np.ndarray:
        """query_id of every record of a TSV file, in file order (int64), without decoding a record: line split + the last field.  A rank of
        an N-GPU job finds its contiguous query block of a shared file with it (``pipeline.tsv_shard``)."""
        out = []
        save, self.prefault = self.prefault, False                # heads and tails only: nothing worth mapping ahead, nothing to release
        try:
            for base, _g, starts, ends in self.iter_spans(path, 1 << 16):
                q = np.empty(len(starts), np.int64)
                rc = self.lib.mmf_query_ids(base, starts.ctypes.data, ends.ctypes.data, len(starts), q.ctypes.data)
                if rc != 0:
                    raise ValueError("mmf_query_ids failed (%d): %s" % (rc, self.lib.mmf_last_error().decode()))
                out.append(q)
        finally:
            self.prefault = save
        return np.concatenate(out) if out else np.zeros(0, np.int64)

    def byte_shard(self, path: str, rank: int, world: int):
        """[b0, b1): rank's share of a TSV file that ``world`` ranks read, cut at QUERY boundaries near the byte offsets size * r / world -- O(1) work per rank
        (a look at the ~30 records around each of its two cut points), no pass over the file, shards balanced by BYTES (= decode work).  The cut for offset p is
        the start of the first record at or after p whose query id differs from the record in front of it; ranks r and r + 1 evaluate the same function for their
        common cut, so the shards partition the file (valid.tsv / testB.tsv: records grouped by query)."""
        import mmap
        size = os.path.getsize(path)
        if size == 0 or world <= 1:
            return 0, size
        with open(path, "rb") as f, mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ) as mm:
            view = np.frombuffer(mm, np.uint8)
            try:
                base = view.ctypes.data
                starts, ends, used = np.empty(64, np.int64), np.empty(64, np.int64), C.c_int64()
                q = np.empty(64, np.int64)

                def cut(p):
                    if p <= 0:
                        return 0
                    if p >= size:
                        return size
                    pos = mm.rfind(b"\n", 0, p) + 1               # start of the line that holds byte p - 1 ... (0 when there is none)
                    prev_q = None
                    while pos < size:
                        n = self.lib.mmf_split_lines(base + pos, size - pos, starts.ctypes.data, ends.ctypes.data, 64, C.byref(used))
                        if n < 0:
                            raise ValueError(self.lib.mmf_last_error().decode())
                        if n and self.lib.mmf_query_ids(base + pos, starts.ctypes.data, ends.ctypes.data, n, q.ctypes.data) != 0:
                            raise ValueError(self.lib.mmf_last_error().decode())
                        for i in range(n):
                            if prev_q is not None and pos + starts[i] >= p and q[i] != prev_q:
                                return int(pos + starts[i])
                            prev_q = q[i]
                        pos += used.value
                    return size
                return cut(size * rank // world), cut(size * (rank + 1) // world)
            finally:
                del view

    def iter_spans(self, path: str, batch_lines: int = 8192, ramp: int = 0, records=None, byte_range=None):
        """Stream a TSV file as record spans: yields (base address, getbytes, starts, ends) per ``batch_lines`` records
        (blank lines and header lines containing 'product_id' skipped, kdd_data.py:70-71).  ``ramp`` > 0: the first batches hold
        ramp, 2 ramp, 4 ramp ... records until ``batch_lines`` is reached -- a consumer that overlaps decode, copy and scoring starts
        after the decode of ``ramp`` records instead of a whole batch (pipeline.stream_scores_tsv).  ``records = (lo, hi)``: only the records
        lo <= index < hi of the file (header and blank lines not counted) are yielded -- a rank's shard; the ones in front are split, not decoded.
        ``byte_range = (b0, b1)`` (``byte_shard``): only the records that start in [b0, b1), without touching the rest of the file.  The file is mmapped and the line
        splitting is native; the spans stay valid until the generator is advanced.  ``self.stats`` accumulates the seconds spent
        mapping / unmapping, prefaulting and splitting (tools/feat_bench.py prints them)."""
        import mmap
        import time
        st = self.stats
        clock = time.perf_counter
        with open(path, "rb") as f:
            size = os.fstat(f.fileno()).st_size
            if size == 0:
                return
            t0 = clock()
            mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
            view = np.frombuffer(mm, np.uint8)
            st["mmap"] = st.get("mmap", 0.0) + clock() - t0
            try:
                base, pos = view.ctypes.data, 0
                if byte_range is not None:
                    pos, size = int(byte_range[0]), min(size, int(byte_range[1]))
                starts, ends = np.empty(batch_lines, np.int64), np.empty(batch_lines, np.int64)
                used = C.c_int64()
                cur = min(batch_lines, ramp) if ramp > 0 else batch_lines
                seen, (rec_lo, rec_hi) = 0, (records if records is not None else (0, 1 << 62))
                skip_buf = None
                mapped, per_rec = pos, 64 << 10              # bytes of the mapping whose pages are in, estimate of a record's size
                prefault = self.prefault and hasattr(self.lib, "mmf_prefault")
                while pos < size and seen < rec_hi:
                    if seen < rec_lo:                         # records of other ranks in front of this shard: find their ends, touch nothing else
                        if skip_buf is None:
                            skip_buf = (np.empty(1 << 16, np.int64), np.empty(1 << 16, np.int64))
                        k = self.lib.mmf_split_lines(base + pos, size - pos, skip_buf[0].ctypes.data, skip_buf[1].ctypes.data,
                                                     min(1 << 16, rec_lo - seen), C.byref(used))
                        if k < 0:
                            raise ValueError(self.lib.mmf_last_error().decode())
                        seen += k
                        pos += used.value
                        mapped = max(mapped, pos)
                        continue
                    cur = min(cur, rec_hi - seen)
                    want = min(size, pos + int(cur * per_rec * 1.25) + (4 << 20))
                    t0 = clock()
                    if prefault and want > mapped:            # the next batch's pages, mapped by several threads instead of by the splitter's faults
                        self.lib.mmf_prefault(self._h, base + mapped, want - mapped, self.threads)
                        mapped = want
                    t1 = clock()
                    n = self.lib.mmf_split_lines(base + pos, size - pos, starts.ctypes.data, ends.ctypes.data, cur, C.byref(used))
                    cur = min(batch_lines, 2 * cur)
                    t2 = clock()
                    st["prefault"] = st.get("prefault", 0.0) + t1 - t0
                    st["split"] = st.get("split", 0.0) + t2 - t1
                    if n < 0:
                        raise ValueError(self.lib.mmf_last_error().decode())
                    seen += max(n, 0)
                    if n:
                        yield base + pos, (lambda a, b, p0=pos: mm[p0 + a:p0 + b]), starts[:n].copy(), ends[:n].copy()
                        per_rec = max(per_rec // 2, used.value // n)
                        if prefault:                          # the consumer is done with this batch's bytes: the NEXT decode unmaps them on the side
                            self.lib.mmf_release_later(self._h, base + pos, used.value)
                    pos += used.value
            finally:
                t0 = clock()
                if hasattr(self.lib, "mmf_release_later"):
                    self.lib.mmf_release_later(self._h, None, 0)      # pending ranges point into this mapping: forget them before it goes away
                del view                                     # release the exported buffer before the mmap closes
                mm.close()
                st["munmap"] = st.get("munmap", 0.0) + clock() - t0

    def iter_file(self, path: str, batch_lines: int = 8192, sen2forest: bool = False, layout: bool = True, ramp: int = 0, records=None, byte_range=None):
        """Stream a TSV file: yields one batch dict per ``batch_lines`` records (see ``iter_spans``)."""
        for base, getbytes, starts, ends in self.iter_spans(path, batch_lines, ramp, records, byte_range):
            a = self._run(base, getbytes, starts, ends, sen2forest)
            yield self._layout(a) if layout else a

    # ---- the three reference batch layouts (same keys / dtypes as featurizer.*_batch) ----
    def batch(self, lines, sen2forest: bool = False) -> dict:
        return self._layout(self.featurize(lines, sen2forest))

    def _layout(self, a: dict) -> dict:
        n, T = a["product_id"].shape[0], self.text_len
        if self.model == "zk":
            return {"num_boxes": a["num_boxes"], "np_boxes_5": a["boxes"], "np_images_features": a["feats"],
                    "np_idx_class_labels": a["label_ids"], "np_idx_query_": a["query_ids"], "len_query_": a["query_len"],
                    "labels": np.ones(n, np.int64), "segment_ids": np.tile(np.array([0] * T + [1] * N_BOX, np.int32), (n, 1)),
                    "query_id": a["query_id"], "product_id": a["product_id"]}
        if self.model == "lds":
            return {"input_ids": a["query_ids"].astype(np.int64), "segment_ids": np.zeros((n, T), np.int64), "boxes": a["boxes"],
                    "features": a["feats"], "labelfeat": a["label_ids"].astype(np.int64), "next_sentence_labels": np.zeros(n, np.int64),
                    "query_id": a["query_id"], "product_id": a["product_id"]}
        nb = np.minimum(a["num_boxes"], N_BOX)
        lens = np.minimum(a["query_len"], T)
        return {"input_ids": a["query_ids"].astype(np.int64), "boxes_label_input_ids": a["label_ids"].astype(np.int64),
                "input_mask": (np.arange(T)[None, :] < lens[:, None]).astype(np.int64),
                "boxes_label_input_mask": (np.arange(LABEL_LEN)[None, None, :] < np.minimum(a["label_len"], LABEL_LEN)[:, :, None]).astype(np.int64),
                "feats": a["feats"], "boxes": a["boxes"],
                "visual_attention_mask": (np.arange(N_BOX)[None, :] < nb[:, None]).astype(np.float32),
                "query_id": a["query_id"], "product_id": a["product_id"]}

    def close(self):
        if self._h:
            self.lib.mmf_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
