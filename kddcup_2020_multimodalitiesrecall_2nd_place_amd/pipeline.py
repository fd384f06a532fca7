"""Predict drivers: TSV records -> featurizer -> HIP scorer -> score file, i.e. what ``code/main.py`` launches per
sub-model (code/imagebert_zk/evaluate_normal.py:222-252, code/imagebert_lds/src/run_pretraining_predict_score.py:566-589,
code/lxmert/src/tasks/kdd_model.py:46-129), batched instead of 1 / 5 / 256 pairs per framework call.
The written files are what ``ensemble.ensemble`` (main.py's merge) consumes.
"""
from __future__ import annotations

import os

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


def kdd_predict(scorer, tsv_lines, label_table, tokenizer, save_path=None, batch_pairs: int = 8192):
    """``KDD.predict(mod, save)`` of the lxmert sub-model (code/lxmert/src/tasks/kdd_model.py:46-129) over already-read TSV lines:
    returns ``(match_pred, match_label, rank_score_pred)`` -- the arg-max class per pair (``:112``), the fed target (1 for every
    row, kdd_data.py:74) and ``{query_id: [(product_id, softmax(logit)[-1]), ...]}`` (``:107-110``) -- and with ``save_path``
    writes the ``query-id,product-id,score`` CSV of ``:114-128`` grouped by query in first-seen order like the reference's dict walk."""
    import collections
    records = [F.read_line(l, label_table, tokenizer) for l in tsv_lines if l.strip() and "product_id" not in l]
    qid, pid, score = score_records(scorer, records, batch_pairs)
    match_pred = [int(s > 0.5) for s in score]           # argmax of a two-class softmax
    match_label = [1] * len(records)
    rank_score_pred = collections.defaultdict(list)
    for q, p, s in zip(qid, pid, score):
        rank_score_pred[int(q)].append((int(p), float(s)))
    if save_path is not None:
        rows = [(q, p, s) for q, items in rank_score_pred.items() for p, s in items]
        scorefile.write_score_csv(save_path, [r[0] for r in rows], [r[1] for r in rows], [r[2] for r in rows])
    return match_pred, match_label, rank_score_pred


_FEATURIZERS: dict = {}
_COPY_STREAMS: dict = {}


def _copy_stream(dev):
    """ONE H2D stream per device for the life of the process: torch's caching allocator keeps a pool per stream, so a fresh stream per call would reserve
    a fresh set of batch buffers per call (2.6 GB per pass of stream_scores_tsv, never reused: tools/soak_tsv.py)."""
    import torch
    key = (dev.type, dev.index)
    if key not in _COPY_STREAMS:
        _COPY_STREAMS[key] = torch.cuda.Stream(dev)
    return _COPY_STREAMS[key]


def _cached_featurizer(vocab_path, label_table, model, threads, want_feats=True, slot=0):
    """One NativeFeaturizer (helper threads + three pinned buffer sets, ~2 GB at 8192-record batches) per (vocabulary, label table, model,
    threads): creating them costs about as much as decoding 30 000 records."""
    from .featurizer_native import NativeFeaturizer
    key = (os.path.abspath(vocab_path), os.path.getmtime(vocab_path), model, threads, want_feats, slot, tuple(sorted((int(k), v) for k, v in label_table.items())))
    if key not in _FEATURIZERS:
        if len(_FEATURIZERS) >= 6:
            _FEATURIZERS.pop(next(iter(_FEATURIZERS))).close()
        _FEATURIZERS[key] = NativeFeaturizer(vocab_path, label_table, model, threads=threads, pinned=True, reuse_buffers=True, pools=3, want_feats=want_feats)
    return _FEATURIZERS[key]


def tsv_shard(nf, tsv_path, rank: int, world: int):
    """A rank's share of a TSV file that all ranks of a node read (BASELINE.json config 4): contiguous QUERY blocks
    (``sharding.query_block`` over the file's queries in file order -- valid.tsv / testB.tsv are grouped by query), found from the records'
    last field alone (``NativeFeaturizer.query_ids``: line split, no decode).  -> ((first record, end record), records per rank): every rank
    computes the same list, so the score gather needs no size exchange (``sharding.gather_scores(counts=...)``)."""
    from . import sharding
    qid = nf.query_ids(tsv_path)
    if qid.size == 0:
        return (0, 0), [0] * world
    first = np.concatenate([[0], np.flatnonzero(qid[1:] != qid[:-1]) + 1, [qid.size]])       # record index at which each query group starts (+ the end)
    n_groups = len(first) - 1
    bounds = [sharding.query_block(n_groups, world, r) for r in range(world)]
    counts = [int(first[hi] - first[lo]) for lo, hi in bounds]
    lo, hi = bounds[rank]
    return (int(first[lo]), int(first[hi])), counts


def decode_threads_for(world: int) -> int:
    """libmmfeat threads of ONE rank when ``world`` ranks share the host (one featurizer per rank, each decoding only its shard): half of an equal share
    of the physical cores, at most the single-consumer default (32), at least 4 -- on the MI355X box's 128-core host 32 / 32 / 16 / 8 threads for 1 / 2 / 4 /
    8 ranks.  More is slower there: eight ranks decode 763 / 584-690 / 400 k records/s in all with 8 / 16 / 32 threads each (profiles/rd6_feat_sweep.txt)."""
    import os as _os
    cpus = len(_os.sched_getaffinity(0)) if hasattr(_os, "sched_getaffinity") else (_os.cpu_count() or 8)
    return int(max(4, min(32, cpus // 4 // max(world, 1))))


def stream_scores_tsv(scorer, tsv_path, vocab_path, label_table, sen2forest: bool = False, batch_pairs: int = 32768, threads: int = 0,
                      ramp: int = 1024, shard=None, shard_by: str = "bytes", stats: dict = None):
    """TSV file -> (query_id, product_id, score) with the three stages overlapped:

      producer thread   libmmfeat decodes batch i+2 into one of three pinned buffer sets (ctypes releases the GIL)
      copy stream       H2D of batch i+1 (pinned -> device, waited for on the host, which then frees that buffer set)
      main stream       the scorer's kernels for batch i (asynchronous; nothing on the host waits for them until the end)

    ``ramp``: the first batches hold ramp, 2 ramp, 4 ramp ... records (0: every batch ``batch_pairs``): the GPU starts after ~4 ms of
    host work instead of the ~27 ms a 8192-record batch takes to decode and copy -- 4 % of a 150 000-record file, a fifth of testB.
    The featurizer (its helper threads and pinned buffer sets) is kept per (vocabulary, model, threads) between calls.

    ``shard = (rank, world)``: this process scores only its share of the file with ``decode_threads_for(world)`` decode threads unless ``threads``
    says otherwise, and returns a 4th value ``counts`` for ``sharding.gather_scores``.  ``shard_by="bytes"`` (default): the file is cut at query
    boundaries near size * r / world (``NativeFeaturizer.byte_shard``: O(1) per rank, shards balanced by decode work; ``counts`` is None -- the gather
    exchanges the sizes).  ``shard_by="queries"``: equal numbers of queries per rank (``tsv_shard``: every rank indexes the whole file first -- an
    index pass per rank that costs more than the decode from ~4 ranks on, profiles/rd6_feat_sweep.txt -- and gets the static ``counts`` list).
    """
    import queue
    import threading

    import torch

    from .featurizer_native import NativeFeaturizer
    if shard is not None and threads <= 0:
        threads = decode_threads_for(shard[1])
    nf = _cached_featurizer(vocab_path, label_table, scorer.cfg.name, threads)
    records = counts = byte_range = None
    if shard is not None and shard_by == "queries":
        records, counts = tsv_shard(nf, tsv_path, shard[0], shard[1])
    elif shard is not None:
        byte_range = nf.byte_shard(tsv_path, shard[0], shard[1])
    dev = scorer.device
    q = queue.Queue(maxsize=1)           # one decoded batch waiting + one being decoded + one being copied = 3 pools

    def produce():
        try:
            for b in nf.iter_file(tsv_path, batch_pairs, sen2forest, ramp=ramp, records=records, byte_range=byte_range):
                q.put(b)
            q.put(None)
        except BaseException as e:       # surfaced in the consumer
            q.put(e)

    th = threading.Thread(target=produce, daemon=True)
    th.start()
    copy_stream = _copy_stream(dev)
    qids, pids, scores = [], [], []
    import time
    clock = time.perf_counter
    t_get = t_h2d = t_enq = 0.0                  # host seconds waiting for a decoded batch / for its H2D copy / enqueueing its kernels (``stats``)
    while True:
        t0 = clock()
        b = q.get()
        t_get += clock() - t0
        if b is None:
            break
        if isinstance(b, BaseException):
            raise b
        qids.append(b["query_id"].copy())
        pids.append(b["product_id"].copy())
        with torch.cuda.stream(copy_stream):
            d = {k: (torch.from_numpy(v).to(dev, non_blocking=True) if isinstance(v, np.ndarray) and v.dtype.kind in "fiu" else v)
                 for k, v in b.items() if k not in ("query_id", "product_id", "keep")}
        t0 = clock()
        copy_stream.synchronize()        # the pinned set may be refilled from here on; the main stream is still busy with batch i
        t1 = clock()
        _, probs = score_batch(scorer, d)
        t_h2d += t1 - t0
        t_enq += clock() - t1
        for t in d.values():             # allocated on the copy stream, consumed on the main one: defer reuse of the memory
            if torch.is_tensor(t):
                t.record_stream(torch.cuda.current_stream(dev))
        scores.append(probs[:, 1])
    th.join()
    cat = lambda xs, dt: np.concatenate(xs) if xs else np.zeros(0, dt)
    t0 = clock()
    score = torch.cat(scores).float().cpu().numpy() if scores else np.zeros(0, np.float32)
    if stats is not None:
        stats.update(wait_decode_s=t_get, wait_h2d_s=t_h2d, enqueue_s=t_enq, drain_s=clock() - t0, batches=len(scores), featurizer=dict(nf.stats))
    if shard is not None:
        return cat(qids, np.int64), cat(pids, np.int64), score, counts
    return cat(qids, np.int64), cat(pids, np.int64), score


def predict_tsv_native(scorer, tsv_path, vocab_path, label_table, out_path, sen2forest: bool = False, batch_pairs: int = 32768,
                       threads: int = 0):
    """``predict_tsv`` with the native featurizer (libmmfeat), decode / H2D / scoring overlapped -- same scores, same file."""
    qid, pid, score = stream_scores_tsv(scorer, tsv_path, vocab_path, label_table, sen2forest, batch_pairs, threads)
    (scorefile.write_score_csv if scorer.cfg.name == "lxmert" else scorefile.write_score_tsv)(out_path, qid, pid, score)
    return qid, pid, score


def ensemble_feed(zk_feed: dict, s2f_feed: dict, lx_feed: dict) -> dict:
    """The fused entry point's feed (``scorers.EnsembleScorer.prepare``) from the three per-model featurizer batches of the
    SAME records: image side and label ids once (zk layout), the two zk query variants, the lxmert-flavour query."""
    return {"feats": zk_feed["np_images_features"], "boxes_5": zk_feed["np_boxes_5"], "num_boxes": zk_feed["num_boxes"],
            "label_ids": zk_feed["np_idx_class_labels"], "query_ids": zk_feed["np_idx_query_"], "len_query": zk_feed["len_query_"],
            "s2f_query_ids": s2f_feed["np_idx_query_"], "s2f_len_query": s2f_feed["len_query_"], "labels": zk_feed["labels"],
            "lx_input_ids": lx_feed["input_ids"], "lx_input_mask": lx_feed["input_mask"]}


class EnsembleScorer:
    """BASELINE.json config 5, TSV records -> merged scores: the three models score the same pair shard on the same GPU in ONE
    library call (``scorers.EnsembleScorer`` -> ``mms_score_ensemble``) and are merged pre-gather with main.py:59's weights, so
    the exchange step stays one fp32 per pair.

    ``zk`` is used twice -- once on the query as given, once on the ``sen2forest`` rewrite (evaluate_normal_sen2fs.py,
    load_data_v4.py:153-154) -- exactly the four tables main.py merges.  The product-uniqueness filter is global over
    queries and therefore runs on rank 0 after the gather (``ensemble.uniqueness_filter``).

    The fused call reads ONE label-id tensor; the TF-flavour and the lxmert-flavour tokenizer must therefore agree on the class
    names (they differ only in ``max_input_chars_per_word`` and the never-split specials, neither of which a class name hits);
    ``score_lines`` checks it on every batch.
    """

    WEIGHTS = (0.2, 0.2, 0.3, 0.3)

    def __init__(self, zk, lds, lxmert):
        from .scorers import EnsembleScorer as Fused
        self.zk, self.lds, self.lxmert = zk, lds, lxmert
        self.fused = Fused(zk, lds, lxmert, self.WEIGHTS)

    def _score(self, zk_feed, s2f_feed, lx_feed):
        merged, mem = self.fused(ensemble_feed(zk_feed, s2f_feed, lx_feed))
        return merged.double().cpu().numpy(), tuple(m for m in mem.cpu().numpy())

    def score_lines(self, tsv_lines, label_table, tok_tf, tok_hf, batch_pairs: int = 8192):
        """tok_tf: WordPieceTokenizer as the TF sub-projects build it; tok_hf: the lxmert (HF) flavour."""
        lines = [l for l in tsv_lines if l.strip() and "product_id" not in l]
        rec = [F.read_line(l, label_table, tok_tf) for l in lines]
        rec_s2f = [F.read_line(l, label_table, tok_tf, sen2forest=True) for l in lines]
        rec_hf = [F.read_line(l, label_table, tok_hf) for l in lines]
        merged, parts = [], [[], [], [], []]
        for s in range(0, len(rec), batch_pairs):
            zf = F.zk_batch(rec[s:s + batch_pairs], self.zk.cfg.text_len)
            lf = F.lxmert_batch(rec_hf[s:s + batch_pairs], self.lxmert.cfg.text_len)
            if not np.array_equal(zf["np_idx_class_labels"], lf["boxes_label_input_ids"]):
                raise ValueError("the two tokenizer flavours disagree on a class name: the fused entry point takes one label-id tensor")
            m, p4 = self._score(zf, F.zk_batch(rec_s2f[s:s + batch_pairs], self.zk.cfg.text_len), lf)
            merged.append(m)
            for k in range(4):
                parts[k].append(p4[k])
        cat = lambda xs, dt: np.concatenate(xs) if xs else np.zeros(0, dt)
        qid = np.array([r.query_id for r in rec], np.int64)
        pid = np.array([r.product_id for r in rec], np.int64)
        return qid, pid, cat(merged, np.float64), tuple(cat(x, np.float32) for x in parts)

    def score_tsv_native(self, tsv_path, vocab_path, label_table, batch_pairs: int = 16384, threads: int = 0):
        """``score_lines`` on a TSV file through libmmfeat, decode and scoring overlapped: a producer thread decodes span batch i+1
        three times (zk flavour, zk with the sen2forest rewrite, lxmert flavour; ctypes releases the GIL, each featurizer rotates
        three pinned buffer sets) while the main thread copies batch i to the device and runs the ONE fused call on it."""
        import queue
        import threading

        # (kept between calls like stream_scores_tsv's; the fused feed reads the 2048-d features from the first pass only, the other two flavours skip their decode)
        nf_zk = _cached_featurizer(vocab_path, label_table, "zk", threads)
        nf_s2f = _cached_featurizer(vocab_path, label_table, "zk", threads, want_feats=False, slot=1)
        nf_lx = _cached_featurizer(vocab_path, label_table, "lxmert", threads, want_feats=False)
        q = queue.Queue(maxsize=1)          # one decoded batch waiting + one being decoded + one being scored = 3 buffer sets

        def produce():
            try:
                for base, getbytes, starts, ends in nf_zk.iter_spans(tsv_path, batch_pairs, ramp=1024):      # (ramp: the first fused call starts after 1024 records)
                    a = nf_zk._run(base, getbytes, starts, ends, False)
                    item = (a["query_id"].copy(), a["product_id"].copy(), nf_zk._layout(a),
                            nf_s2f._layout(nf_s2f._run(base, getbytes, starts, ends, True)),
                            nf_lx._layout(nf_lx._run(base, getbytes, starts, ends, False)))
                    q.put(item)
                q.put(None)
            except BaseException as e:      # surfaced in the consumer
                q.put(e)

        th = threading.Thread(target=produce, daemon=True)
        th.start()
        import torch
        dev = self.zk.device
        copy_stream = _copy_stream(dev)
        qids, pids, merged, members = [], [], [], []
        while True:
            item = q.get()
            if item is None:
                break
            if isinstance(item, BaseException):
                raise item
            qi, pi, zf, sf, lf = item
            qids.append(qi)
            pids.append(pi)
            # as in stream_scores_tsv: H2D of batch i + 1 on a copy stream while the fused call of batch i runs; nothing waits for the kernels until the end
            with torch.cuda.stream(copy_stream):
                prep = self.fused.prepare(ensemble_feed(zf, sf, lf))
            copy_stream.synchronize()           # the pinned sets of this batch may be refilled from here on
            m, mem = self.fused.score_prepared(prep)
            for t in prep[2].values():          # allocated on the copy stream, consumed on the main one
                if torch.is_tensor(t):
                    t.record_stream(torch.cuda.current_stream(dev))
            merged.append(m)
            members.append(mem)
        th.join()
        cat = lambda xs, dt: np.concatenate(xs) if xs else np.zeros(0, dt)
        if not merged:
            return cat(qids, np.int64), cat(pids, np.int64), np.zeros(0, np.float64), tuple(np.zeros(0, np.float32) for _ in range(4))
        mem_all = torch.cat(members, 1).cpu().numpy()
        merged_all = torch.cat(merged).double().cpu().numpy()
        return cat(qids, np.int64), cat(pids, np.int64), merged_all, tuple(mem_all[k] for k in range(4))
