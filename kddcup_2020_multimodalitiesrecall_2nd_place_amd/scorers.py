"""Host-side mirrors of the reference's three model-forward call surfaces.

Each scorer keeps the argument names, order, dtypes and shapes of the reference entry point it
replaces, so the reference's predict drivers can call it unchanged:

* ``ZkScorer.__call__``      <- ``model_triple.model_attention_channel_e`` (code/imagebert_zk/model_triple.py:162-214;
                               fed by evaluate_normal.py:227-238) -> ``(loss, probs[B,2], [loss])``
* ``LdsScorer.__call__``     <- ``bertmodel(..., features, ...)`` (code/imagebert_lds/src/run_pretraining_predict_score.py:288-336)
                               -> ``next_sentence_prob[B,2]``
* ``LxmertScorer.forward``   <- ``KDDModel.forward`` (code/lxmert/src/tasks/kdd_model.py:183-214) -> ``(x_norm, None, logit[B,2])``

All arithmetic of the forward runs in libmmscore (HIP, gfx950).  torch is used here only to own
device buffers and the stream, and for index bookkeeping (de-duplicating label-text tuples).
Inputs may be numpy arrays or torch tensors (host or device); outputs follow the input kind.
"""
from __future__ import annotations

import numpy as np
import torch

from . import lib as _lib
from .config import N_BOX, LABEL_LEN


def _is_np(x):
    return isinstance(x, np.ndarray) or not torch.is_tensor(x)


class _Base:
    def __init__(self, cfg, weights: dict, precision="auto", device: int = 0, chunk_pairs: int = 0,
                 stop_after: int = -1, dedup_labels: bool = True, pack_tokens: bool = True):
        """precision: 1 / 2 / 3 or "auto" = ``weights.auto_precision``: 2 for bf16-representable matrices,
        3 for a real fp32 checkpoint."""
        if not torch.cuda.is_available():
            raise _lib.MmsError("no HIP device visible: the scorers have no CPU path")
        if precision == "auto":
            from .weights import auto_precision
            precision = auto_precision(weights)
        self.cfg = cfg
        self.device = torch.device("cuda", device)
        self.dedup_labels = dedup_labels
        self.precision = precision
        self.handle = _lib.Handle(cfg, precision=precision, device=device, chunk_pairs=chunk_pairs, stop_after=stop_after,
                                  pack_tokens=pack_tokens)
        self.handle.load_weights(weights)
        self.logits = None

    # -- helpers ---------------------------------------------------------------------------------
    def _dev(self, x, dtype):
        t = torch.as_tensor(np.ascontiguousarray(x)) if _is_np(x) else x
        return t.to(device=self.device, dtype=dtype, non_blocking=True).contiguous()

    def _labels(self, label_ids, dtype):
        """[B,10,8] ids -> (uniq [U,8], index [B*10] int32).  Index bookkeeping only."""
        lab = self._dev(label_ids, dtype).reshape(-1, LABEL_LEN)
        if self.dedup_labels:
            uniq, inv = torch.unique(lab, dim=0, return_inverse=True)
            return uniq.contiguous(), inv.to(torch.int32).contiguous()
        return lab, torch.arange(lab.shape[0], device=self.device, dtype=torch.int32)

    def _run(self, struct, n, keep):
        logits = torch.empty((n, 2), device=self.device, dtype=torch.float32)
        probs = torch.empty((n, 2), device=self.device, dtype=torch.float32)
        st = torch.cuda.current_stream(self.device).cuda_stream
        if n > 0:
            self.handle.score(struct, logits.data_ptr(), probs.data_ptr(), st)
        self._keep = keep  # keep inputs alive until the stream has consumed them
        self.logits = logits
        return logits, probs

    def read_hidden(self, rows):
        """Debug: current hidden state (fp32 [rows,768]) of the last chunk."""
        out = torch.empty((rows, 768), device=self.device, dtype=torch.float32)
        self.handle.debug_read_x(out.data_ptr(), rows, torch.cuda.current_stream(self.device).cuda_stream)
        return out

    def close(self):
        self.handle.close()


class ZkScorer(_Base):
    def prepare(self, num_boxes, np_boxes_5, np_images_features, np_idx_class_labels, np_idx_query_, len_query_,
                labels, segment_ids=None):
        T = self.cfg.text_len
        q = self._dev(np_idx_query_, torch.int32)
        B = q.shape[0]
        if segment_ids is None:  # load_data_v4.py:204
            segment_ids = torch.tensor([0] * T + [1] * N_BOX, dtype=torch.int32).repeat(B, 1)
        uniq, idx = self._labels(np_idx_class_labels, torch.int32)
        t = dict(num_boxes=self._dev(num_boxes, torch.int32), boxes=self._dev(np_boxes_5, torch.float32),
                 feats=self._dev(np_images_features, torch.float32), uniq=uniq, idx=idx, q=q,
                 lq=self._dev(len_query_, torch.int32), labels=self._dev(labels, torch.int64),
                 seg=self._dev(segment_ids, torch.int32))
        assert t["feats"].shape == (B, N_BOX, 2048) and t["boxes"].shape == (B, N_BOX, 5) and q.shape[1] == T
        s = _lib.ZkBatch(B, t["num_boxes"].data_ptr(), t["boxes"].data_ptr(), t["feats"].data_ptr(), uniq.data_ptr(),
                         uniq.shape[0], idx.data_ptr(), q.data_ptr(), t["lq"].data_ptr(), t["labels"].data_ptr(),
                         t["seg"].data_ptr())
        return s, B, t

    def score_prepared(self, prepared):
        s, B, keep = prepared
        return self._run(s, B, keep)

    def __call__(self, num_boxes, np_boxes_5, np_images_features, np_idx_class_labels, np_len_class_labels,
                 np_idx_query_, len_query_, labels, segment_ids=None, label_query=None, weight_label_query=None,
                 is_training=False, reuse=None):
        as_np = _is_np(np_images_features)
        prep = self.prepare(num_boxes, np_boxes_5, np_images_features, np_idx_class_labels, np_idx_query_,
                            len_query_, labels, segment_ids)
        logits, probs = self.score_prepared(prep)
        # mean softmax cross-entropy of the margin logits (model_triple.py:83,103) -- reported, unused at predict
        lab = prep[2]["labels"]
        loss = (torch.logsumexp(logits, 1) - logits.gather(1, lab[:, None]).squeeze(1)).mean()
        if as_np:
            return float(loss), probs.cpu().numpy(), [float(loss)]
        return loss, probs, [loss]


class LdsScorer(_Base):
    def prepare(self, features: dict):
        ids = self._dev(features["input_ids"], torch.int64)
        B = ids.shape[0]
        seg = features.get("segment_ids")
        seg = torch.zeros_like(ids) if seg is None else self._dev(seg, torch.int64)
        t = dict(ids=ids, seg=seg, feats=self._dev(features["features"], torch.float32),
                 lab=self._dev(features["labelfeat"], torch.int64))
        assert t["feats"].shape == (B, N_BOX, 2048) and t["lab"].shape == (B, N_BOX, LABEL_LEN)
        s = _lib.LdsBatch(B, ids.data_ptr(), seg.data_ptr(), t["feats"].data_ptr(), t["lab"].data_ptr())
        return s, B, t

    def score_prepared(self, prepared):
        s, B, keep = prepared
        return self._run(s, B, keep)

    def __call__(self, features: dict, **_ignored):
        as_np = _is_np(features["features"])
        _, probs = self.score_prepared(self.prepare(features))
        return probs.cpu().numpy() if as_np else probs


class LxmertScorer(_Base):
    def prepare(self, input_ids, boxes_label_input_ids, input_mask, feats, boxes, visual_attention_mask):
        ids = self._dev(input_ids, torch.int64)
        B = ids.shape[0]
        if input_mask is None:  # modeling.py:878-879
            input_mask = torch.ones_like(ids)
        if visual_attention_mask is None:
            visual_attention_mask = torch.ones((B, N_BOX), dtype=torch.float32)
        uniq, idx = self._labels(boxes_label_input_ids, torch.int64)
        t = dict(ids=ids, mask=self._dev(input_mask, torch.int64), uniq=uniq, idx=idx,
                 feats=self._dev(feats, torch.float32), boxes=self._dev(boxes, torch.float32),
                 vm=self._dev(visual_attention_mask, torch.float32))
        assert t["feats"].shape == (B, N_BOX, 2048) and t["boxes"].shape == (B, N_BOX, 4)
        s = _lib.LxmertBatch(B, ids.data_ptr(), t["mask"].data_ptr(), uniq.data_ptr(), uniq.shape[0], idx.data_ptr(),
                             t["feats"].data_ptr(), t["boxes"].data_ptr(), t["vm"].data_ptr())
        return s, B, t

    def score_prepared(self, prepared):
        s, B, keep = prepared
        return self._run(s, B, keep)

    def forward(self, input_ids, boxes_label_input_ids, segment_ids, input_mask, boxes_label_segment_ids,
                boxes_label_input_mask, feats, boxes, visual_attention_mask):
        """Same positional signature as KDDModel.forward.  segment ids are all-zero in the reference
        feed (kdd_model.py:97-100 passes None) and the label mask is computed but unused there
        (modeling.py:579,900-902); the discarded MLM head (kdd_model.py:201-202) is not computed."""
        as_np = _is_np(feats)
        logits, _ = self.score_prepared(self.prepare(input_ids, boxes_label_input_ids, input_mask, feats, boxes,
                                                     visual_attention_mask))
        return (None, None, logits.cpu().numpy() if as_np else logits)

    __call__ = forward


def make_scorer(cfg, weights, **kw):
    return {"zk": ZkScorer, "lds": LdsScorer, "lxmert": LxmertScorer}[cfg.name](cfg, weights, **kw)


def score_batch(scorer, batch: dict):
    """Run a synth.*_batch dict through the matching scorer; returns (logits, probs) device tensors."""
    n = scorer.cfg.name
    if n == "zk":
        p = scorer.prepare(batch["num_boxes"], batch["np_boxes_5"], batch["np_images_features"], batch["np_idx_class_labels"],
                           batch["np_idx_query_"], batch["len_query_"], batch["labels"], batch["segment_ids"])
    elif n == "lds":
        p = scorer.prepare(batch)
    else:
        p = scorer.prepare(batch["input_ids"], batch["boxes_label_input_ids"], batch["input_mask"], batch["feats"],
                           batch["boxes"], batch["visual_attention_mask"])
    return scorer.score_prepared(p)
