"""CPU ORACLE (second restatement) -- TEST INFRASTRUCTURE ONLY.

torch-CPU restatement of the two TF1 ImageBERT forwards (zk, lds), written independently of
``oracle/np_models.py`` on top of *library* primitives (``F.conv2d`` with explicit SAME padding,
``F.layer_norm``, ``F.softmax``, ``F.embedding``, ``F.gelu(approximate='tanh')``) so the two
restatements can only agree if both follow the cited reference lines.  It is also the timed
``cpu_baseline`` ("port", fp32, all host cores) of ``bench.py`` -- BASELINE.md section 3.

The TF1 reference cannot run here (no TensorFlow); TF semantics restated from documentation:
``slim.conv2d`` defaults (padding='SAME', activation_fn=relu), ``slim.fully_connected(x, n, None)``
linear, ``tf.layers.dense`` kernel [in,out], ``tf.contrib.layers.layer_norm`` eps 1e-12 biased
variance, ``tf.nn.l2_normalize`` = x * rsqrt(max(sum x^2, eps)).
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

H, NH, HD, NB, LL = 768, 12, 64, 10, 8


class _Prepared(dict):
    """Weights already converted by ``prepare`` (``forward`` takes them as they are)."""


def _t(w, dtype):
    if isinstance(w, _Prepared) and w.dtype == dtype:
        return w
    out = _Prepared({k: torch.from_numpy(np.array(v)).to(dtype) for k, v in w.items()})       # np.array: a writeable copy (the seeded weights are read-only)
    out.dtype = dtype
    return out


def prepare(weights, dtype=torch.float32):
    """numpy weight container -> torch tensors, once: a timed loop (bench.py's cpu_baseline) calls ``forward`` with the result instead of
    paying the ~0.5 GB conversion per batch."""
    return _t(weights, dtype)


def _ln(x, w, scope):
    return F.layer_norm(x, (x.shape[-1],), w[scope + "/gamma"], w[scope + "/beta"], eps=1e-12)


def _dense(x, w, scope):
    return F.linear(x, w[scope + "/kernel"].t(), w[scope + "/bias"])


def _encoder_layer(x, w, i, add):
    """code/imagebert_zk/pixelbert.py:928-985 (== pixelmodel.py:909-965)."""
    p = "bert/encoder/layer_%d" % i
    B, S, _ = x.shape
    heads = lambda t: t.view(B, S, NH, HD).permute(0, 2, 1, 3)
    q = heads(_dense(x, w, p + "/attention/self/query"))
    k = heads(_dense(x, w, p + "/attention/self/key"))
    v = heads(_dense(x, w, p + "/attention/self/value"))
    s = torch.matmul(q, k.transpose(-1, -2)) * (1.0 / math.sqrt(HD))
    if add is not None:
        s = s + add[:, None, None, :]
    ctx = torch.matmul(F.softmax(s, dim=-1), v).permute(0, 2, 1, 3).reshape(B, S, H)
    att = _ln(_dense(ctx, w, p + "/attention/output/dense") + x, w, p + "/attention/output/LayerNorm")
    mid = F.gelu(_dense(att, w, p + "/intermediate/dense"), approximate="tanh")
    return _ln(_dense(mid, w, p + "/output/dense") + att, w, p + "/output/LayerNorm")


def _pooled(x, w):
    return torch.tanh(_dense(x[:, 0], w, "bert/pooler/dense"))


def zk_forward(weights, batch, layers, dtype=torch.float32):
    """code/imagebert_zk/model_triple.py:162-214 + pixelbert.py:150-278."""
    w = _t(weights, dtype)
    E = w["bert/embeddings/word_embeddings"]
    ids = torch.as_tensor(batch["np_idx_query_"]).long()
    lab = torch.as_tensor(batch["np_idx_class_labels"]).long()
    B, T = ids.shape
    # kdd_conv1: NHWC [B,10,8,768] -> NCHW [B,768,10,8]; kernel [1,8,in,out] -> [out,in,1,8];
    # SAME for width 8, kernel 8, stride 1: total pad 7 -> 3 left, 4 right (model_triple.py:189)
    x = F.embedding(lab, E).permute(0, 3, 1, 2)
    k1 = w["kdd_conv1/weights"].permute(3, 2, 0, 1)
    y = F.relu(F.conv2d(F.pad(x, (3, 4, 0, 0)), k1, w["kdd_conv1/biases"]))
    lab_emb = y.mean(dim=3).permute(0, 2, 1)                                            # :190
    box = F.linear(torch.as_tensor(batch["np_boxes_5"]).to(dtype), w["kdd_dense1/weights"].t(),
                   w["kdd_dense1/biases"])                                               # :191
    feats = torch.as_tensor(batch["np_images_features"]).to(dtype)
    img = F.relu(F.linear(feats, w["kdd_conv2/weights"][0, 0].t(), w["kdd_conv2/biases"]))  # :192-194
    tok = F.linear(lab_emb + box + img, w["kdd_featureemb/fully_connected/weights"].t(),
                   w["kdd_featureemb/fully_connected/biases"])                           # pixelbert.py:449-452
    x = torch.cat([F.embedding(ids, E), tok], 1)
    x = x + F.embedding(torch.as_tensor(batch["segment_ids"]).long(), w["bert/embeddings/token_type_embeddings"])
    pos = torch.tensor(list(range(T)) + [T] * NB)
    x = x + F.embedding(pos, w["bert/embeddings/position_embeddings"])[None]
    x = _ln(x, w, "bert/embeddings/LayerNorm")
    lq = torch.as_tensor(batch["len_query_"]).long()
    nb = torch.as_tensor(batch["num_boxes"]).long()
    mask = torch.cat([torch.arange(T)[None] < lq[:, None], torch.arange(NB)[None] < nb[:, None]], 1)
    add = (1.0 - mask.to(dtype)) * -10000.0
    for i in range(layers):
        x = _encoder_layer(x, w, i, add)
    pooled = _pooled(x, w)
    # amsoftmax_loss, model_triple.py:56-86
    xn = pooled * torch.rsqrt(torch.clamp((pooled * pooled).sum(1, keepdim=True), min=1e-12))
    K = w["cls/seq_relationship/am_kernel"]
    Kn = K * torch.rsqrt(torch.clamp((K * K).sum(0, keepdim=True), min=1e-10))
    cos = torch.clamp(xn @ Kn, -1, 1)
    onehot = F.one_hot(torch.as_tensor(batch["labels"]).long(), 2).to(dtype)
    gt = (cos * onehot).sum(1, keepdim=True)
    logits = (cos - onehot * ((gt > 0.35).to(dtype) * 0.35)) * 30.0
    return logits.numpy(), F.softmax(logits, dim=-1).numpy()


def lds_forward(weights, batch, layers, dtype=torch.float32):
    """code/imagebert_lds/src/pixelmodel.py:145-270 + run_pretraining_predict_score.py:479-501."""
    w = _t(weights, dtype)
    E = w["bert/embeddings/word_embeddings"]
    ids = torch.as_tensor(batch["input_ids"]).long()
    B, T = ids.shape
    x = F.embedding(ids, E)
    x = x + F.embedding(torch.as_tensor(batch["segment_ids"]).long(), w["bert/embeddings/token_type_embeddings"])
    x = x + w["bert/embeddings/position_embeddings"][:T][None]
    x = _ln(x, w, "bert/embeddings/LayerNorm")
    feat = F.linear(torch.as_tensor(batch["features"]).to(dtype), w["featureemb/fully_connected/weights"].t(),
                    w["featureemb/fully_connected/biases"])
    # pixelmodel.py:489-498 raw reshape-matmul, done literally
    lab = torch.as_tensor(batch["labelfeat"]).long()
    g = F.embedding(lab.reshape(-1), E)
    out = torch.matmul(g.reshape(-1, LL), w["bert/embeddings/word_embeddings_labelembedding"]).squeeze(-1)
    labtok = out.reshape(B, NB, H)
    x = torch.cat([x, feat, labtok], 1)
    for i in range(layers):
        x = _encoder_layer(x, w, i, None)
    pooled = _pooled(x, w)
    logits = F.linear(pooled, w["cls/seq_relationship/output_weights"], w["cls/seq_relationship/output_bias"])
    return logits.numpy(), F.softmax(logits, dim=-1).numpy()


def forward(cfg, weights, batch, dtype=torch.float32):
    with torch.no_grad():
        if cfg.name == "zk":
            return zk_forward(weights, batch, cfg.layers, dtype)
        if cfg.name == "lds":
            return lds_forward(weights, batch, cfg.layers, dtype)
    raise ValueError(cfg.name)
