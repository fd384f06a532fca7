"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  numpy restatement of the reference's three pair forwards.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package; the product path (``kddcup_2020_multimodalitiesrecall_2nd_place_amd``) never does.

Every function cites the reference lines it follows (paths relative to /root/reference/).  The
restatement is dtype-generic: float64 for goldens, float32 for the timed CPU baseline.

Pinning status (also in DESIGN.md):
* lxmert: PINNED -- the reference imports and runs in the build container; the goldens under
  ``tests/golden/lxmert_*.npz`` were produced by the reference itself
  (``tests/golden/make_lxmert_golden.py``) and this restatement reproduces them.
* TF-style encoder layer (shared by zk and lds): PINNED through the reference's
  ``lxrt/modeling.BertLayer`` built with a tanh-GELU callable (``tests/golden/bertlayer_tanh_*.npz``).
* zk / lds embedding stages and heads: PARITY UNPINNED by the reference (TensorFlow 1.x is not
  installable here, the repo has no tests/weights/inputs for them).  They are cross-checked by a
  second, independently written torch restatement (``oracle/torch_models.py``) using library
  conv2d / layer_norm / softmax primitives.
"""
from __future__ import annotations

import math

import numpy as np

HIDDEN, HEADS, HEAD_DIM = 768, 12, 64
N_BOX, LABEL_LEN = 10, 8
LN_EPS = 1e-12


# --------------------------------------------------------------------------------------------
# primitives
# --------------------------------------------------------------------------------------------
def layer_norm(x, gamma, beta, eps=LN_EPS):
    """tf.contrib.layers.layer_norm (pixelbert.py:414-417) == torch.nn.LayerNorm(eps=1e-12)
    (modeling.py:266): biased variance over the last axis, eps inside the rsqrt."""
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * gamma + beta


def gelu_tanh(x):
    """pixelbert.py:326-328 / pixelmodel.py:318-320."""
    c = x.dtype.type(math.sqrt(2.0 / math.pi))
    return x * (0.5 * (1.0 + np.tanh(c * (x + x.dtype.type(0.044715) * x * x * x))))


def _erf(x):
    # vectorised erf without scipy (scipy is not guaranteed in every consumer): use math.erf via
    # numpy's frompyfunc only for small inputs, else scipy if present.
    try:
        from scipy.special import erf  # type: ignore
        return erf(x).astype(x.dtype)
    except Exception:  # pragma: no cover
        return np.vectorize(math.erf, otypes=[x.dtype])(x)


def gelu_erf(x):
    """modeling.py:113-119."""
    return x * x.dtype.type(0.5) * (1.0 + _erf(x / x.dtype.type(math.sqrt(2.0))))


def softmax(x):
    m = x.max(-1, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(-1, keepdims=True)


def attention(q_in, kv_in, wq, bq, wk, bk, wv, bv, key_add=None):
    """Multi-head attention core.  Weights are [in, out].

    pixelbert.py:767-850 (tf.layers.dense kernels [in,out], scores * 1/sqrt(64), additive
    (1-mask)*-10000, softmax over keys, probs @ V) == modeling.py:326-352.
    ``key_add``: [B, Sk] additive term ((1-m)*-10000) or None.
    """
    B, Sq, _ = q_in.shape
    Sk = kv_in.shape[1]
    q = (q_in @ wq + bq).reshape(B, Sq, HEADS, HEAD_DIM).transpose(0, 2, 1, 3)
    k = (kv_in @ wk + bk).reshape(B, Sk, HEADS, HEAD_DIM).transpose(0, 2, 1, 3)
    v = (kv_in @ wv + bv).reshape(B, Sk, HEADS, HEAD_DIM).transpose(0, 2, 1, 3)
    s = (q @ k.transpose(0, 1, 3, 2)) * q.dtype.type(1.0 / math.sqrt(HEAD_DIM))
    if key_add is not None:
        s = s + key_add[:, None, None, :]
    p = softmax(s)
    ctx = (p @ v).transpose(0, 2, 1, 3).reshape(B, Sq, HIDDEN)
    return ctx


def _cast(w: dict, dtype):
    return {k: np.asarray(v, dtype=dtype) for k, v in w.items()}


def _key_add(mask, dtype):
    return ((1.0 - mask.astype(dtype)) * dtype(-10000.0)).astype(dtype)


# --------------------------------------------------------------------------------------------
# TF-style (zk / lds) encoder: pixelbert.py:855-995 == pixelmodel.py:836-974
# --------------------------------------------------------------------------------------------
def tf_encoder_layer(x, w, i, key_add):
    p = "bert/encoder/layer_%d" % i
    a = p + "/attention/self/"
    ctx = attention(x, x, w[a + "query/kernel"], w[a + "query/bias"], w[a + "key/kernel"],
                    w[a + "key/bias"], w[a + "value/kernel"], w[a + "value/bias"], key_add)
    att = ctx @ w[p + "/attention/output/dense/kernel"] + w[p + "/attention/output/dense/bias"]
    att = layer_norm(att + x, w[p + "/attention/output/LayerNorm/gamma"],
                     w[p + "/attention/output/LayerNorm/beta"])                     # :960-966
    mid = gelu_tanh(att @ w[p + "/intermediate/dense/kernel"] + w[p + "/intermediate/dense/bias"])
    out = mid @ w[p + "/output/dense/kernel"] + w[p + "/output/dense/bias"]
    return layer_norm(out + att, w[p + "/output/LayerNorm/gamma"], w[p + "/output/LayerNorm/beta"])


def tf_pooler(x, w):
    """pixelbert.py:258-266: tanh(dense(seq[:, 0]))."""
    return np.tanh(x[:, 0] @ w["bert/pooler/dense/kernel"] + w["bert/pooler/dense/bias"])


# --------------------------------------------------------------------------------------------
# imagebert_zk
# --------------------------------------------------------------------------------------------
def zk_label_text(label_ids, w):
    """model_triple.py:178-190: word_embeddings[ids] [B,10,8,768] -> slim.conv2d(768,[1,8]) ->
    mean over the 8 positions.  slim.conv2d defaults: padding='SAME' (total pad 7 = 3 left +
    4 right), bias, activation_fn=relu.  out[p] = relu(b + sum_k x[p+k-3] @ W[0,k]).
    """
    E = w["bert/embeddings/word_embeddings"]
    W = w["kdd_conv1/weights"][0]          # [8, in, out]
    b = w["kdd_conv1/biases"]
    x = E[label_ids]                        # [B,10,8,768]
    out = np.zeros_like(x)
    for p in range(LABEL_LEN):
        acc = np.zeros(x.shape[:2] + (HIDDEN,), x.dtype) + b
        for k in range(LABEL_LEN):
            src = p + k - 3
            if 0 <= src < LABEL_LEN:
                acc = acc + x[:, :, src, :] @ W[k]
        out[:, :, p, :] = np.maximum(acc, 0)
    return out.mean(2)


def zk_image_tokens(batch, w):
    """model_triple.py:189-195 then pixelbert.py:186,449-452 (kdd_featureemb, linear)."""
    lab = zk_label_text(batch["np_idx_class_labels"].astype(np.int64), w)
    box = batch["np_boxes_5"] @ w["kdd_dense1/weights"] + w["kdd_dense1/biases"]          # :191 linear
    img = np.maximum(batch["np_images_features"] @ w["kdd_conv2/weights"][0, 0]
                     + w["kdd_conv2/biases"], 0)                                       # :192-194 relu
    tok = lab + box + img                                                              # :195
    return tok @ w["kdd_featureemb/fully_connected/weights"] + w["kdd_featureemb/fully_connected/biases"]


def zk_embeddings(batch, w):
    """pixelbert.py:203-227,541-621: text lookup, concat text||image, + token type (fed
    segment_ids), + positions [0..19]+[20]*10, LayerNorm over all 30 rows."""
    ids = batch["np_idx_query_"].astype(np.int64)
    T = ids.shape[1]
    text = w["bert/embeddings/word_embeddings"][ids]
    x = np.concatenate([text, zk_image_tokens(batch, w)], 1)
    x = x + w["bert/embeddings/token_type_embeddings"][batch["segment_ids"].astype(np.int64)]
    pos = np.array(list(range(T)) + [T] * N_BOX)                                       # :613-617
    x = x + w["bert/embeddings/position_embeddings"][pos][None]
    return layer_norm(x, w["bert/embeddings/LayerNorm/gamma"], w["bert/embeddings/LayerNorm/beta"])


def zk_key_mask(batch, text_len):
    """model_triple.py:198-201: sequence_mask(len_query,20) || sequence_mask(num_boxes,10)."""
    qm = np.arange(text_len)[None, :] < batch["len_query_"][:, None]
    bm = np.arange(N_BOX)[None, :] < batch["num_boxes"][:, None]
    return np.concatenate([qm, bm], 1)


def zk_head(pooled, labels, w, scale=30.0, margin=0.35):
    """model_triple.py:56-86: AM-softmax head, label dependent at inference.
    l2_normalize(x) = x * rsqrt(max(sum x^2, eps)), eps 1e-12 (pooled) / 1e-10 (kernel columns)."""
    dt = pooled.dtype.type
    xn = pooled / np.sqrt(np.maximum((pooled ** 2).sum(1, keepdims=True), dt(1e-12)))
    K = w["cls/seq_relationship/am_kernel"]
    Kn = K / np.sqrt(np.maximum((K ** 2).sum(0, keepdims=True), dt(1e-10)))
    cos = np.clip(xn @ Kn, -1, 1)
    onehot = np.eye(2, dtype=pooled.dtype)[labels.astype(np.int64)]
    gt = (cos * onehot).sum(1, keepdims=True)
    m = (gt > dt(margin)).astype(pooled.dtype) * dt(margin)
    logits = (cos - onehot * m) * dt(scale)
    return logits, softmax(logits)


def zk_forward(weights, batch, layers, dtype=np.float64, intermediates=None):
    """model_attention_channel_e (model_triple.py:162-214) -> (logits[B,2], probs[B,2])."""
    w = _cast(weights, dtype)
    b = {k: (np.asarray(v, dtype) if np.issubdtype(np.asarray(v).dtype, np.floating) else np.asarray(v))
         for k, v in batch.items()}
    x = zk_embeddings(b, w)
    if intermediates is not None:
        intermediates["embedding_output"] = x.copy()
    key_add = _key_add(zk_key_mask(b, b["np_idx_query_"].shape[1]), dtype)
    for i in range(layers):
        x = tf_encoder_layer(x, w, i, key_add)
        if intermediates is not None:
            intermediates["layer_%d" % i] = x.copy()
    pooled = tf_pooler(x, w)
    if intermediates is not None:
        intermediates["pooled"] = pooled.copy()
    return zk_head(pooled, b["labels"], w)


# --------------------------------------------------------------------------------------------
# imagebert_lds
# --------------------------------------------------------------------------------------------
def lds_label_tokens(label_ids, w):
    """pixelmodel.py:489-498: gathered [B*80,768] is RAW-reshaped to (-1, 8) and multiplied by
    word_embeddings_labelembedding [8,1]  =>  out[b,box,j] = sum_k wl[k] * E[ids[b,box,j//96]][8*(j%96)+k]."""
    E = w["bert/embeddings/word_embeddings"]
    wl = w["bert/embeddings/word_embeddings_labelembedding"]
    B = label_ids.shape[0]
    g = E[label_ids.reshape(-1)]                       # [B*80, 768]
    out = (g.reshape(-1, LABEL_LEN) @ wl)[:, 0]
    return out.reshape(B, N_BOX, HIDDEN)


def lds_embeddings(batch, w):
    """pixelmodel.py:199-232,506-602: text + type + pos(0..19) -> LN; feature tokens (linear
    2048->768, :439-442) and label tokens are concatenated AFTER the LN (:600-601)."""
    ids = batch["input_ids"].astype(np.int64)
    T = ids.shape[1]
    x = w["bert/embeddings/word_embeddings"][ids]
    x = x + w["bert/embeddings/token_type_embeddings"][batch["segment_ids"].astype(np.int64)]
    x = x + w["bert/embeddings/position_embeddings"][:T][None]
    x = layer_norm(x, w["bert/embeddings/LayerNorm/gamma"], w["bert/embeddings/LayerNorm/beta"])
    feat = batch["features"] @ w["featureemb/fully_connected/weights"] + w["featureemb/fully_connected/biases"]
    lab = lds_label_tokens(batch["labelfeat"].astype(np.int64), w)
    return np.concatenate([x, feat, lab], 1)


def lds_forward(weights, batch, layers, dtype=np.float64, intermediates=None):
    """bertmodel (run_pretraining_predict_score.py:288-336) + get_next_sentence_output (:479-501).
    input_mask is None -> all-ones mask -> no additive term (pixelmodel.py:189-190)."""
    w = _cast(weights, dtype)
    b = {k: (np.asarray(v, dtype) if np.issubdtype(np.asarray(v).dtype, np.floating) else np.asarray(v))
         for k, v in batch.items() if k not in ("query_id", "product_id")}
    x = lds_embeddings(b, w)
    if intermediates is not None:
        intermediates["embedding_output"] = x.copy()
    for i in range(layers):
        x = tf_encoder_layer(x, w, i, None)
        if intermediates is not None:
            intermediates["layer_%d" % i] = x.copy()
    pooled = tf_pooler(x, w)
    if intermediates is not None:
        intermediates["pooled"] = pooled.copy()
    logits = pooled @ w["cls/seq_relationship/output_weights"].T + w["cls/seq_relationship/output_bias"]
    return logits, softmax(logits)


# --------------------------------------------------------------------------------------------
# lxmert (torch Linear weights are [out, in])
# --------------------------------------------------------------------------------------------
def _lin(x, w, name):
    return x @ w[name + ".weight"].T + w[name + ".bias"]


def _ln(x, w, name):
    return layer_norm(x, w[name + ".weight"], w[name + ".bias"])


def pt_attention(q_in, kv_in, w, p, sub, key_add):
    """BertAttention (modeling.py:300-352)."""
    n = "%s.%s." % (p, sub)
    return attention(q_in, kv_in, w[n + "query.weight"].T, w[n + "query.bias"], w[n + "key.weight"].T,
                     w[n + "key.bias"], w[n + "value.weight"].T, w[n + "value.bias"], key_add)


def pt_att_block(q_in, kv_in, w, p, sub, key_add):
    """BertSelfattLayer / BertCrossattLayer (modeling.py:355-392): att -> dense -> LN(x + input)."""
    ctx = pt_attention(q_in, kv_in, w, p, sub, key_add)
    return _ln(_lin(ctx, w, p + ".output.dense") + q_in, w, p + ".output.LayerNorm")


def pt_ffn(x, w, inter, out):
    """BertIntermediate + BertOutput (modeling.py:395-420), erf GELU."""
    mid = gelu_erf(_lin(x, w, inter + ".dense"))
    return _ln(_lin(mid, w, out + ".dense") + x, w, out + ".LayerNorm")


def pt_bert_layer(x, w, p, key_add):
    """BertLayer (modeling.py:423-434)."""
    att = pt_att_block(x, x, w, p + ".attention", "self", key_add)
    return pt_ffn(att, w, p + ".intermediate", p + ".output")


def pt_embeddings(ids, w):
    """BertEmbeddings (modeling.py:269-297): word + pos(0..S-1) + type(0) -> LN."""
    b = "lxrt_encoder.model.bert.embeddings."
    S = ids.shape[-1]
    x = w[b + "word_embeddings.weight"][ids] + w[b + "position_embeddings.weight"][:S] \
        + w[b + "token_type_embeddings.weight"][0]
    return _ln(x, w, b + "LayerNorm")


def lxmert_visn_tokens(batch, w):
    """VisualFeatEncoder (modeling.py:496-533): (LN(Wf f) + LN(Wb b) + LN(Wl conv(label)))/3.
    label text goes through BertEmbeddings per box (modeling.py:915), then Conv2d(8->1,k=1) over
    the token-position axis (:526)."""
    v = "lxrt_encoder.model.bert.encoder.visn_fc."
    x = _ln(_lin(batch["feats"], w, v + "visn_fc"), w, v + "visn_layer_norm")
    y = _ln(_lin(batch["boxes"], w, v + "box_fc"), w, v + "box_layer_norm")
    lab = pt_embeddings(batch["boxes_label_input_ids"].astype(np.int64), w)        # [B,10,8,768]
    cw = w[v + "label_conv.weight"].reshape(LABEL_LEN)
    z = np.einsum("bnth,t->bnh", lab, cw) + w[v + "label_conv.bias"][0]
    z = _ln(_lin(z, w, v + "label_fc"), w, v + "label_layer_norm")
    return (x + y + z) / x.dtype.type(3.0)


def lxmert_forward(weights, batch, l_layers, r_layers, x_layers, dtype=np.float64, intermediates=None):
    """KDDModel.forward (kdd_model.py:183-214) with default flags -> logit_fc(pooled);
    LXRTModel.forward (modeling.py:872-927); LXRTEncoder.forward (:568-593)."""
    w = _cast(weights, dtype)
    b = {k: (np.asarray(v, dtype) if np.issubdtype(np.asarray(v).dtype, np.floating) else np.asarray(v))
         for k, v in batch.items()}
    e = "lxrt_encoder.model.bert.encoder."
    lang_add = _key_add(b["input_mask"], dtype)
    visn_add = _key_add(b["visual_attention_mask"], dtype)
    lang = pt_embeddings(b["input_ids"].astype(np.int64), w)
    visn = lxmert_visn_tokens(b, w)
    if intermediates is not None:
        intermediates["lang_emb"], intermediates["visn_emb"] = lang.copy(), visn.copy()
    for i in range(l_layers):
        lang = pt_bert_layer(lang, w, "%slayer.%d" % (e, i), lang_add)
    for i in range(r_layers):
        visn = pt_bert_layer(visn, w, "%sr_layers.%d" % (e, i), visn_add)
    if intermediates is not None:
        intermediates["lang_l"], intermediates["visn_r"] = lang.copy(), visn.copy()
    for i in range(x_layers):
        p = "%sx_layers.%d" % (e, i)
        # LXRTXLayer (modeling.py:444-493): ONE visual_attention module for both directions
        la = pt_att_block(lang, visn, w, p + ".visual_attention", "att", visn_add)
        va = pt_att_block(visn, lang, w, p + ".visual_attention", "att", lang_add)
        la = pt_att_block(la, la, w, p + ".lang_self_att", "self", lang_add)
        va = pt_att_block(va, va, w, p + ".visn_self_att", "self", visn_add)
        lang = pt_ffn(la, w, p + ".lang_inter", p + ".lang_output")
        visn = pt_ffn(va, w, p + ".visn_inter", p + ".visn_output")
        if intermediates is not None:
            intermediates["lang_x%d" % i], intermediates["visn_x%d" % i] = lang.copy(), visn.copy()
    pooled = np.tanh(_lin(lang[:, 0], w, "lxrt_encoder.model.bert.pooler.dense"))   # :596-608
    if intermediates is not None:
        intermediates["pooled"] = pooled.copy()
    h = gelu_erf(_lin(pooled, w, "logit_fc.0"))                                      # kdd_model.py:167-172
    h = _ln(h, w, "logit_fc.2")
    logits = _lin(h, w, "logit_fc.3")
    return logits, softmax(logits)


def forward(cfg, weights, batch, dtype=np.float64, intermediates=None):
    if cfg.name == "zk":
        return zk_forward(weights, batch, cfg.layers, dtype, intermediates)
    if cfg.name == "lds":
        return lds_forward(weights, batch, cfg.layers, dtype, intermediates)
    if cfg.name == "lxmert":
        return lxmert_forward(weights, batch, cfg.l_layers, cfg.r_layers, cfg.x_layers, dtype, intermediates)
    raise ValueError(cfg.name)
