"""Flat weight container (checkpoint-variable name -> fp32 array) and a seeded generator.

No checkpoint ships with the reference (weights sit behind a Baidu-pan link, README.md:47-48), so
parity and throughput are measured on seeded synthetic weights of the exact reference shapes.  The
tensor NAMES are the reference's own checkpoint variable names, so a real-checkpoint importer is a
rename-free dict fill:

* zk / lds: TF1 variable names as created by ``code/imagebert_zk/pixelbert.py:200-266``,
  ``model_triple.py:62,189-193``, ``code/imagebert_lds/src/pixelmodel.py:196-270,439-498`` and
  ``run_pretraining_predict_score.py:484-490`` (dense ``kernel`` is [in, out]).
* lxmert: ``KDDModel.state_dict()`` keys (torch ``Linear.weight`` is [out, in]); the unused MLM /
  AM-softmax heads (``kdd_model.py:174-181``) are not generated.

The generator is counter-based (splitmix64 keyed by tensor name) so the same tensors exist in this
container and on the GPU box without shipping them.  Matrices that feed MFMA are rounded to
bf16-representable fp32 values: bf16 is this build's *storage format* for GEMM weights, and both the
oracle and the HIP path consume exactly the same numbers (DESIGN.md "precision modes").
"""
from __future__ import annotations

import hashlib

import numpy as np

from .config import FEAT_DIM, HIDDEN, LABEL_LEN, LdsConfig, LxmertConfig, ZkConfig

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def _key(name: str, seed: int) -> np.uint64:
    h = hashlib.sha256(("%d|%s" % (seed, name)).encode()).digest()
    return np.uint64(int.from_bytes(h[:8], "little"))


def uniform01(name: str, n: int, seed: int, stream: int = 0, start: int = 0) -> np.ndarray:
    """n doubles in (0, 1), a pure function of (name, seed, stream, index); ``start``: index of the first one."""
    with np.errstate(over="ignore"):
        ctr = np.arange(start, start + n, dtype=np.uint64) * np.uint64(2) + np.uint64(stream)
        bits = _splitmix64(_splitmix64(ctr ^ _key(name, seed)))
    return ((bits >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


_CHUNK = 1 << 20
_pool = None


def _normal_chunk(name, seed, std, mean, start, n, bf16=False):
    u1 = uniform01(name, n, seed, 0, start)
    u2 = uniform01(name, n, seed, 1, start)
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    out = (mean + std * z).astype(np.float32)
    return round_to_bf16(out) if bf16 else out


def normal(name: str, shape, seed: int, std: float = 1.0, mean: float = 0.0, bf16: bool = False) -> np.ndarray:
    """Every element is a pure function of (name, seed, index), so large tensors are generated in chunks on a thread pool (numpy releases
    the GIL inside its ufuncs): the full-size models take seconds instead of 15 .. 20 s each (half of the GPU suite's wall time was this)."""
    global _pool
    n = int(np.prod(shape))
    if n <= 2 * _CHUNK:
        return _normal_chunk(name, seed, std, mean, 0, n, bf16).reshape(shape)
    if _pool is None:
        import os
        from concurrent.futures import ThreadPoolExecutor
        _pool = ThreadPoolExecutor(max(1, min(32, (os.cpu_count() or 2) - 1)))
    starts = list(range(0, n, _CHUNK))
    parts = list(_pool.map(lambda s0: _normal_chunk(name, seed, std, mean, s0, min(_CHUNK, n - s0), bf16), starts))
    return np.concatenate(parts).reshape(shape)


def round_to_bf16(x: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even fp32 -> bf16, returned as fp32 (low 16 mantissa bits zero)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    b = x.view(np.uint32).astype(np.uint64)
    b = (b + np.uint64(0x7FFF) + ((b >> np.uint64(16)) & np.uint64(1))) & np.uint64(0xFFFF0000)
    return b.astype(np.uint32).view(np.float32).reshape(x.shape)


class _Gen:
    def __init__(self, seed: int, bf16_matrices: bool):
        self.seed, self.bf16 = seed, bf16_matrices
        self.out: dict[str, np.ndarray] = {}

    def mat(self, name, shape, std):
        self.out[name] = normal(name, shape, self.seed, std, bf16=self.bf16)      # rounded to bf16 inside the (parallel) chunks

    def vec(self, name, shape, std, mean=0.0):
        self.out[name] = normal(name, shape, self.seed, std, mean)

    def ln_tf(self, scope):
        self.vec(scope + "/gamma", (HIDDEN,), 0.1, 1.0)
        self.vec(scope + "/beta", (HIDDEN,), 0.1)

    def ln_pt(self, scope, n=HIDDEN):
        self.vec(scope + ".weight", (n,), 0.1, 1.0)
        self.vec(scope + ".bias", (n,), 0.1)


# std choices: fan-in scaled so post-LN activations, attention logits (std ~ 2) and the final
# 2-way logits stay O(1) through the whole depth -- a 1e-3 *relative* logit check is then meaningful.
_STD_H = 0.04      # fan-in 768
_STD_I = 0.02      # fan-in 3072
_STD_F = 0.03      # fan-in 2048, inputs relu(N(0,1))
_STD_B = 0.02      # biases
_STD_E = 0.05      # embedding tables


def _tf_encoder(g: _Gen, layers: int, inter: int):
    for i in range(layers):
        p = "bert/encoder/layer_%d" % i
        for n in ("query", "key", "value"):
            g.mat("%s/attention/self/%s/kernel" % (p, n), (HIDDEN, HIDDEN), _STD_H)
            g.vec("%s/attention/self/%s/bias" % (p, n), (HIDDEN,), _STD_B)
        g.mat(p + "/attention/output/dense/kernel", (HIDDEN, HIDDEN), _STD_H)
        g.vec(p + "/attention/output/dense/bias", (HIDDEN,), _STD_B)
        g.ln_tf(p + "/attention/output/LayerNorm")
        g.mat(p + "/intermediate/dense/kernel", (HIDDEN, inter), _STD_H)
        g.vec(p + "/intermediate/dense/bias", (inter,), _STD_B)
        g.mat(p + "/output/dense/kernel", (inter, HIDDEN), _STD_I)
        g.vec(p + "/output/dense/bias", (HIDDEN,), _STD_B)
        g.ln_tf(p + "/output/LayerNorm")
    g.mat("bert/pooler/dense/kernel", (HIDDEN, HIDDEN), _STD_H)
    g.vec("bert/pooler/dense/bias", (HIDDEN,), _STD_B)


def _tf_embeddings(g: _Gen, cfg):
    g.vec("bert/embeddings/word_embeddings", (cfg.vocab, HIDDEN), _STD_E)
    g.vec("bert/embeddings/token_type_embeddings", (cfg.type_vocab, HIDDEN), _STD_E)
    g.vec("bert/embeddings/position_embeddings", (cfg.max_pos, HIDDEN), _STD_E)
    g.ln_tf("bert/embeddings/LayerNorm")


def make_zk_weights(cfg: ZkConfig = ZkConfig(), seed: int = 20200823, bf16_matrices: bool = True, _gen=None):
    g = _gen or _Gen(seed, bf16_matrices)
    _tf_embeddings(g, cfg)
    g.mat("kdd_conv1/weights", (1, LABEL_LEN, HIDDEN, HIDDEN), 0.25)  # inputs are 0.05-std embeddings
    g.vec("kdd_conv1/biases", (HIDDEN,), 0.1)
    g.vec("kdd_dense1/weights", (cfg.box_dim, HIDDEN), 0.5)
    g.vec("kdd_dense1/biases", (HIDDEN,), _STD_B)
    g.mat("kdd_conv2/weights", (1, 1, FEAT_DIM, HIDDEN), _STD_F)
    g.vec("kdd_conv2/biases", (HIDDEN,), 0.1)
    g.mat("kdd_featureemb/fully_connected/weights", (HIDDEN, HIDDEN), _STD_H)
    g.vec("kdd_featureemb/fully_connected/biases", (HIDDEN,), _STD_B)
    _tf_encoder(g, cfg.layers, cfg.inter)
    g.vec("cls/seq_relationship/am_kernel", (HIDDEN, 2), 0.05)
    return g.out


def make_lds_weights(cfg: LdsConfig = LdsConfig(), seed: int = 20200823, bf16_matrices: bool = True, _gen=None):
    g = _gen or _Gen(seed + 1, bf16_matrices)
    _tf_embeddings(g, cfg)
    g.vec("bert/embeddings/word_embeddings_labelembedding", (LABEL_LEN, 1), 2.0)
    g.mat("featureemb/fully_connected/weights", (FEAT_DIM, HIDDEN), _STD_F)
    g.vec("featureemb/fully_connected/biases", (HIDDEN,), 0.1)
    _tf_encoder(g, cfg.layers, cfg.inter)
    g.vec("cls/seq_relationship/output_weights", (2, HIDDEN), 0.05)
    g.vec("cls/seq_relationship/output_bias", (2,), 0.1)
    return g.out


def _pt_att(g: _Gen, p: str, sub: str):
    for n in ("query", "key", "value"):
        g.mat("%s.%s.%s.weight" % (p, sub, n), (HIDDEN, HIDDEN), _STD_H)
        g.vec("%s.%s.%s.bias" % (p, sub, n), (HIDDEN,), _STD_B)
    g.mat(p + ".output.dense.weight", (HIDDEN, HIDDEN), _STD_H)
    g.vec(p + ".output.dense.bias", (HIDDEN,), _STD_B)
    g.ln_pt(p + ".output.LayerNorm")


def _pt_ffn(g: _Gen, inter_name: str, out_name: str, inter: int):
    g.mat(inter_name + ".dense.weight", (inter, HIDDEN), _STD_H)
    g.vec(inter_name + ".dense.bias", (inter,), _STD_B)
    g.mat(out_name + ".dense.weight", (HIDDEN, inter), _STD_I)
    g.vec(out_name + ".dense.bias", (HIDDEN,), _STD_B)
    g.ln_pt(out_name + ".LayerNorm")


def make_lxmert_weights(cfg: LxmertConfig = LxmertConfig(), seed: int = 20200823,
                        bf16_matrices: bool = True, _gen=None):
    g = _gen or _Gen(seed + 2, bf16_matrices)
    b = "lxrt_encoder.model.bert."
    g.vec(b + "embeddings.word_embeddings.weight", (cfg.vocab, HIDDEN), _STD_E)
    g.vec(b + "embeddings.position_embeddings.weight", (cfg.max_pos, HIDDEN), _STD_E)
    g.vec(b + "embeddings.token_type_embeddings.weight", (cfg.type_vocab, HIDDEN), _STD_E)
    g.ln_pt(b + "embeddings.LayerNorm")
    v = b + "encoder.visn_fc."
    g.mat(v + "visn_fc.weight", (HIDDEN, FEAT_DIM), _STD_F)
    g.vec(v + "visn_fc.bias", (HIDDEN,), 0.1)
    g.ln_pt(v + "visn_layer_norm")
    g.vec(v + "box_fc.weight", (HIDDEN, cfg.box_dim), 0.5)
    g.vec(v + "box_fc.bias", (HIDDEN,), _STD_B)
    g.ln_pt(v + "box_layer_norm")
    g.vec(v + "label_conv.weight", (1, LABEL_LEN, 1, 1), 0.5)
    g.vec(v + "label_conv.bias", (1,), 0.1)
    g.mat(v + "label_fc.weight", (HIDDEN, HIDDEN), _STD_H)
    g.vec(v + "label_fc.bias", (HIDDEN,), _STD_B)
    g.ln_pt(v + "label_layer_norm")
    for kind, n in (("layer", cfg.l_layers), ("r_layers", cfg.r_layers)):
        for i in range(n):
            p = "%sencoder.%s.%d" % (b, kind, i)
            _pt_att(g, p + ".attention", "self")
            _pt_ffn(g, p + ".intermediate", p + ".output", cfg.inter)
    for i in range(cfg.x_layers):
        p = "%sencoder.x_layers.%d" % (b, i)
        _pt_att(g, p + ".visual_attention", "att")
        _pt_att(g, p + ".lang_self_att", "self")
        _pt_att(g, p + ".visn_self_att", "self")
        _pt_ffn(g, p + ".lang_inter", p + ".lang_output", cfg.inter)
        _pt_ffn(g, p + ".visn_inter", p + ".visn_output", cfg.inter)
    g.mat(b + "pooler.dense.weight", (HIDDEN, HIDDEN), _STD_H)
    g.vec(b + "pooler.dense.bias", (HIDDEN,), _STD_B)
    g.mat("logit_fc.0.weight", (2 * HIDDEN, HIDDEN), _STD_H)
    g.vec("logit_fc.0.bias", (2 * HIDDEN,), _STD_B)
    g.ln_pt("logit_fc.2", 2 * HIDDEN)
    g.vec("logit_fc.3.weight", (2, 2 * HIDDEN), 0.03)
    g.vec("logit_fc.3.bias", (2,), 0.1)
    return g.out


_MEMO: dict = {}
_MEMO_BYTES = 3 << 30      # host memory the memo may pin (twelve full models were ~10 GB); the oldest entries go first


def make_weights(cfg, seed: int = 20200823, bf16_matrices: bool = True, memo: bool = True):
    """Seeded weights of ``cfg`` (synthetic: tests, bench.py, smoke()).  Memoised per process up to ``_MEMO_BYTES`` of host memory -- a pure function of its
    arguments, and the test suite builds the same full-size models dozens of times.  Callers get their own dict over SHARED, READ-ONLY arrays: replace an
    entry to change a tensor (``w[k] = w[k] * 2``); an in-place edit (``w[k] *= 2``) raises.  ``memo=False`` returns private, writeable arrays and leaves
    the memo alone ("Synthetic weights")."""
    maker = {"zk": make_zk_weights, "lds": make_lds_weights, "lxmert": make_lxmert_weights}[cfg.name]
    if not memo:
        return maker(cfg, seed, bf16_matrices)
    key = (repr(cfg), seed, bool(bf16_matrices))
    if key not in _MEMO:
        w = maker(cfg, seed, bf16_matrices)
        for v in w.values():
            v.flags.writeable = False
        size = lambda d: sum(v.nbytes for v in d.values())
        total = size(w) + sum(size(d) for d in _MEMO.values())
        while _MEMO and total > _MEMO_BYTES:
            total -= size(_MEMO.pop(next(iter(_MEMO))))
        _MEMO[key] = w
    return dict(_MEMO[key])


# ------------------------------------------------------------------------------------------------------------------
# checkpoint importers.  The container's tensor names ARE the reference's checkpoint
# names, so importing is a filtered dict copy plus validation; nothing here needs TensorFlow or the checkpoints
# themselves (which are not shipped: README.md:47-48).
# ------------------------------------------------------------------------------------------------------------------
def expected_shapes(cfg) -> dict:
    """{name: shape} the scorer of ``cfg`` requires (what ``mms_finalize`` checks on the C side)."""
    return _shapes_from_generator(cfg)


def _shapes_from_generator(cfg):
    class _Rec(_Gen):
        def mat(self, name, shape, std):
            self.out[name] = tuple(shape)

        def vec(self, name, shape, std, mean=0.0):
            self.out[name] = tuple(shape)
    g = _Rec(0, False)       # the make_* functions only use .mat / .vec / .ln_* of the generator they are handed
    {"zk": make_zk_weights, "lds": make_lds_weights, "lxmert": make_lxmert_weights}[cfg.name](cfg, _gen=g)
    return dict(g.out)


def kdd_state_dict_shapes(cfg) -> dict:
    """{name: shape} of ``KDDModel().state_dict()`` in the reference's own order (code/lxmert/src/tasks/kdd_model.py:154-181: ``logit_W`` is
    registered before any sub-module output shows up, then ``lxrt_encoder`` -- embeddings, ``visn_fc``, ``layer``, ``x_layers``, ``r_layers``,
    pooler (lxrt/modeling.py:540-571,856-861) --, ``logit_fc``, and the MLM / relationship heads ``cls.*`` (modeling.py:648-676; the decoder
    matrix is the word-embedding table, tied).  The scorer reads ``expected_shapes(cfg)`` of these; ``logit_W`` and ``cls.*`` are carried so that a
    checkpoint round-trips.  Pinned to the reference's key list by tests/test_kdd_dropin.py (fixture meta of lxmert_fp32ckpt.npz)."""
    used = _shapes_from_generator(cfg)
    b = "lxrt_encoder.model.bert."
    out = {"logit_W": (HIDDEN, 2)}
    groups = ("embeddings.", "encoder.visn_fc.", "encoder.layer.", "encoder.x_layers.", "encoder.r_layers.", "pooler.")
    for g in groups:                      # the generator emits r_layers before x_layers; the module registers x_layers first
        for k, shp in used.items():
            if k.startswith(b + g):
                out[k] = shp
    for k, shp in used.items():
        if k.startswith("logit_fc."):
            out[k] = shp
    assert len(out) == len(used) + 1, "a scorer tensor is outside the KDDModel key groups"
    out["cls.predictions.bias"] = (cfg.vocab,)
    out["cls.predictions.transform.dense.weight"] = (HIDDEN, HIDDEN)
    out["cls.predictions.transform.dense.bias"] = (HIDDEN,)
    out["cls.predictions.transform.LayerNorm.weight"] = (HIDDEN,)
    out["cls.predictions.transform.LayerNorm.bias"] = (HIDDEN,)
    out["cls.predictions.decoder.weight"] = (cfg.vocab, HIDDEN)
    out["cls.seq_relationship.weight"] = (2, HIDDEN)
    out["cls.seq_relationship.bias"] = (2,)
    return out


def from_torch_state_dict(cfg, state_dict) -> dict:
    """lxmert: ``torch.load('BEST.pth')`` / ``KDDModel.state_dict()`` (code/lxmert/src/tasks/kdd_model.py:131-152) ->
    container.  Unused heads (``cls.*``, ``logit_W``) and DataParallel ``module.`` prefixes are dropped."""
    want = _shapes_from_generator(cfg)
    out = {}
    for k, v in state_dict.items():
        k = k[7:] if k.startswith("module.") else k
        if k in want:
            a = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
            out[k] = np.ascontiguousarray(a, dtype=np.float32)
    validate(cfg, out)
    return out


def from_tf_variables(cfg, reader, ema: bool = None) -> dict:
    """zk / lds: ``reader`` is anything with ``get_tensor(name)`` and ``has_tensor(name)`` (a
    ``tf.train.load_checkpoint`` reader, or a dict wrapper).  zk restores the EMA shadow variables
    (code/imagebert_zk/evaluate_normal.py:204-206,212): ``<name>/ExponentialMovingAverage`` is preferred when present
    (``ema=None``) or required (``ema=True``); lds restores the raw variables (run_pretraining_predict_score.py:558-563)."""
    want = _shapes_from_generator(cfg)
    out = {}
    for name in want:
        shadow = name + "/ExponentialMovingAverage"
        if ema is not False and reader.has_tensor(shadow):
            src = shadow
        elif ema is True:
            raise KeyError("EMA shadow variable missing: " + shadow)
        else:
            src = name
        out[name] = np.ascontiguousarray(reader.get_tensor(src), dtype=np.float32)
    validate(cfg, out)
    return out


def from_tf_checkpoint(cfg, prefix: str, ema: bool = None) -> dict:
    """zk / lds weights straight from a TensorFlow checkpoint bundle (``<prefix>.index`` + ``<prefix>.data-*``), the files
    ``saver.restore(sess, ckpt)`` reads in the reference (evaluate_normal.py:204-212, run_pretraining_predict_score.py:558-563).
    No TensorFlow needed: ``tf_checkpoint.BundleReader`` parses the bundle.  ``ema`` as in ``from_tf_variables``."""
    from .tf_checkpoint import BundleReader
    return from_tf_variables(cfg, BundleReader(prefix), ema=ema)


class DictReader:
    """Minimal ``get_tensor`` / ``has_tensor`` adapter over a {name: array} dict (e.g. an .npz export of a checkpoint)."""

    def __init__(self, d):
        self.d = d

    def has_tensor(self, name):
        return name in self.d

    def get_tensor(self, name):
        return self.d[name]


def validate(cfg, weights: dict):
    want = _shapes_from_generator(cfg)
    missing = sorted(set(want) - set(weights))
    bad = sorted(k for k in want if k in weights and tuple(weights[k].shape) != want[k])
    if missing or bad:
        raise ValueError("checkpoint does not match %s: missing %s; wrong shape %s" % (
            cfg.name, missing[:5] + (["..."] if len(missing) > 5 else []),
            [(k, tuple(weights[k].shape), want[k]) for k in bad[:5]]))


def auto_precision(weights: dict) -> int:
    """Precision mode a checkpoint needs for 1e-3 logit parity with its fp32 reference: 2 (weights stored as bf16, two MFMA
    passes) when every matrix is bf16-representable -- the synthetic weights of this repo --, else 3 (weights kept as hi + lo
    planes, three passes: a real fp32 checkpoint loses ~5e-3 to the bf16 rounding alone)."""
    for k, v in weights.items():
        v = np.asarray(v)
        dims = sorted(d for d in v.shape if d > 1)
        # the GEMM operands: dense / conv kernels with both dimensions >= 256 (embedding tables, 5->768 box projections, 768->2
        # heads stay fp32 in the HIP path whatever the mode)
        if len(dims) >= 2 and dims[-2] >= 256 and "embedding" not in k and v.dtype.kind == "f":
            # the library receives fp32 (lib.Handle.load_weights casts): test the values it will see, whatever the export's dtype
            # (a float64 / float16 .npz is as much a real checkpoint as a float32 one)
            v32 = np.ascontiguousarray(v, dtype=np.float32)
            if np.any(v32.view(np.uint32) & np.uint32(0xFFFF)):
                return 3
    return 2


def bf16_rounding_report(weights: dict) -> dict:
    """How far a real fp32 checkpoint is from this build's bf16 GEMM-weight storage format: max relative rounding
    step per matrix (2^-9 worst case).  DESIGN.md 'precision modes' explains what that means for logit parity."""
    rep = {}
    for k, v in weights.items():
        if v.ndim >= 2 and v.size >= 768 * 2:
            r = round_to_bf16(v)
            denom = np.maximum(np.abs(v), 1e-30)
            rep[k] = float(np.max(np.abs(r - v) / denom))
    return rep
