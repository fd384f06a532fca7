"""Shape/behaviour contracts of the three pair scorers on the hot path.

Every constant here restates a value hard-coded in the reference (paths relative to
/root/reference/):

* hidden 768 / 12 heads / intermediate 3072 / vocab 21128 / 512 positions / 2 token types:
  ``code/user_data/bert_config.json``.
* zk  (ImageBERT-"B"): 20 query WordPieces + 10 boxes, label text length 8, box geometry 5-d
  (``code/imagebert_zk/load_data_v4.py:28-29``, ``model_triple.py:191-201``), tanh-GELU
  (``pixelbert.py:315-328``), key mask from (len_query, num_boxes).
* lds (ImageBERT-"A"): 20 text + 10 feature + 10 label tokens, *no* attention mask
  (``code/imagebert_lds/src/pixelmodel.py:189-190,600-601``).
* lxmert: 23 query tokens, 10 boxes with 4-d geometry, 9 language / 5 relational / 5 cross layers,
  erf-GELU (``code/lxmert/src/tasks/kdd_model.py:21-23``, ``param.py:79-81``,
  ``lxrt/modeling.py:113-119``).

``hidden``/``heads`` are fixed (the HIP kernels are specialised for 768 = 12 x 64); layer counts,
vocabulary, intermediate width and position-table length may be shrunk for cheap parity cases.
"""
from dataclasses import dataclass, replace

HIDDEN = 768
HEADS = 12
HEAD_DIM = 64
FEAT_DIM = 2048
N_BOX = 10
LABEL_LEN = 8
LN_EPS = 1e-12
MASK_ADD = -10000.0  # additive key mask, pixelbert.py:813 / modeling.py:895

# ids fixed by code/user_data/vocab.txt (0-based line numbers)
PAD_ID, UNK_ID, CLS_ID, SEP_ID, MASK_ID = 0, 100, 101, 102, 103


@dataclass(frozen=True)
class ZkConfig:
    name: str = "zk"
    text_len: int = 20
    layers: int = 12
    vocab: int = 21128
    inter: int = 3072
    max_pos: int = 512
    type_vocab: int = 2
    box_dim: int = 5
    am_scale: float = 30.0   # model_triple.py:57
    am_margin: float = 0.35  # model_triple.py:58

    @property
    def seq(self):
        return self.text_len + N_BOX

    def shrunk(self, **kw):
        return replace(self, **kw)


@dataclass(frozen=True)
class LdsConfig:
    name: str = "lds"
    text_len: int = 20
    layers: int = 12
    vocab: int = 21128
    inter: int = 3072
    max_pos: int = 512
    type_vocab: int = 2

    @property
    def seq(self):
        return self.text_len + 2 * N_BOX

    def shrunk(self, **kw):
        return replace(self, **kw)


@dataclass(frozen=True)
class LxmertConfig:
    name: str = "lxmert"
    text_len: int = 23
    l_layers: int = 9
    r_layers: int = 5
    x_layers: int = 5
    vocab: int = 21128
    inter: int = 3072
    max_pos: int = 512
    type_vocab: int = 2
    box_dim: int = 4

    def shrunk(self, **kw):
        return replace(self, **kw)


# Algorithmic FLOPs per pair (SURVEY.md section 8(d) / BASELINE.md section 2): padded-shape matmul
# FLOPs of the reference graph, label-text encoder counted as a lookup, LXMERT MLM head excluded.
def flops_per_pair(cfg) -> float:
    H, I = HIDDEN, cfg.inter

    def layer(sq, sk=None):
        sk = sq if sk is None else sk
        proj = 2 * sq * H * H * 4            # q,k,v,out projections (k,v on sk rows handled by caller)
        att = 2 * 2 * sq * sk * H            # QK^T + PV over all heads
        ffn = 2 * 2 * sq * H * I
        return proj + att + ffn

    if cfg.name == "zk":
        S = cfg.seq
        emb = 2 * N_BOX * FEAT_DIM * H + 2 * N_BOX * H * H + 2 * N_BOX * cfg.box_dim * H
        head = 2 * H * H + 2 * H * 2
        return cfg.layers * layer(S) + emb + head
    if cfg.name == "lds":
        S = cfg.seq
        emb = 2 * N_BOX * FEAT_DIM * H
        head = 2 * H * H + 2 * H * 2
        return cfg.layers * layer(S) + emb + head
    if cfg.name == "lxmert":
        L, V = cfg.text_len, N_BOX
        emb = 2 * V * FEAT_DIM * H + 2 * V * cfg.box_dim * H + 2 * V * H * H  # visn_fc, box_fc, label_fc
        xl = 0
        # cross: q on own rows, k/v on other rows, out dense on own rows
        for sq, sk in ((L, V), (V, L)):
            xl += 2 * sq * H * H * 2 + 2 * sk * H * H * 2 + 2 * 2 * sq * sk * H
        xl += layer(L) + layer(V)
        head = 2 * H * H + 2 * H * 2 * H + 2 * 2 * H * 2
        return cfg.l_layers * layer(L) + cfg.r_layers * layer(V) + cfg.x_layers * xl + emb + head
    raise ValueError(cfg.name)
