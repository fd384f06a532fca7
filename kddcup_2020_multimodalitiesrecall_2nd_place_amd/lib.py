"""ctypes binding of the C ABI in ``include/mmscore.h`` (libmmscore.so, HIP/gfx950).

There is NO fallback: if the shared library is missing or no HIP device is usable, importing the
scorers raises.  PyTorch is used by the callers only as the owner of device buffers / streams; this
module passes raw pointers.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libmmscore.so")
# `make -C csrc lab`: the product sources built with -DMMS_LAB plus the superseded A/B kernels and timing-only diagnostics.
# Only the scripts under tools/ load it (``lib.load(lib.LAB_LIB_PATH)`` before anything else touches the library).
LAB_LIB_PATH = os.path.join(_HERE, "csrc", "libmmscore_lab.so")

MODEL_ZK, MODEL_LDS, MODEL_LXMERT = 0, 1, 2
MODEL_IDS = {"zk": MODEL_ZK, "lds": MODEL_LDS, "lxmert": MODEL_LXMERT}
ACT_NONE, ACT_RELU, ACT_GELU_TANH, ACT_GELU_ERF, ACT_TANH = 0, 1, 2, 3, 4


class MmsError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "model", "layers", "r_layers", "x_layers", "vocab", "inter", "max_pos", "type_vocab", "text_len",
        "precision", "chunk_pairs", "stop_after", "device", "pack_tokens", "fuse_layernorm", "fuse_attention")]


class ZkBatch(C.Structure):
    _fields_ = [("n_pairs", C.c_int64), ("num_boxes", C.c_void_p), ("boxes_5", C.c_void_p), ("feats", C.c_void_p),
                ("uniq_label_ids", C.c_void_p), ("n_uniq_labels", C.c_int64), ("label_index", C.c_void_p),
                ("query_ids", C.c_void_p), ("len_query", C.c_void_p), ("labels", C.c_void_p),
                ("segment_ids", C.c_void_p), ("label_ids", C.c_void_p)]


class LdsBatch(C.Structure):
    _fields_ = [("n_pairs", C.c_int64), ("input_ids", C.c_void_p), ("segment_ids", C.c_void_p),
                ("features", C.c_void_p), ("labelfeat", C.c_void_p)]


class LxmertBatch(C.Structure):
    _fields_ = [("n_pairs", C.c_int64), ("input_ids", C.c_void_p), ("input_mask", C.c_void_p),
                ("uniq_label_ids", C.c_void_p), ("n_uniq_labels", C.c_int64), ("label_index", C.c_void_p),
                ("feats", C.c_void_p), ("boxes", C.c_void_p), ("visual_attention_mask", C.c_void_p),
                ("label_ids", C.c_void_p), ("x_norm", C.c_void_p)]


class EnsembleBatch(C.Structure):
    _fields_ = [("n_pairs", C.c_int64), ("feats", C.c_void_p), ("boxes_5", C.c_void_p), ("num_boxes", C.c_void_p),
                ("label_ids", C.c_void_p), ("query_ids", C.c_void_p), ("len_query", C.c_void_p),
                ("s2f_query_ids", C.c_void_p), ("s2f_len_query", C.c_void_p), ("labels", C.c_void_p),
                ("lx_input_ids", C.c_void_p), ("lx_input_mask", C.c_void_p)]


ABI_VERSION = 6     # include/mmscore.h MMS_ABI_VERSION

# enum Engine of csrc/regimes.h (mms_dbg_gemm / mms_dbg_gemm_bench name ONE GEMM engine per call; tests/test_abi.py holds the two lists together)
ENGINES = {"ENG_AUTO": 0, "ENG_TILE_128": 1, "ENG_TILE_DMA": 3, "ENG_TILE": 4, "ENG_SKINNY": 5, "ENG_TILE_256": 16, "ENG_PP": 20, "ENG_PP_PERSIST": 26,
           "ENG_PPW": 27, "ENG_DW": 28, "ENG_SKINNY_K4": 54, "ENG_SKINNY_PARTS": 55, "ENG_SKINNY_K8": 58}

EXPORTS = ("mms_version", "mms_global_error", "mms_create", "mms_destroy", "mms_last_error", "mms_load_weight",
           "mms_finalize", "mms_score_zk", "mms_score_lds", "mms_score_lxmert", "mms_score_ensemble", "mms_gemm_timing", "mms_gemm_timing_class",
           "mms_debug_read_x", "mms_dbg_gemm", "mms_dbg_gemm_f8", "mms_dbg_gemm_ln", "mms_dbg_proj_ln_splitk", "mms_dbg_attention", "mms_dbg_qkv_attn", "mms_dbg_layernorm",
           "mms_dbg_gemm_bench", "mms_dbg_counter", "mms_fused_timing", "mms_side_lane_flops")
LAB_EXPORTS = ("mms_dbg_gemm_mx", "mms_lab_ln_trace")      # libmmscore_lab.so only

_lib = None


def load(path=None):
    """dlopen libmmscore.so and declare prototypes.  Raises MmsError if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise MmsError("HIP extension missing: %s (run `python -c 'import __graft_entry__ as g; g.build()'`)" % path)
    # torch must be imported first: it bundles the HIP runtime (libamdhip64.so.7) that owns the device
    # buffers/streams we are handed; libmmscore binds to that already-loaded copy by SONAME.
    import torch  # noqa: F401
    lib = C.CDLL(path)
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    lib.mms_version.restype = C.c_int
    if lib.mms_version() != ABI_VERSION:     # the ctypes structs below mirror include/mmscore.h at exactly this revision
        raise MmsError("%s has ABI revision %d, this package's ctypes structs are revision %d: rebuild (`make -C .../csrc`)"
                       % (path, lib.mms_version(), ABI_VERSION))
    lib.mms_global_error.restype = C.c_char_p
    lib.mms_last_error.restype = C.c_char_p
    lib.mms_last_error.argtypes = [vp]
    lib.mms_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    lib.mms_destroy.argtypes = [vp]
    lib.mms_destroy.restype = None
    lib.mms_load_weight.argtypes = [vp, C.c_char_p, vp, C.POINTER(i64), i32]
    lib.mms_finalize.argtypes = [vp]
    lib.mms_score_zk.argtypes = [vp, C.POINTER(ZkBatch), vp, vp, vp]
    lib.mms_score_lds.argtypes = [vp, C.POINTER(LdsBatch), vp, vp, vp]
    lib.mms_score_lxmert.argtypes = [vp, C.POINTER(LxmertBatch), vp, vp, vp]
    lib.mms_score_ensemble.argtypes = [vp, vp, vp, C.POINTER(EnsembleBatch), C.POINTER(C.c_float), vp, vp, vp]
    lib.mms_dbg_gemm_f8.argtypes = [vp, i64, i64, vp, i64, vp, i32, i32, vp, vp]
    if hasattr(lib, "mms_dbg_gemm_mx"):      # lab build
        lib.mms_dbg_gemm_mx.argtypes = [vp, i64, i64, vp, i64, vp, i32, i32, vp, vp]
    lib.mms_dbg_gemm_ln.argtypes = [vp, i64, i64, vp, vp, vp, vp, vp, i32, vp, C.POINTER(i32), vp]
    lib.mms_dbg_proj_ln_splitk.argtypes = [vp, i64, i64, vp, vp, vp, vp, vp, i32, vp, vp]
    lib.mms_gemm_timing.argtypes = [vp, i32, i32, C.POINTER(C.c_double), C.POINTER(i64), C.POINTER(C.c_double)]
    lib.mms_gemm_timing_class.argtypes = [vp, i32, C.POINTER(C.c_double), C.POINTER(i64), C.POINTER(C.c_double)]
    lib.mms_debug_read_x.argtypes = [vp, vp, i64, vp]
    lib.mms_dbg_gemm.argtypes = [vp, i64, i64, i64, vp, i64, vp, vp, i32, i32, i32, i32, vp, vp]
    lib.mms_dbg_attention.argtypes = [vp, vp, vp, i64, i32, i32, vp, vp, vp]
    lib.mms_dbg_layernorm.argtypes = [vp, vp, vp, i64, vp, vp]
    lib.mms_dbg_qkv_attn.argtypes = [vp, i64, i64, vp, vp, vp, vp, i64, i32, i32, vp, vp, vp, vp, i32, vp, C.POINTER(i32), vp]
    lib.mms_dbg_counter.argtypes = [vp, i32]
    lib.mms_fused_timing.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(i64), C.POINTER(C.c_double)]
    lib.mms_side_lane_flops.argtypes = [vp, C.POINTER(C.c_double)]
    lib.mms_dbg_counter.restype = i64
    lib.mms_dbg_gemm_bench.argtypes = [i64, i64, i64, i32, i32, i32, i32, i32, i32, C.POINTER(C.c_float)]
    _lib = lib
    return lib


class Handle:
    """Owns one ``mms_handle`` (one model on one GPU)."""

    def __init__(self, cfg, precision: int = 2, device: int = 0, chunk_pairs: int = 0, stop_after: int = -1,
                 pack_tokens: bool = True, fuse_layernorm=0, fuse_attention: int = 0):
        self.lib = load()
        self.cfg = cfg
        c = Config()
        c.model = MODEL_IDS[cfg.name]
        if cfg.name == "lxmert":
            c.layers, c.r_layers, c.x_layers = cfg.l_layers, cfg.r_layers, cfg.x_layers
        else:
            c.layers, c.r_layers, c.x_layers = cfg.layers, 0, 0
        c.vocab, c.inter, c.max_pos, c.type_vocab, c.text_len = cfg.vocab, cfg.inter, cfg.max_pos, cfg.type_vocab, cfg.text_len
        c.precision, c.chunk_pairs, c.stop_after, c.device = precision, chunk_pairs, stop_after, device
        c.pack_tokens = int(bool(pack_tokens))
        c.fuse_layernorm = 3 if fuse_layernorm is True else int(fuse_layernorm)      # mask: 1 attention-output, 2 FFN-down (True: both)
        c.fuse_attention = int(fuse_attention)
        self._h = C.c_void_p()
        rc = self.lib.mms_create(C.byref(c), C.byref(self._h))
        if rc != 0:
            raise MmsError("mms_create failed (%d): %s" % (rc, self.lib.mms_global_error().decode()))

    def _check(self, rc, what):
        if rc != 0:
            raise MmsError("%s failed (%d): %s" % (what, rc, self.lib.mms_last_error(self._h).decode()))

    def load_weights(self, weights: dict):
        for name, arr in weights.items():
            a = np.ascontiguousarray(arr, dtype=np.float32)
            shape = (C.c_int64 * a.ndim)(*a.shape)
            self._check(self.lib.mms_load_weight(self._h, name.encode(), a.ctypes.data, shape, a.ndim),
                        "mms_load_weight(%s)" % name)
        self._check(self.lib.mms_finalize(self._h), "mms_finalize")

    def score(self, batch_struct, logits_ptr, probs_ptr, stream_ptr):
        fn = {ZkBatch: self.lib.mms_score_zk, LdsBatch: self.lib.mms_score_lds, LxmertBatch: self.lib.mms_score_lxmert}[type(batch_struct)]
        self._check(fn(self._h, C.byref(batch_struct), logits_ptr, probs_ptr, stream_ptr), fn.__name__)

    def gemm_timing(self, enable: bool, reset: bool, read: bool = False):
        ms, n, fl = C.c_double(0), C.c_int64(0), C.c_double(0)
        if read:
            self._check(self.lib.mms_gemm_timing(self._h, int(enable), int(reset), C.byref(ms), C.byref(n), C.byref(fl)), "mms_gemm_timing")
        else:
            self._check(self.lib.mms_gemm_timing(self._h, int(enable), int(reset), None, None, None), "mms_gemm_timing")
        return ms.value, n.value, fl.value

    def gemm_timing_class(self, cls: int):
        """(ms, launches, flops) of the timed GEMM launches with the plain (0) / the fused LayerNorm (1) epilogue."""
        ms, n, fl = C.c_double(0), C.c_int64(0), C.c_double(0)
        self._check(self.lib.mms_gemm_timing_class(self._h, cls, C.byref(ms), C.byref(n), C.byref(fl)), "mms_gemm_timing_class")
        return ms.value, n.value, fl.value

    def side_lane_flops(self):
        """Executed FLOPs of the launches that ran on the side lane since the last gemm_timing reset (counted, not timed)."""
        fl = C.c_double(0)
        self._check(self.lib.mms_side_lane_flops(self._h, C.byref(fl)), "mms_side_lane_flops")
        return fl.value

    def fused_timing(self):
        """(ms, launches, projection flops) of the fused QKV + attention launches timed since the last gemm_timing reset."""
        ms, n, fl = C.c_double(0), C.c_int64(0), C.c_double(0)
        self._check(self.lib.mms_fused_timing(self._h, C.byref(ms), C.byref(n), C.byref(fl)), "mms_fused_timing")
        return ms.value, n.value, fl.value

    def counter(self, which: int) -> int:
        """mms_dbg_counter: 0 = fused QKV + attention launches, 1 = LayerNorm-fused GEMM launches, 2 = split-K launches (small calls), 3 = skinny-GEMM launches (launches of <= 128 padded rows), 4 = fork / join pairs of the second launch lane (lxmert; zk handle: member lanes of the fused three-model call) since creation."""
        return int(self.lib.mms_dbg_counter(self._h, which))

    def debug_read_x(self, dst_ptr, rows, stream_ptr):
        self._check(self.lib.mms_debug_read_x(self._h, dst_ptr, rows, stream_ptr), "mms_debug_read_x")

    def close(self):
        if self._h:
            self.lib.mms_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
