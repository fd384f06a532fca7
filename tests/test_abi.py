"""CPU suite: the C-ABI library loads and exports every symbol include/mmscore.h declares
(no compute calls without a GPU), and the host binding refuses to run without the extension."""
import ctypes
import os
import re

import pytest

from kddcup_2020_multimodalitiesrecall_2nd_place_amd import lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "mmscore.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mms_[a-z_0-9]+)\s*\(", src)))


def test_header_declares_expected_surface():
    names = _declared()
    for n in ("mms_create", "mms_load_weight", "mms_finalize", "mms_score_zk", "mms_score_lds", "mms_score_lxmert",
              "mms_destroy", "mms_last_error"):
        assert n in names


def test_library_exports_every_declared_symbol():
    assert os.path.exists(lib.LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    so = ctypes.CDLL(lib.LIB_PATH)
    for n in _declared():
        assert hasattr(so, n), n
    assert sorted(lib.EXPORTS) == _declared()
    assert so.mms_version() == 1


def test_create_rejects_bad_config_without_touching_a_device():
    l = lib.load()
    c = lib.Config()
    c.model = 7
    h = ctypes.c_void_p()
    assert l.mms_create(ctypes.byref(c), ctypes.byref(h)) == 1
    assert b"model" in l.mms_global_error()
    assert l.mms_create(None, ctypes.byref(h)) == 1


def test_scorers_fail_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from helpers import small_cfg
    from kddcup_2020_multimodalitiesrecall_2nd_place_amd import scorers, weights
    cfg = small_cfg("lds", layers=1)
    with pytest.raises(lib.MmsError):
        scorers.LdsScorer(cfg, weights.make_weights(cfg))


def test_missing_library_is_an_error(monkeypatch):
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", "/nonexistent/libmmscore.so")
    with pytest.raises(lib.MmsError):
        lib.load()
