"""CPU suite: the C-ABI library loads and exports every symbol include/mmscore.h declares
(no compute calls without a GPU), and the host binding refuses to run without the extension."""
import ctypes
import os
import re
import subprocess

import pytest

from kddcup_2020_multimodalitiesrecall_2nd_place_amd import lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "mmscore.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mms_[a-z_0-9]+)\s*\(", src)))


def _header_abi_version():
    return int(re.search(r"#define\s+MMS_ABI_VERSION\s+(\d+)", open(os.path.join(ROOT, "include", "mmscore.h")).read()).group(1))


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
    assert so.mms_version() == lib.ABI_VERSION == _header_abi_version()


def test_product_library_carries_no_lab_engines_or_global_switches():
    """the shelved engines and the process-global GEMM-variant switch live in libmmscore_lab.so only."""
    so = ctypes.CDLL(lib.LIB_PATH)
    for n in lib.LAB_EXPORTS + ("mms_set_gemm_variant",):
        assert not hasattr(so, n), n
    syms = subprocess.run(["nm", "-D", "--defined-only", lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "gemm_dw" not in syms and "launch_gemm_mxRK" not in syms and "set_gemm_variant" not in syms


def test_create_rejects_out_of_range_route_options():
    l = lib.load()
    for field, bad in (("fuse_attention", 3), ("fuse_attention", -1), ("fuse_layernorm", 4), ("fuse_layernorm", -2)):
        c = lib.Config()
        c.model, c.layers, c.vocab, c.inter, c.max_pos, c.type_vocab, c.text_len, c.precision = 0, 1, 100, 128, 64, 2, 20, 2
        setattr(c, field, bad)
        h = ctypes.c_void_p()
        assert l.mms_create(ctypes.byref(c), ctypes.byref(h)) == 1 and field.encode() in l.mms_global_error(), field


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


def test_ctypes_structs_match_the_c_header_layout(tmp_path):
    """include/mmscore.h is plain C: compile it with gcc, print sizeof / offsetof of every struct the binding mirrors, and compare
    with the ctypes Structures of lib.py -- a drifted field order or type would otherwise only show up as garbage on a GPU."""
    import re
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    src = open(os.path.join(ROOT, "include", "mmscore.h")).read()
    src_nc = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    pairs = {"mms_config": lib.Config, "mms_zk_batch": lib.ZkBatch, "mms_lds_batch": lib.LdsBatch, "mms_lxmert_batch": lib.LxmertBatch,
             "mms_ensemble_batch": lib.EnsembleBatch}
    prog = ['#include <stdio.h>', '#include <stddef.h>', '#include "mmscore.h"', 'int main(void) {']
    fields = {}
    for cname in pairs:
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), src_nc, flags=re.S).group(1)
        names = re.findall(r"([A-Za-z_0-9]+)\s*;", body)
        fields[cname] = names
        prog.append('printf("%s %%zu", sizeof(%s));' % (cname, cname))
        for n in names:
            prog.append('printf(" %s:%%zu", offsetof(%s, %s));' % (n, cname, n))
        prog.append('printf("\\n");')
    prog += ["return 0; }"]
    c = tmp_path / "layout.c"
    c.write_text("\n".join(prog))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)])
    out = subprocess.check_output([str(exe)], text=True).splitlines()
    for line in out:
        toks = line.split()
        cname, size = toks[0], int(toks[1])
        st = pairs[cname]
        assert ctypes.sizeof(st) == size, (cname, ctypes.sizeof(st), size)
        py = [(n, getattr(st, n).offset) for n, _ in st._fields_]
        cc = [(t.split(":")[0], int(t.split(":")[1])) for t in toks[2:]]
        assert py == cc, (cname, py, cc)


def test_c_host_links_and_calls_the_library(tmp_path):
    """The C / C++ host of INTEGRATION.md: compiles against include/mmscore.h with plain gcc, links libmmscore.so, and gets the
    documented error behaviour (status code + message, nothing thrown across the ABI) without a GPU."""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    csrc = os.path.dirname(lib.LIB_PATH)
    c = tmp_path / "host.c"
    c.write_text(r'''
#include <stdio.h>
#include <string.h>
#include "mmscore.h"
int main(void) {
    mms_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.model = 9; cfg.layers = 12; cfg.vocab = 21128; cfg.inter = 3072; cfg.max_pos = 512; cfg.type_vocab = 2; cfg.text_len = 20; cfg.precision = 2;
    mms_handle* h = 0;
    int rc = mms_create(&cfg, &h);
    printf("version %d rc %d msg %s\n", mms_version(), rc, mms_global_error());
    cfg.model = MMS_MODEL_LDS; cfg.text_len = 31;                 /* 51-token sequences: beyond the attention kernels */
    rc = mms_create(&cfg, &h);
    printf("rc %d msg %s\n", rc, mms_global_error());
    mms_ensemble_batch b; memset(&b, 0, sizeof b);
    printf("ensemble(null handles) rc %d\n", mms_score_ensemble(0, 0, 0, &b, 0, 0, 0, 0));
    return 0;
}
''')
    exe = tmp_path / "host"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe), "-L", csrc,
                           "-lmmscore", "-Wl,-rpath," + csrc, "-Wl,-rpath,/opt/rocm/lib", "-Wl,--allow-shlib-undefined"])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.splitlines()
    assert lines[0].startswith("version %d rc 1" % lib.ABI_VERSION) and "model" in lines[0]
    assert lines[1].startswith("rc 1") and "48-token" in lines[1]
    assert lines[2].endswith("rc 1")


def test_header_lists_the_size_regime_bounds_the_library_uses():
    """csrc/regimes.h is the ONE table of launch-size bounds and engine numbers: include/mmscore.h carries it verbatim (generated), api.hip and
    gemm_dispatch.hip define no bound of their own and compare engines by name, lib.py mirrors the enum."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_regime_doc", os.path.join(ROOT, "tools", "gen_regime_doc.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    consts, table, engines = gen.parse()
    assert {t[0]: t[3] for t in table} == {"SKINNY_ROWS": 128, "TINY_ROWS": 1024, "FUSED_ATTN_ROWS": 1024, "TALL_ROWS": 4096, "SPLITK_HALF_ROWS": 4096,
                                           "PP_WIDE_ROWS": 5120, "SPLITK_ROWS": 8192, "SPLITK2_LO_ROWS": 11264, "PP_ROWS": 16384, "LNF_ROWS": 98304,
                                           "ENS_LANE_ROWS": 200000, "LANE_ROWS": 400000}
    assert [t[3] for t in table] == sorted(t[3] for t in table)
    hdr = open(os.path.join(ROOT, "include", "mmscore.h")).read()
    assert gen.render(hdr) == hdr, "include/mmscore.h is stale: run python tools/gen_regime_doc.py"
    for t in table:                                    # every bound, by name and value, between the markers
        assert re.search(r"rows\s+%s\s+%d\s+%s\b" % (re.escape(t[1]), t[3], t[0]), hdr), t[0]
    csrc = os.path.join(ROOT, "kddcup_2020_multimodalitiesrecall_2nd_place_amd", "csrc")
    api, disp = open(os.path.join(csrc, "api.hip")).read(), open(os.path.join(csrc, "gemm_dispatch.hip")).read()
    for src in (api, disp):                            # no second definition of a bound, no engine compared by number
        for name in consts:
            assert not re.search(r"constexpr\s+int(?:64_t)?\s+%s\s*=" % name, src), name
        assert not re.search(r"\b(?:variant|engine|e)\s*[!=]=\s*\d", src)
    for name in ("PP_ROWS_DEFAULT", "PP_WIDE_ROWS_DEFAULT"):
        assert name in disp
    for name in ("SKINNY_ROWS_DEFAULT", "TINY_ROWS_DEFAULT", "FUSED_ATTN_ROWS_DEFAULT", "TALL_ROWS", "SPLITK_HALF_ROWS", "SPLITK_ROWS", "SPLITK2_LO_ROWS", "LNF_ROWS_DEFAULT",
                 "ENS_LANE_ROWS_DEFAULT", "LANE_ROWS_DEFAULT"):
        assert name in api, name
    assert {n: v for n, v, _ in engines if n != "ENG_DIAG_BASE"} == lib.ENGINES


def test_test_hooks_are_fenced_out_of_the_product_header(tmp_path):
    """A C host that includes mmscore.h sees the scoring ABI only; the mms_dbg_* hooks need -DMMS_TEST_HOOKS (they are still exported by the library)."""
    hdr = os.path.join(ROOT, "include")
    src = tmp_path / "t.c"
    src.write_text('#include "mmscore.h"\nint main(void) { return (int)(long)&mms_dbg_counter; }\n')
    plain = subprocess.run(["gcc", "-I", hdr, "-fsyntax-only", "-Werror=implicit-function-declaration", str(src)], capture_output=True, text=True)
    assert plain.returncode != 0 and "mms_dbg_counter" in plain.stderr
    hooks = subprocess.run(["gcc", "-I", hdr, "-DMMS_TEST_HOOKS", "-fsyntax-only", str(src)], capture_output=True, text=True)
    assert hooks.returncode == 0, hooks.stderr
    src.write_text('#include "mmscore.h"\nint main(void) { return mms_version() == MMS_ABI_VERSION ? 0 : 1; }\n')
    assert subprocess.run(["gcc", "-I", hdr, "-fsyntax-only", "-Wall", "-Werror", str(src)], capture_output=True, text=True).returncode == 0
