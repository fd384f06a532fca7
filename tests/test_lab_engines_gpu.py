"""The lab build's shelved engines keep their kernel tests, in a process of their own: libmmscore_lab.so is loaded there instead of the
product library (the product binary no longer carries gemm_dw.hip or the 1.5-pass MX kernel: VERDICT r3 item 8)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_shelved_engines_in_the_lab_build():
    from kddcup_2020_multimodalitiesrecall_2nd_place_amd import lib
    if not os.path.exists(lib.LAB_LIB_PATH):
        pytest.skip("libmmscore_lab.so not built (make -C .../csrc lab)")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pytest_lab.py"), os.path.join(ROOT, "tests", "lab", "shelved_engines.py"),
                          "-q", "-x", "-p", "no:cacheprovider"], cwd=os.path.join(ROOT, "tests"), capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, (out.stdout[-3000:], out.stderr[-2000:])
    assert " passed" in out.stdout and "failed" not in out.stdout
