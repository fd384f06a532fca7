"""lxmert's two launch lanes (api.hip mms_handle::side; lxrt/modeling.py:444-493: the language and the vision stream's sub-layers between two cross attentions are
independent, and so are the language layers and the box stream's layers in front of the first cross attention): the same kernels on the same operands, so a call on two
lanes gives the bits of the call on one lane wherever no launch reaches the fused-LayerNorm regime (>= 16 384 rows), and fp32 round-off of the other LayerNorm route above."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_two_lanes_give_the_bits_of_one_lane(tmp_path):
    from kddcup_2020_multimodalitiesrecall_2nd_place_amd import lib
    if not os.path.exists(lib.LAB_LIB_PATH):
        pytest.skip("libmmscore_lab.so not built (make -C .../csrc lab)")
    res = {}
    for tag, env in (("one", {"MMS_LANE_ROWS": "0"}), ("two", {}), ("query", {"MMS_LANE_ROWS": "1"})):
        f = str(tmp_path / (tag + ".npz"))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "lab", "lanes_worker.py"), f], env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
        res[tag] = np.load(f)
    for k in res["one"].files:
        a, b, q = res["one"][k], res["two"][k], res["query"][k]
        rows = {"B7": 7, "B60": 60, "B300": 300, "B700": 700, "B3000": 3000, "distinct": 40}[k] * 23
        if rows < 16384:      # (language rows = pairs x text_len: the longest launch of the call)
            assert np.array_equal(a, b) and np.array_equal(a, q), (k, float(np.abs(a - b).max()), float(np.abs(a - q).max()))
        else:                 # one lane: LayerNorm in the GEMM epilogue; two lanes: the LayerNorm kernel
            assert np.abs(a - b).max() < 2e-4 and np.abs(a - q).max() < 2e-4, (k, float(np.abs(a - b).max()))
