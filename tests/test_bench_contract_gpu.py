"""GPU suite: bench.py prints exactly one JSON line with the fields the driver and the judge read (task contract, section 4)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_prints_one_contract_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--queries", "40", "--cpu-budget", "4"],
                         cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 1 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["unit"] == "pairs/s" and d["value"] > 0 and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert abs(d["value"] - 40 * 30 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-3
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert "traffic" in r and r["achieved"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "pairs/s" and c["value"] > 0 and c["cores"] >= 1 and c["sample"]
    assert d["value"] > 20 * c["value"]
    assert c["hip_vs_port_max_vecrel"] < 1e-3                     # the CPU port's sample doubles as a check of the HIP logits
    assert d["ms_per_step_median_hipevent"] > 0
    sec = d["secondary"]
    assert sec["value_incl_h2d"]["value"] > 0 and sec["value_incl_h2d"]["value"] <= d["value"] * 1.05
    p3 = sec["precision3"]
    assert p3["precision_mode"] == 3 and p3["value"] > 0 and p3["parity_max_vecrel_vs_fp32_port"] < 1e-3
    assert sec["lds"]["value"] > 0 and sec["lxmert"]["value"] > 0
    # every secondary model line carries its own parity against the oracle's fp32 port (64 pairs of the timed batch)
    for k in ("lds", "lxmert", "dense", "all_boxes"):
        assert sec[k]["parity_max_vecrel_vs_fp32_port"] < 1e-3, (k, sec[k])
    sw = sec["box_sweep"]
    assert len(sw) == 5 and all(x["value"] > 0 and 0 < x["live_token_fraction"] <= 1 for x in sw)
    assert [x["mean_boxes_per_pair"] for x in sw] == sorted(x["mean_boxes_per_pair"] for x in sw) and sw[-1]["mean_boxes_per_pair"] == 10.0
    assert sec["dense"]["live_token_fraction"] == 1.0
    sh = sec["shard_rates"]
    assert sh["bench_strong_n8"]["predicted_strong_8"] == round(8 * sh["bench_strong_n8"]["value_one_gpu"], 1) and sh["testB_n8"]["pairs_rank0"] > 0
    cl = sec["call_latency_ms"]["pairs_per_call"]      # the reference's own call sizes through the same handle
    assert 0 < cl["1"] < cl["256"] and cl["1"] < 5.0, cl
    assert d["value_fp32_checkpoint"] == p3     # mode 3 also as a top-level value
    # the line runs the LIBRARY defaults: what make_scorer(cfg, weights) gives a user
    assert d["config"]["fuse_attention"] == 2 and d["config"]["fuse_layernorm"] == 3
    ws = r["whole_step"]
    assert 0 < ws["frac"] <= r["frac_incl_fused"] + 1e-6 and abs(ws["achieved"] - ws["executed_flops"] / (ws["ms_per_step"] * 1e-3) / 1e12) < 0.05
    fq = r["fused_qkv_attention"]          # the Q|K|V projections ran inside the fused kernel, timed apart from the plain GEMM launches
    assert fq["launches"] > 0 and fq["achieved"] > 0 and abs(fq["frac"] - fq["achieved"] / r["peak"]) < 1e-3
    assert r["achieved_incl_fused"] > 0 and min(r["achieved"], fq["achieved"]) <= r["achieved_incl_fused"] <= max(r["achieved"], fq["achieved"])


@pytest.mark.gpu
def test_bench_ensemble_and_fp8_lines():
    for extra in (["--model", "ensemble"], ["--precision", "4"]):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--queries", "30", "--no-cpu",
                              "--no-secondary"] + extra, cwd=ROOT, capture_output=True, text=True, timeout=1500)
        assert out.returncode == 0, out.stderr[-2000:]
        d = json.loads([l for l in out.stdout.splitlines() if l.strip()][-1])
        assert d["value"] > 0 and d["roofline"]["achieved"] > 0
        assert ("ensemble" in d["config"]["workload"]) == (extra[0] == "--model")
