"""Prints the numbers DESIGN.md section 6 quotes from the committed evidence set (profiles/<prefix>bench_*.json, <prefix>batch_sweep.json): python tools/design_table.py [prefix]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = sys.argv[1] if len(sys.argv) > 1 else "rd5z_"


def load(name):
    return json.loads(open(os.path.join(ROOT, "profiles", P + name)).read().strip().splitlines()[-1])


def row(name):
    d = load("bench_%s.json" % name)
    r = d.get("roofline", {})
    g = lambda k, f="achieved": (r.get(k) or {}).get(f)
    return dict(value=d["value"], ms=d["ms_per_step"], all=r.get("achieved"), plain=g("plain_epilogue"), lnf=g("layernorm_fused"), fused=g("fused_qkv_attention"),
                fused_ms=g("fused_qkv_attention", "avg_launch_ms"), whole=g("whole_step", "frac"), whole_tf=g("whole_step"))


for n in ("default", "zk", "zk_fuseattn1", "zk_fuseattn0", "zk_fuseln0", "zk_mode3", "zk_mode3_fuseattn2", "zk_testB", "zk_valid", "zk_2ranks_shared_gpu", "lds", "lds_fuseattn0", "lds_fuseattn1",
          "lxmert", "lxmert_fuseattn0", "lxmert_fuseattn1", "ensemble", "zk_fp8", "ensemble_fp8"):
    try:
        print("%-22s %s" % (n, row(n)))
    except Exception as e:
        print(n, "missing", e)
d = load("bench_default.json")
s = d["secondary"]
print("secondary:", {k: v.get("value") for k, v in s.items() if isinstance(v, dict) and "value" in v})
print("call_latency_ms:", s["call_latency_ms"]["pairs_per_call"])
print("cpu:", d["cpu_baseline"]["value"], d["cpu_baseline"].get("hip_vs_port_max_vecrel"), "lds", s["lds"]["cpu_port"]["value"], "lxmert", s["lxmert"]["cpu_port"]["value"])
print("parity:", {k: s[k]["parity_max_vecrel_vs_fp32_port"] for k in ("precision3", "lds", "lxmert", "dense")})
print("box sweep:", [(b["mean_boxes_per_pair"], b["live_token_fraction"], b["value"]) for b in s["box_sweep"]])
print("shard rates:", {k: (v["value_one_gpu"], v["predicted_strong_8"]) for k, v in s["shard_rates"].items()})
sw = json.load(open(os.path.join(ROOT, "profiles", P + "batch_sweep.json")))["sweep"]
for m, rows in sw.items():
    print("sweep", m, [(r["pairs_per_call"], r["ms_per_call"], r["of_large_batch_rate"]) for r in rows])
