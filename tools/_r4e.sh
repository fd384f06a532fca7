#!/bin/bash
O=gpurun_out/r4e; mkdir -p $O
python -m pytest tests -m gpu -q --durations=25 > $O/pytest.txt 2>&1; tail -40 $O/pytest.txt | grep -v "^$" | tail -34
python bench.py --batch-sweep > $O/batch_sweep.json 2> $O/batch_sweep.err; tail -c 400 $O/batch_sweep.err
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.err
python - <<'P'
import json
try:
    r=json.load(open("gpurun_out/r4e/bench_default.json")); print(r["value"], r["ms_per_step"], r["roofline"]["achieved"], r["config"])
    s=r["secondary"]; print({k:(v.get("value") if isinstance(v,dict) else [ (x["mean_boxes_per_pair"],x["live_token_fraction"],x["value"]) for x in v]) for k,v in s.items()})
    print({k:v.get("parity_max_vecrel_vs_fp32_port") for k,v in s.items() if isinstance(v,dict)})
except Exception as e: print("bench failed", e)
try:
    b=json.load(open("gpurun_out/r4e/batch_sweep.json"))
    for m,rows in b["sweep"].items(): print(m, [(r["pairs_per_call"], r["ms_per_call"], r["pairs_per_s"]) for r in rows])
except Exception as e: print("sweep failed", e)
P
