#!/bin/bash
O=gpurun_out/r4c; mkdir -p $O
python tools/ln_trace.py 454656 768 > $O/ln_trace_k768.txt 2>&1; python tools/ln_trace.py 454656 3072 > $O/ln_trace_k3072.txt 2>&1
head -8 $O/ln_trace_k768.txt; head -8 $O/ln_trace_k3072.txt
for m in 0 1 2 3; do python bench.py --no-cpu --no-secondary --fuse-ln $m > $O/bench_zk_fuseln$m.json 2>$O/err$m.txt; done
python - <<'P'
import json
for m in range(4):
    try:
        r=json.load(open("gpurun_out/r4c/bench_zk_fuseln%d.json"%m)); print(m, r["value"], r["ms_per_step"], r["roofline"]["achieved"], r["roofline"]["avg_launch_ms"], r["roofline"]["launches"])
    except Exception as e: print(m, "failed", e)
P
