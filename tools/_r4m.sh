#!/bin/bash
O=gpurun_out/r4m; mkdir -p $O
python -m pytest tests/test_parity_gpu.py tests/test_round2_gpu.py tests/test_round3_gpu.py tests/test_fused_attention_gpu.py -q -k "zk or edge or surfaces or pipeline or workload or testB or large_launch" > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
python bench.py --no-cpu --no-secondary > $O/bench_zk.json 2>/dev/null; python bench.py --no-cpu --no-secondary --all-boxes > $O/bench_zk_allboxes.json 2>/dev/null
python - <<'P'
import json
for f in ("bench_zk","bench_zk_allboxes"):
    r=json.load(open("gpurun_out/r4m/%s.json"%f)); print(f, r["value"], r["ms_per_step"], r["roofline"]["whole_step"]["frac"])
P
