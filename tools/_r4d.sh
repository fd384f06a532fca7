#!/bin/bash
O=gpurun_out/r4d; mkdir -p $O
python -m pytest tests/test_round2_gpu.py -q -x -k "fused_layernorm" > $O/pytest_ln.txt 2>&1; tail -3 $O/pytest_ln.txt
python tools/ln_trace.py 454656 768 > $O/ln_trace_k768.txt 2>&1; python tools/ln_trace.py 454656 3072 > $O/ln_trace_k3072.txt 2>&1
head -8 $O/ln_trace_k768.txt; head -8 $O/ln_trace_k3072.txt
GB_M=454656 GB_NSPLIT=2 GB_SHAPES="attout,768,768,0,1,1;ffn_down,768,3072,0,1,1" python tools/gemm_bench.py 62 63 > $O/gemm_bench_ln.txt 2>&1; cat $O/gemm_bench_ln.txt
for m in 0 1 3; do python bench.py --no-cpu --no-secondary --fuse-ln $m > $O/bench_zk_fuseln$m.json 2>$O/err$m.txt; done
python - <<'P'
import json
for m in (0,1,3):
    try:
        r=json.load(open("gpurun_out/r4d/bench_zk_fuseln%d.json"%m)); print(m, r["value"], r["ms_per_step"], r["roofline"]["achieved"], r["roofline"]["avg_launch_ms"], r["roofline"]["launches"])
    except Exception as e: print(m, "failed", e)
P
