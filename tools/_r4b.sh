#!/bin/bash
# round 4: fused LayerNorm route with the residual as K stages -- kernel tests, micro-bench A/B (62 = GEMM + LayerNorm kernel, 63 = one fused launch), bench A/B
O=gpurun_out/r4b; mkdir -p $O
python -m pytest tests/test_round2_gpu.py -q -x -k "fused_layernorm" > $O/pytest_ln.txt 2>&1; tail -3 $O/pytest_ln.txt
GB_M=454656 GB_NSPLIT=2 GB_SHAPES="attout,768,768,0,1,1;ffn_down,768,3072,0,1,1" python tools/gemm_bench.py 62 63 > $O/gemm_bench_ln.txt 2>&1; cat $O/gemm_bench_ln.txt
python bench.py --no-cpu --no-secondary > $O/bench_zk.json 2>$O/err1.txt
python bench.py --no-cpu --no-secondary --fuse-ln > $O/bench_zk_fuseln.json 2>$O/err2.txt
python - <<'P'
import json
for f in ("bench_zk","bench_zk_fuseln"):
    try:
        r=json.load(open("gpurun_out/r4b/%s.json"%f)); print(f, r["value"], r["ms_per_step"], r["roofline"]["achieved"], r["roofline"]["avg_launch_ms"], r["roofline"]["launches"])
    except Exception as e: print(f, "failed", e)
P
