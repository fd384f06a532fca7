#!/bin/bash
# lab: persistent-GEMM grid size sweep on the headline workload (is the full 256-CU grid the best one?)
# usage (GPU box, lab library built): bash tools/grid_sweep.sh > gpurun_out/grid_sweep.txt
cd "$(dirname "$0")/.."
for g in 256 248 240 224 208 192 160 128; do
  echo "== MMS_PP_GRID=$g"
  MMS_PP_GRID=$g timeout 300 python tools/bench_lab.py --steps 5 --warmup 2 --no-cpu --no-secondary 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print(r['value'], r['ms_per_step'], r['roofline']['achieved'])"
done
