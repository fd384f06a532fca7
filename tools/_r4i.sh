#!/bin/bash
O=gpurun_out/r4i; mkdir -p $O
python -m pytest tests/test_round2_gpu.py tests/test_multirank_gpu.py tests/test_parity_gpu.py tests/test_round3_gpu.py -q -k "testB or sharded or eight or shallow or edge or empty or large_launch or call_surfaces or lds_merged" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
python bench.py --batch-sweep > $O/batch_sweep.json 2>/dev/null
python - <<'P'
import json
b=json.load(open("gpurun_out/r4i/batch_sweep.json"))
for m,rows in b["sweep"].items(): print(m, [(r["pairs_per_call"], r["ms_per_call"], r["pairs_per_s"], r["of_large_batch_rate"]) for r in rows])
P
