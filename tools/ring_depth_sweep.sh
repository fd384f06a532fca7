#!/bin/bash
# lab: LDS ring depth of the single-plane ping-pong GEMMs (mode 1 and fp8; 32 KiB stages): 3 / 4 / 5 slots
# usage (GPU box; make lab libmmscore_lab_n4.so libmmscore_lab_n5.so built): bash tools/ring_depth_sweep.sh
cd "$(dirname "$0")/.."
C=kddcup_2020_multimodalitiesrecall_2nd_place_amd/csrc
run() {
  timeout 300 python tools/bench_lab.py --steps 5 --warmup 2 --no-cpu --no-secondary "$@" 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print(r['value'], r['ms_per_step'], r['roofline']['achieved'])"
}
for lib in libmmscore_lab.so libmmscore_lab_n4.so libmmscore_lab_n5.so; do
  for p in 1 4; do
    echo "== $lib precision $p"; MMS_LAB_LIB=$C/$lib run --precision $p
  done
done
echo "== parity of the 5-slot build (fp8 GEMM vs numpy, precision-4 deviation test, shallow logits)"
MMS_LAB_LIB=$C/libmmscore_lab_n5.so timeout 600 python tools/pytest_lab.py tests/test_gemm_routes_gpu.py tests/test_model_routes_gpu.py -q -x -p no:cacheprovider -k "fp8" 2>&1 | tail -3
MMS_LAB_LIB=$C/libmmscore_lab_n4.so timeout 600 python tools/pytest_lab.py tests/test_gemm_routes_gpu.py tests/test_model_routes_gpu.py -q -x -p no:cacheprovider -k "fp8" 2>&1 | tail -3
