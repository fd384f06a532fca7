#!/bin/bash
# lab: wave grid of the ping-pong GEMM, 2(M) x 4(N) (libmmscore_lab.so) against 4(M) x 2(N) (libmmscore_lab_x.so built with
# LABX="-DMMS_PP_WN=2"): value, ms / step, GEMM TFLOP/s per precision mode, then parity tests on the WN=2 build
cd "$(dirname "$0")/.."
C=kddcup_2020_multimodalitiesrecall_2nd_place_amd/csrc
run() {
  timeout 300 python tools/bench_lab.py --steps 5 --warmup 2 --no-cpu --no-secondary "$@" 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print(r['value'], r['ms_per_step'], r['roofline']['achieved'])"
}
for rep in 1 2; do
for lib in libmmscore_lab.so libmmscore_lab_x.so; do
  for p in 2 1 4; do
    echo "== $lib precision $p"; MMS_LAB_LIB=$C/$lib run --precision $p
  done
done
done
echo "== $lib lds"; MMS_LAB_LIB=$C/libmmscore_lab.so run --model lds; MMS_LAB_LIB=$C/libmmscore_lab_x.so run --model lds
echo "== parity (WN=2 build)"
MMS_LAB_LIB=$C/libmmscore_lab_x.so timeout 900 python tools/pytest_lab.py tests/test_parity_gpu.py tests/test_model_routes_gpu.py tests/test_gemm_routes_gpu.py -q -x -p no:cacheprovider -k "full_size or full_depth or testB or fp8 or dense_and or stagewise" 2>&1 | tail -4
