#!/bin/bash
# lab: FFN-up / FFN-down in row blocks over one reused intermediate buffer (does the Infinity Cache keep it?)
# usage (GPU box, lab library built): bash tools/ffn_block_sweep.sh > gpurun_out/ffn_block_sweep.txt
cd "$(dirname "$0")/.."
run() {
  timeout 300 python tools/bench_lab.py --steps 5 --warmup 2 --no-cpu --no-secondary "$@" 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print(r['value'], r['ms_per_step'], r['roofline']['achieved'], r['roofline']['launches'])"
}
echo "== baseline"; run
for b in 10880 21760 43520 65280 87040 130560; do
  echo "== MMS_FFN_BLOCK=$b"; MMS_FFN_BLOCK=$b MMS_FFN_LIVE=460000 run
done
echo "== lds baseline"; run --model lds
echo "== lds MMS_FFN_BLOCK=21760"; MMS_FFN_BLOCK=21760 run --model lds
