#!/bin/bash
O=gpurun_out/r4l; mkdir -p $O
python -m pytest tests/test_parity_gpu.py tests/test_multirank_gpu.py tests/test_round2_gpu.py -q -k "call_sizes or empty or sharded or eight or testB or shallow or edge or call_surfaces" > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
for m in zk lds lxmert; do for b in 1 5; do python tools/small_batch.py $m $b 200 2>/dev/null | tail -1; done; done | tee $O/small_batch.txt
