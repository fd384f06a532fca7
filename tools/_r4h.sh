#!/bin/bash
O=gpurun_out/r4h; mkdir -p $O
python -m pytest tests/test_round2_gpu.py tests/test_multirank_gpu.py -q -x -k "testB or sharded or eight" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for m in zk lds lxmert; do for b in 256; do python tools/small_batch.py $m $b 100 2>/dev/null | tail -1; done; done | tee $O/small_batch2.txt
