#!/bin/bash
# GPU box: same-box A/B of two builds of libmmscore.so through bench.py (the library file is swapped between runs, alternating).
# usage: tools/ab_bench.sh <other libmmscore.so> <rounds> <bench.py args...>     -> "A <pairs/s>" = the tree's build, "B <pairs/s>" = the other one
set -e
other=$1; rounds=$2; shift 2
C=kddcup_2020_multimodalitiesrecall_2nd_place_amd/csrc
cp $C/libmmscore.so /tmp/libmmscore_A.so
for r in $(seq 1 $rounds); do
  for which in A B; do
    if [ $which = A ]; then cp /tmp/libmmscore_A.so $C/libmmscore.so; else cp $other $C/libmmscore.so; fi
    v=$(python bench.py --no-secondary --no-cpu "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline'].get('fused_qkv_attention',{}).get('achieved'))")
    echo "$which $v"
  done
done
cp /tmp/libmmscore_A.so $C/libmmscore.so
