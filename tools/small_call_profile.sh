#!/bin/bash
# GPU box: rocprofv3 kernel stats of 100 scoring calls at one call size per model -> gpurun_out/<tag>_<model><B>_kernel_stats.csv + a summary
# usage: tools/small_call_profile.sh <tag> "zk 256" "lxmert 256" ...
tag=$1; shift
out=$PWD/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
for spec in "$@"; do
  set -- $spec; m=$1; b=$2
  d=/tmp/scp_${m}${b}; rm -rf $d
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $d -o r -- python $out/../tools/small_batch.py $m $b 100 > $out/${tag}_${m}${b}.log 2>&1)
  f=$(find $d -name '*kernel_stats.csv' | head -1)
  cp $f $out/${tag}_${m}${b}_kernel_stats.csv
  python - $f $m $b <<'P' | tee -a $out/${tag}_summary.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows); n = sum(int(r['Calls']) for r in rows)
print("## %s B=%s: %d launches, %.3f ms GPU-busy per call (203 calls), %.1f launches per call" % (sys.argv[2], sys.argv[3], n, tot / 203e6, n / 203.0))
for r in rows[:12]:
    print("   %-86s calls %6s avg %8.1f us %5.1f%%" % (r['Name'][:86], r['Calls'], float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
P
  grep "per call" $out/${tag}_${m}${b}.log | tee -a $out/${tag}_summary.txt
done
