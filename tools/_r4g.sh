#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r4g; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for t in "zk 256" "zk 1" "lxmert 256"; do set -- $t; tag=$1$2
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o $tag -- python $GRAFT_REPO_ROOT/tools/small_batch.py $1 $2 100 > $O/$tag.log 2>&1
f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1); cp "$f" $O/${tag}_kernel_stats.csv; tail -1 $O/$tag.log
python - "$f" <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows); calls=sum(int(r["Calls"]) for r in rows)
print("kernels: %d launches, %.3f ms total -> per scoring call (103 calls): %.3f ms GPU-busy, %.1f launches" % (calls, tot/1e6, tot/1e6/103, calls/103))
for r in sorted(rows, key=lambda r:-float(r["TotalDurationNs"]))[:14]: print("   %-70s calls %6s avg %8.1f us  %5.1f%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["Percentage"])))
P
done
