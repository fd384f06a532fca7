#!/bin/bash
O=gpurun_out/r4f; mkdir -p $O
python -m pytest tests/test_round3_gpu.py -q -k "large_launch" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for m in zk lds lxmert; do for b in 1 256; do python tools/small_batch.py $m $b 100 2>/dev/null | tail -1; done; done | tee $O/small_batch.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_zk256 -o zk256 -- python $GRAFT_REPO_ROOT/tools/small_batch.py zk 256 100 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/prof_zk1 -o zk1 -- python $GRAFT_REPO_ROOT/tools/small_batch.py zk 1 100 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
for t in zk256 zk1; do f=$(find /tmp/prof_$t -name "*kernel_stats.csv" | head -1); cp $f $O/${t}_kernel_stats.csv; python - $f <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows); calls=sum(int(r["Calls"]) for r in rows)
print(sys.argv[1].split("/")[2], "kernels: %d launches, %.3f ms total -> per scoring call (103 calls): %.3f ms GPU-busy, %.1f launches" % (calls, tot/1e6, tot/1e6/103, calls/103))
for r in sorted(rows, key=lambda r:-float(r["TotalDurationNs"]))[:8]: print("   %-60s calls %6s avg %8.1f us  %5.1f%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["Percentage"])))
P
done
python bench.py --no-cpu --steps 3 > $O/bench_default.json 2>/dev/null; python - <<'P'
import json
r=json.load(open("gpurun_out/r4f/bench_default.json")); print(r["value"], json.dumps(r["roofline"])[:1500])
P
