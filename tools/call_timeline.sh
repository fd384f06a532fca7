#!/bin/bash
# GPU box: the launch sequence of ONE scoring call (rocprofv3 --kernel-trace): kernel, grid, duration, gap to the previous kernel's end.
# usage: tools/call_timeline.sh <model> <pairs> [out]      (prints the last call of 12)
m=$1; b=$2; out=${3:-/dev/stdout}
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
d=/tmp/ctl_$m$b; rm -rf $d
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $d -o t -- python $R/tools/small_batch.py $m $b 4 > /dev/null 2>&1)
f=$(find $d -name '*kernel_trace.csv' | head -1)
python - $f $m $b > $out <<'P'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
# a call = the launches between two of its first kernels; take the last complete one
per = len(rows) // 11          # 3 warm-up + 4 + 4 calls
call = rows[-per:]
t_prev = None
print("# %s, %s pairs: %d launches in the call, %.3f ms from first start to last end" % (sys.argv[2], sys.argv[3], len(call), (int(call[-1]['End_Timestamp']) - int(call[0]['Start_Timestamp'])) / 1e6))
busy = 0
for r in call:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = 0 if t_prev is None else (s - t_prev) / 1e3
    busy += (e - s) / 1e3
    print("%8.1f us  gap %6.1f  grid %7s x %4s  %s" % ((e - s) / 1e3, gap, r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size', '?'), r['Workgroup_Size_X'] if 'Workgroup_Size_X' in r else r.get('Workgroup_Size', '?'), r['Kernel_Name'][:110]))
    t_prev = e if t_prev is None else max(t_prev, e)
print("# busy %.1f us" % busy)
P
