#!/bin/bash
# usage (GPU box): tools/pmc_clock.sh <tag> [lab] [bench args] -- shader clock the GEMM launches actually run at:
# GRBM_GUI_ACTIVE (busy cycles of the graphics clock domain) per dispatch / that dispatch's duration from the kernel trace.
# "lab" as second argument runs the lab library (MMS_PP_GRID etc. honoured).
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
entry=$R/bench.py
if [ "$1" = "lab" ]; then entry=$R/tools/bench_lab.py; shift; fi
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmc
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $R/gpurun_out/pmc -o ${tag}_clk -- python $entry --steps 1 --warmup 1 --no-cpu --no-secondary "$@" > $R/gpurun_out/pmc/${tag}_clk.log 2>&1
python - <<PY
import csv, json, collections
dur = {}
for r in csv.DictReader(open("$R/gpurun_out/pmc/${tag}_clk_kernel_trace.csv")):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
cyc = collections.defaultdict(dict)
for r in csv.DictReader(open("$R/gpurun_out/pmc/${tag}_clk_counter_collection.csv")):
    cyc[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
agg = collections.defaultdict(lambda: [0.0, 0.0, 0.0, 0])
for d, c in cyc.items():
    if d not in dur: continue
    ns, name = dur[d]
    k = "gemm_pp" if "gemm_pp_kernel" in name else ("gemm_other" if "gemm" in name else ("attn" if "attn" in name else ("ln" if "k_ln" in name else "other")))
    if ns < 200000: continue          # short launches: timestamp granularity
    a = agg[k]; a[0] += c.get("GRBM_GUI_ACTIVE", 0.0); a[1] += c.get("GRBM_COUNT", 0.0); a[2] += ns; a[3] += 1
out = {k: {"dispatches": v[3], "ms": v[2] / 1e6, "gui_active_cycles_per_ns": v[0] / v[2], "grbm_count_cycles_per_ns": v[1] / v[2]} for k, v in agg.items()}
json.dump(out, open("$R/gpurun_out/pmc/${tag}_clock.json", "w"), indent=1)
print(json.dumps(out))
PY
