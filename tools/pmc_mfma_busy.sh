#!/bin/bash
# usage (GPU box): tools/pmc_mfma_busy.sh <tag> [bench args] -- matrix-pipe occupancy of the GEMM launches of one bench pass (product library):
# SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x the launch's shader cycles), shader cycles = GRBM_GUI_ACTIVE / 8 XCDs of the same dispatch.
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmc
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc -o ${tag}_mfma -- python $R/bench.py --steps 1 --warmup 1 --no-cpu --no-secondary "$@" > $R/gpurun_out/pmc/${tag}_mfma.log 2>&1
python - <<PY
import csv, json, collections
dur = {}
for r in csv.DictReader(open("$R/gpurun_out/pmc/${tag}_mfma_kernel_trace.csv")):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
cnt = collections.defaultdict(dict)
for r in csv.DictReader(open("$R/gpurun_out/pmc/${tag}_mfma_counter_collection.csv")):
    cnt[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for d, c in cnt.items():
    if d not in dur or ("gemm_pp_kernel" not in dur[d][1] and "qkv_attn" not in dur[d][1]) or dur[d][0] < 200000: continue
    name = dur[d][1].split("(")[0].replace("void ", "")
    a = agg[name]
    for k, v in c.items(): a[k] += v
    a["ns"] += dur[d][0]; a["n"] += 1
out = {}
for k, a in agg.items():
    cyc = a["GRBM_GUI_ACTIVE"] / 8.0
    out[k] = {"dispatches": int(a["n"]), "ms": a["ns"] / 1e6, "shader_ghz": cyc / a["ns"],
              "mfma_busy_frac": a["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc) if cyc else None,
              "busy_cycles_per_mfma": a["SQ_VALU_MFMA_BUSY_CYCLES"] / a["SQ_INSTS_MFMA"] if a["SQ_INSTS_MFMA"] else None}
json.dump(out, open("$R/gpurun_out/pmc/${tag}_mfma_busy.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
