#!/bin/bash
# usage (on the GPU box): tools/pmc_gemm.sh <tag> N K act planes resid variant nsplit
# collects two SQ counter passes for one GEMM shape/variant into gpurun_out/pmc/<tag>_p{1,2}
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmc
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_MFMA \
  --output-format csv -d $R/gpurun_out/pmc -o ${tag}_p1 -- python $R/tools/gemm_one.py "$@" 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_LDS \
  --output-format csv -d $R/gpurun_out/pmc -o ${tag}_p2 -- python $R/tools/gemm_one.py "$@" 2 > /dev/null 2>&1
python - <<PY
import csv, glob, collections
for p in ("p1","p2"):
    f = glob.glob("$R/gpurun_out/pmc/${tag}_%s_counter_collection.csv" % p)
    if not f: print("no counter file for", p); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"][:60]
        if "gemm" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k, d in agg.items():
        print("${tag}", p, k)
        for c, v in sorted(d.items()): print("    %-28s %16.0f  (per dispatch %14.0f)" % (c, v, v / max(1, n[(k, c)])))
PY
