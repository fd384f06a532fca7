#!/bin/bash
# usage (GPU box): tools/pmc_traffic.sh <tag> [bench args]  -- HBM traffic of the GEMM kernels over one bench pass.
# Two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) as MI355X_MICROARCH.md prescribes; the gfx950 FETCH_SIZE
# x2 correction for wide coalesced reads is applied in the summary (WRITE_SIZE is reported uncorrected).
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmc
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc -o ${tag}_$c -- python $R/bench.py --steps 1 --warmup 0 --no-cpu --no-secondary "$@" > $R/gpurun_out/pmc/${tag}_$c.log 2>&1
done
python - <<PY
import csv, json, collections
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open("$R/gpurun_out/pmc/${tag}_%s_counter_collection.csv" % c)):
        k = "qkv_attn" if "qkv_attn" in r["Kernel_Name"] else "gemm_ln" if ("gemm_pp_kernel" in r["Kernel_Name"] and "true, true>" in r["Kernel_Name"].replace("1, 1>", "true, true>")) else "gemm" if "gemm" in r["Kernel_Name"] else ("attn" if "attn" in r["Kernel_Name"] else ("ln" if "k_ln" in r["Kernel_Name"] else "other"))
        agg[k] += float(r["Counter_Value"]); n[k] += 1
    out[c] = {k: {"sum_kb": agg[k], "dispatches": n[k]} for k in agg}
g_f, g_w = dict(out["FETCH_SIZE"]["gemm"]), dict(out["WRITE_SIZE"]["gemm"])
ln_f, ln_w = out["FETCH_SIZE"].get("gemm_ln"), out["WRITE_SIZE"].get("gemm_ln")
if ln_f:      # the LayerNorm-fused projections are GEMM launches of the bench's roofline entry too: pooled here, and reported apart below
    for d, e in ((g_f, ln_f), (g_w, ln_w)):
        d["sum_kb"] += e["sum_kb"]; d["dispatches"] += e["dispatches"]
res = {"tag": "$tag", "gemm_launches": g_f["dispatches"],
       "fetch_bytes_per_launch_raw": g_f["sum_kb"] * 1024 / g_f["dispatches"],
       "fetch_bytes_per_launch_x2_corrected": 2 * g_f["sum_kb"] * 1024 / g_f["dispatches"],
       "write_bytes_per_launch_uncalibrated": g_w["sum_kb"] * 1024 / g_w["dispatches"],
       "all": out}
res["hbm_bytes_per_launch"] = res["fetch_bytes_per_launch_x2_corrected"] + res["write_bytes_per_launch_uncalibrated"]
if ln_f:
    res["ln_fused_launches"] = ln_f["dispatches"]
    res["ln_fused_hbm_bytes_per_launch"] = 2 * ln_f["sum_kb"] * 1024 / ln_f["dispatches"] + ln_w["sum_kb"] * 1024 / ln_w["dispatches"]
    res["ln_fused_fetch_x2_bytes_per_launch"] = 2 * ln_f["sum_kb"] * 1024 / ln_f["dispatches"]
if "qkv_attn" in out["FETCH_SIZE"]:     # the fused QKV + attention launches (mms_config.fuse_attention), same corrections
    q_f, q_w = out["FETCH_SIZE"]["qkv_attn"], out["WRITE_SIZE"]["qkv_attn"]
    res["fused_launches"] = q_f["dispatches"]
    res["fused_hbm_bytes_per_launch"] = 2 * q_f["sum_kb"] * 1024 / q_f["dispatches"] + q_w["sum_kb"] * 1024 / q_w["dispatches"]
for line in open("$R/gpurun_out/pmc/${tag}_FETCH_SIZE.log"):     # which route the measured pass took (bench.py's own line)
    if line.startswith("{"):
        res["fuse_attention"] = json.loads(line)["config"].get("fuse_attention", 0)
        res["fuse_layernorm"] = json.loads(line)["config"].get("fuse_layernorm", 0)
json.dump(res, open("$R/gpurun_out/pmc/${tag}_traffic.json", "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "all"}))
PY
