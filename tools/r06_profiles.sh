#!/bin/bash
# usage (GPU box): tools/r06_profiles.sh <tag>   -- the evidence set behind DESIGN.md section 6 (round 6): bench lines, rocprofv3 kernel stats of the same commands,
# PMC HBM traffic (separate FETCH_SIZE / WRITE_SIZE passes; a third pass splits the L2's memory-side reads into DRAM-bound and not), matrix-pipe occupancy, the
# per-tile timeline of the fused QKV + attention kernel, the call-size sweep.  Everything lands in gpurun_out/<tag>/; copy what is cited into profiles/.
R=${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-r06}
O=$R/gpurun_out/$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for m in zk lds lxmert; do
  python $R/bench.py --model $m --no-secondary $([ $m = zk ] || echo --no-cpu) > $O/bench_$m.json 2> $O/bench_$m.err
done
( time python $R/bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_wall.txt
python $R/bench.py --model ensemble --no-cpu --no-secondary > $O/bench_ensemble.json 2> $O/bench_ensemble.err
python $R/bench.py --precision 4 --no-cpu --no-secondary > $O/bench_zk_fp8.json 2> $O/bench_zk_fp8.err
python $R/bench.py --model ensemble --precision 4 --no-cpu --no-secondary > $O/bench_ensemble_fp8.json 2> $O/bench_ensemble_fp8.err
python $R/bench.py --workload valid --no-cpu --no-secondary > $O/bench_zk_valid.json 2> $O/bench_zk_valid.err
python $R/bench.py --workload testB --no-cpu --no-secondary > $O/bench_zk_testB.json 2> $O/bench_zk_testB.err
python $R/bench.py --fuse-ln 0 --no-cpu --no-secondary > $O/bench_zk_fuseln0.json 2> $O/bench_zk_fuseln0.err
for m in zk lxmert lds; do for f in 0 1; do   # same-box A/B of mms_config.fuse_attention: off / exact-fp32 attention arithmetic (the lines above run the library default, 2)
  python $R/bench.py --model $m --fuse-attn $f --no-cpu --no-secondary > $O/bench_${m}_fuseattn$f.json 2> $O/bench_${m}_fuseattn$f.err
done; done
python $R/bench.py --precision 3 --fp32-weights --no-cpu --no-secondary > $O/bench_zk_mode3.json 2> $O/bench_zk_mode3.err
python $R/bench.py --precision 3 --fp32-weights --fuse-attn 2 --no-cpu --no-secondary > $O/bench_zk_mode3_fuseattn2.json 2> $O/bench_zk_mode3_fuseattn2.err
MMS_BENCH_SHARE_GPU=1 MMS_BENCH_BACKEND=gloo python $R/bench.py --gpus 2 --no-cpu > $O/bench_zk_2ranks_shared_gpu.json 2> $O/bench_zk_2ranks_shared_gpu.err
for m in zk lxmert ensemble; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$m -o $m -- python $R/bench.py --model $m --steps 3 --warmup 1 --no-cpu --no-secondary > $O/prof_$m.log 2>&1
  find $O/prof_$m -name "*kernel_stats.csv" -exec cp {} $O/bench_${m}_kernel_stats.csv \;
  rm -rf $O/prof_$m
done
for m in zk lds lxmert; do
  $R/tools/pmc_traffic.sh $m --model $m --no-secondary > $O/pmc_traffic_$m.log 2>&1
  cp $R/gpurun_out/pmc/${m}_traffic.json $O/pmc_traffic_$m.json
done
# where the fused kernel's L2 misses go: all memory-side read requests vs the ones bound for DRAM (the rest are served by the Infinity Cache)
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $R/gpurun_out/pmc -o zk_ea -- python $R/bench.py --steps 1 --warmup 0 --no-cpu --no-secondary > $O/pmc_ea_zk.log 2>&1
python - <<PY > $O/pmc_ea_zk.txt 2>&1
import csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open("$R/gpurun_out/pmc/zk_ea_counter_collection.csv")):
    k = "qkv_attn" if "qkv_attn" in r["Kernel_Name"] else "gemm_ln" if ("gemm_pp_kernel" in r["Kernel_Name"] and "true, true>" in r["Kernel_Name"].replace("1, 1>", "true, true>")) else "gemm" if "gemm" in r["Kernel_Name"] else "other"
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
print("# memory-side read requests of the L2 per launch (TCC_EA0_RDREQ: 32-B and 64-B requests; x 64 B, the 32-B ones x 32 B) and the share bound for DRAM; zk, 1 pass")
for k, c in agg.items():
    d = n[(k, "TCC_EA0_RDREQ_sum")]
    rq, dr, r32 = c["TCC_EA0_RDREQ_sum"] / d, c["TCC_EA0_RDREQ_DRAM_sum"] / d, c["TCC_EA0_RDREQ_32B_sum"] / d
    print("%-10s launches %4d  requests/launch %.3e (%.2f GB at 64 B, 32-B requests %.1f %%)  DRAM-bound %.3e = %.1f %%" % (k, d, rq, ((rq - r32) * 64 + r32 * 32) / 1e9, 100 * r32 / max(rq, 1), dr, 100 * dr / max(rq, 1)))
PY
$R/tools/pmc_mfma_busy.sh zk > $O/mfma_busy_zk.log 2>&1; cp $R/gpurun_out/pmc/zk_mfma_busy.json $O/mfma_busy_zk.json 2>/dev/null
$R/tools/pmc_mfma_busy.sh lxmert --model lxmert > $O/mfma_busy_lxmert.log 2>&1; cp $R/gpurun_out/pmc/lxmert_mfma_busy.json $O/mfma_busy_lxmert.json 2>/dev/null
rm -rf $R/gpurun_out/pmc
for fl in 0 16 32; do      # per-tile timeline of qkv_attn2_kernel (lab build): all work / no attention / no staging stores
  echo "== MMS_QA_FLAGS=$fl" >> $O/qa_trace.txt
  QA_FUSE=2 MMS_QA_FLAGS=$fl python $R/tools/qa_trace.py 2>&1 | grep "^tile [2-5]" >> $O/qa_trace.txt
done
echo "== 4 x 2 wave layout of the same route (MMS_QA_LAYOUT=42), all work" >> $O/qa_trace.txt
QA_FUSE=2 MMS_QA_LAYOUT=42 python $R/tools/qa_trace.py 2>&1 | grep -A1 "^tile [2-5]" >> $O/qa_trace.txt
python $R/bench.py --batch-sweep > $O/batch_sweep.json 2> $O/batch_sweep.err
# host side (round 6): TSV file -> scores, single model and BASELINE.json config 5; featurizer alone by thread count; eight ranks' featurizers on the one host
python $R/tools/e2e_tsv_bench.py 150000 zk 0,32 2>&1 | grep -v "amdgpu.ids\|Warning\|return {k" > $O/e2e_tsv.txt
python $R/tools/e2e_tsv_bench.py 120000 ensemble 2>&1 | grep -v "amdgpu.ids\|Warning\|return {k" >> $O/e2e_tsv.txt
python $R/tools/feat_bench.py --records 60000 --threads 8,16,32,64,128 --tiers 2 --pinned 2>&1 | grep -v amdgpu.ids > $O/feat_bench.txt
for r in 8 4 2; do python $R/tools/feat_bench.py --records 240000 --ranks $r --pinned 2>&1 | grep -v amdgpu.ids >> $O/feat_bench.txt; done
(cd $R && python -m pytest tests/test_shard_ranking_gpu.py tests/test_kdd_dropin.py tests/test_rccl_world1_gpu.py -m gpu -q -s 2>&1 | grep "^\[" > $O/round6_test_prints.txt); (cd /tmp)
python $R/tools/soak_small.py 20 > $O/soak_small_calls.txt 2>&1
(cd $R && bash tools/small_call_profile.sh $T/kmix "zk 256" "lds 256" "lxmert 256" "zk 1" "lds 5" > /dev/null 2>&1); cp $O/kmix_summary.txt $O/small_call_kernel_mix.txt 2>/dev/null; (cd /tmp)
cd $R && timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/pytest_gpu_tail.txt
ls -la $O
