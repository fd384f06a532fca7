#!/bin/bash
# usage (GPU box): tools/r04_profiles.sh <tag>   -- the evidence set behind DESIGN.md section 6: bench lines, rocprofv3 kernel stats of the
# same commands, PMC HBM traffic (separate FETCH_SIZE / WRITE_SIZE passes) for the three models.  Everything lands in gpurun_out/<tag>/.
R=${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-r04}
O=$R/gpurun_out/$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for m in zk lds lxmert; do
  python $R/bench.py --model $m --no-secondary $([ $m = zk ] || echo --no-cpu) > $O/bench_$m.json 2> $O/bench_$m.err
done
python $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
python $R/bench.py --model ensemble --no-cpu --no-secondary > $O/bench_ensemble.json 2> $O/bench_ensemble.err
python $R/bench.py --precision 4 --no-cpu --no-secondary > $O/bench_zk_fp8.json 2> $O/bench_zk_fp8.err
python $R/bench.py --model ensemble --precision 4 --no-cpu --no-secondary > $O/bench_ensemble_fp8.json 2> $O/bench_ensemble_fp8.err
python $R/bench.py --workload valid --no-cpu --no-secondary > $O/bench_zk_valid.json 2> $O/bench_zk_valid.err
python $R/bench.py --fuse-ln 0 --no-cpu --no-secondary > $O/bench_zk_fuseln0.json 2> $O/bench_zk_fuseln0.err      # the two-kernel LayerNorm route (round-3 launch plan)
python $R/bench.py --fuse-ln 1 --no-cpu --no-secondary > $O/bench_zk_fuseln1.json 2> $O/bench_zk_fuseln1.err      # attention-output projections only
for f in 0 1; do   # mms_config.fuse_attention off / exact-fp32 attention (the library default, which every line above runs, is 2: split-bf16 attention MFMAs)
  python $R/bench.py --fuse-attn $f --no-cpu --no-secondary > $O/bench_zk_fuseattn$f.json 2> $O/bench_zk_fuseattn$f.err
done
python $R/bench.py --precision 3 --fp32-weights --no-cpu --no-secondary > $O/bench_zk_mode3.json 2> $O/bench_zk_mode3.err
MMS_BENCH_SHARE_GPU=1 MMS_BENCH_BACKEND=gloo python $R/bench.py --gpus 2 --no-cpu > $O/bench_zk_2ranks_shared_gpu.json 2> $O/bench_zk_2ranks_shared_gpu.err
python $R/bench.py --workload testB --no-cpu --no-secondary > $O/bench_zk_testB.json 2> $O/bench_zk_testB.err
for m in zk ensemble; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$m -o $m -- python $R/bench.py --model $m --steps 3 --warmup 1 --no-cpu --no-secondary > $O/prof_$m.log 2>&1
  cp $O/prof_$m/*kernel_stats.csv $O/bench_${m}_kernel_stats.csv 2>/dev/null || find $O/prof_$m -name "*kernel_stats.csv" -exec cp {} $O/bench_${m}_kernel_stats.csv \;
  rm -rf $O/prof_$m
done
for m in zk lds lxmert; do
  $R/tools/pmc_traffic.sh $m --model $m --no-secondary > $O/pmc_traffic_$m.log 2>&1
  cp $R/gpurun_out/pmc/${m}_traffic.json $O/pmc_traffic_$m.json
done
$R/tools/pmc_mfma_busy.sh zk > $O/mfma_busy_zk.log 2>&1; cp $R/gpurun_out/pmc/zk_mfma_busy.json $O/mfma_busy_zk.json 2>/dev/null
rm -rf $R/gpurun_out/pmc
python $R/bench.py --batch-sweep > $O/batch_sweep.json 2> $O/batch_sweep.err
ls -la $O
