for m in zk lds lxmert; do python bench.py --no-cpu --model $m --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m', d['value'], d['roofline'])"; done
MMS_GEMM_VARIANT=4 python bench.py --no-cpu --model zk --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('zk v4', d['value'], d['roofline']['achieved'])"
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.txt 2>&1; tail -3 gpurun_out/gpu_tests.txt
