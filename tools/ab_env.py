"""A/B of lab-build environment knobs on the bench workload: python tools/ab_env.py model VAR=v1,v2,... [VAR2=...]  (lab library)."""
import itertools, os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
model = sys.argv[1]
knobs = [a.split("=") for a in sys.argv[2:]]
names = [k for k, _ in knobs]
for combo in itertools.product(*[v.split(",") for _, v in knobs]):
    env = dict(os.environ, MMS_USE_LAB="1", **dict(zip(names, combo)))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_lab.py"), "--model", model, "--no-cpu", "--no-secondary", "--steps", "4"],
                         env=env, capture_output=True, text=True)
    try:
        d = json.loads(out.stdout.strip().splitlines()[-1])
        print(dict(zip(names, combo)), "%.1f pairs/s  %.2f ms  gemm %.1f TF" % (d["value"], d["ms_per_step"], d["roofline"]["achieved"]), flush=True)
    except Exception:
        print(dict(zip(names, combo)), "FAILED", out.stderr[-400:], flush=True)
