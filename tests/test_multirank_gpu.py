"""GPU suite: the N > 1 path with the HIP scorer on a single-GPU box (ranks share the device, gloo exchange) -- SURVEY.md section 8(e),
BASELINE.json config 4 (testB-like ragged shards) -- and bench.py's own rank spawning."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_hip_scoring_equals_single_rank_bitwise(world, tmp_path):
    out = tmp_path / "res.json"
    port = _port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "multirank_worker.py"), str(out)], env=env, cwd=ROOT))
    for p in procs:
        assert p.wait(timeout=900) == 0
    res = json.load(open(out))
    assert res["ok"] is True and sum(res["counts"]) == res["pairs"]


def test_ranks_decode_and_score_only_their_query_block_of_a_shared_tsv(tmp_path):
    """File-fed N > 1 path: three ranks read ONE TSV file, each decodes (libmmfeat) and scores its contiguous query block only; the gathered
    (query id, product id, score) triples equal rank 0's pass over the whole file."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_featurizer_native import _random_lines
    lines = [l for l in _random_lines(150, 77) if int(l.split("\t")[3]) > 0 and all(ord(c) < 128 for c in l.split("\t")[7])]
    path = tmp_path / "shared.tsv"
    path.write_bytes(("product_id\tx\n" + "\n".join(lines) + "\n").encode("utf-8"))
    out = tmp_path / "res_tsv.json"
    port = _port()
    procs = []
    for r in range(3):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="3", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "multirank_worker.py"), str(out), "tsv", str(path)], env=env, cwd=ROOT))
    for p in procs:
        assert p.wait(timeout=900) == 0
    res = json.load(open(out))
    assert res["ok"] is True and res["pairs"] == len(lines) and len(res["counts"]) == 3, res


def test_eight_ranks_fused_ensemble_gather_and_rank0_post_processing(tmp_path):
    """The N = 8 code path end to end on ONE device (no 8-GPU node is available to the builder; no RCCL run exists, DESIGN.md
    section 7): 8 processes, contiguous query blocks of fewer than 3 queries each, the fused three-model scorer per rank, one
    all-gather with static counts, then main.py's global uniqueness filter + top-5 on rank 0 -- the same submission rows as one rank doing it all
    (scores equal to fp32 round-off: 3-query shards run lxmert's distinct-query stage in the tiny-launch regime, the whole job does not)."""
    out = tmp_path / "res8.json"
    port = _port()
    procs = []
    for r in range(8):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="8", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "multirank_worker.py"), str(out), "ensemble"], env=env, cwd=ROOT))
    for p in procs:
        assert p.wait(timeout=1500) == 0
    res = json.load(open(out))
    assert res["ok"] is True and sum(res["counts"]) == res["pairs"] and len(res["counts"]) == 8 and 0 < res["queries"] <= 21, res


def test_bench_gpus8_reports_weak_and_strong_in_one_line():
    """`python bench.py --gpus 8` (self-spawned, 8 ranks sharing device 0 over gloo here; one GPU each over RCCL on the driver's node):
    ONE JSON line carrying the weak-scaling value (N x Q queries) and, under "strong", the metric's own Q-query job cut over the 8 ranks."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(MMS_BENCH_SHARE_GPU="1", MMS_BENCH_BACKEND="gloo")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--queries", "24",
                          "--cands", "10"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=2400)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["config"]["pairs_total"] == 8 * 24 * 10 and d["value"] > 0
    st = d["strong"]
    assert st["scaling"] == "strong" and st["pairs_total"] == 24 * 10 and sum(st["pairs_per_gpu"]) == 240 and len(st["pairs_per_gpu"]) == 8
    assert st["value"] > 0 and abs(st["value"] - 240 * d["steps"] / (st["ms_per_step"] * 1e-3 * d["steps"])) / st["value"] < 1e-3
    # the line says where its ranks ran: 8 entries, here all on device 0 (test mode, declared as such)
    rk = d["ranks"]
    assert [r["rank"] for r in rk] == list(range(8)) and all(r["hip_device"] == 0 and r["pci_bus_id"] and r["name"] for r in rk)
    assert len({r["pid"] for r in rk}) == 8
    di = d["distributed"]
    assert di["backend"] == "gloo" and di["shared_device_test_mode"] is True and di["distinct_devices"] == 1


def test_bench_refuses_nccl_ranks_on_one_device():
    """Two RCCL ranks that resolve to ONE device (a launcher that did not give each rank its own GPU) must fail loudly, not print a
    scaling number.  Here: 2 self-spawned ranks whose HIP_VISIBLE_DEVICES shows both the same single GPU."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MMS_BENCH_SHARE_GPU")}
    env.update(MMS_BENCH_BACKEND="gloo", MMS_BENCH_FORCE_LOCAL0="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--queries", "4",
                          "--cands", "4"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert out.returncode != 0 and "distinct devices" in out.stderr, (out.returncode, out.stderr[-1500:])
    assert not [l for l in out.stdout.splitlines() if l.strip().startswith("{")]


def test_bench_gpus2_spawns_two_ranks_itself():
    """`python bench.py --gpus 2` with NO launcher and no WORLD_SIZE: the bench starts its own ranks.  On this
    one-GPU box the ranks share device 0 and exchange over gloo; on the driver's node they get one GPU each and RCCL."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(MMS_BENCH_SHARE_GPU="1", MMS_BENCH_BACKEND="gloo")
    for wl, scaling in (("bench", "weak"), ("testB", "strong")):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--queries", "40",
                              "--workload", wl], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
        assert out.returncode == 0, out.stderr[-3000:]
        lines = [l for l in out.stdout.splitlines() if l.strip()]
        assert len(lines) == 1, lines
        d = json.loads(lines[0])
        assert d["n_gpus"] == 2 and d["scaling"] == scaling and d["value"] > 0
        total = d["config"]["pairs_total"]
        assert abs(d["value"] - total * d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"])) / d["value"] < 1e-3
        assert "cpu_baseline" not in d                      # rank 0 at N = 1 only
        if wl == "bench":
            assert total == 2 * 40 * 30
    # a --gpus that contradicts the launcher's WORLD_SIZE is an error, not a silent 1-rank run
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], cwd=ROOT,
                         env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0 and "contradicts" in (bad.stderr + bad.stdout)


def test_bench_under_torch_distributed_run():
    """The driver's N > 1 launch line: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...` -- ranks come from the launcher's environment (here: 2 ranks on one device over gloo)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(MMS_BENCH_SHARE_GPU="1", MMS_BENCH_BACKEND="gloo")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--queries", "30"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["pairs_total"] == 2 * 30 * 30 and d["value"] > 0
