"""bench.py's multi-GPU entry (CPU): `--gpus N` started bare launches N ranks itself; the plan it would run is printed by
--dry-launch; a launcher whose WORLD_SIZE differs from --gpus is refused (no line with n_gpus != --gpus is ever printed)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, env=env, timeout=120)


def test_dry_launch_plans_one_rank_per_gpu():
    r = _run(["--gpus", "8", "--steps", "7", "--warmup", "2", "--dry-launch"])
    assert r.returncode == 0, r.stderr
    plan = json.loads(r.stdout.strip().splitlines()[-1])
    cmd = plan["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    tail = cmd[cmd.index(BENCH) + 1:]
    assert tail == ["--gpus", "8", "--steps", "7", "--warmup", "2"]          # the ranks run the same workload; --dry-launch is gone
    assert [k["RANK"] for k in plan["ranks"]] == [str(i) for i in range(8)]
    assert all(k["WORLD_SIZE"] == "8" and k["LOCAL_RANK"] == k["RANK"] and k["device"] == f"cuda:{k['RANK']}" for k in plan["ranks"])
    assert len({k["MASTER_PORT"] for k in plan["ranks"]}) == 1 and int(plan["ranks"][0]["MASTER_PORT"]) > 0


def test_dry_launch_single_gpu_runs_in_process():
    r = _run(["--gpus", "1", "--dry-launch"])
    assert r.returncode == 0, r.stderr
    plan = json.loads(r.stdout.strip().splitlines()[-1])
    assert plan["cmd"] is None and plan["ranks"][0]["WORLD_SIZE"] == "1"


def test_world_size_must_equal_gpus():
    r = _run(["--gpus", "4", "--steps", "1"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0
    assert "refusing" in (r.stderr + r.stdout)
    assert not any(l.startswith("{") for l in r.stdout.splitlines())           # no JSON line


def test_under_a_launcher_the_plan_is_this_process():
    r = _run(["--gpus", "2", "--dry-launch"], {"WORLD_SIZE": "2", "RANK": "1", "LOCAL_RANK": "1"})
    assert r.returncode == 0, r.stderr
    plan = json.loads(r.stdout.strip().splitlines()[-1])
    assert plan["cmd"] is None and plan["ranks"][0]["RANK"] == "1"
