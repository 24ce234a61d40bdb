"""bench.py --gpus N starts its N ranks itself (no external launcher), refuses a world size that does
not match --gpus, and the parent never touches the GPU: rehearsed here over gloo without a GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run(args, env_extra=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, env=env, timeout=timeout)


def last_json(out):
    lines = [l for l in out.strip().splitlines() if l.startswith("{")]
    return json.loads(lines[-1])


def test_gpus_2_spawns_two_ranks():
    port = 29600 + os.getpid() % 300
    r = run(["--gpus", "2", "--plumbing-check", "--master-port", str(port)])
    assert r.returncode == 0, r.stderr[-2000:]
    rec = last_json(r.stdout)
    assert rec["plumbing"] is True and rec["n_gpus"] == 2


def test_single_rank_needs_no_launcher():
    r = run(["--gpus", "1", "--plumbing-check"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert last_json(r.stdout)["n_gpus"] == 1


def test_world_size_mismatch_fails():
    # launched as ONE rank of an external launcher while asking for 4 GPUs: refuse, do not report n_gpus 1
    r = run(["--gpus", "4", "--plumbing-check"], env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0
    assert "world size 1" in (r.stderr + r.stdout)


def test_strong_scaling_and_rank_emulation_flags():
    port = 29650 + os.getpid() % 300
    r = run(["--gpus", "2", "--scaling", "strong", "--plumbing-check", "--master-port", str(port)])
    assert r.returncode == 0, r.stderr[-2000:]
    rec = last_json(r.stdout)
    assert rec["n_gpus"] == 2 and rec["scaling"] == "strong"
    r = run(["--emulate-rank", "3/8", "--config", "4", "--plumbing-check"])
    assert r.returncode == 0 and last_json(r.stdout)["emulate_rank"] == "3/8"
    for bad in ("8/8", "x", "-1/4"):
        r = run(["--emulate-rank", bad, "--plumbing-check"])
        assert r.returncode != 0 and "emulate-rank" in (r.stderr + r.stdout)
    # one rank's block is emulated on ONE GPU: not together with several ranks
    r = run(["--gpus", "2", "--emulate-rank", "0/8", "--plumbing-check"], env_extra={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "one GPU" in (r.stderr + r.stdout)
