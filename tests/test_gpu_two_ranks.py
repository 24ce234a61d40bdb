"""HIP kernels + sharding + gather composed: bench.py as two ranks on the one GPU of the box (gloo rendezvous,
both ranks launching on the same card) must reproduce the one-rank generation bit for bit under strong
scaling, and the rank-emulation mode must run exactly one rank's block.

The ranks are child processes of a torch.distributed.run child of this test (bench.py's own launcher): two
GPU processes besides this one, within the box's limit."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run(args, timeout=600):
    """One bench.py run: the printed line stays under its cap (N > 1 too), and the FULL record -- the side file the line
    names -- is what the assertions below read."""
    import tempfile
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    with tempfile.TemporaryDirectory() as tmp:
        env["OCD_BENCH_DETAIL"] = os.path.join(tmp, "detail.json")
        r = subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, env=env, timeout=timeout)
        assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1 and len(lines[0]) < 8192, r.stdout[-2000:]
        line = json.loads(lines[0])
        assert line["detail"] == env["OCD_BENCH_DETAIL"]
        with open(line["detail"]) as f:
            full = json.load(f)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "dtype"):
        assert line[k] == full[k], k
    assert line["roofline"]["kernel_ms"] == full["roofline"]["kernel_ms"]
    return full


COMMON = ["--steps", "3", "--warmup", "1", "--no-extras", "--no-cpu-baseline"]


def test_two_ranks_reproduce_one_rank_under_strong_scaling():
    one = run(["--gpus", "1", "--scaling", "strong"] + COMMON)
    port = 29700 + os.getpid() % 200
    two = run(["--gpus", "2", "--backend", "gloo", "--scaling", "strong", "--master-port", str(port)] + COMMON)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert one["scaling"] == two["scaling"] == "strong"
    assert one["config"]["episodes_per_generation"] == two["config"]["episodes_per_generation"] == 2048
    assert two["config"]["episodes_per_gpu"] == 1024
    # the same 64 candidates, evaluated as two blocks on two processes and gathered: identical costs
    assert two["generation_cost_checksum"] == one["generation_cost_checksum"]


def test_weak_scaling_doubles_the_population():
    port = 29900 + os.getpid() % 90
    two = run(["--gpus", "2", "--backend", "gloo", "--master-port", str(port)] + COMMON)
    assert two["n_gpus"] == 2 and two["scaling"] == "weak"
    assert two["config"]["episodes_per_generation"] == 4096 and two["config"]["episodes_per_gpu"] == 2048


def test_rank_emulation_runs_one_block_of_config_4():
    d = run(["--config", "4", "--emulate-rank", "3/8"] + COMMON)
    assert d["n_gpus"] == 1 and d["scaling"] == "strong"
    assert d["config"]["episodes_per_generation"] == 16384 and d["config"]["episodes_per_gpu"] == 2048
    assert "rank 3 of a 8-way" in d["config"]["sharding"]
    # value counts this GPU's episodes only
    assert abs(d["value"] - 2048 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6


def test_rccl_runs_in_a_one_rank_group():
    """--force-collective: a one-rank RCCL ("nccl") process group, and every timed step goes through
    sharding.gather_returns on device tensors -- the code path N ranks run (bench.py init_group, sharding.py
    all_gather_into_tensor) executed on the one GPU of this box; same costs as the plain run."""
    plain = run(["--gpus", "1"] + COMMON)
    forced = run(["--gpus", "1", "--force-collective", "--master-port", str(30300 + os.getpid() % 200)] + COMMON)
    co = forced["collective"]
    assert co["ranks_seen"] == 1 and co["backend"].startswith("nccl") and co["in_timed_step"] is True
    assert co["all_gather_us"] > 0 and co["bytes_per_rank"] == 2048 * 4
    assert forced["roofline"]["launch"]["returns_written_to"] == "device memory"
    assert forced["generation_cost_checksum"] == plain["generation_cost_checksum"]
    assert plain["collective"] is None            # --no-extras: no collective anywhere in a plain one-GPU run


def test_two_ranks_carry_baseline_configs_4_and_5_with_parity():
    """at N > 1 the line holds BASELINE configs 4 / 5 AS STATED -- the fixed population split into
    N candidate blocks, the gather inside the timed step -- and every BASELINE block an untimed oracle parity object whose
    counts are summed over the ranks.  Two gloo ranks on the one card: 8 192 / 16 384 episodes per rank, the same
    generation costs as the whole population on one rank, every checked episode bit-identical to the oracle."""
    port = 30500 + os.getpid() % 200
    two = run(["--gpus", "2", "--backend", "gloo", "--master-port", str(port), "--steps", "3", "--warmup", "1",
               "--no-cpu-baseline"], timeout=900)
    assert two["n_gpus"] == 2 and two["scaling"] == "weak" and two["config"]["episodes_per_gpu"] == 2048
    c4, c5 = two["config4"], two["config5"]
    assert c4["scaling"] == c5["scaling"] == "strong"
    assert c4["episodes_per_generation"] == 16384 and c4["episodes_per_gpu"] == 8192 and "H=15" in c4["workload"]
    assert c5["episodes_per_generation"] == 32768 and c5["episodes_per_gpu"] == 16384 and "H=25" in c5["workload"]
    for b in (c4, c5):
        assert b["collective"]["ranks_seen"] == 2 and b["collective"]["in_timed_step"] is True
        assert len(b["roofline"]["launch"]["kernel_ms_per_rank"]) == 2 and min(b["roofline"]["launch"]["kernel_ms_per_rank"]) > 0
        assert abs(b["value"] - b["episodes_per_generation"] / (b["ms_per_step"] * 1e-3)) / b["value"] < 1e-6
    for b in (two, two["config2"], c4, c5):
        par = b["parity"]
        assert par["episodes_checked"] >= 128 and par["bitwise_equal"] == par["episodes_checked"] == par["within_1e-4_rel"]
        assert par["argmin_flips"] == 0 and par["plan_steps_checked"] > 0
    assert two["parity"]["ranks"] == c4["parity"]["ranks"] == 2 and two["config2"]["parity"]["ranks"] == 1
    # the reference's Pool over independent runs as one process per GPU: 28 runs dealt over the two ranks, each rank's 14 in
    # lockstep (one launch of 14 x 27 episodes per generation and rank), nothing exchanged until the end
    x28 = two["reference_h5_x28"]
    assert x28["runs"] == 28 and x28["ranks"] == 2 and x28["lockstep"] is True
    assert x28["episodes_per_generation_per_rank"] == [378, 378] and x28["episodes_per_generation"] == 756
    assert len(x28["cma_generation_ms_per_rank"]) == 2 and x28["cma_generation_ms"] >= max(x28["cma_generation_ms_per_rank"]) - 1e-9
    assert two["parity"]["episodes_checked"] >= 256 and c5["parity"]["episodes_checked"] >= 256
    # the same populations whole on one rank: identical generation costs
    for c, blk in ((4, c4), (5, c5)):
        one = run(["--gpus", "1", "--config", str(c), "--scaling", "strong", "--no-parity"] + COMMON)
        assert one["config"]["episodes_per_generation"] == blk["episodes_per_generation"]
        assert one["generation_cost_checksum"] == blk["generation_cost_checksum"], c
