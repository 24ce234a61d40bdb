"""optimize_cmaes_lockstep under torch.distributed on the CPU (gloo, world_size 2), with the per-rank GPU work stubbed out:
the runs are dealt over the ranks, ONE all_gather_object hands every rank every run's results -- and a rank whose runs fail
still reaches that collective, so that EVERY rank raises instead of the others waiting for the backend's timeout
(reference: the Pool over init groups, experiments/run_mpc_ord.py:83-90)."""
import os
import sys

import pytest
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import lockstep_failure_helper as helper  # noqa: E402


def _run(fail_rank, port_off):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 1500) + port_off
    procs = [ctx.Process(target=helper.worker, args=(r, 2, port, fail_rank, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return got


def test_runs_are_dealt_over_the_ranks_and_gathered():
    got = _run(fail_rank=-1, port_off=0)
    for rank, status, best, seeds, ranks in got:
        assert status == "ok" and ranks == 2
        assert best == [0.0, 1.0, 2.0, 3.0, 4.0] and seeds == [10, 11, 12, 13, 14]      # every rank holds every run's results


@pytest.mark.parametrize("fail_rank", [0, 1])
def test_a_failing_rank_makes_every_rank_raise_instead_of_hanging(fail_rank):
    got = _run(fail_rank=fail_rank, port_off=7 + fail_rank)
    for rank, status, msg, _, _ in got:
        assert status == "raised" and f"rank {fail_rank}: ValueError: boom on rank {fail_rank}" in msg, (rank, status, msg)
