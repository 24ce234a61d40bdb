"""N>1 path on CPU: two gloo ranks shard the episode batch, all-gather the returns once, and every
rank ends up with exactly the single-process result (here computed by the CPU oracle)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from l4dc_mpc_ocd_amd import scenarios, sharding

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("P,W", [(16, 2), (7, 2), (5, 4), (3, 8), (1, 2)])
def test_candidate_blocks_partition(P, W):
    blocks = [sharding.candidate_block(P, W, r) for r in range(W)]
    assert blocks[0][0] == 0 and blocks[-1][1] == P
    assert all(blocks[i][1] == blocks[i + 1][0] for i in range(W - 1))
    sizes = [b - a for a, b in blocks]
    assert max(sizes) - min(sizes) <= 1
    N, S = 3, 2
    assert [sharding.episode_range(P, N, S, W, r) for r in range(W)] == [(a * N * S, b * N * S) for a, b in blocks]


def _worker(rank, world, port, P, N, q):
    sys.path.insert(0, HERE)
    import oracle_lib
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        scn = scenarios.replanning(horizon=5, n_iter=8)
        S = scn.desc.n_samples
        inits = scn.init_dist.sample(N, seed=3)
        w32 = np.stack([scenarios.planner_weights_fp32(c) for c in scn.candidate_weights(P, seed=4)])
        e0, e1 = sharding.episode_range(P, N, S, world, rank)
        local = oracle_lib.load().rollout(scn.desc, inits, w32, ep_begin=e0, ep_end=e1, n_threads=1)["returns"]
        full = sharding.gather_returns(torch.from_numpy(local), P, N, S)
        q.put((rank, full.numpy().copy(), sharding.fitness_from_returns(full.numpy(), P, N, S)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("P", [4, 3])        # even and ragged candidate split
def test_two_rank_gather_equals_single_process(oracle, P):
    N, world = 2, 2
    scn = scenarios.replanning(horizon=5, n_iter=8)
    inits = scn.init_dist.sample(N, seed=3)
    w32 = np.stack([scenarios.planner_weights_fp32(c) for c in scn.candidate_weights(P, seed=4)])
    ref = oracle.rollout(scn.desc, inits, w32)["returns"]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + P
    procs = [ctx.Process(target=_worker, args=(r, world, port, P, N, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, full, fit in got:
        assert np.array_equal(full, ref), rank
        assert np.array_equal(fit, sharding.fitness_from_returns(ref, P, N, scn.desc.n_samples))


def test_gather_is_identity_without_process_group():
    t = torch.arange(6, dtype=torch.float32)
    assert sharding.gather_returns(t, 3, 2, 1) is t
