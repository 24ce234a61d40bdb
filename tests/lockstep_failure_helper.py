"""Worker of tests/test_lockstep_ranks_cpu.py: one gloo rank of _lockstep_over_ranks with the per-rank work stubbed out (no GPU)."""
import os
import sys
import types

import numpy as np


def worker(rank, world, port, fail_rank, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from l4dc_mpc_ocd_amd.interact_drive.reward_design import mpc_ord

        class Stub:                                   # the attributes _lockstep_over_ranks reads and writes
            def __init__(self, k):
                self.k, self.history, self.iter, self.stop_reason = k, mpc_ord.list2(), 0, {}
                self.history.seed = 0
                self.n_nonfinite, self.n_resampled, self.generation_seconds = [], 0, []
                self.es = types.SimpleNamespace(best_f=float(k))

        def local(ords, seeds, sigma0s, *rest):       # what a rank does with ITS runs
            if dist.get_rank() == fail_rank:
                raise ValueError(f"boom on rank {dist.get_rank()}")
            res = mpc_ord.LockstepResult(list(ords))
            for o, seed in zip(ords, seeds):
                o.history.append((np.full(3, o.k, dtype=np.float64), float(seed)))
                o.history.seed, o.iter, o.stop_reason = seed, 1, {"maxiter": 1}
                res.best.append(np.full(3, o.k, dtype=np.float64))
            res.generation_seconds, res.generation_wall_seconds, res.episodes_per_generation = [0.001], [0.001], [27 * len(ords)]
            res.launch = {"mapping": "stub"}
            return res

        mpc_ord._lockstep_local = local
        ords = [Stub(k) for k in range(5)]
        try:
            res = mpc_ord._lockstep_over_ranks(ords, [10 + k for k in range(5)], [0.1] * 5, None, 1, None, None, 4)
            q.put((rank, "ok", [float(b[0]) for b in res.best], [o.history.seed for o in res.runs], res.ranks))
        except RuntimeError as exc:
            q.put((rank, "raised", str(exc).splitlines()[0], None, None))
    finally:
        dist.destroy_process_group()
