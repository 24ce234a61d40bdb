"""Child of tests/test_gpu_lockstep.py::test_lockstep_deals_the_runs_over_ranks: one rank of a gloo group on the one GPU of
the box; every rank builds the same seven runs, MPC_ORD.optimize_cmaes_many deals them over the ranks, rank 0 writes what
EVERY rank must hold afterwards."""
import os
import pickle
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def runs_of(scn):
    return [(list(scn.init_dist.sample([1, 3, 2][r % 3], seed=90 + r)), 7 + 2 * r, [0.05, 0.2][r % 2]) for r in range(7)]


def main():
    import torch
    import torch.distributed as dist
    from l4dc_mpc_ocd_amd import scenarios
    from l4dc_mpc_ocd_amd.interact_drive.experiments.run_mpc_ord import make_mpc_ord
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    name = sys.argv[2]
    scn = scenarios.SCENARIOS[name](horizon=5)
    base = make_mpc_ord(name, horizon=5, n_inits=1, seed=1)
    res = base.optimize_cmaes_many(runs_of(scn), maxiter=4)
    out = dict(rank=dist.get_rank(), ranks=res.ranks, lockstep=res.lockstep,
               histories=[[(np.asarray(w), float(r)) for w, r in o.history] for o in res.runs],
               seeds=[o.history.seed for o in res.runs], stops=[o.stop_reason for o in res.runs],
               best=[np.asarray(b) for b in res.best], iters=[o.iter for o in res.runs],
               made_here=[o.es is not None for o in res.runs],
               episodes=[st["episodes_per_generation"] if st else None for st in res.per_rank])
    with open(f"{sys.argv[1]}.rank{dist.get_rank()}", "wb") as f:
        pickle.dump(out, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
