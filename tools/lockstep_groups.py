#!/usr/bin/env python3
"""Generation wall-clock of R lockstep CMA-ES runs of the reference's shape (finite_horizon H = 5, pop 9 x 3 inits) against
the number of launch groups and host threads (csrc/ocd_cma.c: ocd_cma_run_many, ABI 7 / 8).  GPU box.
usage: lockstep_groups.py [R] [gens] ["g:t:chunk[:scan],..."]   (default: groups 1, 2, 3, 4, 2, 1 with one thread, chunk 32;
scan = the engine's scan_mode option: 0 the launcher's choice, 2 DPP rows, 3 one wavefront per workgroup)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 28
    gens = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    from l4dc_mpc_ocd_amd import scenarios
    from l4dc_mpc_ocd_amd.interact_drive.experiments import run_mpc_ord as rmo
    scn = scenarios.finite_horizon(horizon=5)
    runs = [(list(scn.init_dist.sample(3, seed=300 + r)), 1 + r, 0.05) for r in range(R)]
    ref = None
    plan = [(g, 1, 32, 0) for g in (1, 2, 3, 4, 2, 1)]
    if len(sys.argv) > 3:
        plan = [(tuple(int(v) for v in item.split(":")) + (0,))[:4] for item in sys.argv[3].split(",")]
    for groups, threads, chunk, scan in plan:
        m = rmo.make_mpc_ord("finite_horizon", horizon=5, n_inits=3, seed=1)
        m._engine().set_option("scan_mode", scan)
        res = m.optimize_cmaes_many(runs, maxiter=gens, termination={"tolfacupx": float("inf"), "tolupsigma": float("inf")}, groups=groups or None,
                                    host_threads=threads or None, chunk=chunk or None)
        wall = np.median(np.array(res.generation_wall_seconds[-32:])) * 1e3
        nat = np.median(np.array(res.generation_seconds[-32:])) * 1e3
        chk = float(sum(o.history[-1][1] for o in res.runs))
        ref = chk if ref is None else ref
        print(f"R={R} groups={groups} (used {res.groups}) threads={res.host_threads} chunk={chunk} scan={scan}: generation {wall:.4f} ms wall, {nat:.4f} ms native timers; host split "
              f"{ {k: round(v, 4) for k, v in res.host_split_ms().items()} }; same histories: {chk == ref}")


if __name__ == "__main__":
    main()
