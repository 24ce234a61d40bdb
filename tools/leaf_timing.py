#!/usr/bin/env python3
"""Kernel time of the planner with a terminal value (leaf_evaluation, naive_planner.py:20,69-70) against the same
scenario without one: finite_horizon at the reference's horizon H = 5 (and 6), batches of 3 ... 2 048 episodes.
usage (GPU box): python tools/leaf_timing.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from l4dc_mpc_ocd_amd import scenarios
    from l4dc_mpc_ocd_amd.engine import Engine
    rng = np.random.default_rng(8)
    grid = [np.linspace(-0.22, 0.22, 20), np.linspace(-1.2, 1.5, 60), np.linspace(0.0, 2.0, 30)]   # coarse_value_iteration.py:43-55
    vals = rng.standard_normal((20, 60, 30)).astype(np.float32)
    for H in (5, 6):
        for P, N in ((1, 3), (16, 8), (64, 32)):
            scn = scenarios.finite_horizon(horizon=H)
            inits = scn.init_dist.sample(N, seed=1)
            w32 = np.stack([scenarios.planner_weights_fp32(c) for c in scn.candidate_weights(P, seed=2)])
            E = P * N
            init_dev = torch.as_tensor(inits, dtype=torch.float32).cuda()
            w_dev = torch.as_tensor(w32).cuda()
            ret = torch.empty(E, dtype=torch.float32, device="cuda")
            row = []
            for leaf in (False, True):
                eng = Engine(scn, "cuda:0")
                if leaf:
                    eng.set_leaf_value(grid, vals, 1)
                eng.time_rollout(init_dev, w_dev, 0, E, ret, 3)
                row.append(eng.time_rollout(init_dev, w_dev, 0, E, ret, 10))
            print(f"finite_horizon H={H} n_iter={scn.desc.n_iter} episodes={E}: no terminal value {row[0]:.3f} ms, "
                  f"with terminal value {row[1]:.3f} ms ({row[1] / row[0]:.2f}x)", flush=True)


if __name__ == "__main__":
    main()
