#!/usr/bin/env python3
"""Kernel time of the REFERENCE's own shapes (pycma's population 9 x 3 inits = 27 episodes) per lane mapping:
finite_horizon H=5 (K=3), finite_horizon H=6 + extra_inits (K=6, n_iter 200), local_opt H=5 + extra_inits (K=6).
usage: python tools/small_shapes.py [--reps 10] [--pop 9] [--inits 3]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.sweep import launch_tag  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--pop", type=int, default=9)
    ap.add_argument("--inits", type=int, default=3)
    ap.add_argument("--modes", default="0,1,2,3")
    a = ap.parse_args()
    import torch
    from l4dc_mpc_ocd_amd import scenarios
    from l4dc_mpc_ocd_amd.engine import Engine
    shapes = [("finite_horizon H=5 K=3", scenarios.finite_horizon(horizon=5)),
              ("finite_horizon H=6 K=6 n_iter=200", scenarios.finite_horizon(horizon=6, extra_inits=True)),
              ("local_opt H=5 K=6", scenarios.local_opt(horizon=5, extra_inits=True)),
              ("replanning H=5 K=3 (S=2, T=20)", scenarios.replanning(horizon=5))]
    for name, scn in shapes:
        inits = scn.init_dist.sample(a.inits, seed=7)
        w32 = np.stack([scenarios.planner_weights_fp32(x) for x in scn.candidate_weights(a.pop, seed=8)])
        eng = Engine(scn, "cuda:0")
        init_dev = torch.as_tensor(inits, dtype=torch.float32).cuda()
        w_dev = torch.as_tensor(w32).cuda()
        E = a.pop * a.inits * scn.desc.n_samples
        ret = torch.empty(E, dtype=torch.float32, device="cuda")
        base = None
        for mode in [int(m) for m in a.modes.split(",")]:
            eng.set_option("scan_mode", mode)
            eng.time_rollout(init_dev, w_dev, 0, E, ret, 3)
            ms = eng.time_rollout(init_dev, w_dev, 0, E, ret, a.reps)
            r = ret.cpu().numpy().copy()
            if base is None:
                base = r
            same = np.array_equal(r.view(np.uint32), base.view(np.uint32))
            print(f"{name}: E={E} scan_mode={mode}: {ms:.3f} ms/launch  bitwise==mode0: {same}  {launch_tag(eng)}", flush=True)


if __name__ == "__main__":
    main()
