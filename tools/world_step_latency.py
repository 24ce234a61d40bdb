#!/usr/bin/env python3
"""Latency of the object-by-object path (BASELINE config 1's shape): world.step() = PlannerCar._get_next_control ->
NaivePlanner.generate_plan (one plan launch) + Car.step of every car (world.py:79-109, planner_car.py:54-85), and
MPC_ORD.eval_weights through the scalar API.  `--engine FILE` runs the same loop on another engine.py (A/B of the
host plumbing: round 4's per-call device synchronisation + copies against pinned zero-copy staging + an event wait)."""
import argparse
import importlib.util
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--engine", default="")
    ap.add_argument("--episodes", type=int, default=12)
    a = ap.parse_args()
    import l4dc_mpc_ocd_amd  # noqa: F401
    if a.engine:
        spec = importlib.util.spec_from_file_location("l4dc_mpc_ocd_amd.engine", a.engine)
        mod = importlib.util.module_from_spec(spec)
        sys.modules["l4dc_mpc_ocd_amd.engine"] = mod
        spec.loader.exec_module(mod)
    from l4dc_mpc_ocd_amd.interact_drive.reward_design.mpc_ord import MPC_ORD, finite_horizon_env
    car, world, init_states = finite_horizon_env(horizon=5, env_seeds=[1000001, 1000002, 1000003])
    m = MPC_ORD(world, car, init_states, 15)
    m.eval_weights(m.designer_weights)
    ts = []
    for _ in range(40):
        t0 = time.perf_counter()
        m.eval_weights(m.designer_weights)
        ts.append(time.perf_counter() - t0)
    steps = []
    for e in range(a.episodes):
        car.init_state = type(car.state)(init_states[e % 3])
        world.reset()
        for _ in range(15):
            t0 = time.perf_counter()
            world.step()
            steps.append(time.perf_counter() - t0)
    steps = np.array(steps[15:]) * 1e3
    print(f"engine {a.engine or 'current'}: eval_weights {np.median(ts) * 1e3:.4f} ms; world.step() median {np.median(steps):.4f} ms, "
          f"mean {steps.mean():.4f}, p90 {np.percentile(steps, 90):.4f} over {steps.size} steps")


if __name__ == "__main__":
    main()
