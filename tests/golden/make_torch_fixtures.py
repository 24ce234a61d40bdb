#!/usr/bin/env python3
"""Second-opinion fixtures for the GPU path: tests/golden/torch_<scenario>_h<H>.npz.

Made in the BUILD container (torch-CPU autograd, float64) by tests/golden/torch_restatement.py -- the restatement
written from the reference's Python (merging.py:44-83, math_utils.py:28-31,87-95,166-178,
simulation_utils.py:9-21, naive_planner.py:44-77,107-164), NOT from the C oracle and NOT from the kernels.
The reference itself cannot produce vectors here (TensorFlow is not installed), so this is the
most independent check the HIP path can be given: tests/test_gpu_torch_fixtures.py compares ocd_reward_batch,
ocd_mpc_reward_batch and ocd_plan_batch with these values (tolerances there: fp32 kernels vs float64 autograd),
tests/test_torch_fixtures.py does the same for the oracle on the CPU.

Per (scenario, H) fixture -- H in {5, 10} for the four scenarios, replanning H = 15 and merging H = 25 (BASELINE configs 4 / 5) -- B = 32 world states -- the first 8 inside a scripted car's collision bump,
the next 8 beyond the fence threshold, every 5th control row beyond the clip range:
    world_states [B,C,4] f32, weights [B,D] f32 (normalised candidates), controls [B,H,2] f32, other_plans
    features [B,D] f64, R [B] f64, grad [B,H,2] f64, traj [B,H,4] f64          (objective at the given controls)
    sgd_states (indices), sgd_plans [b,K,H,2] f64, sgd_losses [b,K] f64, sgd_best [b] i32
        (generate_plan with n_iter = 25 plain-SGD steps per control initialisation, lr 0.1)
    sgd_stable [b,K] bool: the same SGD run in torch FLOAT32 lands on the float64 end point (loss within 2e-5
        relative, plan within 2e-5): only those (state, initialisation) pairs are compared with a tolerance --
        where torch's own fp32 run leaves the fp64 one (steep bumps: the iteration amplifies rounding), no fp32
        implementation can be held to it.  Decided by torch alone, not by the code under test.

usage: python tests/golden/make_torch_fixtures.py [scenario:H ...]     (rewrites the ten files, or the named ones; deterministic)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import torch_restatement as tr  # noqa: E402
from l4dc_mpc_ocd_amd import scenarios  # noqa: E402

B = 32
N_SGD = 6
SGD_ITERS = 25


def world_states(scn, rng):
    d = scn.desc
    ws = np.zeros((B, d.n_cars, 4))
    for j in range(1, d.n_cars):
        ws[:, j, :] = np.array(d.other_init[j - 1][:])
    ws[:, 0, :] = scn.init_dist.sample(B, seed=int(rng.integers(1 << 30)))
    ws[:, 0, 0] += 0.8 * rng.uniform(-0.12, 0.12, B)
    ws[:, 0, 3] += rng.uniform(-0.3, 0.3, B)
    ws[:, 1:, 0] += rng.uniform(-0.05, 0.05, (B, d.n_cars - 1))
    ws[:, 1:, 1] += rng.uniform(-0.2, 0.2, (B, d.n_cars - 1))
    # 0..7: the ego starts inside car 1's collision bump (and moves along with it for a few steps)
    ws[:8, 0, 0] = ws[:8, 1, 0] + rng.uniform(-0.06, 0.06, 8)
    ws[:8, 0, 1] = ws[:8, 1, 1] + rng.uniform(-0.12, 0.12, 8)
    ws[:8, 0, 2] = ws[:8, 1, 2] + rng.uniform(-0.1, 0.1, 8)
    # 8..15: beyond the fence threshold on either side (|x| > 0.05 * lanes - 0.05)
    lo = 0.05 * d.n_lanes - 0.05
    ws[8:16, 0, 0] = np.where(rng.random(8) < 0.5, -1.0, 1.0) * (lo + rng.uniform(0.005, 0.045, 8))
    return ws.astype(np.float32)


def make(name, H, seed):
    scn = scenarios.SCENARIOS[name](horizon=H, n_iter=SGD_ITERS)
    d = scn.desc
    rng = np.random.default_rng(seed)
    ws = world_states(scn, rng)
    w = np.stack([scenarios.planner_weights_fp32(c) for c in scn.candidate_weights(B, seed=seed + 1)])
    u = np.stack([rng.uniform(-1.0, 1.0, (B, H)), rng.uniform(-1.5, 1.5, (B, H))], axis=2).astype(np.float32)
    u[::5, 0] = (5.0, -6.0)                                   # beyond the clip range: gradient gated to zero
    u[2::5, H - 1] = (-9.0, 4.5)
    other = scn.other_plans()
    feats = np.stack([tr.features(d, s) for s in ws])
    R, G, T = [], [], []
    for b in range(B):
        r, g, t = tr.mpc_reward_and_grad(d, ws[b], w[b], u[b], other)
        R.append(r); G.append(g); T.append(t)
    # SGD end points: a collision state, a fence state and four plain ones
    sgd_states = np.array([1, 9, 16, 19, 24, 29], dtype=np.int32)[:N_SGD]
    import torch
    plans, losses, best, stable = [], [], [], []
    for b in sgd_states:
        o, l, k = tr.generate_plan(d, ws[b], w[b], other)
        o32, l32, _ = tr.generate_plan(d, ws[b], w[b], other, dtype=torch.float32)
        ok = (np.abs(l32 - l) <= 2e-5 * np.maximum(1e-2, np.abs(l))) & \
             (np.abs(o32 - o).reshape(len(l), -1).max(axis=1) <= 2e-5)
        plans.append(o); losses.append(l); best.append(k); stable.append(ok)
    out = dict(world_states=ws, weights=w, controls=u,
               other_plans=np.zeros((0,), dtype=np.float32) if other is None else np.asarray(other, dtype=np.float32),
               features=feats, R=np.array(R), grad=np.array(G), traj=np.array(T),
               sgd_states=sgd_states, sgd_plans=np.array(plans), sgd_losses=np.array(losses),
               sgd_best=np.array(best, dtype=np.int32), sgd_stable=np.array(stable), sgd_iters=np.int32(SGD_ITERS),
               horizon=np.int32(H))
    path = os.path.join(HERE, f"torch_{name}_h{H}.npz")
    np.savez_compressed(path, **out)
    L = d.n_lanes
    print(f"{os.path.basename(path)}: collision>0 in {int((feats[:, L + 2] > 0).sum())} states, fence>0 in "
          f"{int((feats[:, L + 3] > 0).sum())}, |grad| max {np.abs(np.array(G)).max():.3g}, best inits {best}, "
          f"fp32-stable SGD pairs {int(np.sum(stable))}/{np.size(stable)}")


# (scenario, H) -> seed.  H = 5 / 10: every scenario; H = 15 / 25 (round 5): BASELINE configs 4 / 5's own horizons, on their
# scenarios -- naive_planner.py:33-77 unrolled 15 / 25 deep, objective and gradient against float64 autograd
CASES = {(name, H): 4000 + 10 * i + H for i, name in enumerate(("finite_horizon", "local_opt", "replanning", "merging"))
         for H in (5, 10)}
CASES[("replanning", 15)] = 4035
CASES[("merging", 25)] = 4055


def main():
    want = [tuple(a.split(":")) for a in sys.argv[1:]]
    for (name, H), seed in CASES.items():
        if not want or (name, str(H)) in want:
            make(name, H, seed=seed)


if __name__ == "__main__":
    main()
