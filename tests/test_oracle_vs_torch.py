"""Cross-check the C oracle against the independent torch-autograd restatement.

The reference's tests do not pin ThreeLaneTestCar.features, the scenarios or
the objective's gradient ("parity unpinned"); this is the second
opinion.  fp32 oracle vs fp64 torch: tolerances are rounding-level, not exact.
"""
import numpy as np
import pytest

import torch_restatement as tr
from l4dc_mpc_ocd_amd import scenarios


def _random_world_states(scn, n, rng, spread=1.0):
    d = scn.desc
    out = np.zeros((n, d.n_cars, 4))
    for j in range(1, d.n_cars):
        out[:, j, :] = np.array(d.other_init[j - 1][:])
    ego = scn.init_dist.sample(n, seed=int(rng.integers(1 << 30)))
    out[:, 0, :] = ego
    # spread the ego around so that lanes / fences / collision bumps are all exercised
    out[:, 0, 0] += spread * rng.uniform(-0.12, 0.12, n)
    out[:, 0, 3] += rng.uniform(-0.3, 0.3, n)
    out[:, 1:, 0] += rng.uniform(-0.05, 0.05, (n, d.n_cars - 1))
    out[:, 1:, 1] += rng.uniform(-0.2, 0.2, (n, d.n_cars - 1))
    return out.astype(np.float32)


@pytest.mark.parametrize("name", ["finite_horizon", "local_opt", "replanning", "merging"])
def test_features_match_torch(oracle, name):
    scn = scenarios.SCENARIOS[name](horizon=5)
    rng = np.random.default_rng(7)
    states = _random_world_states(scn, 200, rng)
    # make some states collide with / sit next to a scripted car
    states[:40, 0, 0] = states[:40, 1, 0] + rng.uniform(-0.07, 0.07, 40)
    states[:40, 0, 1] = states[:40, 1, 1] + rng.uniform(-0.14, 0.14, 40)
    nz_col = nz_fence = 0
    for ws in states:
        _, feats, _ = oracle.reward(scn.desc, ws, scn.designer_weights)
        ref = tr.features(scn.desc, ws)
        np.testing.assert_allclose(feats, ref, rtol=2e-5, atol=1e-7)
        L = scn.desc.n_lanes
        nz_col += feats[L + 2] > 0
        nz_fence += feats[L + 3] > 0
    assert nz_col >= 20 and nz_fence >= 10     # the interesting branches were exercised


@pytest.mark.parametrize("name,H", [("finite_horizon", 5), ("local_opt", 10), ("replanning", 6), ("merging", 8)])
def test_objective_and_gradient_match_torch(oracle, name, H):
    scn = scenarios.SCENARIOS[name](horizon=H)
    rng = np.random.default_rng(11)
    states = _random_world_states(scn, 24, rng, spread=0.8)
    other = scn.other_plans()
    worst = 0.0
    for i, ws in enumerate(states):
        w = scenarios.planner_weights_fp32(scn.candidate_weights(4, seed=i)[i % 4])
        u = np.stack([rng.uniform(-1.0, 1.0, H), rng.uniform(-1.5, 1.5, H)], axis=1).astype(np.float32)
        if i % 5 == 0:
            u[0] = (5.0, -6.0)      # beyond the clip range: gradient must be gated to zero
        r, g, traj = oracle.mpc_reward(scn.desc, ws, w, u, other)
        r64, g64, traj64 = tr.mpc_reward_and_grad(scn.desc, ws, w, u, other)
        np.testing.assert_allclose(traj, traj64, rtol=1e-5, atol=1e-6)
        assert abs(r - r64) <= 2e-5 * max(1.0, abs(r64))
        scale = max(1e-3, np.abs(g64).max())
        err = np.abs(g - g64).max() / scale
        worst = max(worst, err)
        assert err < 2e-4, (i, g, g64)
        if i % 5 == 0:
            assert g[0, 0] == 0.0 and g[0, 1] == 0.0
    assert worst < 2e-4


def test_plan_matches_torch_sgd(oracle):
    """A short SGD run from each control initialisation lands where torch-autograd SGD lands."""
    scn = scenarios.finite_horizon(horizon=5, n_iter=25)
    rng = np.random.default_rng(3)
    ws = _random_world_states(scn, 3, rng, spread=0.3)
    for s in ws:
        w = scenarios.planner_weights_fp32(scn.designer_weights)
        out = oracle.plan_batch(scn.desc, s, w)
        opts, losses, best = tr.generate_plan(scn.desc, s, w)
        np.testing.assert_allclose(out["all_plans"][0], opts, rtol=5e-3, atol=5e-4)
        np.testing.assert_allclose(out["all_losses"][0], losses, rtol=1e-4, atol=1e-6)
        assert out["best_init"][0] == best
