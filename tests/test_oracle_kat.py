"""Pin the CPU oracle against every known-answer test the reference holds for this path.

Sources (reference tree):
  interact_drive/tests/test_simulation_utils.py:113-158     dynamics KATs (assertAlmostEqual, 7 places)
  interact_drive/math_utils.py:19-26,65-71,140-147          doctests of _f, smooth_threshold, smooth_bump
  interact_drive/planner/tests/targetSpeedRewardMaximizerCar.py:17-27   reward doctest
  interact_drive/planner/tests/test_naivePlanner.py:21-32,50-63         planner KATs (atol 1e-5)
  interact_drive/reward_design/tests/test_first_order_ioc.py:29-60, tests/linearTargetSpeedPlannerCar.py:36-44
                                                             "this leads to zero controls" (linear target-speed car)
"""
import numpy as np
import pytest

from l4dc_mpc_ocd_amd import scenarios

PI_2 = np.pi / 2


def almost_equal(a, b, places=7):
    # unittest.assertAlmostEqual: round(a-b, places) == 0
    return round(float(a) - float(b), places) == 0


@pytest.mark.parametrize("state,friction,expect", [
    ((0., 0., 1., PI_2), 0.0, (0., 1., 1., PI_2)),        # test_next_car_state_0
    ((0., 0., 1., PI_2), 1.0, (0., 0.5, 0., PI_2)),       # test_next_car_state_1
    ((0., 0., 1., PI_2), 0.5, (0., 0.75, 0.5, PI_2)),     # test_next_car_state_2
    ((0., 0., 1., 0.), 0.5, (0.75, 0., 0.5, 0.)),         # test_next_car_state_3
])
def test_dynamics_kat(oracle, state, friction, expect):
    out = oracle.dynamics_step(state, (0., 0.), dt=1.0, friction=friction)
    for got, want in zip(out, expect):
        assert almost_equal(got, want), (out, expect)


def test_dynamics_control_clipping(oracle):
    # car_dynamics_step clips acc to [-8, 4] and ang_vel to [-4, 4] (simulation_utils.py:11-12)
    hi = oracle.dynamics_step((0, 0, 1, 0), (100., 100.), dt=1.0, friction=0.0)
    lo = oracle.dynamics_step((0, 0, 1, 0), (-100., -100.), dt=1.0, friction=0.0)
    assert almost_equal(hi[2], 5.0) and almost_equal(hi[3], 4.0)
    assert almost_equal(lo[2], -7.0) and almost_equal(lo[3], -4.0)


def test_f_doctest(oracle):
    assert oracle.f(0.0) == 0.0
    assert oracle.f(1.0) > 0
    assert oracle.f(0.01) > 0
    assert oracle.f(1e10) - 1 < 1e-5


def test_smooth_threshold_doctest(oracle):
    assert oracle.smooth_threshold(0.0, 0.0, 1.0) == 1.0
    assert oracle.smooth_threshold(-1.0, 0.0, 1.0) == 0.0
    assert oracle.smooth_threshold(-0.5, 0.0, 1.0) == 0.5


def test_smooth_bump_doctest(oracle):
    assert oracle.smooth_bump(0.0, -1.0, 1.0) == 1.0
    assert oracle.smooth_bump(-1.0, -1.0, 1.0) == 0.0
    assert oracle.smooth_bump(1.0, -1.0, 1.0) == 0.0
    assert oracle.smooth_bump(0.5, -1.0, 1.0) > 0


@pytest.mark.parametrize("v,expect", [(1.0, -1.0), (0.0, 0.0), (2.0, -4.0)])
def test_target_speed_reward_doctest(oracle, v, expect):
    scn = scenarios.target_speed_kat(horizon=5, n_iter=1, learning_rate=5.0, friction=0.0, target_speed=0.0)
    r, _, _ = oracle.reward(scn.desc, [[0., 0., v, PI_2]], None)
    assert r == expect


def test_planner_kat_zero_friction(oracle):
    """test_zero_friction_correct_speed: every planned control == (0, 0), atol 1e-5.
    Also pins the first-index tie-break: the (0, -+0.65) initialisations reach the same loss."""
    scn = scenarios.target_speed_kat(horizon=5, n_iter=100, learning_rate=5.0, friction=0.0)
    out = oracle.plan_batch(scn.desc, [[0., 0., 1., PI_2]], None)
    assert out["plans"].shape == (1, 5, 2)
    np.testing.assert_allclose(out["plans"][0], np.zeros((5, 2)), atol=1e-5)
    assert out["best_init"][0] == 0
    assert out["all_losses"][0, 0] == out["all_losses"][0, 1] == out["all_losses"][0, 2]
    np.testing.assert_allclose(out["all_plans"][0, 1, :, 1], -0.65, atol=1e-6)


def test_planner_kat_friction(oracle):
    """test_friction_correct_speed: every planned control == (friction * 1.0**2, 0), atol 1e-5."""
    friction = 0.5
    scn = scenarios.target_speed_kat(horizon=3, n_iter=500, learning_rate=5.0, friction=friction)
    out = oracle.plan_batch(scn.desc, [[0., 0., 1., PI_2]], None)
    for t in range(3):
        np.testing.assert_allclose(out["plans"][0, t], np.array([friction * 1.0 ** 2, 0.]), atol=1e-5)


def test_planner_with_other_controls_runs(oracle):
    """TestPlanVsFixedPlanCar.test_no_interaction: generate_plan with other_controls for a
    FixedPlanCar (layout [C-1, H, 2]); the reference asserts nothing, only that it runs."""
    scn = scenarios.finite_horizon(horizon=3, n_iter=20)
    ws = [[0., 0., 1., PI_2], [0.1, 0., 1., PI_2]]
    other = np.zeros((1, 3, 2), dtype=np.float32)
    out = oracle.plan_batch(scn.desc, ws, scn.designer_weights, other_plans=other)
    assert np.all(np.isfinite(out["plans"]))


def test_linear_target_speed_car_leads_to_zero_controls(oracle):
    """test_first_order_ioc.py:29-49: LinearTargetSpeedPlannerCar, weights (2, -1), target_speed 0, v = 1, friction 0,
    horizon 5, n_iter 10, lr 5.0 -- "this leads to zero controls": d/dv (w0 v + w1 v^2) = w0 + 2 w1 v vanishes at
    v = 1 for w = (2, -1)/sqrt(5), exactly in fp32 (2 w1 == -w0 bit for bit), so SGD never moves the controls and
    the six world steps of the test's trajectory all apply (0, 0)."""
    scn = scenarios.linear_target_speed(horizon=5, n_iter=10, learning_rate=5.0, friction=0.0, target_speed=0.0)
    w = scenarios.normalize_like_reference(np.array([2., -1.], dtype=np.float32)).astype(np.float32)   # LinearRewardCar.__init__
    assert w[0] == -2 * w[1]
    r, feats, _ = oracle.reward(scn.desc, [[0., 0., 1., PI_2]], w)
    assert np.array_equal(feats, [1.0, 1.0]) and r == np.float32(np.float32(w[0] * 1.0) + np.float32(w[1] * 1.0))
    out = oracle.plan_batch(scn.desc, [[0., 0., 1., PI_2]], w)
    assert np.array_equal(out["plans"][0], np.zeros((5, 2), dtype=np.float32))          # exactly zero controls
    assert out["best_init"][0] == 0 and out["all_losses"][0, 0] == out["all_losses"][0, 1] == out["all_losses"][0, 2]
    assert np.array_equal(out["all_plans"][0, 2, :, 1], np.full(5, 0.65, dtype=np.float32))   # heading never enters
    ro = oracle.rollout(scn.desc, np.array([[0., 0., 1., PI_2]]), w[None], want_traj=True)
    assert np.array_equal(ro["ctrl"][0], np.zeros((6, 2), dtype=np.float32))
    assert np.array_equal(ro["traj"][0][:, 0, 2], np.ones(7, dtype=np.float32))          # the speed stays 1
    np.testing.assert_allclose(ro["traj"][0][:, 0, 1], 0.1 * np.arange(7), rtol=1e-6)    # y advances by v dt


def test_linear_target_speed_gradient_matches_torch(oracle):
    """The same car with friction (test_first_order_ioc.py:62-79: friction 0.2, n_iter 200): objective and gradient
    against torch autograd on the reference's feature expressions."""
    import torch_restatement as tr
    scn = scenarios.linear_target_speed(horizon=5, n_iter=200, learning_rate=5.0, friction=0.2, target_speed=0.0)
    rng = np.random.default_rng(5)
    for _ in range(20):
        ws = np.array([[rng.uniform(-0.2, 0.2), rng.uniform(-1, 1), rng.uniform(0.2, 2.0), PI_2 + rng.uniform(-0.4, 0.4)]], dtype=np.float32)
        w = scenarios.normalize_like_reference(rng.standard_normal(2)).astype(np.float32)
        u = np.stack([rng.uniform(-2, 2, 5), rng.uniform(-1, 1, 5)], axis=1).astype(np.float32)
        r, g, traj = oracle.mpc_reward(scn.desc, ws, w, u)
        r64, g64, t64 = tr.mpc_reward_and_grad(scn.desc, ws, w, u)
        np.testing.assert_allclose(traj, t64, rtol=1e-5, atol=1e-6)
        assert abs(r - r64) <= 2e-5 * max(1.0, abs(r64))
        assert np.abs(g - g64).max() <= 2e-5 * max(1e-3, np.abs(g64).max())
        assert np.all(g[:, 1] == 0.0)                      # the heading rate never enters the reward
