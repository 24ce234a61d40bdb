"""The HIP path, through the C ABI (ocd_rollout_episodes, ocd_plan_batch), against EPISODE fixtures that do not come
from the builder's oracle: float64 torch runs of a restatement of the reference's fitness loop with the scenario
constants typed from the reference files (tests/golden/make_torch_episode_fixtures.py).  Returns within 1e-4
relative (the north_star tolerance), trajectories and controls within 1e-4, the kept control initialisation exact,
on the episodes torch itself reproduces in float32.  Reference: mpc_ord.py:67-151, world.py:79-109,
replanning_world.py:11-36, fixed_plan_car.py:25-39, planner_car.py:54-85, naive_planner.py:107-164."""
import numpy as np
import pytest

import torch_episode_check as tec

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", tec.cases())
def test_hip_matches_torch_episodes(hip, case):
    from l4dc_mpc_ocd_amd.engine import Engine
    scn, z = tec.load(case)
    eng = Engine(scn, "cuda:0")

    def rollout_fn(inits, w32):
        return eng.rollout(inits, w32, want_traj=True)

    def plan_fn(ws, w):
        return eng.plan_batch(ws, w)["best_init"]

    print(case, tec.check(scn, z, rollout_fn, plan_fn))


@pytest.mark.parametrize("mode", [1, 2, 3, 4])
def test_every_planner_variant_reproduces_the_torch_episodes(hip, mode):
    """The four lane mappings of the planner kernel on the replanning episodes (three cars, plans, teleport)."""
    from l4dc_mpc_ocd_amd.engine import Engine
    scn, z = tec.load("replanning_h5")
    eng = Engine(scn, "cuda:0")
    eng.set_option("scan_mode", mode)
    print(mode, tec.check(scn, z, lambda inits, w32: eng.rollout(inits, w32, want_traj=True)))


def test_mpc_ord_eval_weights_returns_the_float64_cost(hip):
    """MPC_ORD.eval_weights of the host mirror (the pycma fitness callable, mpc_ord.py:109-151) on the fixture's
    candidates and init states: the value pycma would see, against the restatement's."""
    from l4dc_mpc_ocd_amd.interact_drive.reward_design.mpc_ord import MPC_ORD, finite_horizon_env
    scn, z = tec.load("finite_horizon_h5")
    car, world, _ = finite_horizon_env(horizon=5, env_seeds=[1])
    ord_ = MPC_ORD(world, car, [s for s in z["init_states"]], 15)
    P, N = z["candidates"].shape[0], z["init_states"].shape[0]
    full = z["stable"].reshape(P, N).all(axis=1)
    assert full.any()
    for p in np.nonzero(full)[0]:
        got = ord_.eval_weights(list(z["candidates"][p]))
        assert abs(got - z["cost"][p]) <= 1e-4 * max(1.0, abs(z["cost"][p])), (p, got, z["cost"][p])
