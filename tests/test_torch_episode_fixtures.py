"""The CPU oracle against the EPISODE fixtures (tests/golden/torch_episode_*.npz): whole receding-horizon episodes
of a float64 torch restatement of the reference's fitness loop (mpc_ord.py:67-151 over world.py:79-109,
replanning_world.py:11-36, fixed_plan_car.py:25-39, planner_car.py:54-85) whose scenario constants are typed from
the reference files.  The HIP path gets the identical check in tests/test_gpu_torch_episode_fixtures.py."""
import numpy as np
import pytest

import torch_episode_check as tec


@pytest.mark.parametrize("case", tec.cases())
def test_oracle_matches_torch_episodes(oracle, case):
    scn, z = tec.load(case)
    d = scn.desc

    def rollout_fn(inits, w32):
        return oracle.rollout(d, inits, w32, want_traj=True)

    def plan_fn(ws, w):
        return oracle.plan_batch(d, ws, w, scn.other_plans())["best_init"]

    print(case, tec.check(scn, z, rollout_fn, plan_fn))


def test_all_ten_episode_fixtures_are_present():
    assert tec.cases() == ["finite_horizon_h10", "finite_horizon_h5", "finite_horizon_h6", "local_opt_h10", "local_opt_h5",
                           "local_opt_h5_extra", "merging_h10", "merging_h5", "replanning_h10", "replanning_h5"]


def test_host_mirror_fitness_matches_the_float64_cost(oracle):
    """sharding.fitness_from_returns (the float64 reduction MPC_ORD.eval_population uses) on the oracle's returns
    against eval_weights' own return value in the fixture (mpc_ord.py:126-151)."""
    from l4dc_mpc_ocd_amd import scenarios, sharding
    for case in ("finite_horizon_h5", "replanning_h5"):
        scn, z = tec.load(case)
        w32 = np.stack([scenarios.planner_weights_fp32(c) for c in z["candidates"]])
        ret = oracle.rollout(scn.desc, z["init_states"], w32)["returns"]
        P, N, S = w32.shape[0], z["init_states"].shape[0], scn.desc.n_samples
        cost = sharding.fitness_from_returns(ret, P, N, S)
        full = z["stable"].reshape(P, -1).all(axis=1)
        assert full.any()
        np.testing.assert_allclose(cost[full], z["cost"][full], rtol=1e-4)


@pytest.mark.parametrize("case", [c for c in tec.cases() if not c.endswith("h10")])
def test_float64_build_of_the_oracle_follows_the_float64_episodes(case):
    """The oracle compiled in double (`make -C oracle fp64`: every float a double, constants keep their fp32 values)
    against the float64 torch episodes: with rounding out of the way on both sides the two restatements agree to 1e-6
    on EVERY episode of the reference's horizons (measured 3e-7; the residue is fp32(0.1) vs 0.1-style constants) --
    the fp32-unstable ones included, i.e. the 1e-4 of the float32 comparison is rounding, not a difference of algorithm."""
    import oracle_lib
    o64 = oracle_lib.load("fp64")
    scn, z = tec.load(case)
    w = z["planner_w32"].astype(np.float64)
    inits = z["init_states"].astype(np.float32).astype(np.float64)         # tf.constant(init, dtype=tf.float32)
    out = o64.rollout(scn.desc, inits, w, want_traj=True)
    E = out["returns"].shape[0]
    rerr = np.abs(out["returns"] - z["sample_reward"]) / np.maximum(1e-2, np.abs(z["sample_reward"]))
    terr = np.abs(out["traj"] - z["states"]).reshape(E, -1).max(axis=1)
    cerr = np.abs(out["ctrl"] - z["controls"]).reshape(E, -1).max(axis=1)
    assert rerr.max() <= 1e-6 and terr.max() <= 1e-6 and cerr.max() <= 2e-6, (rerr.max(), terr.max(), cerr.max())
