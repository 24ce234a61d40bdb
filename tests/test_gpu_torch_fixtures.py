"""The HIP path, through the C ABI (ocd_reward_batch, ocd_mpc_reward_batch, ocd_plan_batch), against fixtures
that do NOT come from the builder's oracle: float64 torch-autograd runs of a restatement of the reference's
Python (tests/golden/make_torch_fixtures.py).  Every other -m gpu comparison is HIP vs oracle (bit-exact, same
author); this one is the independent second opinion on the GPU path itself (reference: the features nobody
pins, experiments/merging.py:44-83, math_utils.py:28-31,87-95,166-178; naive_planner.py:44-77,107-164)."""
import numpy as np
import pytest

import torch_fixture_check as tfc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,H", tfc.fixtures())
def test_hip_matches_torch_fixture(hip, name, H):
    from l4dc_mpc_ocd_amd.engine import Engine
    scn, z = tfc.load(name, H)
    eng = Engine(scn, "cuda:0")

    def reward_fn(ws, w):
        return eng.reward_batch(ws, w)[0]

    def objective_fn(ws, w, u):
        out = eng.mpc_reward_batch(ws, w, u, want_traj=True)
        return out["reward"], out["grad"], out["traj"]

    def plan_fn(ws, w):
        return eng.plan_batch(ws, w, want_all=True)

    worst = tfc.check(scn, z, reward_fn, objective_fn, plan_fn)
    print(name, H, worst)


@pytest.mark.parametrize("mode", [1, 2, 3, 4])
def test_every_planner_variant_lands_on_the_torch_sgd_end_points(hip, mode):
    """The four lane mappings of the planner kernel (LDS windows, DPP rows, all-in-one wavefront, chunked) against
    the float64 SGD end points at H = 10."""
    from l4dc_mpc_ocd_amd.engine import Engine
    scn, z = tfc.load("merging", 10)
    eng = Engine(scn, "cuda:0")
    eng.set_option("scan_mode", mode)
    idx = z["sgd_states"]
    out = eng.plan_batch(z["world_states"][idx], z["weights"][idx], want_all=True)
    st = z["sgd_stable"]
    lerr = np.abs(out["all_losses"] - z["sgd_losses"]) / np.maximum(1e-2, np.abs(z["sgd_losses"]))
    perr = np.abs(out["all_plans"] - z["sgd_plans"]).reshape(st.shape + (-1,)).max(axis=2)
    assert st.mean() >= 0.8 and lerr[st].max() <= 1e-4 and perr[st].max() <= 1e-4, (lerr, perr)
    ok = st.all(axis=1)
    assert np.array_equal(out["best_init"][ok], z["sgd_best"][ok])
