"""The CPU oracle against the float64 torch-autograd fixtures (tests/golden/torch_*.npz): features, objective,
gradient, trajectory and 25-step SGD end points of the four scenarios at H = 5 and 10, collision / fence /
clipped-control states included.  The HIP path gets the identical check in tests/test_gpu_torch_fixtures.py."""
import numpy as np
import pytest

import torch_fixture_check as tfc


@pytest.mark.parametrize("name,H", tfc.fixtures())
def test_oracle_matches_torch_fixture(oracle, name, H):
    scn, z = tfc.load(name, H)
    d = scn.desc
    other = scn.other_plans()

    def reward_fn(ws, w):
        return np.stack([oracle.reward(d, s, w)[1] for s in ws])

    def objective_fn(ws, w, u):
        res = [oracle.mpc_reward(d, ws[b], w[b], u[b], other) for b in range(ws.shape[0])]
        return np.array([r[0] for r in res]), np.stack([r[1] for r in res]), np.stack([r[2] for r in res])

    def plan_fn(ws, w):
        return oracle.plan_batch(d, ws, w, other)

    worst = tfc.check(scn, z, reward_fn, objective_fn, plan_fn)
    print(name, H, worst)


def test_all_ten_fixtures_are_present():
    assert len(tfc.fixtures()) == 10 and ("replanning", 15) in tfc.fixtures() and ("merging", 25) in tfc.fixtures()
