"""The hand-written forms of the reward evaluation against each other, form by form ("Engineering": five
evaluations of the same feature arithmetic held together by plan-level tests only).  ocd_debug_feature_variants runs
reward_state (the definition), reward_one (straight line / shortened divisions / sub-skips), reward_fc (full / shortened),
reward_every = reward_fcc (full / shortened) and the work-item form of the chunked kernel's shared-SIMD builds (one item per
active feature through an LDS list; a state inside both cars' boxes as a pair of neighbouring items) on the same world states and reports per state which forms'
preconditions hold; every valid (state, form) pair must give reward_state's value and adjoint bit for bit, and
reward_state itself the CPU oracle's.  Reference: merging.py:44-83, math_utils.py:28-31,87-95,166-178."""
import numpy as np
import pytest

from l4dc_mpc_ocd_amd import abi, scenarios

pytestmark = pytest.mark.gpu
FORMS = ["reward_state", "reward_one", "reward_one shortened", "reward_one sub-skips", "reward_fc", "reward_fc shortened",
         "reward_every", "reward_every shortened", "work items", "reward_two", "reward_two reciprocal quotients"]


def same(a, b):
    return (a == b) | (np.isnan(a) & np.isnan(b))


def world_states(scn, rng, n):
    d = scn.desc
    C = d.n_cars
    lo, wd = np.float32(d.fence_lo), np.float32(d.fence_width)
    ws = np.zeros((n, C, 4), dtype=np.float32)
    for j in range(1, C):
        ws[:, j] = np.array(d.other_init[j - 1][:], dtype=np.float32) + rng.uniform(-0.05, 0.05, (n, 4)).astype(np.float32)
    k = rng.integers(0, 8, n)
    ws[:, 0, 0] = rng.uniform(-0.3, 0.3, n)
    ws[:, 0, 1] = ws[:, 1, 1] + rng.uniform(-0.4, 0.4, n)
    ws[:, 0, 2] = rng.uniform(0.0, 1.5, n)
    ws[:, 0, 3] = np.pi / 2 + rng.uniform(-0.5, 0.5, n)
    # inside car 1's collision box
    sel = k == 0
    ws[sel, 0, 0] = ws[sel, 1, 0] + rng.uniform(-0.079, 0.079, sel.sum())
    ws[sel, 0, 1] = ws[sel, 1, 1] + rng.uniform(-0.149, 0.149, sel.sum())
    # the fence region and its edges, both sides
    sel = k == 1
    ws[sel, 0, 0] = rng.choice([-1.0, 1.0], sel.sum()) * (lo + wd * rng.uniform(-0.1, 1.3, sel.sum()))
    sel = k == 2                                                     # by units in the last place around the fence's lower edge
    edge = np.full(sel.sum(), lo, dtype=np.float32)
    for _ in range(6):
        step = rng.integers(-1, 2, sel.sum())
        edge = np.where(step > 0, np.nextafter(edge, np.float32(np.inf)), np.where(step < 0, np.nextafter(edge, np.float32(-np.inf)), edge))
    ws[sel, 0, 0] = edge * rng.choice([-1.0, 1.0], sel.sum()).astype(np.float32)
    if C > 2:
        sel = k == 3                                                 # both cars on one spot, the ego inside both boxes
        ws[sel, 2, :2] = ws[sel, 1, :2] + rng.uniform(-0.02, 0.02, (sel.sum(), 2)) * (rng.random((sel.sum(), 1)) < 0.7)   # (30 %: exactly: a tie of reduce_max)
        ws[sel, 0, 0] = ws[sel, 1, 0] + rng.uniform(-0.05, 0.05, sel.sum())
        ws[sel, 0, 1] = ws[sel, 1, 1] + rng.uniform(-0.1, 0.1, sel.sum())
    sel = k == 4                                                     # exactly on / denormally close to a car's centre
    ws[sel, 0, 0] = ws[sel, 1, 0] + rng.choice([0.0, 1e-40, -1e-38, 1e-33], sel.sum()).astype(np.float32)
    ws[sel, 0, 1] = ws[sel, 1, 1] + rng.choice([0.0, 1e-40, 1e-31], sel.sum()).astype(np.float32)
    sel = k == 5                                                     # far off the road: beyond the guard of the shortened forms
    ws[sel, 0, 0] = rng.choice([-1.0, 1.0], sel.sum()) * 10.0 ** rng.uniform(0, 30, sel.sum())
    sel = k == 6                                                     # a car so far away that its bump width rounds away
    ws[sel, 1, 0] = 10.0 ** rng.uniform(6, 30, sel.sum())
    return ws


@pytest.mark.parametrize("name,H", [("finite_horizon", 5), ("local_opt", 10), ("replanning", 5), ("merging", 10)])
def test_every_form_equals_reward_state_where_its_precondition_holds(hip, oracle, name, H):
    from l4dc_mpc_ocd_amd.engine import Engine
    scn = scenarios.SCENARIOS[name](horizon=H)
    d = scn.desc
    eng = Engine(scn, "cuda:0")
    rng = np.random.default_rng({"finite_horizon": 31, "local_opt": 32, "replanning": 33, "merging": 34}[name])
    n_valid = np.zeros(len(FORMS), dtype=np.int64)
    for rep in range(4):
        ws = world_states(scn, rng, 4096)
        wts = rng.standard_normal(d.n_features)
        if rep == 1:
            wts[rng.integers(0, d.n_features)] = 0.0
        w = (wts / np.linalg.norm(wts)).astype(np.float32)
        if rep == 3:                                                 # lanes at equal distance: reduce_min ties
            ws[:, 0, 0] = np.where(rng.random(4096) < 0.3, np.float32(0.5 * (d.lane_center[0] + d.lane_center[1])), ws[:, 0, 0])
        out, valid = eng.feature_variants(ws, w)
        assert valid[:, 0].all() and valid[:, 6].all()
        # the definition against the oracle: features' weighted sum (the adjoint is held by the objective tests)
        _, r_ref = oracle.reward_batch(d, ws, w)
        assert same(out[:, 0, 0], r_ref).all()
        for k in range(1, len(FORMS)):
            sel = valid[:, k]
            n_valid[k] += int(sel.sum())
            bad = ~same(out[sel, k], out[sel, 0]).all(axis=1)
            assert not bad.any(), (name, FORMS[k], int(bad.sum()), ws[sel][bad][:3], out[sel, k][bad][:3], out[sel, 0][bad][:3])
    # every form was exercised, the one-feature forms on most states (reward_two: two scripted cars only)
    two_cars = d.n_cars == 3
    assert (n_valid[1:9] > 500).all() and (not two_cars or (n_valid[9:] > 6000).all()), dict(zip(FORMS, n_valid))
    assert two_cars or (n_valid[9:] == 0).all()
    assert n_valid[1] > 4000 and n_valid[4] > 6000 and n_valid[8] > 6000
    print(name, dict(zip(FORMS, n_valid.tolist())))


def test_unsupported_shapes_are_refused(hip):
    from l4dc_mpc_ocd_amd.engine import Engine
    scn = scenarios.finite_horizon(horizon=5)
    d = abi.ScenarioDesc.from_buffer_copy(bytes(scn.desc))
    d.reward_kind = abi.OCD_REWARD_TARGET_SPEED
    d.n_lanes = 0
    eng = Engine(scenarios.Scenario("ts", d, scn.init_dist, None), "cuda:0")
    with pytest.raises(abi.OcdError):
        eng.feature_variants(np.zeros((4, 2, 4), dtype=np.float32), np.zeros(7, dtype=np.float32))
