"""Differential test over RANDOM scenario descriptors (not only the reference's four scenarios):
friction, learning rate, time step, target speed, lane geometry, bump sizes, scripted plans, teleport
step, control initialisations -- HIP path vs CPU oracle, bit for bit, plans and episodes."""
import numpy as np
import pytest

from l4dc_mpc_ocd_amd import abi, scenarios

pytestmark = pytest.mark.gpu

# (H, scripted cars, lanes): EVERY planning horizon 1..OCD_MAX_HORIZON (the reference takes any,
# naive_planner.py:19-26), cycling through 1-3 scripted cars x 1-4 lanes; horizons with a specialised
# kernel (OCD_KERNEL_TABLE) appear with their own (cars, lanes) pairs as well.
_PAIRS = [(1, 3), (2, 2), (2, 3), (3, 3), (1, 2), (3, 2), (1, 1), (2, 4), (1, 4), (3, 1), (2, 1), (3, 4)]
SHAPES = [(H, *_PAIRS[H % len(_PAIRS)]) for H in range(1, abi.OCD_MAX_HORIZON + 1)] + \
         [(5, 1, 3), (6, 1, 3), (10, 1, 3), (15, 1, 3), (25, 1, 3), (5, 2, 2), (10, 2, 2), (15, 2, 2),
          (5, 2, 3), (10, 2, 3), (25, 2, 3)]


def random_scenario(rng, H, NO, L):
    base = scenarios.finite_horizon(horizon=H)
    d = abi.ScenarioDesc.from_buffer_copy(bytes(base.desc))
    d.n_cars = NO + 1
    d.n_lanes = L
    d.n_iter = int(rng.integers(3, 25))
    d.extra_inits = int(rng.integers(0, 2))
    d.check_plans = int(rng.integers(0, 2))
    d.episode_len = int(rng.integers(2, 7))
    d.n_samples = int(rng.integers(1, 3))
    d.teleport_step = int(rng.integers(0, 4))
    d.teleport_period = int(rng.integers(0, abi.OCD_MAX_SAMPLES + 1))
    for s in range(abi.OCD_MAX_SAMPLES):
        d.teleport_car[s] = int(rng.integers(1, NO + 1)) if rng.random() < 0.7 else -1
    dt = float(rng.choice([0.05, 0.1, 0.2]))
    d.dt = dt
    d.dt_sq = np.float32(dt ** 2)
    d.learning_rate = float(rng.uniform(0.01, 0.6))
    d.ego_friction = float(rng.choice([0.0, 0.1, 0.2, 0.5]))
    d.target_speed = float(rng.uniform(0.5, 1.5))
    centers = np.sort(rng.uniform(-0.15, 0.15, L))
    for i in range(L):
        d.lane_center[i] = centers[i]
    d.lane_origin_y = float(rng.uniform(-6.0, -4.0))          # StraightLane.p[1]; the normal's y component stays 0
    d.lane_normal_y = 0.0 if rng.random() < 0.5 else -0.0
    d.fence_lo = np.float32(rng.uniform(0.0, 0.15))
    d.fence_width = float(rng.uniform(0.02, 0.08))
    d.fence_shape = float(np.float32(5.0 / float(np.float32(d.fence_width))))
    d.bump_half_x = float(rng.uniform(0.04, 0.12))
    d.bump_half_y = float(rng.uniform(0.1, 0.3))
    for j in range(NO):
        init = (rng.uniform(-0.1, 0.1), rng.uniform(-1.2, -0.4), rng.uniform(0.3, 1.0), np.pi / 2 + rng.uniform(-0.1, 0.1))
        for k in range(4):
            d.other_init[j][k] = init[k]
        d.other_friction[j] = float(rng.choice([0.0, 0.2]))
        n_plan = int(rng.integers(0, 6))
        d.other_plan_len[j] = n_plan
        for t in range(n_plan):
            d.other_plan[j][t][0], d.other_plan[j][t][1] = rng.uniform(-1, 1), rng.uniform(-3, 3)
        d.other_default[j][0], d.other_default[j][1] = rng.uniform(-0.2, 0.2), rng.uniform(-0.5, 0.5)
        if rng.random() < 0.5:      # what a check_plans planner assumes beyond the plan (planner_car.py:66-75)
            d.other_assumed_default[j][0], d.other_assumed_default[j][1] = d.other_default[j][0], d.other_default[j][1]
        else:
            d.other_assumed_default[j][0], d.other_assumed_default[j][1] = 0.0, 0.0
    w = rng.standard_normal(L + 4)
    w[L + 1:] = -np.abs(w[L + 1:]) * 3            # collision / fence / min-lane are costs, as in the reference
    w32 = (w / np.linalg.norm(w)).astype(np.float32)
    for i in range(L + 4):
        d.designer_weights[i] = w32[i]
    scn = scenarios.Scenario(f"random_H{H}_NO{NO}_L{L}", d, base.init_dist, None)
    return scn


@pytest.mark.parametrize("case", range(len(SHAPES) + 20))
def test_random_descriptor_bitwise(hip, oracle, case):
    from l4dc_mpc_ocd_amd.engine import Engine
    H, NO, L = SHAPES[case % len(SHAPES)]
    rng = np.random.default_rng(1000 + case)
    scn = random_scenario(rng, H, NO, L)
    d = scn.desc
    if case % 5 == 1 and L >= 2:                                    # coinciding lane centres: reduce_min ties in every pass
        d.lane_center[1] = d.lane_center[0]
    if case % 5 == 2 and NO >= 2:                                   # coinciding scripted cars: lanes with two active collisions
        for k in range(4):
            d.other_init[1][k] = d.other_init[0][k]
    eng = Engine(scn, "cuda:0")
    eng.set_option("scan_mode", int(rng.integers(0, 5)))            # any variant the shape allows
    phase = int(rng.integers(0, 3))
    eng.set_option("reset_phase", phase)
    eng.set_option("no_latency_build", case % 2)                    # latency and throughput builds of the DPP variants
    B = int(rng.integers(1, 12))
    ws = np.zeros((B, NO + 1, 4), dtype=np.float32)
    ws[:, 0, 0] = rng.uniform(-0.2, 0.2, B); ws[:, 0, 1] = rng.uniform(-1.4, -0.5, B)
    ws[:, 0, 2] = rng.uniform(0.2, 1.3, B); ws[:, 0, 3] = np.pi / 2 + rng.uniform(-0.4, 0.4, B)
    for j in range(NO):
        ws[:, j + 1] = np.array(d.other_init[j][:]) + rng.uniform(-0.05, 0.05, (B, 4))
    wts = rng.standard_normal((B, L + 4))
    wts[:, L + 1:] = -np.abs(wts[:, L + 1:]) * 2
    w32 = (wts / np.linalg.norm(wts, axis=1, keepdims=True)).astype(np.float32)
    ref = oracle.plan_batch(d, ws, w32, other_plans=scn.other_plans())
    out = eng.plan_batch(ws, w32, want_all=True)
    for k in ("all_losses", "all_plans", "plans", "best_loss"):
        a, b = out[k], ref[k]
        assert a.shape == b.shape and np.all((a == b) | (np.isnan(a) & np.isnan(b))), (scn.name, k)
    assert np.array_equal(out["best_init"], ref["best_init"])
    inits = ws[: min(B, 3), 0]
    ro = eng.rollout(inits, w32[:2], want_traj=True)
    rr = oracle.rollout(d, inits, w32[:2], want_traj=True, reset_phase=phase)
    for k in ("ctrl", "traj", "returns"):
        a, b = ro[k], rr[k]
        assert a.shape == b.shape and np.all((a == b) | (np.isnan(a) & np.isnan(b))), (scn.name, k)
