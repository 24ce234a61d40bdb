"""ABI 3 on the GPU, against the CPU oracle:

* the SCORED reward carries StraightLane.dist2median's y-term (world.py:216-217): an ego beyond the finite numbers scores NaN,
  as the reference's expression does (tests/test_lane_offset_y_term.py holds the oracle to the float64 torch episode);
* an index row of ocd_rollout_indexed that names no candidate / init row is an error, not a clamp (the C client checks the
  pinned-host path: tests/test_abi_c_client.py)."""
import numpy as np
import pytest

from l4dc_mpc_ocd_amd import abi, scenarios

pytestmark = pytest.mark.gpu
PI_2 = np.pi / 2


def same_bits(a, b):
    a, b = np.asarray(a, dtype=np.float32), np.asarray(b, dtype=np.float32)
    return bool(((a == b) | (np.isnan(a) & np.isnan(b))).all())


@pytest.mark.parametrize("name,H", [("finite_horizon", 5), ("local_opt", 10), ("replanning", 15), ("merging", 25)])
def test_episodes_that_leave_the_finite_numbers_score_nan_like_the_oracle(hip, oracle, name, H):
    from l4dc_mpc_ocd_amd.engine import Engine
    scn = scenarios.SCENARIOS[name](horizon=H, n_iter=12)
    d = scn.desc
    inits = np.asarray(scn.init_dist.sample(8, seed=3), dtype=np.float32)
    inits[1, 1] = np.inf                 # y = +inf: every lane distance of the scored reward is NaN
    inits[2, 1] = -np.inf
    inits[3, 1] = np.nan
    inits[4, 2] = 3e19                   # v^2 overflows in the first real step: y = -inf from the second score on
    inits[5, 0] = np.inf                 # x = +inf: NaN through the zero-weight lane features either way
    w = np.stack([scenarios.planner_weights_fp32(c) for c in scn.candidate_weights(3, seed=4)])
    eng = Engine(scn, "cuda:0")
    ref = oracle.rollout(d, inits, w, want_traj=True)
    for mode in (0, 1, 2, 3, 4):
        eng.set_option("scan_mode", mode)
        got = eng.rollout(inits, w, want_traj=True)
        for k in ("returns", "traj", "ctrl"):
            assert same_bits(got[k], ref[k]), (name, mode, k)
    ret = ref["returns"].reshape(3, 8, d.n_samples)
    assert np.isnan(ret[:, 1:6]).all() and np.isfinite(ret[:, [0, 6, 7]]).all()


def test_reward_batch_is_the_scored_form(hip, oracle):
    from l4dc_mpc_ocd_amd.engine import Engine
    scn = scenarios.merging(horizon=5)
    d = scn.desc
    rng = np.random.default_rng(2)
    ws = np.zeros((64, d.n_cars, 4), dtype=np.float32)
    for j in range(d.n_cars - 1):
        ws[:, j + 1] = [d.other_init[j][k] for k in range(4)]
    ws[:, 0, 0] = rng.uniform(-0.2, 0.2, 64)
    ws[:, 0, 1] = rng.uniform(-2.0, -1.0, 64)
    ws[:, 0, 2] = rng.uniform(0.5, 1.0, 64)
    ws[:, 0, 3] = PI_2
    ws[::4, 0, 1] = [np.inf, -np.inf, np.nan, 3e38] * 4
    wt = scn.designer_weights.astype(np.float32)
    feats, rew = Engine(scn, "cuda:0").reward_batch(ws, wt)
    for b in range(64):
        r, f, _ = oracle.reward(d, ws[b], wt, want_grad=False)
        assert same_bits(rew[b], r) and same_bits(feats[b], f), b
    bad = ~np.isfinite(ws[:, 0, 1])
    assert np.isnan(rew[bad]).all() and np.isfinite(rew[~bad]).all()


def test_a_lane_normal_with_a_y_component_is_unsupported(hip):
    from l4dc_mpc_ocd_amd.engine import Engine
    scn = scenarios.finite_horizon(horizon=5)
    scn.desc.lane_normal_y = 0.25
    with pytest.raises(Exception, match="lane_normal_y"):
        Engine(scn, "cuda:0")


def test_a_device_index_row_out_of_range_is_an_error_and_a_nan(hip, oracle):
    """Engine.rollout_indexed(check_index=False): the index lives in device memory, so the kernel is what finds the row."""
    import torch
    from l4dc_mpc_ocd_amd.engine import Engine
    scn = scenarios.replanning(horizon=5, n_iter=10)
    d = scn.desc
    inits = np.asarray(scn.init_dist.sample(4, seed=1), dtype=np.float32)
    w = np.stack([scenarios.planner_weights_fp32(c) for c in scn.candidate_weights(3, seed=2)])
    idx = np.array([(p, n, (p * 4 + n) * d.n_samples + s) for p in range(3) for n in range(4) for s in range(d.n_samples)], dtype=np.int32)
    eng = Engine(scn, "cuda:0")
    good = eng.rollout_indexed(inits, w, idx)["returns"]
    assert same_bits(good, oracle.rollout(d, inits, w)["returns"].reshape(-1))
    for col, val in ((0, 3), (0, -1), (1, 4), (1, -7), (2, -1)):
        bad = idx.copy()
        bad[5, col] = val
        with pytest.raises(ValueError):                             # the Python wrapper's own check
            eng.rollout_indexed(inits, w, bad)
        out = eng.rollout_indexed(inits, w, bad, to_numpy=False, check_index=False)
        torch.cuda.synchronize()
        ret = out["returns"].cpu().numpy()
        assert np.isnan(ret[5]) and same_bits(np.delete(ret, 5), np.delete(good, 5)), (col, val)
        row = abi.C.c_int64(-2)
        assert hip.ocd_scenario_index_error(eng._h, abi.C.byref(row)) == abi.OCD_ERR_INVALID_ARG and row.value == 5
        assert b"row 5" in hip.ocd_last_error()
        assert hip.ocd_scenario_index_error(eng._h, None) == abi.OCD_OK
    # the wrapper raises by itself after the wait
    bad = idx.copy()
    bad[9, 0] = 99
    with pytest.raises(Exception, match="row 9"):
        eng.rollout_indexed(inits, w, bad, check_index=False)
    assert same_bits(eng.rollout_indexed(inits, w, idx)["returns"], good)
