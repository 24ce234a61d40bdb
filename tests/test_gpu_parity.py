"""GPU parity: the HIP path (through the C ABI) against the CPU oracle, bit for bit.

The arithmetic contract makes kernel and oracle execute the
same IEEE binary32 operations in the same order, so every comparison here is
exact equality (np.array_equal on the raw values), not a tolerance.  The
BASELINE north_star tolerance (returns within 1e-4 rel) is implied.
"""
import numpy as np
import pytest

from l4dc_mpc_ocd_amd import scenarios

pytestmark = pytest.mark.gpu
PI_2 = np.pi / 2


@pytest.fixture(scope="module")
def eng_factory(hip):
    from l4dc_mpc_ocd_amd.engine import Engine
    cache = {}

    def make(scn):
        return Engine(scn, "cuda:0")
    return make


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_bitwise(a, b, what=""):
    a = np.asarray(a, dtype=np.float32)
    b = np.asarray(b, dtype=np.float32)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    # +0 / -0 compare equal on purpose; NaN must match NaN
    same = (a == b) | (np.isnan(a) & np.isnan(b))
    if not same.all():
        idx = np.argwhere(~same)[:5]
        raise AssertionError(f"{what}: {(~same).sum()} of {a.size} values differ, first at {idx.tolist()}: "
                             f"{a[tuple(idx[0])]!r} vs {b[tuple(idx[0])]!r}")


def test_device_math_is_bitwise_the_oracles(oracle, eng_factory):
    eng = eng_factory(scenarios.finite_horizon(horizon=5))
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-95, 3, 60000), rng.uniform(-8, 8, 40000), rng.uniform(-2e3, 2e3, 20000),
                        [0.0, -0.0, -87.0, -87.00001, 1.0, np.pi / 2, -np.pi / 2, 88.5]]).astype(np.float32)
    e, s, c = eng.debug_math(x)
    e_ref = np.array([oracle.expf(v) for v in x], dtype=np.float32)
    s_ref = np.array([oracle.sinf(v) for v in x], dtype=np.float32)
    c_ref = np.array([oracle.cosf(v) for v in x], dtype=np.float32)
    assert np.array_equal(bits(e), bits(e_ref))
    assert np.array_equal(bits(s), bits(s_ref))
    assert np.array_equal(bits(c), bits(c_ref))


def _world_states(scn, n, seed, spread=1.0):
    rng = np.random.default_rng(seed)
    d = scn.desc
    out = np.zeros((n, d.n_cars, 4))
    for j in range(1, d.n_cars):
        out[:, j, :] = np.array(d.other_init[j - 1][:])
    out[:, 0, :] = scn.init_dist.sample(n, seed=seed + 1)
    out[:, 0, 0] += spread * rng.uniform(-0.12, 0.12, n)
    out[:, 0, 3] += rng.uniform(-0.3, 0.3, n)
    out[:, 1:, 0] += rng.uniform(-0.05, 0.05, (n, d.n_cars - 1))
    out[:, 1:, 1] += rng.uniform(-0.2, 0.2, (n, d.n_cars - 1))
    # a third of the states start next to / inside a scripted car's collision bump
    m = n // 3
    out[:m, 0, 0] = out[:m, 1, 0] + rng.uniform(-0.07, 0.07, m)
    out[:m, 0, 1] = out[:m, 1, 1] + rng.uniform(-0.3, 0.05, m)
    return out.astype(np.float32)


@pytest.mark.parametrize("name", ["finite_horizon", "local_opt", "replanning", "merging"])
def test_reward_batch_bitwise(oracle, eng_factory, name):
    scn = scenarios.SCENARIOS[name](horizon=5)
    eng = eng_factory(scn)
    ws = _world_states(scn, 500, seed=5)
    feats, rew = eng.reward_batch(ws, scn.designer_weights)
    f_ref, r_ref = oracle.reward_batch(scn.desc, ws, scn.designer_weights)
    assert_bitwise(feats, f_ref, "features")
    assert_bitwise(rew, r_ref, "reward")


def test_planner_kats_on_gpu(oracle, eng_factory):
    """The reference's planner known-answer tests (test_naivePlanner.py:21-63) through the HIP path."""
    scn = scenarios.target_speed_kat(horizon=5, n_iter=100, learning_rate=5.0, friction=0.0)
    out = eng_factory(scn).plan_batch([[0., 0., 1., PI_2]], None, want_all=True)
    np.testing.assert_allclose(out["plans"][0], np.zeros((5, 2)), atol=1e-5)
    assert out["best_init"][0] == 0                      # first index wins the three-way tie
    ref = oracle.plan_batch(scn.desc, [[0., 0., 1., PI_2]], None)
    assert_bitwise(out["all_plans"], ref["all_plans"])

    scn = scenarios.target_speed_kat(horizon=3, n_iter=500, learning_rate=5.0, friction=0.5)
    out = eng_factory(scn).plan_batch([[0., 0., 1., PI_2]], None, want_all=True)
    for t in range(3):
        np.testing.assert_allclose(out["plans"][0, t], np.array([0.5, 0.]), atol=1e-5)
    ref = oracle.plan_batch(scn.desc, [[0., 0., 1., PI_2]], None)
    assert_bitwise(out["all_plans"], ref["all_plans"])
    assert_bitwise(out["all_losses"], ref["all_losses"])


@pytest.mark.parametrize("name,H,extra,n_iter", [
    ("finite_horizon", 5, False, 100), ("finite_horizon", 10, False, 60), ("finite_horizon", 6, True, 40),
    ("local_opt", 10, True, 50), ("local_opt", 3, False, 30), ("replanning", 5, False, 100),
    ("replanning", 15, False, 40), ("merging", 8, False, 40), ("merging", 25, False, 20),
    ("finite_horizon", 32, False, 10), ("finite_horizon", 16, False, 20),
])
def test_plan_batch_bitwise(oracle, eng_factory, name, H, extra, n_iter):
    kw = dict(horizon=H, n_iter=n_iter)
    if name in ("finite_horizon", "local_opt"):
        kw["extra_inits"] = extra
    scn = scenarios.SCENARIOS[name](**kw)
    eng = eng_factory(scn)
    B = 37                                               # ragged: not a multiple of the segments per wavefront
    ws = _world_states(scn, B, seed=H)
    w = np.stack([scenarios.planner_weights_fp32(c) for c in scn.candidate_weights(B, seed=H + 1)])
    out = eng.plan_batch(ws, w, want_all=True)
    ref = oracle.plan_batch(scn.desc, ws, w, other_plans=scn.other_plans())
    assert_bitwise(out["all_losses"], ref["all_losses"], "all_losses")
    assert_bitwise(out["all_plans"], ref["all_plans"], "all_plans")
    assert np.array_equal(out["best_init"], ref["best_init"])
    assert_bitwise(out["plans"], ref["plans"], "plans")
    assert_bitwise(out["best_loss"], ref["best_loss"], "best_loss")
    assert np.all(np.isfinite(out["plans"]))


@pytest.mark.parametrize("name,H,mode", [("finite_horizon", 6, 0), ("local_opt", 10, 0), ("local_opt", 10, 4), ("finite_horizon", 6, 1)])
def test_extra_inits_coast_at_the_cars_own_speed(oracle, eng_factory, name, H, mode):
    """naive_planner.py:114: the extra initialisations use friction * self.car.state[2] ** 2 -- the car object's
    speed, not the init_state argument's.  ocd_plan_batch_from takes that speed apart from the world state."""
    scn = scenarios.SCENARIOS[name](horizon=H, n_iter=30, extra_inits=True)
    eng = eng_factory(scn)
    eng.set_option("scan_mode", mode)
    B = 21
    ws = _world_states(scn, B, seed=3 * H)
    w = np.stack([scenarios.planner_weights_fp32(c) for c in scn.candidate_weights(B, seed=H + 2)])
    own = (ws[:, 0, 2] + np.random.default_rng(H).uniform(0.2, 0.8, B)).astype(np.float32)
    out = eng.plan_batch(ws, w, want_all=True, init_speed=own)
    ref = oracle.plan_batch(scn.desc, ws, w, other_plans=scn.other_plans(), init_speed=own)
    assert_bitwise(out["all_plans"], ref["all_plans"], "all_plans")
    assert_bitwise(out["all_losses"], ref["all_losses"], "all_losses")
    assert np.array_equal(out["best_init"], ref["best_init"])
    same_speed = eng.plan_batch(ws, w, want_all=True)
    assert_bitwise(same_speed["all_plans"][:, :3], out["all_plans"][:, :3], "the three plain initialisations do not change")
    assert not np.array_equal(same_speed["all_plans"][:, 3:], out["all_plans"][:, 3:])
    assert_bitwise(eng.plan_batch(ws, w, want_all=True, init_speed=ws[:, 0, 2])["all_plans"], same_speed["all_plans"], "own speed == state speed")


def test_plan_batch_shared_weights_and_single_problem(oracle, eng_factory):
    scn = scenarios.finite_horizon(horizon=5, n_iter=30)
    eng = eng_factory(scn)
    ws = _world_states(scn, 1, seed=1)
    w = scenarios.planner_weights_fp32(scn.designer_weights)
    out = eng.plan_batch(ws, w, want_all=True)           # [D] weights shared by every problem
    ref = oracle.plan_batch(scn.desc, ws, w)
    assert_bitwise(out["all_plans"], ref["all_plans"])
    out0 = eng.plan_batch(np.zeros((0, 2, 4), dtype=np.float32), w)   # empty batch is a no-op
    assert out0["plans"].shape == (0, 5, 2)


@pytest.mark.parametrize("name,H,P,N", [
    ("finite_horizon", 5, 2, 3), ("local_opt", 10, 3, 2), ("replanning", 5, 2, 3), ("merging", 5, 2, 2),
    ("replanning", 15, 1, 2),
])
def test_rollout_bitwise(oracle, eng_factory, name, H, P, N):
    scn = scenarios.SCENARIOS[name](horizon=H)
    eng = eng_factory(scn)
    inits = scn.init_dist.sample(N, seed=10 + H)
    w = np.stack([scenarios.planner_weights_fp32(c) for c in scn.candidate_weights(P, seed=20 + H)])
    out = eng.rollout(inits, w, want_traj=True)
    ref = oracle.rollout(scn.desc, inits, w, want_traj=True)
    assert_bitwise(out["ctrl"], ref["ctrl"], "applied controls")
    assert_bitwise(out["traj"], ref["traj"], "trajectories")
    assert_bitwise(out["returns"], ref["returns"], "returns")
    assert out["returns"].shape == (P * N * scn.desc.n_samples,)


@pytest.mark.parametrize("scan_mode", [1, 2, 3, 4])
@pytest.mark.parametrize("name,H,n_iter", [("finite_horizon", 10, 40), ("replanning", 15, 25), ("merging", 5, 40),
                                           ("local_opt", 16, 15), ("finite_horizon", 3, 30), ("merging", 10, 20),
                                           ("merging", 25, 12), ("local_opt", 25, 12)])
def test_all_scan_variants_bitwise(oracle, eng_factory, hip, scan_mode, name, H, n_iter):
    """scan_mode 1 (LDS windows), 2 (DPP row shifts, H <= 16), 3 (all K initialisations in one
    wavefront, K*H <= 64) and 4 (a lane owns a chunk of consecutive steps) are four implementations of the same
    recurrences; all must reproduce the oracle bit for bit, plans, losses and episodes.  (A shape without
    the requested kernel runs the LDS kernel.)"""
    scn = scenarios.SCENARIOS[name](horizon=H, n_iter=n_iter)
    eng = eng_factory(scn)
    B = 21
    ws = _world_states(scn, B, seed=H + 3)
    ws[3, 0, 2] = -40.0                                   # one problem that overflows inside the horizon
    w = np.stack([scenarios.planner_weights_fp32(c) for c in scn.candidate_weights(B, seed=H + 4)])
    ref = oracle.plan_batch(scn.desc, ws, w, other_plans=scn.other_plans())
    inits = scn.init_dist.sample(3, seed=H + 5)
    rr = oracle.rollout(scn.desc, inits, w[:2], want_traj=True)
    eng.set_option("scan_mode", scan_mode)
    try:
        out = eng.plan_batch(ws, w, want_all=True)
        ro = eng.rollout(inits, w[:2], want_traj=True)
        eng.set_option("segs_per_wave", 3)          # several trajectories per wavefront as well
        out3 = eng.plan_batch(ws, w, want_all=True)
        eng.set_option("no_latency_build", 1)       # the throughput builds of the DPP variants (small launches pick LAT)
        out4 = eng.plan_batch(ws, w, want_all=True)
        ro4 = eng.rollout(inits, w[:2], want_traj=True)
    finally:
        eng.set_option("scan_mode", 0)
        eng.set_option("segs_per_wave", 0)
        eng.set_option("no_latency_build", 0)
    assert_bitwise(ro4["traj"], rr["traj"], "traj"); assert_bitwise(ro4["returns"], rr["returns"], "returns")
    for o in (out, out3, out4):
        assert_bitwise(o["all_losses"], ref["all_losses"], "losses"); assert_bitwise(o["all_plans"], ref["all_plans"], "plans")
        assert np.array_equal(o["best_init"], ref["best_init"]); assert_bitwise(o["best_loss"], ref["best_loss"], "best loss")
    assert_bitwise(ro["ctrl"], rr["ctrl"], "controls"); assert_bitwise(ro["traj"], rr["traj"], "traj")
    assert_bitwise(ro["returns"], rr["returns"], "returns")


@pytest.mark.parametrize("name,H,chunk", [("finite_horizon", 10, 2), ("finite_horizon", 10, 5), ("local_opt", 15, 2),
                                          ("local_opt", 15, 3), ("local_opt", 15, 5), ("replanning", 15, 2),
                                          ("replanning", 15, 3), ("replanning", 10, 2), ("replanning", 10, 5),
                                          ("merging", 25, 2), ("merging", 25, 3), ("merging", 25, 5), ("local_opt", 25, 2), ("local_opt", 25, 3),
                                          ("merging", 10, 2), ("finite_horizon+", 10, 2), ("local_opt+", 15, 2),
                                          ("local_opt+", 15, 3), ("finite_horizon+", 25, 3), ("local_opt+", 25, 5)])
def test_chunk_sizes_bitwise(oracle, eng_factory, hip, name, H, chunk):
    """V_CHUNK with every compiled chunk size S, including sizes that do not divide the horizon (the last lane of a
    segment then owns H - (NC-1)*S steps and padding): latency build, throughput build (no_latency_build), the
    three-wavefront build (a launch of >= 3 wavefronts per SIMD is too big for a test; its code differs only in
    register allocation), full and partial packing, an overflowing problem among the others.  "+" = extra_inits
    (six control initialisations per trajectory: naive_planner.py:112-116)."""
    extra = name.endswith("+")
    scn = scenarios.SCENARIOS[name.rstrip("+")](horizon=H, n_iter=18, **({"extra_inits": True} if extra else {}))
    eng = eng_factory(scn)
    B = 23
    ws = _world_states(scn, B, seed=H + chunk)
    ws[5, 0, 2] = -40.0                                   # overflows inside the horizon
    w = np.stack([scenarios.planner_weights_fp32(c) for c in scn.candidate_weights(B, seed=H + 4)])
    ref = oracle.plan_batch(scn.desc, ws, w, other_plans=scn.other_plans())
    inits = scn.init_dist.sample(3, seed=H + 5)
    rr = oracle.rollout(scn.desc, inits, w[:2], want_traj=True)
    eng.set_option("scan_mode", 4)
    eng.set_option("chunk_size", chunk)
    try:
        outs = []
        for segs, nolat in ((0, 0), (1, 0), (0, 1), (2, 1)):
            eng.set_option("segs_per_wave", segs)
            eng.set_option("no_latency_build", nolat)
            outs.append(eng.plan_batch(ws, w, want_all=True))
            ll = eng.last_launch()
            assert (ll["scan_mode"], ll["chunk"], ll["specialised_horizon"]) == (4, chunk, H), ll
            cap = 64 // (scn.desc.n_ctrl_inits * -(-H // chunk))
            assert ll["build_wavefronts_per_simd"] == (0 if nolat else 1)
            assert segs == 0 or ll["trajectories_per_wavefront"] == min(segs, cap)
            ro = eng.rollout(inits, w[:2], want_traj=True)
            assert_bitwise(ro["ctrl"], rr["ctrl"], "controls"); assert_bitwise(ro["traj"], rr["traj"], "traj")
            assert_bitwise(ro["returns"], rr["returns"], "returns")
    finally:
        eng.set_option("scan_mode", 0); eng.set_option("chunk_size", 0)
        eng.set_option("segs_per_wave", 0); eng.set_option("no_latency_build", 0)
    for o in outs:
        assert_bitwise(o["all_losses"], ref["all_losses"], "losses"); assert_bitwise(o["all_plans"], ref["all_plans"], "plans")
        assert np.array_equal(o["best_init"], ref["best_init"]); assert_bitwise(o["best_loss"], ref["best_loss"], "best loss")
        assert_bitwise(o["plans"], ref["plans"], "plans")


@pytest.mark.parametrize("segs,mode", [(1, 0), (2, 0), (6, 1), (0, 0), (1, 3), (2, 3), (3, 2), (0, 4), (3, 4), (1, 4)])
def test_results_do_not_depend_on_packing(oracle, eng_factory, hip, segs, mode):
    """segs_per_wave (trajectories per wavefront) is a pure performance knob: packed lanes, parked
    lanes and the wave-uniform feature skips must not change a bit."""
    scn = scenarios.finite_horizon(horizon=10, n_iter=40)
    eng = eng_factory(scn)
    ws = _world_states(scn, 29, seed=77)
    w = np.stack([scenarios.planner_weights_fp32(c) for c in scn.candidate_weights(29, seed=78)])
    ref = oracle.plan_batch(scn.desc, ws, w)
    eng.set_option("segs_per_wave", segs)
    eng.set_option("scan_mode", mode)                  # 6 per wavefront exists only with LDS windows
    eng.set_option("no_feature_skips", segs % 2)       # skips on and off
    eng.set_option("no_unified_features", (segs // 2) % 2)   # shared exp(-1/u) path on and off
    try:
        out = eng.plan_batch(ws, w, want_all=True)
        inits = scn.init_dist.sample(5, seed=79)
        ro = eng.rollout(inits, w[:2], want_traj=True)
    finally:
        eng.set_option("segs_per_wave", 0)
        eng.set_option("no_feature_skips", 0)
        eng.set_option("no_unified_features", 0)
        eng.set_option("scan_mode", 0)
    assert_bitwise(out["all_plans"], ref["all_plans"]); assert_bitwise(out["all_losses"], ref["all_losses"])
    rr = oracle.rollout(scn.desc, inits, w[:2], want_traj=True)
    assert_bitwise(ro["traj"], rr["traj"]); assert_bitwise(ro["returns"], rr["returns"])
    from l4dc_mpc_ocd_amd import abi
    with pytest.raises(abi.OcdError):
        eng.set_option("nonsense", 1)


@pytest.mark.parametrize("segs,mode", [(1, 0), (0, 0), (6, 1), (0, 1), (1, 3), (2, 3), (2, 2), (0, 4), (2, 4)])
def test_non_finite_trajectories_bitwise(oracle, eng_factory, hip, segs, mode):
    """Hard braking drives v negative and the drag term -f*v^2 then runs away to -inf inside the
    horizon (car_dynamics_step has no speed floor): rewards become -inf, losses +inf, some adjoints
    NaN.  The kernels must reproduce the oracle through inf and NaN, in both scan variants."""
    scn = scenarios.finite_horizon(horizon=10, n_iter=30)
    eng = eng_factory(scn)
    B = 24
    ws = _world_states(scn, B, seed=5)
    ws[:, 0, 2] = np.linspace(-45.0, -5.0, B)          # large negative speeds: overflow after a few steps
    ws[::5, 0, 2] = 0.8                                # with some ordinary problems mixed in
    w = np.stack([scenarios.planner_weights_fp32(c) for c in scn.candidate_weights(B, seed=6)])
    ref = oracle.plan_batch(scn.desc, ws, w)
    assert np.isinf(ref["all_losses"]).any() or np.isnan(ref["all_losses"]).any()
    eng.set_option("segs_per_wave", segs)
    eng.set_option("scan_mode", mode)
    eng.set_option("no_latency_build", segs % 2)       # both builds of the DPP variants
    try:
        out = eng.plan_batch(ws, w, want_all=True)
        ro = eng.rollout_from_state(ws, w, first_step=0, n_steps=8)
    finally:
        eng.set_option("segs_per_wave", 0)
        eng.set_option("no_latency_build", 0)
    assert_bitwise(out["all_losses"], ref["all_losses"], "losses")
    assert_bitwise(out["all_plans"], ref["all_plans"], "plans")
    assert np.array_equal(out["best_init"], ref["best_init"])
    rr = oracle.rollout_from_state(scn.desc, ws, w, 0, 8)
    assert not np.all(np.isfinite(rr["traj"]))          # the WORLD state itself overflows: exact-select fallback
    assert_bitwise(ro["traj"], rr["traj"], "traj"); assert_bitwise(ro["returns"], rr["returns"], "returns")


def test_rollout_episode_range_matches_full(oracle, eng_factory):
    """Sharding contract: any [ep_begin, ep_end) slice equals the same slice of the full run."""
    scn = scenarios.replanning(horizon=5)
    eng = eng_factory(scn)
    inits = scn.init_dist.sample(3, seed=4)
    w = np.stack([scenarios.planner_weights_fp32(c) for c in scn.candidate_weights(3, seed=5)])
    full = eng.rollout(inits, w)["returns"]
    assert full.shape == (18,)
    for b, e in [(0, 5), (5, 11), (11, 18), (7, 8), (9, 9)]:
        part = eng.rollout(inits, w, ep_begin=b, ep_end=e)["returns"]
        assert_bitwise(part, full[b:e], f"slice {b}:{e}")


def test_baseline_config2_bitwise(oracle, eng_factory):
    """BASELINE config 2 (finite_horizon, pop 16 x 8 inits, H=10) at full size, every episode."""
    scn, inits, cands = scenarios.baseline_config(2)
    w = np.stack([scenarios.planner_weights_fp32(c) for c in cands])
    out = eng_factory(scn).rollout(inits, w)
    ref = oracle.rollout(scn.desc, inits, w)
    assert_bitwise(out["returns"], ref["returns"], "cfg2 returns")
    rel = np.abs(out["returns"] - ref["returns"]) / np.abs(ref["returns"])
    assert rel.max() <= 1e-4                            # the north_star tolerance, trivially


def test_errors_are_reported_not_thrown(hip, eng_factory):
    import ctypes as C
    from l4dc_mpc_ocd_amd import abi
    bad = scenarios.finite_horizon(horizon=abi.OCD_MAX_HORIZON + 1).desc   # beyond the header limit
    h = C.c_void_p()
    assert hip.ocd_scenario_create(C.byref(bad), C.byref(h)) == abi.OCD_ERR_INVALID_ARG
    assert b"horizon" in hip.ocd_last_error()
    bad = scenarios.finite_horizon(horizon=5).desc
    bad.n_cars = 9
    h = C.c_void_p()
    assert hip.ocd_scenario_create(C.byref(bad), C.byref(h)) == abi.OCD_ERR_INVALID_ARG
    assert b"n_cars" in hip.ocd_last_error()


@pytest.mark.parametrize("mode", [0, 1, 2, 3, 4])
def test_finite_speed_with_overflowing_drag_bitwise(oracle, eng_factory, mode):
    """|v| ~ 3e19 is finite but fr * v * v overflows: the LDS variant's masked fma needs a finite increment
    (fma(-inf, 0, v) is NaN), so it must take its exact-select fallback exactly when the other variants
    keep v; every variant has to reproduce the oracle."""
    scn = scenarios.finite_horizon(horizon=10, n_iter=5)
    eng = eng_factory(scn)
    eng.set_option("scan_mode", mode)
    ws = _world_states(scn, 9, seed=11)
    ws[:, 0, 2] = [3e19, -3e19, 1.9e19, 1.0, 2.5e19, -1e20, 0.5, 4e19, 1e18]
    w = np.stack([scenarios.planner_weights_fp32(c) for c in scn.candidate_weights(9, seed=12)])
    ref = oracle.plan_batch(scn.desc, ws, w)
    out = eng.plan_batch(ws, w, want_all=True)
    assert_bitwise(out["all_plans"], ref["all_plans"], "plans"); assert_bitwise(out["all_losses"], ref["all_losses"], "losses")
    rr = oracle.rollout_from_state(scn.desc, ws, w, 0, 3)
    ro = eng.rollout_from_state(ws, w, first_step=0, n_steps=3)
    assert_bitwise(ro["traj"], rr["traj"], "traj"); assert_bitwise(ro["returns"], rr["returns"], "returns")
