"""The oracle's terminal value (ValueFeature.interpolate_value, value_interpolation.py:28-61) against an
independent float64 numpy trilinear interpolation and central finite differences.  The reference holds no
vector for it (no scenario sets leaf_evaluation): PARITY UNPINNED, this is the second opinion."""
import numpy as np

from l4dc_mpc_ocd_amd import scenarios


def trilinear64(grid, vals, p):
    idx, frac = [], []
    for g, x in zip(grid, p):
        g = np.asarray(g, dtype=np.float64)
        if x < g[0] or x > g[-1]:
            return np.nan
        c = min(int(np.searchsorted(g, x, side="right") - 1), len(g) - 2)
        idx.append(c); frac.append((x - g[c]) / (g[c + 1] - g[c]))
    out = 0.0
    for i0 in range(2):
        for i1 in range(2):
            for i2 in range(2):
                wgt = (frac[0] if i0 else 1 - frac[0]) * (frac[1] if i1 else 1 - frac[1]) * (frac[2] if i2 else 1 - frac[2])
                out += wgt * float(vals[idx[0] + i0, idx[1] + i1, idx[2] + i2])
    return out


def test_terminal_value_matches_numpy_trilinear(oracle):
    rng = np.random.default_rng(0)
    grid = [np.linspace(-0.3, 0.3, 9).astype(np.float32), np.sort(rng.uniform(-2.5, 2.5, 14)).astype(np.float32),
            np.linspace(0.0, 3.0, 7).astype(np.float32)]
    vals = rng.standard_normal((9, 14, 7)).astype(np.float32)
    scn = scenarios.finite_horizon(horizon=1)                       # H = 1: the objective IS the terminal value
    ws = np.zeros((2, 4), dtype=np.float32)
    ws[1] = np.array(scn.desc.other_init[0][:])
    w = scenarios.planner_weights_fp32(scn.raw_designer_weights)
    dt, f = float(scn.desc.dt), float(scn.desc.ego_friction)
    for proj_kind in (0, 1):
        oracle.set_leaf_value(grid, vals, proj_kind)
        try:
            for _ in range(200):
                ws[0] = [rng.uniform(-0.25, 0.25), rng.uniform(-2.0, 2.0), rng.uniform(0.3, 2.0), np.pi / 2 + rng.uniform(-0.5, 0.5)]
                u = np.array([[rng.uniform(-1, 1), rng.uniform(-1, 1)]], dtype=np.float32)
                r, g, tr = oracle.mpc_reward(scn.desc, ws, w, u)
                x, y, v, th = [float(t) for t in tr[0]]
                coarse = (x, y, v if proj_kind == 0 else v * np.sin(th))
                want = trilinear64(grid, vals, coarse)
                assert np.isnan(want) == np.isnan(r)
                if np.isnan(want):
                    continue
                assert abs(float(r) - want) <= 2e-5 * max(1.0, abs(want))
                # gradient w.r.t. the control by central differences of the float64 interpolation of the oracle's own
                # post-step state map (car_dynamics_step): only where the step stays inside one grid cell
                eps = 1e-3
                def R(du):
                    a, om = float(u[0, 0]) + du[0], float(u[0, 1]) + du[1]
                    x0, y0, v0, t0 = [float(t) for t in ws[0]]
                    acc = a - f * v0 * v0
                    dd = v0 * dt + 0.5 * acc * dt * dt
                    xs, ys, vs, ts = x0 + np.cos(t0) * dd, y0 + np.sin(t0) * dd, v0 + acc * dt, t0 + om * dt
                    return trilinear64(grid, vals, (xs, ys, vs if proj_kind == 0 else vs * np.sin(ts)))
                fd = np.array([(R((eps, 0)) - R((-eps, 0))) / (2 * eps), (R((0, eps)) - R((0, -eps))) / (2 * eps)])
                if np.all(np.isfinite(fd)) and abs(R((eps, 0)) + R((-eps, 0)) - 2 * want) < 1e-7 and abs(R((0, eps)) + R((0, -eps)) - 2 * want) < 1e-7:
                    np.testing.assert_allclose(g[0], fd, rtol=2e-2, atol=2e-4)
        finally:
            oracle.set_leaf_value(None, None)


def test_out_of_grid_terminal_state_gives_nan_value_and_zero_gradient(oracle):
    """value_interpolation.py:59-60: outside the grid the traced function returns the constant float('nan'), so the
    objective is NaN but the terminal step sends NO gradient back: the controls keep the finite gradient of the
    other horizon steps, and generate_plan returns finite controls."""
    rng = np.random.default_rng(3)
    grid = [np.linspace(-0.3, 0.3, 5).astype(np.float32), np.linspace(-1.0, 1.0, 6).astype(np.float32),
            np.linspace(0.0, 3.0, 4).astype(np.float32)]
    vals = rng.standard_normal((5, 6, 4)).astype(np.float32)
    scn = scenarios.finite_horizon(horizon=4)
    ws = np.zeros((2, 4), dtype=np.float32)
    ws[0] = [0.02, 5.0, 1.0, np.pi / 2]                              # y = 5 is far outside the y grid [-1, 1]
    ws[1] = np.array(scn.desc.other_init[0][:])
    w = scenarios.planner_weights_fp32(scn.raw_designer_weights)
    u = rng.uniform(-0.5, 0.5, (4, 2)).astype(np.float32)
    r0, g0, _ = oracle.mpc_reward(scn.desc, ws, w, u)               # no terminal value: the reference gradient
    oracle.set_leaf_value(grid, vals, 0)
    try:
        r, g, _ = oracle.mpc_reward(scn.desc, ws, w, u)
        assert np.isnan(r) and np.all(np.isfinite(g))
        # the terminal step contributes nothing: the gradient is that of the first H-1 rewards alone, i.e. the plain
        # objective's minus what its own last-step reward contributed -- and the last control gets exactly zero
        assert np.array_equal(g[-1], np.zeros(2, dtype=np.float32))
        plan = oracle.plan_batch(scn.desc, ws[None], w)
        assert np.all(np.isfinite(plan["plans"])) and np.all(np.isnan(plan["best_loss"]))
    finally:
        oracle.set_leaf_value(None, None)
    assert np.isfinite(r0) and np.all(np.isfinite(g0))
