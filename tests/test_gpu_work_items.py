"""The work-item lists of the chunked kernel's shared-SIMD builds at their limits (csrc/ocd_chunk_kernel.hip: horizon_pass,
csrc/ocd_device.h: reward_base_grad / feature_item_grad): the gradient passes of those builds append one item per active
(lane, step, feature) to a list in LDS, evaluate it 64 items at a time and hand the adjoint terms back.  The planner tests
of tests/test_gpu_parity.py run that path on ordinary states (every chunk size, no_latency_build); here the states are
chosen to fill it -- every state of every trajectory on the fence AND inside a car's box (two items per pair: the list
overflows and the step falls back to the evaluation of every feature), every state inside BOTH cars' boxes (pairs of items
from an even slot; exact ties of reduce_max) -- and the plans must still be the CPU oracle's, bit for bit.
Reference: naive_planner.py:33-77,81-164, merging.py:44-83 (reduce_max over the cars: merging.py:78)."""
import numpy as np
import pytest

from l4dc_mpc_ocd_amd import abi, scenarios

pytestmark = pytest.mark.gpu


def assert_bitwise(a, b, what=""):
    a, b = np.asarray(a, dtype=np.float32), np.asarray(b, dtype=np.float32)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    same = (a == b) | (np.isnan(a) & np.isnan(b))
    assert same.all(), f"{what}: {(~same).sum()} of {a.size} values differ, first at {np.argwhere(~same)[:4].tolist()}"


def crowded_states(scn, n, rng, kind):
    """World states whose whole horizon stays where the features are: slow egos (the plan barely moves them) ...
      'fence+car': on the fence band with a resting car on top of them;  'both cars': inside both cars' boxes, a third of
      them with the two cars on EXACTLY the same spot (the two collision products tie);  'everything': both at once."""
    d = scn.desc
    C = d.n_cars
    lo, wd = float(d.fence_lo), float(d.fence_width)
    ws = np.zeros((n, C, 4), dtype=np.float32)
    ws[:, 0, 1] = rng.uniform(-0.2, 0.2, n)
    ws[:, 0, 2] = rng.uniform(0.0, 0.05, n)
    ws[:, 0, 3] = np.pi / 2 + rng.uniform(-0.05, 0.05, n)
    on_fence = kind in ("fence+car", "everything")
    ws[:, 0, 0] = rng.choice([-1.0, 1.0], n) * (lo + wd * rng.uniform(0.05, 0.9, n)) if on_fence else rng.uniform(-0.1, 0.1, n)
    for j in range(1, C):
        ws[:, j, 0] = ws[:, 0, 0] + rng.uniform(-0.04, 0.04, n)
        ws[:, j, 1] = ws[:, 0, 1] + rng.uniform(-0.08, 0.08, n)
        ws[:, j, 2] = 0.0                                            # resting: the planner's prediction keeps them there
        ws[:, j, 3] = np.pi / 2
    if C > 2:
        if kind == "fence+car":
            ws[:, 2, 1] += 5.0                                       # the second car far ahead: one car per state
        else:
            tie = rng.random(n) < 0.34
            ws[tie, 2, :2] = ws[tie, 1, :2]
    return ws


CASES = [("local_opt", 10, 2, "fence+car"), ("local_opt", 10, 5, "fence+car"), ("replanning", 15, 2, "fence+car"),
         ("replanning", 15, 3, "both cars"), ("replanning", 15, 2, "both cars"), ("replanning", 15, 5, "both cars"),
         ("replanning", 15, 3, "everything"), ("merging", 25, 5, "fence+car"), ("merging", 25, 5, "both cars"),
         ("merging", 25, 3, "everything"), ("merging", 10, 2, "both cars")]


@pytest.mark.parametrize("name,H,chunk,kind", CASES)
def test_crowded_states_through_the_item_lists_bitwise(hip, oracle, name, H, chunk, kind):
    from l4dc_mpc_ocd_amd.engine import Engine
    scn = scenarios.SCENARIOS[name](horizon=H, n_iter=12)
    d = scn.desc
    rng = np.random.default_rng(7000 + H * 10 + chunk + len(kind))
    cap = 64 // (d.n_ctrl_inits * -(-H // chunk))
    B = 3 * cap + 1                                                  # full wavefronts and a partly filled one
    ws = crowded_states(scn, B, rng, kind)
    w = np.stack([scenarios.planner_weights_fp32(c) for c in scn.candidate_weights(B, seed=H + chunk)])
    ref = oracle.plan_batch(d, ws, w, other_plans=scn.other_plans())
    # the states are where they are meant to be: the oracle's own features say so
    feats = np.array([oracle.reward(d, ws[b], w[b])[1] for b in range(B)])
    L = d.n_lanes
    if kind != "both cars":
        assert (feats[:, L + 3] != 0).mean() > 0.9                   # fence feature active
    assert (feats[:, L + 2] != 0).mean() > 0.9                       # collision feature active
    eng = Engine(scn, "cuda:0")
    eng.set_option("scan_mode", 4)
    eng.set_option("chunk_size", chunk)
    eng.set_option("no_latency_build", 1)                            # the builds with the item lists
    out = eng.plan_batch(ws, w, want_all=True)
    ll = eng.last_launch()
    assert (ll["scan_mode"], ll["chunk"], ll["build_wavefronts_per_simd"]) == (4, chunk, 0), ll
    assert_bitwise(out["all_losses"], ref["all_losses"], "losses")
    assert_bitwise(out["all_plans"], ref["all_plans"], "plans of every initialisation")
    assert np.array_equal(out["best_init"], ref["best_init"])
    assert_bitwise(out["plans"], ref["plans"], "plans")
    # the latency build of the same launch (every pair through the straight-line evaluation): the same bits
    eng.set_option("no_latency_build", 0)
    out2 = eng.plan_batch(ws, w, want_all=True)
    assert eng.last_launch()["build_wavefronts_per_simd"] == 1
    assert_bitwise(out2["all_plans"], ref["all_plans"], "plans, latency build")
