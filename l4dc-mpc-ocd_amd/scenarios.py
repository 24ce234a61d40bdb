"""Scenario descriptors: the constants the reference's scenario factories hard-code.

Reference (file:line relative to the reference tree):
  finite_horizon  interact_drive/reward_design/mpc_ord.py:162-207
  local_opt       experiments/local_opt_scenario.py:6-55
  replanning      experiments/replanning_world.py:11-95
  merging         experiments/merging.py:86-99
  eval horizons / samples / tuned weights   experiments/run_mpc_ord.py:19-44
  planner KAT car interact_drive/planner/tests/test_naivePlanner.py:16-63

A scenario is data only (an ``ocd_scenario_desc`` plus host-side metadata); the
world, cars and planner objects of ``interact_drive`` are thin views over it.
"""
from __future__ import annotations

import dataclasses
from typing import Dict, Optional, Sequence, Tuple

import numpy as np

from . import abi

_PI_2 = np.pi / 2


def normalize_like_reference(weights: np.ndarray, times: int = 1) -> np.ndarray:
    """``weights / np.linalg.norm(weights)`` applied ``times`` times in the array's own dtype."""
    w = np.asarray(weights)
    for _ in range(times):
        w = w / np.linalg.norm(w)
    return w


def planner_weights_fp32(weights: Sequence[float]) -> np.ndarray:
    """Candidate weights -> the fp32 vector the planning car ends up with.

    MPC_ORD.eval_weights normalises (mpc_ord.py:120), eval_weights_for_init
    normalises again (mpc_ord.py:71) and the LinearRewardCar.weights setter a
    third time before the fp32 assign (linear_reward_car.py:45-47); all three
    in float64.
    """
    w = np.asarray(weights, dtype=np.float64)
    if w.ndim == 2:
        w = w[0]
    return normalize_like_reference(w, 3).astype(np.float32)


def planner_weights_fp32_batch(weights_2d, out: Optional[np.ndarray] = None) -> np.ndarray:
    """[P, D] candidate weights -> [P, D] fp32, row by row exactly `planner_weights_fp32` (np.linalg.norm of a
    1-D float64 vector is sqrt(x.dot(x)) with numpy's BLAS dot), without its per-row Python overhead: the host side
    of a CMA-ES generation is otherwise dominated by this.  Fastest: one native call (csrc/ocd_cma.c
    ocd_normalise_weights) with the summation order that reproduces numpy's dot on THIS machine -- found once per row
    length by a self-check on 4 096 random rows against the per-row numpy chain; if no native order matches, the
    batched numpy form below (itself self-checked), then the per-row loop.  `out`: a [P, D] fp32 buffer to fill
    (e.g. pinned memory the kernel reads)."""
    W = np.ascontiguousarray(weights_2d, dtype=np.float64)         # (the native call only reads it)
    if W.ndim != 2:
        raise ValueError("weights_2d must be [P, D]")
    variant = _NATIVE_NORMALISE.get(W.shape[1], -1)
    if variant == -1:
        variant = _native_normalise_variant(W.shape[1])
    if variant is not None:
        res = out if (out is not None and out.dtype == np.float32 and out.shape == W.shape and out.flags.c_contiguous) \
            else np.empty(W.shape, dtype=np.float32)
        if _CMA_LIB[0].ocd_normalise_weights(W.ctypes.data, W.shape[0], W.shape[1], variant, res.ctypes.data) != 0:
            raise ValueError("ocd_normalise_weights: bad arguments")
        if out is not None and res is not out:
            out[...] = res
        return res
    W = np.array(W, dtype=np.float64)                              # the numpy forms work in place: a private copy
    if _ROW_DOTS_BATCHED_OK.get(W.shape[1]) is None:
        row_dots(W)                                                 # runs the one-off self-check for this row length
    if _ROW_DOTS_BATCHED_OK[W.shape[1]]:
        # the same three operations per pass (dot, sqrt, divide) in place: a CMA-ES generation pays this on its
        # critical path, and at 64 x 7 the cost is numpy call overhead, not arithmetic
        row, col = W[:, None, :], W[:, :, None]
        n2 = np.empty((W.shape[0], 1, 1), dtype=np.float64)
        nrm = n2.reshape(-1, 1)
        for _ in range(3):
            np.matmul(row, col, out=n2)
            np.sqrt(n2, out=n2)
            np.divide(W, nrm, out=W)
    else:
        for _ in range(3):
            W = W / np.sqrt(_row_dots_loop(W))[:, None]
    res = W.astype(np.float32)
    if out is not None:
        out[...] = res
        return out
    return res


_CMA_LIB = [None]
_NATIVE_NORMALISE = {}          # row length -> variant of ocd_normalise_weights that reproduces numpy here, or None


def _native_normalise_variant(D: int):
    """Which summation order of csrc/ocd_cma.c:ocd_normalise_weights gives, bit for bit, the per-row numpy chain
    `planner_weights_fp32` on this machine (numpy's BLAS decides): checked once per row length on 4 096 random rows
    of widely varying scale; None = use the numpy path."""
    if D in _NATIVE_NORMALISE:
        return _NATIVE_NORMALISE[D]
    found = None
    try:
        if _CMA_LIB[0] is None:
            from .interact_drive.reward_design.cmaes import load_cma_library
            _CMA_LIB[0] = load_cma_library()
        lib = _CMA_LIB[0]
        if 1 <= D <= 64:
            rng = np.random.RandomState(20241004 + D)
            T = np.ascontiguousarray(rng.standard_normal((4096, D)) * np.exp(rng.uniform(-4.0, 4.0, (4096, 1))))
            ref = np.stack([planner_weights_fp32(r) for r in T])
            got = np.empty((4096, D), dtype=np.float32)
            for variant in (1, 0):
                if lib.ocd_normalise_weights(T.ctypes.data, 4096, D, variant, got.ctypes.data) == 0 and \
                        np.array_equal(got.view(np.uint32), ref.view(np.uint32)):
                    found = variant
                    break
    except Exception:                                              # no library, no compiler: the numpy path is complete
        found = None
    _NATIVE_NORMALISE[D] = found
    return found


_ROW_DOTS_BATCHED_OK = {}


def _row_dots_loop(W: np.ndarray) -> np.ndarray:
    n2 = np.empty(W.shape[0], dtype=np.float64)
    for i, row in enumerate(W):
        n2[i] = row.dot(row)
    return n2


def _row_dots_batched(W: np.ndarray) -> np.ndarray:
    return np.matmul(W[:, None, :], W[:, :, None])[:, 0, 0]        # P products (1 x D)(D x 1): numpy's dot kernel per row


def row_dots(W: np.ndarray) -> np.ndarray:
    """row.dot(row) of every row of a float64 [P, D] array, bit for bit what the per-row call returns (the BLAS
    dot is neither a plain left-to-right sum nor numpy's einsum: 28 % / 43 % of random rows differ).  The batched
    matmul runs the same dot kernel per row without the Python loop (64 rows: 2.4 us instead of 35 us); that
    it reproduces the per-row call with THIS numpy / BLAS is checked once per row length on 4 096 random rows,
    and the loop is used if it ever does not."""
    W = np.ascontiguousarray(W, dtype=np.float64)
    D = W.shape[1]
    ok = _ROW_DOTS_BATCHED_OK.get(D)
    if ok is None:
        rng = np.random.RandomState(20240229 + D)
        T = rng.standard_normal((4096, D)) * np.exp(rng.uniform(-4.0, 4.0, (4096, 1)))
        ok = bool(np.array_equal(_row_dots_batched(T), _row_dots_loop(T)))
        _ROW_DOTS_BATCHED_OK[D] = ok
    return _row_dots_batched(W) if ok else _row_dots_loop(W)


def designer_weights_fp32(raw: Sequence[float], raw_dtype=np.float64, pre_normalised: bool = False) -> np.ndarray:
    """Scenario weights -> MPC_ORD.designer_weights (fp32).

    LinearRewardCar.__init__ normalises in the input dtype and stores fp32
    (linear_reward_car.py:34); MPC_ORD.__init__ divides that fp32 vector by its
    fp32 norm once more (mpc_ord.py:24).
    """
    w = np.asarray(raw, dtype=raw_dtype)
    if pre_normalised:
        w = w / np.linalg.norm(w)
    car_w = (w / np.linalg.norm(w)).astype(np.float32)
    return (car_w / np.linalg.norm(car_w)).astype(np.float32)


@dataclasses.dataclass
class InitDistribution:
    """Truncated normals the ego init state is drawn from: (mean, std, (lo, hi)) for x, y, v."""
    x: Tuple[float, float, Tuple[float, float]]
    y: Tuple[float, float, Tuple[float, float]]
    v: Tuple[float, float, Tuple[float, float]]

    def sample(self, n: int, seed: int) -> np.ndarray:
        """[n, 4] float64 init states by inverse-CDF sampling from default_rng(seed).

        The reference samples with scipy.stats.truncnorm under np.random.seed
        (mpc_ord.py:171-181); that stream is scipy-version dependent, so init
        states are treated as explicit input data here.
        """
        from scipy.special import ndtr, ndtri
        rng = np.random.default_rng(seed)
        out = np.empty((n, 4), dtype=np.float64)
        for col, (mean, std, (lo, hi)) in enumerate((self.x, self.y, self.v)):
            a, b = (lo - mean) / std, (hi - mean) / std
            u = rng.random(n)
            z = ndtri(ndtr(a) + u * (ndtr(b) - ndtr(a)))
            out[:, col] = np.clip(z * std + mean, lo, hi)
        out[:, 3] = _PI_2
        return out


@dataclasses.dataclass
class Scenario:
    name: str
    desc: abi.ScenarioDesc
    init_dist: Optional[InitDistribution]
    raw_designer_weights: Optional[np.ndarray]
    tuned_weights: Optional[np.ndarray] = None
    default_init: Optional[np.ndarray] = None
    car_weights: Optional[np.ndarray] = None      # exactly what the reference factory hands to the car constructor
    planner_args: Optional[dict] = None           # kwargs the factory passes on to NaivePlanner

    @property
    def n_features(self) -> int:
        return self.desc.n_features

    @property
    def designer_weights(self) -> np.ndarray:
        return np.array(self.desc.designer_weights[: self.n_features], dtype=np.float32)

    def other_plans(self) -> Optional[np.ndarray]:
        """[C-1, H, 2] controls the planner assumes for the scripted cars (planner_car.py:58-80),
        or None when the ego does not check plans (constant-velocity model)."""
        d = self.desc
        if not d.check_plans:
            return None
        H, no = d.horizon, d.n_cars - 1
        out = np.zeros((no, H, 2), dtype=np.float32)
        for j in range(no):
            for t in range(H):
                src = d.other_plan[j][t] if t < d.other_plan_len[j] else d.other_assumed_default[j]
                out[j, t, 0], out[j, t, 1] = src[0], src[1]
        return out

    def candidate_weights(self, pop: int, seed: int, sigma: float = 0.05) -> np.ndarray:
        """[pop, D] synthetic CMA-ES-like candidates: designer weights + sigma*N(0, I).
        Row 0 is the designer weight vector itself (the "Iteration 0" evaluation, mpc_ord.py:39)."""
        rng = np.random.default_rng(seed)
        base = self.designer_weights.astype(np.float64)
        w = base[None, :] + sigma * rng.standard_normal((pop, base.size))
        w[0] = base
        return w


def _base_desc(horizon: int, n_iter: int, extra_inits: bool, episode_len: int) -> abi.ScenarioDesc:
    d = abi.ScenarioDesc()
    d.abi_version = abi.OCD_ABI_VERSION
    d.reward_kind = abi.OCD_REWARD_LANE_FEATURES
    d.horizon = horizon
    d.n_iter = n_iter
    d.extra_inits = int(bool(extra_inits))
    d.check_plans = 0
    d.episode_len = episode_len
    d.n_samples = 1
    d.teleport_step = 0
    d.teleport_period = 0
    for s in range(abi.OCD_MAX_SAMPLES):
        d.teleport_car[s] = -1
    for k, v in enumerate((10., 0., 0., 0.)):      # replanning_world.py:34 (unused unless teleport_step > 0)
        d.teleport_state[k] = v
    dt = 0.1                                  # CarWorld.dt (world.py:19)
    d.dt = dt
    d.dt_sq = np.float32(dt ** 2)             # dt ** 2 squared as a Python float (simulation_utils.py:14)
    d.learning_rate = 0.1                     # naive_planner.py:20
    d.ego_friction = 0.2                      # car.py:33
    d.target_speed = 1.0                      # merging.py:23
    d.fence_width = 0.05                      # merging.py:80
    d.fence_shape = 5.0 / 0.05                # math_utils.py:85: c / width
    d.bump_half_x = 0.08                      # merging.py:72
    d.bump_half_y = 0.15                      # merging.py:73
    return d


def _set_lanes(d: abi.ScenarioDesc, centers: Sequence[float]) -> None:
    d.n_lanes = len(centers)
    for i, c in enumerate(centers):
        d.lane_center[i] = c
    # smooth_threshold(0.05*num_lanes, width=0.05): x_diff = x - (threshold - width) (math_utils.py:89)
    d.fence_lo = np.float32(0.05 * len(centers) - 0.05)
    # StraightLane((x, -5.), (x, 10.), 0.1): p[1] = -5, n = (-m[1], m[0]) = (-1.0, 0.0) (world.py:150,157,184-187)
    d.lane_origin_y = -5.0
    d.lane_normal_y = 0.0


def _set_other(d: abi.ScenarioDesc, j: int, init, friction: float,
               plan: Sequence[Sequence[float]] = (), default=(0.0, 0.0), assumed_default=None) -> None:
    """assumed_default: what a check_plans planner assumes beyond the plan (planner_car.py:66-75):
    the car's default_control if it has a plan and one, else (0, 0); defaults to `default` for a
    FixedPlanCar (plan given) and to (0, 0) otherwise."""
    for k in range(4):
        d.other_init[j][k] = np.float32(init[k])
    d.other_friction[j] = friction
    d.other_plan_len[j] = len(plan)
    for t, u in enumerate(plan):
        d.other_plan[j][t][0], d.other_plan[j][t][1] = u[0], u[1]
    d.other_default[j][0], d.other_default[j][1] = default[0], default[1]
    if assumed_default is None:
        assumed_default = default if len(plan) else (0.0, 0.0)
    d.other_assumed_default[j][0], d.other_assumed_default[j][1] = assumed_default[0], assumed_default[1]


def _set_designer(d: abi.ScenarioDesc, w32: np.ndarray) -> None:
    for i, v in enumerate(w32):
        d.designer_weights[i] = v


_THREE_LANES = (0.0 + -1.0 * 0.1 * 1, 0.0, 0.0 + -1.0 * 0.1 * -1)     # world.py:150-151
_TWO_LANES = (-0.05, -0.05 + -1.0 * 0.1 * -1)                        # world.py:157-158


def finite_horizon(horizon: int = 5, extra_inits: bool = False, n_iter: Optional[int] = None) -> Scenario:
    """mpc_ord.py:162-207; eval horizon 15, 1 sample (run_mpc_ord.py:29-36)."""
    if n_iter is None:
        n_iter = 200 if horizon == 6 else 100          # mpc_ord.py:192
    d = _base_desc(horizon, n_iter, extra_inits, episode_len=15)
    d.n_cars = 2
    _set_lanes(d, _THREE_LANES)
    _set_other(d, 0, (0, -0.6, 0.5, _PI_2), friction=0.0)          # FixedVelocityCar
    raw = np.array([-5, 0., 0., 0., -6., -50, -50])
    _set_designer(d, designer_weights_fp32(raw))
    return Scenario(
        "finite_horizon", d,
        InitDistribution((0, 0.04, (-0.1, 0.1)), (-0.9, 0.02, (-0.95, -0.85)), (0.8, 0.03, (0.7, 0.9))),
        raw,
        tuned_weights=np.array([-0.21963165, -0.01184596, 0.34379187, -0.04687411, -0.06364365,
                                -0.54138792, -0.7308079]),
        car_weights=raw, planner_args=dict(n_iter=n_iter, extra_inits=extra_inits))


def local_opt(horizon: int = 5, extra_inits: bool = False, n_iter: int = 100) -> Scenario:
    """local_opt_scenario.py:6-55; eval horizon 15, 1 sample (run_mpc_ord.py:20-27)."""
    d = _base_desc(horizon, n_iter, extra_inits, episode_len=15)
    d.n_cars = 2
    _set_lanes(d, _THREE_LANES)
    _set_other(d, 0, (0, -0.9, 1., _PI_2), friction=0.0)
    raw = np.array([-5, 0., 0., -10, 0, -50, -50])
    _set_designer(d, designer_weights_fp32(raw, pre_normalised=True))
    return Scenario(
        "local_opt", d,
        InitDistribution((-0.1, 0.005, (-0.12, -0.08)), (-0.9, 0.04, (-1., -0.8)), (1.0, 0.03, (0.9, 1.1))),
        raw,
        tuned_weights=np.array([-0.09686739, 0.25720383, -0.58355971, -0.23075428, -0.41237239,
                                -0.4758984, -0.36625558]),
        car_weights=raw / np.linalg.norm(raw), planner_args=dict(extra_inits=extra_inits))


def replanning(horizon: int = 5, n_iter: int = 100) -> Scenario:
    """replanning_world.py:11-95; eval horizon 20, 2 samples (run_mpc_ord.py:37-43)."""
    d = _base_desc(horizon, n_iter, False, episode_len=20)
    d.n_cars = 3
    d.n_samples = 2
    d.check_plans = 1
    d.target_speed = 1.2
    _set_lanes(d, _TWO_LANES)
    plan1 = [(0., 0.), (0.7, 2.7), (0., 0.), (0.0, -2.7)]
    plan2 = [(0., 0.), (0.7, -2.7), (0., 0.), (0.0, 2.7)]
    _set_other(d, 0, (0., -0.7, 0.8, _PI_2), friction=0.2, plan=plan1)   # FixedPlanCar inherits friction 0.2
    _set_other(d, 1, (0., -0.7, 0.8, _PI_2), friction=0.2, plan=plan2)
    # critical_t = 4; reset() toggles unlucky_car_idx 1 <-> 2 and setup_world() resets once, so the
    # first sample of every init removes car 1 and the second removes car 2 (replanning_world.py:18-36,93)
    # The toggle happens on EVERY reset, whatever the init or sample: a cycle of period 2 over the flat
    # episode index (with the reference's num_samples = 2 that is the sample index).
    d.teleport_step = 4
    d.teleport_period = 2
    for s in range(abi.OCD_MAX_SAMPLES):
        d.teleport_car[s] = 1 + (s % 2)
    raw = np.array([-3, 0, 0, -2, -10, -10], dtype=np.float32)
    _set_designer(d, designer_weights_fp32(raw, raw_dtype=np.float32, pre_normalised=True))
    return Scenario(
        "replanning", d,
        InitDistribution((-0.0, 0.02, (-0.005, 0.005)), (-0.9, 0.04, (-1., -0.8)), (1.0, 0.05, (0.8, 1.2))),
        raw,
        tuned_weights=np.array([-0.55899817, -0.4436692, -0.3724511, -0.19964276, -0.5438697, 0.12770043]),
        car_weights=raw / np.linalg.norm(raw), planner_args=dict(n_iter=n_iter))


def merging(horizon: int = 5, n_iter: int = 100) -> Scenario:
    """merging.py:86-99.  The reference never runs this scenario through MPC_ORD; episode length
    15 (merging.py:120) and the init distribution are build-defined."""
    d = _base_desc(horizon, n_iter, False, episode_len=15)
    d.n_cars = 3
    _set_lanes(d, _THREE_LANES)
    _set_other(d, 0, (0.1, -1.8, 0.8, _PI_2), friction=0.0)
    _set_other(d, 1, (0.1, -1.3, 0.8, _PI_2), friction=0.0)
    raw = np.array([-1, 0., 0., -10., -10., -10, -5])
    _set_designer(d, designer_weights_fp32(raw))
    return Scenario(
        "merging", d,
        InitDistribution((0.0, 0.04, (-0.1, 0.1)), (-1.8, 0.04, (-1.9, -1.7)), (0.8, 0.03, (0.7, 0.9))),
        raw, default_init=np.array([0, -1.8, 0.8, _PI_2]), car_weights=raw)


def target_speed_kat(horizon: int, n_iter: int, learning_rate: float, friction: float,
                     target_speed: float = 1.0) -> Scenario:
    """The single-car world of the reference's planner tests (test_naivePlanner.py:16-63)."""
    d = _base_desc(horizon, n_iter, False, episode_len=0)
    d.reward_kind = abi.OCD_REWARD_TARGET_SPEED
    d.n_cars = 1
    d.n_lanes = 0
    d.learning_rate = learning_rate
    d.ego_friction = friction
    d.target_speed = target_speed
    return Scenario("target_speed_kat", d, None, None, default_init=np.array([0., 0., 1., _PI_2]))


def linear_target_speed(horizon: int = 5, n_iter: int = 10, learning_rate: float = 5.0, friction: float = 0.0,
                        target_speed: float = 0.0, weights=(2.0, -1.0), episode_len: int = 6) -> Scenario:
    """The single-car world of the reference's inverse-optimal-control tests
    (reward_design/tests/test_first_order_ioc.py:29-60, tests/linearTargetSpeedPlannerCar.py): features
    [v, (v - target)^2], weights (2, -1) normalised by LinearRewardCar, start (0, 0, 1, pi/2)."""
    d = _base_desc(horizon, n_iter, False, episode_len=episode_len)
    d.reward_kind = abi.OCD_REWARD_LINEAR_TARGET_SPEED
    d.n_cars = 1
    d.n_lanes = 0
    d.learning_rate = learning_rate
    d.ego_friction = friction
    d.target_speed = target_speed
    raw = np.asarray(weights, dtype=np.float32)                  # tf.constant([2., -1.], dtype=tf.float32)
    _set_designer(d, designer_weights_fp32(raw, raw_dtype=np.float32))
    return Scenario("linear_target_speed", d, None, raw, default_init=np.array([0., 0., 1., _PI_2]), car_weights=raw)


SCENARIOS = {
    "finite_horizon": finite_horizon,
    "local_opt": local_opt,
    "replanning": replanning,
    "merging": merging,
}

# BASELINE.json configs: scenario factory kwargs, population, inits, seeds
BASELINE_CONFIGS: Dict[int, dict] = {
    1: dict(scenario="finite_horizon", horizon=5, pop=1, n_inits=3),
    2: dict(scenario="finite_horizon", horizon=10, pop=16, n_inits=8),
    3: dict(scenario="local_opt", horizon=10, pop=64, n_inits=32),
    4: dict(scenario="replanning", horizon=15, pop=128, n_inits=64),
    5: dict(scenario="merging", horizon=25, pop=256, n_inits=128),
}


def baseline_config(cfg: int):
    """(scenario, init_states [N,4] float64, candidate weights [P,D] float64) of BASELINE config ``cfg``."""
    c = BASELINE_CONFIGS[cfg]
    scn = SCENARIOS[c["scenario"]](horizon=c["horizon"])
    inits = scn.init_dist.sample(c["n_inits"], seed=1000 + cfg)
    cands = scn.candidate_weights(c["pop"], seed=2000 + cfg)
    return scn, inits, cands
