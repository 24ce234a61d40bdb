"""Object graph (world, cars, planner arguments) -> ocd_scenario_desc, and the Engine cache."""
import numpy as np

from .. import abi
from ..scenarios import Scenario

_engines = {}


def describe(world, car, horizon: int, learning_rate: float = 0.1, n_iter: int = 100,
             extra_inits: bool = False, episode_len: int = 0, n_samples: int = 1,
             designer_weights=None) -> abi.ScenarioDesc:
    """Build the descriptor the kernels read from a CarWorld and its planning car.

    Raises NotImplementedError for worlds the compiled kernels do not cover (planning car not at
    index 0, non-vertical lanes, a reward model other than the two compiled ones, ...).
    """
    from .car.fixed_control_car import FixedControlCar
    from .car.fixed_plan_car import FixedPlanCar

    if car.index != 0 or (world.cars and world.cars[0] is not car):
        raise NotImplementedError("the planning car must be the first car added to the world")
    kind = getattr(car, "_ocd_reward_kind", None)
    if kind is None:
        raise NotImplementedError(
            f"{type(car).__name__}: only ThreeLaneTestCar-style lane features and the two target-speed test "
            "rewards are compiled for the GPU planner")
    d = abi.ScenarioDesc()
    d.abi_version = abi.OCD_ABI_VERSION
    d.reward_kind = kind
    d.n_cars = len(world.cars)
    if d.n_cars > abi.OCD_MAX_CARS:
        raise NotImplementedError(f"at most {abi.OCD_MAX_CARS} cars")
    d.horizon = int(horizon)
    d.n_iter = int(n_iter)
    d.extra_inits = int(bool(extra_inits))
    d.check_plans = int(bool(getattr(car, "check_plans", False)))
    d.episode_len = int(episode_len)
    d.n_samples = int(n_samples)
    d.teleport_step = int(getattr(world, "_teleport_step", 0))
    d.teleport_period = int(getattr(world, "_teleport_period", 0))
    tc = world._teleport_cars() if hasattr(world, "_teleport_cars") else [-1] * abi.OCD_MAX_SAMPLES
    for s in range(abi.OCD_MAX_SAMPLES):
        d.teleport_car[s] = tc[s]
    for k, v in enumerate(getattr(world, "_teleport_state", (10., 0., 0., 0.))):
        d.teleport_state[k] = v
    dt = float(world.dt)
    d.dt = dt
    d.dt_sq = np.float32(dt ** 2)
    d.learning_rate = float(learning_rate)
    d.ego_friction = float(car.friction)
    d.target_speed = float(np.float32(getattr(car, "target_speed", 0.0)))
    d.bump_half_x = 0.08
    d.bump_half_y = 0.15
    d.fence_width = 0.05
    d.fence_shape = 5.0 / 0.05
    if kind == abi.OCD_REWARD_LANE_FEATURES:
        lanes = world.lanes
        if not 1 <= len(lanes) <= abi.OCD_MAX_LANES:
            raise NotImplementedError(f"1..{abi.OCD_MAX_LANES} lanes")
        for i, lane in enumerate(lanes):
            if not (float(lane.n[0]) == -1.0 and float(lane.n[1]) == 0.0):
                raise NotImplementedError("only lanes running along +y (normal (-1, 0)) are compiled")
            d.lane_center[i] = float(lane.p[0])
            if float(lane.p[1]) != float(lanes[0].p[1]):
                raise NotImplementedError("lanes that start at different y (StraightLane.p[1]) are not compiled")
        # dist2median's y-term (y - p[1]) * n[1] of the scored reward (world.py:216-217; include/ocd.h ABI 3)
        d.lane_origin_y = float(lanes[0].p[1])
        d.lane_normal_y = float(lanes[0].n[1])
        d.n_lanes = len(lanes)
        num_lanes = getattr(car, "num_lanes", len(lanes))
        d.fence_lo = np.float32(0.05 * num_lanes - 0.05)
        if d.n_cars < 2:
            raise NotImplementedError("ThreeLaneTestCar.features needs at least one other car (reduce_max)")
    else:
        d.n_lanes = 0
    for j, other in enumerate(world.cars[1:]):
        init = np.asarray(other.init_state, dtype=np.float32)
        for k in range(4):
            d.other_init[j][k] = init[k]
        d.other_friction[j] = float(other.friction)
        if isinstance(other, FixedPlanCar):
            if len(other.plan) > abi.OCD_MAX_PLAN:
                raise NotImplementedError(f"FixedPlanCar.plan longer than {abi.OCD_MAX_PLAN}")
            d.other_plan_len[j] = len(other.plan)
            for t, u in enumerate(other.plan):
                d.other_plan[j][t][0], d.other_plan[j][t][1] = float(u[0]), float(u[1])
            if other.default_control is None and episode_len > len(other.plan):
                # the reference's FixedPlanCar.step sets control = None past the plan and then fails
                raise NotImplementedError("FixedPlanCar without default_control stepped beyond its plan")
            dc = other.default_control if other.default_control is not None else other.plan[-1]
            d.other_default[j][0], d.other_default[j][1] = float(dc[0]), float(dc[1])
            # what a check_plans planner assumes beyond the plan (planner_car.py:66-75)
            if other.default_control is not None:
                d.other_assumed_default[j][0], d.other_assumed_default[j][1] = float(dc[0]), float(dc[1])
        elif isinstance(other, FixedControlCar):
            # no .plan attribute: a check_plans planner assumes (0, 0) for it (planner_car.py:76-77)
            d.other_plan_len[j] = 0
            d.other_default[j][0], d.other_default[j][1] = float(other.control[0]), float(other.control[1])
        else:
            raise NotImplementedError(f"{type(other).__name__}: scripted cars must be FixedControl/Velocity/PlanCar")
    if designer_weights is not None and kind != abi.OCD_REWARD_TARGET_SPEED:
        # only episode scoring reads these (mpc_ord.py:99); plans and rewards take weights per call
        w = np.asarray(designer_weights, dtype=np.float32)
        if w.shape[0] != d.n_features:
            raise ValueError(f"weights has {w.shape[0]} entries, the car has {d.n_features} features")
        for i, v in enumerate(w):
            d.designer_weights[i] = v
    return d


def engine_for(desc: abi.ScenarioDesc, name: str = "custom", leaf=None):
    """One Engine per distinct descriptor (+ terminal-value table): the handle owns the device-side constants."""
    from ..engine import Engine
    raw = bytes(desc)
    key = raw if leaf is None else (raw, leaf.key())
    eng = _engines.get(key)
    if eng is None:
        copy = abi.ScenarioDesc.from_buffer_copy(raw)
        eng = Engine(Scenario(name, copy, None, None))
        if leaf is not None:
            eng.set_leaf_value(leaf.disc_grid, leaf.values, leaf.proj_kind)
        if len(_engines) > 64:
            _engines.clear()
        _engines[key] = eng
    return eng
