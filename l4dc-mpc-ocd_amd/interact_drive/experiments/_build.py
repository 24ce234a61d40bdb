"""Scenario descriptor -> world / car objects (the inverse of _describe.describe).

The scenario constants live once, in scenarios.py; the reference-shaped factories
(finite_horizon_env, local_opt_env, replanning setup_world, merging setup_world) are thin wrappers
around `world_from_scenario`.
"""
import numpy as np

from ..car import FixedPlanCar, FixedVelocityCar
from ..world import CarWorld, ThreeLaneCarWorld, TwoLaneCarWorld


def _world_for(desc, world_cls=None, **world_kwargs):
    if world_cls is not None:
        return world_cls(**world_kwargs)
    return {2: TwoLaneCarWorld, 3: ThreeLaneCarWorld}.get(desc.n_lanes, CarWorld)(**world_kwargs)


def world_from_scenario(scn, init_state, debug=True, world_cls=None, car_cls=None, **world_kwargs):
    """Build (planning car, scripted cars, world) for `scn` with the ego starting at `init_state`."""
    from .merging import ThreeLaneTestCar
    d = scn.desc
    world = _world_for(d, world_cls, **world_kwargs)
    ego_kwargs = dict(horizon=d.horizon, weights=scn.car_weights, debug=debug, friction=float(d.ego_friction),
                      target_speed=float(d.target_speed), num_lanes=d.n_lanes, check_plans=bool(d.check_plans))
    if scn.planner_args is not None:
        ego_kwargs["planner_args"] = dict(scn.planner_args)
    ego = (car_cls or ThreeLaneTestCar)(world, np.asarray(init_state), **ego_kwargs)
    others = []
    for j in range(d.n_cars - 1):
        start = np.array(d.other_init[j][:], dtype=np.float64)
        start[3] = np.pi / 2 if abs(start[3] - np.pi / 2) < 1e-6 else start[3]     # keep the exact double
        common = dict(horizon=d.horizon, color='gray', opacity=0.8, debug=debug)
        if d.other_plan_len[j] == 0 and float(d.other_friction[j]) == 0.0:
            others.append(FixedVelocityCar(world, start, **common))
        else:
            plan = [np.array(d.other_plan[j][t][:], dtype=np.float32) for t in range(d.other_plan_len[j])]
            others.append(FixedPlanCar(world, start, plan=plan,
                                       default_control=np.array(d.other_default[j][:], dtype=np.float32),
                                       friction=float(d.other_friction[j]), **common))
    world.add_cars([ego] + others)
    world.reset()
    return ego, others, world
