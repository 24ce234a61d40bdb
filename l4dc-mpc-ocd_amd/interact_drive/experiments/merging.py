"""ThreeLaneTestCar and the merging scenario.  Mirrors experiments/merging.py:20-99."""
from typing import Union

import numpy as np

from ..car import LinearRewardCar, PlannerCar
from ..world import CarWorld, ThreeLaneCarWorld  # noqa: F401
from ... import abi


class ThreeLaneTestCar(LinearRewardCar, PlannerCar):
    """Planning car with the lane / collision / fence features (merging.py:32-83):
    [bounded (v sin(heading) - target)^2, 10*dist^2 to each lane, min of those,
     max collision bump over the other cars, fence threshold * |x|]."""

    _ocd_reward_kind = abi.OCD_REWARD_LANE_FEATURES

    def __init__(self, env: CarWorld, init_state, horizon: int, weights: Union[np.ndarray, list],
                 target_speed=1., color='orange', friction=0.2, opacity=1.0, planner_args=None, debug=False,
                 num_lanes=3, **kwargs):
        super().__init__(env, init_state, horizon=horizon, weights=weights, color=color, friction=friction,
                         opacity=opacity, planner_args=planner_args, debug=debug, **kwargs)
        self.target_speed = np.float32(target_speed)
        self.num_lanes = num_lanes


def setup_world():
    """The merging scenario of experiments/merging.py, built from scenarios.merging."""
    from ._build import world_from_scenario
    from ... import scenarios
    scn = scenarios.merging(horizon=5)
    our_car, (other_car_1, other_car_2), world = world_from_scenario(
        scn, scn.default_init, debug=False, visualizer_args=dict(name="Merging", heatmap_show=True))
    return our_car, other_car_1, other_car_2, world
