"""ThreeLaneTestCar and the merging scenario.  Mirrors experiments/merging.py:20-99."""
from typing import Union

import numpy as np

from ..car import FixedVelocityCar, LinearRewardCar, PlannerCar
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
    """merging.py:86-99."""
    world = ThreeLaneCarWorld(visualizer_args=dict(name="Merging", heatmap_show=True))
    our_car = ThreeLaneTestCar(world, np.array([0, -1.8, 0.8, np.pi / 2]), horizon=5,
                               weights=np.array([-1, 0., 0., -10., -10., -10, -5]))
    other_car_1 = FixedVelocityCar(world, np.array([0.1, -1.8, 0.8, np.pi / 2]), horizon=5, color='gray', opacity=0.8)
    other_car_2 = FixedVelocityCar(world, np.array([0.1, -1.3, 0.8, np.pi / 2]), horizon=5, color='gray', opacity=0.8)
    world.add_cars([our_car, other_car_1, other_car_2])
    world.reset()
    return our_car, other_car_1, other_car_2, world
