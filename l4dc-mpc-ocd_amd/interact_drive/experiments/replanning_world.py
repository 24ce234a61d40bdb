"""Mirrors experiments/replanning_world.py:11-95."""
from typing import Optional

import numpy as np

from .merging import ThreeLaneTestCar
from ._sampling import make_get_init_state
from ..car import FixedPlanCar
from ..tensor import Tensor
from ..world import TwoLaneCarWorld


class ReplanningCarWorld(TwoLaneCarWorld):
    """One of the other cars disappears (is teleported far away) at a fixed timestep."""

    def __init__(self, dt=0.1, critical_t=4, **kwargs):
        super().__init__(dt=dt, **kwargs)
        self.critical_t = critical_t
        self.unlucky_car_idx = 1
        self.t = 0

    def reset(self):
        super().reset()
        self.unlucky_car_idx = 2 if self.unlucky_car_idx == 1 else 1
        self.t = 0

    def step(self, dt: Optional[float] = None):
        self.t += 1
        if self.t == self.critical_t:
            self.cars[self.unlucky_car_idx].state = Tensor([10., 0., 0., 0.])
        return super().step()

    # descriptor hooks: which car the s-th reset() from now on will remove
    @property
    def _teleport_step(self):
        return self.critical_t

    def _teleport_cars(self):
        nxt = 2 if self.unlucky_car_idx == 1 else 1
        oth = 1 if nxt == 2 else 2
        return [nxt, oth, nxt, oth]


og_weights = np.array([-3, 0, 0, -2, -10, -10], dtype=np.float32)
og_weights /= np.linalg.norm(og_weights)
tuned_weights = np.array([-0.55899817, -0.4436692, -0.37245109, -0.19964276, -0.5438697, 0.12770044], dtype=np.float32)
tuned_weights /= np.linalg.norm(tuned_weights)


def setup_world(env_seeds=[1], debug=True):
    get_init_state = make_get_init_state((-0.0, 0.02, (-0.005, 0.005)), (-0.9, 0.04, (-1., -0.8)),
                                         (1.0, 0.05, (0.8, 1.2)))
    init_states = [get_init_state(s) for s in env_seeds]
    world = ReplanningCarWorld()
    our_car = ThreeLaneTestCar(world, init_states[0], horizon=5, weights=og_weights, target_speed=1.2,
                               planner_args={'n_iter': 100}, check_plans=True, num_lanes=2, debug=debug)
    f32 = np.float32
    other_car_1 = FixedPlanCar(world, np.array([0., -0.7, 0.8, np.pi / 2]),
                               plan=[np.array([0., 0.], dtype=f32), np.array([0.7, 2.7], dtype=f32),
                                     np.array([0., 0.], dtype=f32), np.array([0.0, -2.7], dtype=f32)],
                               default_control=np.array([0.0, 0.0], dtype=f32), horizon=5, color='gray',
                               opacity=0.8, debug=debug)
    other_car_2 = FixedPlanCar(world, np.array([0., -0.7, 0.8, np.pi / 2]),
                               plan=[np.array([0., 0.], dtype=f32), np.array([0.7, -2.7], dtype=f32),
                                     np.array([0., 0.], dtype=f32), np.array([0.0, 2.7], dtype=f32)],
                               default_control=np.array([0.0, 0.0], dtype=f32), horizon=5, color='gray',
                               opacity=0.8, debug=debug)
    world.add_cars([our_car, other_car_1, other_car_2])
    world.reset()
    return our_car, world, init_states
