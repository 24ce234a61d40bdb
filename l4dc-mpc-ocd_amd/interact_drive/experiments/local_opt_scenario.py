"""Mirrors experiments/local_opt_scenario.py:6-55."""
import numpy as np

from .merging import ThreeLaneCarWorld, ThreeLaneTestCar
from ._sampling import make_get_init_state
from ..car import FixedVelocityCar


def local_opt_env(env_seeds=[1], extra_inits=False, debug=True):
    get_init_state = make_get_init_state((-0.1, 0.005, (-0.12, -0.08)), (-0.9, 0.04, (-1., -0.8)),
                                         (1.0, 0.03, (0.9, 1.1)))
    init_states = [get_init_state(s) for s in env_seeds]
    world = ThreeLaneCarWorld(visualizer_args=dict(name="Switch Lanes"))
    weights = np.array([-5, 0., 0., -10, 0, -50, -50])
    our_car = ThreeLaneTestCar(world, init_states[0], horizon=5, weights=weights / np.linalg.norm(weights),
                               planner_args=dict(extra_inits=extra_inits), debug=debug)
    other_car = FixedVelocityCar(world, np.array([0, -0.9, 1., np.pi / 2]), horizon=5, color="gray",
                                 opacity=0.8, debug=debug)
    world.add_cars([our_car, other_car])
    world.reset()
    return our_car, world, init_states
