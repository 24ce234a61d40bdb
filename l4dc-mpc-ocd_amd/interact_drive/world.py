"""Driving scenarios.  Mirrors interact_drive/world.py (render() is out of scope: no GL)."""
from typing import Dict, Iterable, List, Optional, Tuple

import numpy as np

from .tensor import Tensor


class CarWorld(object):
    """Contains the cars and lanes of a scenario and a synchronous step() (world.py:9-109)."""

    def __init__(self, dt: float = 0.1, lanes: Optional[List] = None, obstacles: Optional[List] = None,
                 visualizer_args: Optional[Dict] = None, **kwargs):
        self.cars = []
        self.dt = dt
        self.lanes = [] if lanes is None else lanes
        self.obstacles = [] if obstacles is None else obstacles
        self.visualizer_args = dict() if visualizer_args is None else visualizer_args
        self.visualizer = None

    def add_car(self, car):
        car.index = len(self.cars)
        self.cars.append(car)

    def add_cars(self, cars: Iterable):
        for car in cars:
            self.add_car(car)

    @property
    def state(self):
        return [c.state for c in self.cars]

    @state.setter
    def state(self, new_state: Iterable):
        for c, x in zip(self.cars, new_state):
            c.state = x

    def reset(self):
        for car in self.cars:
            car.reset()

    def step(self, dt: Optional[float] = None) -> Tuple[List[Tensor], List[Tensor], List[Tensor]]:
        """All cars choose controls, then all step (world.py:79-109).

        Returns (past_state, controls, state).
        """
        past_state = self.state
        if dt is None:
            dt = self.dt
        for car in self.cars:
            if not car.control_already_determined_for_current_step:
                car.set_next_control()
        for car in self.cars:
            car.step(dt)
        return past_state, [c.control for c in self.cars], self.state

    def render(self, mode: str = "human", heatmap_show=False):
        raise NotImplementedError("rendering (pyglet/OpenGL) is outside the accelerated planner path")

    # ---- world-step bookkeeping the descriptor needs (overridden by ReplanningCarWorld) ----
    _teleport_step = 0

    def _teleport_cars(self):
        return [-1, -1, -1, -1]


class ThreeLaneCarWorld(CarWorld):
    """Three straight lanes (world.py:143-152)."""

    def __init__(self, dt=0.1, **kwargs):
        lane = StraightLane((0.0, -5.), (0.0, 10.), 0.1)
        lanes = [lane.shifted(1), lane, lane.shifted(-1)]
        super().__init__(dt=dt, lanes=lanes, **kwargs)


class TwoLaneCarWorld(CarWorld):
    """Two straight lanes (world.py:155-159)."""

    def __init__(self, dt=0.1, **kwargs):
        lane = StraightLane((-0.05, -5.), (-0.05, 10.), 0.1)
        lanes = [lane, lane.shifted(-1)]
        super().__init__(dt=dt, lanes=lanes, **kwargs)


class StraightLane(object):
    """Lane with median p->q and width w (world.py:162-221)."""

    def __init__(self, p: Tuple[float, float], q: Tuple[float, float], w: float):
        self.p = np.asarray(p)
        self.q = np.asarray(q)
        self.w = w
        self.m = (self.q - self.p) / np.linalg.norm(self.q - self.p)
        self.n = np.asarray([-self.m[1], self.m[0]])

    def shifted(self, n_lanes: int):
        return StraightLane(self.p + self.n * self.w * n_lanes, self.q + self.n * self.w * n_lanes, self.w)

    def dist2median(self, point):
        """Squared distance of a point to the median, fp32 like the reference's tensor arithmetic."""
        f = np.float32
        r = ((f(point[0]) - f(self.p[0])) * f(self.n[0]) + (f(point[1]) - f(self.p[1])) * f(self.n[1]))
        return Tensor(f(r) * f(r))

    def on_road(self, point):
        raise NotImplementedError
