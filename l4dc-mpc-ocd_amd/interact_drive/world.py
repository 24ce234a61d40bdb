"""Worlds of the reference API: a list of cars, straight lanes, a synchronous step.

Names follow interact_drive/world.py (CarWorld, ThreeLaneCarWorld, TwoLaneCarWorld, StraightLane;
.cars, .lanes, .dt, .state, add_car(s), reset, step) because the reference's scenario factories and
scripts use them; the stepping itself is batched (`car.advance`: one dynamics launch for all cars of
equal friction) and rendering is not part of this path.
"""
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

from .tensor import Tensor


class StraightLane(object):
    """A lane whose median runs from p to q, width w.  `m` is the unit direction, `n` the unit normal
    (m rotated by +90 degrees); `shifted(k)` moves the lane k widths along n."""

    def __init__(self, p: Sequence[float], q: Sequence[float], w: float):
        self.p, self.q, self.w = np.asarray(p), np.asarray(q), w
        along = self.q - self.p
        self.m = along / np.linalg.norm(along)
        self.n = np.asarray([-self.m[1], self.m[0]])

    def shifted(self, n_lanes: int) -> "StraightLane":
        offset = self.n * self.w * n_lanes
        return StraightLane(self.p + offset, self.q + offset, self.w)

    def dist2median(self, point) -> Tensor:
        """Squared signed distance of `point` to the median line, in fp32 like the planner computes it
        (host helper only; the planner evaluates lane features on the device)."""
        f = np.float32
        along_n = (f(point[0]) - f(self.p[0])) * f(self.n[0]) + (f(point[1]) - f(self.p[1])) * f(self.n[1])
        return Tensor(f(along_n) * f(along_n))

    def on_road(self, point):
        raise NotImplementedError


def _parallel_lanes(x0: float, width: float, shifts: Sequence[int]) -> List[StraightLane]:
    base = StraightLane((x0, -5.), (x0, 10.), width)
    return [base if k == 0 else base.shifted(k) for k in shifts]


class CarWorld(object):
    """Cars + lanes + time step.  `step()` lets every car fix its control, then moves all of them."""

    _teleport_step = 0           # descriptor hooks; ReplanningCarWorld overrides them

    def __init__(self, dt: float = 0.1, lanes: Optional[List] = None, obstacles: Optional[List] = None,
                 visualizer_args: Optional[Dict] = None, **kwargs):
        self.dt = dt
        self.cars: List = []
        self.lanes = list(lanes) if lanes is not None else []
        self.obstacles = list(obstacles) if obstacles is not None else []
        self.visualizer_args = dict(visualizer_args or {})
        self.visualizer = None

    def _teleport_cars(self):
        return [-1, -1, -1, -1]

    # ---- cars ----
    def add_car(self, car):
        car.index = len(self.cars)
        self.cars.append(car)

    def add_cars(self, cars: Iterable):
        for car in cars:
            self.add_car(car)

    @property
    def state(self):
        return [car.state for car in self.cars]

    @state.setter
    def state(self, new_state: Iterable):
        for car, s in zip(self.cars, new_state):
            car.state = s

    def reset(self):
        for car in self.cars:
            car.reset()

    # ---- time ----
    def step(self, dt: Optional[float] = None) -> Tuple[List[Tensor], List[Tensor], List[Tensor]]:
        """(past_state, controls, state): controls are chosen against the pre-step state, then all cars
        move at once."""
        from .car import advance
        before = self.state
        for car in self.cars:
            if not car.control_already_determined_for_current_step:
                car.set_next_control()
        advance(self.cars, self.dt if dt is None else dt)
        return before, [car.control for car in self.cars], self.state

    def render(self, mode: str = "human", heatmap_show=False):
        raise NotImplementedError("rendering (pyglet/OpenGL) is outside the accelerated planner path")


class ThreeLaneCarWorld(CarWorld):
    """Three lanes of width 0.1 along +y, medians at x = -0.1, 0, +0.1."""

    def __init__(self, dt=0.1, **kwargs):
        super().__init__(dt=dt, lanes=_parallel_lanes(0.0, 0.1, (1, 0, -1)), **kwargs)


class TwoLaneCarWorld(CarWorld):
    """Two lanes of width 0.1 along +y, medians at x = -0.05, +0.05."""

    def __init__(self, dt=0.1, **kwargs):
        super().__init__(dt=dt, lanes=_parallel_lanes(-0.05, 0.1, (0, -1)), **kwargs)
