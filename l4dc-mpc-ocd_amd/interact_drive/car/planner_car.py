"""Planner cars.  Mirrors interact_drive/car/planner_car.py:12-85."""
from typing import Iterable, Union

import numpy as np

from .car import Car
from ..tensor import Tensor


class PlannerCar(Car):
    """A car that performs model predictive control with a NaivePlanner on the GPU."""

    def __init__(self, env, init_state: Union[np.ndarray, Iterable], horizon: int, color: str = 'orange',
                 opacity: float = 1.0, friction: float = 0.2, planner_args: dict = None,
                 check_plans: bool = False, **kwargs):
        super().__init__(env, init_state, color=color, opacity=opacity, friction=friction, **kwargs)
        self.horizon = horizon
        self.planner = None
        self.plan = []
        if planner_args is None:
            planner_args = {}
        self.planner_args = planner_args
        self.check_plans = check_plans

    def initialize_planner(self, planner_args):
        from ..planner.naive_planner import NaivePlanner
        self.planner = NaivePlanner(self.env, self, self.horizon, **planner_args)

    def _get_next_control(self):
        """planner_car.py:54-85 (the other cars' plans are read from index 0 at every step)."""
        if self.planner is None:
            self.initialize_planner(self.planner_args)
        if self.check_plans:
            other_plans = []
            for i, other_car in enumerate(self.env.cars):
                if i == self.index:
                    other_plan = Tensor(np.zeros((self.horizon, 1)))
                else:
                    other_plan = []
                    for j in range(self.horizon):
                        if hasattr(other_car, 'plan') and other_car.plan is not None:
                            if j < len(other_car.plan):
                                other_plan.append(other_car.plan[j])
                            else:
                                if hasattr(other_car, 'default_control') and other_car.default_control is not None:
                                    other_plan.append(other_car.default_control)
                                else:
                                    other_plan.append(np.zeros(2, dtype=np.float32))
                        else:
                            other_plan.append(np.zeros(2, dtype=np.float32))
                    other_plan = Tensor(np.stack([np.asarray(u, dtype=np.float32) for u in other_plan], axis=0))
                other_plans.append(other_plan)
            self.plan = self.planner.generate_plan(other_controls=other_plans)
        else:
            self.plan = self.planner.generate_plan()
        return Tensor(self.plan[0])
