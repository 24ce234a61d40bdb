"""Mirrors interact_drive/car/fixed_control_car.py:12-36."""
from .car import Car
from ..tensor import Tensor


class FixedControlCar(Car):
    def __init__(self, env, init_state, control, color: str = 'gray', opacity: float = 1.0, **kwargs):
        super().__init__(env, init_state, color, opacity, **kwargs)
        self.control = Tensor(control)
        self.control_already_determined_for_current_step = True

    def step(self, dt):
        if self.debug:
            self.past_traj.append((self.state, self.control))
        self.state = self.dynamics_fn(self.state, self.control, dt)

    def reward_fn(self, world_state, self_control):
        return 0

    def _get_next_control(self):
        return self.control
