"""Mirrors interact_drive/car/fixed_plan_car.py:10-43."""
from .car import Car


class FixedPlanCar(Car):
    """Follows a fixed list of controls, then default_control."""

    def __init__(self, env, init_state, plan, default_control=None, color: str = 'gray', opacity=1.0, **kwargs):
        super().__init__(env, init_state, color, opacity, **kwargs)
        self.control_already_determined_for_current_step = True
        self.plan = plan
        self.control = default_control
        self.default_control = default_control
        self.t = 0

    def step(self, dt):
        super().step(dt)
        self.t += 1
        if self.t < len(self.plan):
            self.set_next_control(self.plan[self.t])
        else:
            self.set_next_control(self.default_control)

    def _get_next_control(self):
        return self.control

    def reset(self):
        super().reset()
        self.t = 0
        self.set_next_control(self.plan[self.t])

    def reward_fn(self, world_state, self_control):
        return 0
