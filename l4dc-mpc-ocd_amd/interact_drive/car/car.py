"""Base class for cars.  Mirrors interact_drive/car/car.py:12-123."""
from typing import Iterable, Union

import numpy as np

from ..simulation_utils import get_dynamics_fn
from ..tensor import Tensor


class Car(object):
    def __init__(self, env, init_state: Union[np.ndarray, Iterable], color: str, opacity: float = 1.0,
                 friction: float = 0.2, index: int = 0, debug: bool = False, **kwargs):
        self.env = env
        self.friction = friction
        self.dynamics_fn = get_dynamics_fn(friction)
        self.init_state = Tensor(init_state)
        self.state = Tensor(init_state)
        self.debug = debug
        self.past_traj = []
        self.color = color
        self.opacity = opacity
        self.index = index
        self.control = None
        self.control_already_determined_for_current_step = False

    def reset(self):
        self.state = self.init_state
        if self.debug:
            self.past_traj = []

    def step(self, dt):
        """Updates the state of the car based on self.control (car.py:76-87)."""
        if self.debug:
            self.past_traj.append((self.state, self.control))
        self.control_already_determined_for_current_step = False
        self.state = self.dynamics_fn(self.state, self.control, dt)

    def reward_fn(self, world_state, self_control):
        raise NotImplementedError

    def _get_next_control(self) -> Tensor:
        raise NotImplementedError

    def set_next_control(self, control: Union[None, np.ndarray, Iterable] = None):
        """car.py:109-123."""
        if control is not None:
            self.control = Tensor(control)
        else:
            if not self.control_already_determined_for_current_step:
                self.control = self._get_next_control()
        self.control_already_determined_for_current_step = True
