"""Mirrors interact_drive/car/fixed_velocity_car.py:12-24."""
import numpy as np

from .fixed_control_car import FixedControlCar


class FixedVelocityCar(FixedControlCar):
    """Goes forward at the velocity of its initial state (zero friction, zero controls)."""

    def __init__(self, env, init_state, color: str = 'gray', opacity=1.0, **kwargs):
        kwargs.pop("friction", None)
        super().__init__(env, init_state, np.array([0., 0.]), color, opacity, friction=0., **kwargs)
