"""Import-path compatibility: ``interact_drive.car.fixed_velocity_car.FixedVelocityCar`` (implementation in _cars.py)."""
from ._cars import FixedVelocityCar  # noqa: F401
