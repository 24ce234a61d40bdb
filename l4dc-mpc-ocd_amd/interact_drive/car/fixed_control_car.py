"""Import-path compatibility: ``interact_drive.car.fixed_control_car.FixedControlCar`` (implementation in _cars.py)."""
from ._cars import FixedControlCar  # noqa: F401
