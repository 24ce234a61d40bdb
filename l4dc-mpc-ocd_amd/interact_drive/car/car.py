"""Import-path compatibility: ``interact_drive.car.car.Car`` (implementation in _cars.py)."""
from ._cars import Car  # noqa: F401
