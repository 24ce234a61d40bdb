"""Import-path compatibility: ``interact_drive.car.fixed_plan_car.FixedPlanCar`` (implementation in _cars.py)."""
from ._cars import FixedPlanCar  # noqa: F401
