"""Import-path compatibility: ``interact_drive.car.planner_car.PlannerCar`` (implementation in _cars.py)."""
from ._cars import PlannerCar  # noqa: F401
