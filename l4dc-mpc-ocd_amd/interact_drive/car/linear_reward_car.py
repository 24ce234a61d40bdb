"""Import-path compatibility: ``interact_drive.car.linear_reward_car.LinearRewardCar`` (implementation in _cars.py)."""
from ._cars import LinearRewardCar  # noqa: F401
