"""Cars of the reference API (implementation: _cars.py)."""
from ._cars import (Car, FixedControlCar, FixedPlanCar, FixedVelocityCar, LinearRewardCar,  # noqa: F401
                    PlannerCar, advance)
