from .car import Car  # noqa: F401
from .planner_car import PlannerCar  # noqa: F401
from .linear_reward_car import LinearRewardCar  # noqa: F401
from .fixed_control_car import FixedControlCar  # noqa: F401
from .fixed_velocity_car import FixedVelocityCar  # noqa: F401
from .fixed_plan_car import FixedPlanCar  # noqa: F401
