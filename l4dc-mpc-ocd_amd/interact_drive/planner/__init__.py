from .car_planner import CarPlanner, CoordinateAscentPlanner  # noqa: F401
from .naive_planner import NaivePlanner  # noqa: F401
