"""Planner base types named as in the reference (interact_drive/planner/car_planner.py)."""


class CarPlanner(object):
    """A planner is bound to one car of one world."""

    def __init__(self, world, car):
        self.world, self.car = world, car

    def generate_plan(self):
        raise NotImplementedError


class CoordinateAscentPlanner(CarPlanner):
    """Declared but never implemented by the reference; kept so that imports resolve."""
