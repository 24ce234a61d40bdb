"""Mirrors interact_drive/planner/car_planner.py:7-24."""


class CarPlanner(object):
    def __init__(self, world, car):
        self.world = world
        self.car = car

    def generate_plan(self):
        raise NotImplementedError


class CoordinateAscentPlanner(CarPlanner):
    """Empty in the reference too (car_planner.py:20-24)."""

    def __init__(self, world, car):
        super().__init__(world, car)
