"""`ReplanningCarWorld` / `setup_world` of the reference's experiments/replanning_world.py."""
import numpy as np

from ._build import world_from_scenario
from ._sampling import make_get_init_state
from ..tensor import Tensor
from ..world import TwoLaneCarWorld
from ... import scenarios


class ReplanningCarWorld(TwoLaneCarWorld):
    """At world step `critical_t` one of the two scripted cars vanishes (is moved to (10, 0, 0, 0));
    which one alternates with every reset()."""

    GONE = (10., 0., 0., 0.)

    def __init__(self, dt=0.1, critical_t=4, **kwargs):
        super().__init__(dt=dt, **kwargs)
        self.critical_t, self.unlucky_car_idx, self.t = critical_t, 1, 0

    def reset(self):
        super().reset()
        self.unlucky_car_idx = 3 - self.unlucky_car_idx          # 1 <-> 2
        self.t = 0

    def step(self, dt=None):
        self.t += 1
        if self.t == self.critical_t:
            self.cars[self.unlucky_car_idx].state = Tensor(self.GONE)
        return super().step()

    # descriptor hooks (see _describe.describe): the car the s-th reset() from now on removes
    @property
    def _teleport_step(self):
        return self.critical_t

    _teleport_period = 2          # reset() toggles on every call, whatever the init or sample

    def _teleport_cars(self):
        first = 3 - self.unlucky_car_idx
        return [first, 3 - first, first, 3 - first]


_scn = scenarios.replanning()
og_weights = np.array(_scn.car_weights, dtype=np.float32)
tuned_weights = np.array(_scn.tuned_weights, dtype=np.float32)
tuned_weights /= np.linalg.norm(tuned_weights)


def setup_world(env_seeds=[1], debug=True):
    scn = scenarios.replanning(horizon=5)
    dist = scn.init_dist
    init_states = [make_get_init_state(dist.x, dist.y, dist.v)(s) for s in env_seeds]
    our_car, _, world = world_from_scenario(scn, init_states[0], debug=debug, world_cls=ReplanningCarWorld)
    return our_car, world, init_states
