"""LinearTargetSpeedPlannerCar.  Mirrors interact_drive/reward_design/tests/linearTargetSpeedPlannerCar.py:11-44:
the planning car of the reference's inverse-optimal-control tests (test_first_order_ioc.py:29-135), two features
[velocity, (velocity - target_speed)^2] under LinearRewardCar weights, planned by the GPU NaivePlanner."""
import numpy as np

from ...car import LinearRewardCar, PlannerCar
from .... import abi


class LinearTargetSpeedPlannerCar(LinearRewardCar, PlannerCar):
    _ocd_reward_kind = abi.OCD_REWARD_LINEAR_TARGET_SPEED

    def __init__(self, env, init_state, weights, horizon: int, target_speed: float, friction: float = 0.2, **kwargs):
        super().__init__(env, init_state, horizon=horizon, weights=weights, friction=friction, **kwargs)
        self.target_speed = np.float32(target_speed)
