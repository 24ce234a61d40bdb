"""The car classes of the reference API, as thin stateful views over the GPU planner.

Class and attribute names follow interact_drive/car/*.py (Car, PlannerCar, LinearRewardCar,
FixedControlCar, FixedVelocityCar, FixedPlanCar; .state, .control, .init_state, .friction, .plan,
.default_control, .weights, .control_already_determined_for_current_step) because reference scripts
touch them directly.  The implementation is organised differently: cars never integrate themselves
one by one -- `advance(cars, dt)` moves any set of cars with one batched dynamics launch per friction
value, and a scripted car is a small table (`script_row`) rather than behaviour spread over methods.
"""
from typing import Iterable, List, Optional, Sequence, Union

import numpy as np

from ..tensor import Tensor
from ..._devops import dynamics_batch

ArrayLike = Union[np.ndarray, Iterable]


def advance(cars: Sequence["Car"], dt: float) -> None:
    """Real dynamics for `cars` (car_dynamics_step, reference simulation_utils.py:9-21): one device call
    per distinct friction coefficient, then each car's post-step hook (scripted cars load their next
    control)."""
    by_friction = {}
    for c in cars:
        by_friction.setdefault(float(c.friction), []).append(c)
    for friction, group in by_friction.items():
        states = np.stack([np.asarray(c.state, dtype=np.float32) for c in group])
        controls = np.stack([np.asarray(c.control, dtype=np.float32) for c in group])
        nxt = dynamics_batch(states, controls, dt, friction)
        for c, s in zip(group, nxt):
            if c.debug:
                c.past_traj.append((c.state, c.control))
            c.state = Tensor(s)
    for c in cars:
        c._after_step()


class Car(object):
    """State (x, y, speed, heading), control (acceleration, angular velocity), friction."""

    #: cars whose control never comes from a planner keep this flag raised between steps
    _keeps_control = False

    def __init__(self, env, init_state: ArrayLike, color: str = 'gray', opacity: float = 1.0,
                 friction: float = 0.2, index: int = 0, debug: bool = False, **kwargs):
        self.env, self.index, self.debug = env, index, debug
        self.color, self.opacity = color, opacity
        self.friction = friction
        self.init_state = Tensor(init_state)
        self.state = Tensor(init_state)
        self.control: Optional[Tensor] = None
        self.control_already_determined_for_current_step = self._keeps_control
        self.past_traj: List = []

    # the reference exposes the per-car integrator as an attribute
    @property
    def dynamics_fn(self):
        from ..simulation_utils import get_dynamics_fn
        return get_dynamics_fn(self.friction)

    def reset(self):
        self.state = self.init_state
        if self.debug:
            self.past_traj = []

    def step(self, dt):
        advance([self], dt)

    def _after_step(self):
        self.control_already_determined_for_current_step = self._keeps_control

    def set_next_control(self, control: Optional[ArrayLike] = None):
        if control is not None:
            self.control = Tensor(control)
        elif not self.control_already_determined_for_current_step:
            self.control = self._get_next_control()
        self.control_already_determined_for_current_step = True

    def _get_next_control(self) -> Tensor:
        raise NotImplementedError

    def reward_fn(self, world_state, self_control):
        raise NotImplementedError

    # ---- what the planner's model of this car looks like (a row of the descriptor) ----
    def script_row(self, horizon: int) -> Optional[np.ndarray]:
        """[horizon, 2] controls a planning car should assume for this car, or None (constant velocity)."""
        return None


class FixedControlCar(Car):
    """Always applies the same control."""
    _keeps_control = True

    def __init__(self, env, init_state: ArrayLike, control: ArrayLike, color: str = 'gray',
                 opacity: float = 1.0, **kwargs):
        super().__init__(env, init_state, color, opacity, **kwargs)
        self.control = Tensor(control)

    def _get_next_control(self):
        return self.control

    def reward_fn(self, world_state, self_control):
        return 0


class FixedVelocityCar(FixedControlCar):
    """Zero friction and zero control: keeps the velocity of its initial state."""

    def __init__(self, env, init_state: ArrayLike, color: str = 'gray', opacity=1.0, **kwargs):
        kwargs.pop("friction", None)
        super().__init__(env, init_state, (0., 0.), color, opacity, friction=0., **kwargs)


class FixedPlanCar(Car):
    """Plays back `plan[t]` at world step t, then `default_control`."""
    _keeps_control = True

    def __init__(self, env, init_state: ArrayLike, plan: List[ArrayLike], default_control: Optional[ArrayLike] = None,
                 color: str = 'gray', opacity=1.0, **kwargs):
        super().__init__(env, init_state, color, opacity, **kwargs)
        self.plan = plan
        self.default_control = default_control
        self.control = default_control
        self.t = 0

    def _scripted(self, t: int):
        return self.plan[t] if t < len(self.plan) else self.default_control

    def _after_step(self):
        self.t += 1
        nxt = self._scripted(self.t)
        if nxt is not None:                    # default_control=None keeps the last control, like the reference
            self.control = Tensor(nxt)
        self.control_already_determined_for_current_step = True

    def reset(self):
        super().reset()
        self.t = 0
        self.control = Tensor(self.plan[0])
        self.control_already_determined_for_current_step = True

    def _get_next_control(self):
        return self.control

    def reward_fn(self, world_state, self_control):
        return 0

    def script_row(self, horizon: int):
        # the reference's PlannerCar reads plan[j] from j = 0 at EVERY world step (planner_car.py:66-75)
        rows = [self._scripted(j) if self._scripted(j) is not None else (0., 0.) for j in range(horizon)]
        return np.asarray(rows, dtype=np.float32).reshape(horizon, 2)


class PlannerCar(Car):
    """Chooses its control by model-predictive planning (NaivePlanner on the GPU)."""

    def __init__(self, env, init_state: ArrayLike, horizon: int, color: str = 'orange', opacity: float = 1.0,
                 friction: float = 0.2, planner_args: Optional[dict] = None, check_plans: bool = False, **kwargs):
        super().__init__(env, init_state, color=color, opacity=opacity, friction=friction, **kwargs)
        self.horizon = horizon
        self.planner = None
        self.plan: List = []
        self.planner_args = dict(planner_args or {})
        self.check_plans = check_plans

    def initialize_planner(self, planner_args):
        from ..planner.naive_planner import NaivePlanner
        self.planner = NaivePlanner(self.env, self, self.horizon, **planner_args)

    def assumed_other_controls(self):
        """One entry per car of the world: what this car's planner assumes the others will do
        (a placeholder for itself), or None when it does not look at their plans."""
        if not self.check_plans:
            return None
        table = []
        for other in self.env.cars:
            if other is self:
                table.append(Tensor(np.zeros((self.horizon, 1))))
                continue
            row = other.script_row(self.horizon) if hasattr(other, "script_row") else None
            table.append(Tensor(np.zeros((self.horizon, 2)) if row is None else row))
        return table

    def _get_next_control(self):
        if self.planner is None:
            self.initialize_planner(self.planner_args)
        self.plan = self.planner.generate_plan(other_controls=self.assumed_other_controls())
        return Tensor(self.plan[0])


class LinearRewardCar(Car):
    """reward = weights . features(world_state); weights are kept L2-normalised in fp32."""

    def __init__(self, env, init_state: ArrayLike, weights: ArrayLike, color: str = 'gray', opacity: float = 1.0,
                 friction: float = 0.2, **kwargs):
        super().__init__(env, init_state, color=color, opacity=opacity, friction=friction, **kwargs)
        w = np.asarray(weights)
        self.weights_tf = Tensor(w / np.linalg.norm(w))

    @property
    def weights(self):
        return self.weights_tf.numpy()

    @weights.setter
    def weights(self, weights):
        w = np.asarray(weights)
        self.weights_tf.assign(w / np.linalg.norm(w))

    def _evaluate(self, state, weights):
        from .._describe import describe, engine_for
        ws = np.stack([np.asarray(s, dtype=np.float32) for s in state])[None]
        w = self.weights if weights is None else np.asarray(weights, dtype=np.float32)
        return engine_for(describe(self.env, self, getattr(self, "horizon", 1))).reward_batch(ws, w)

    def features(self, state, control) -> Tensor:
        return Tensor(self._evaluate(state, None)[0][0])

    def reward_fn(self, state, control, weights=None):
        return Tensor(self._evaluate(state, weights)[1][0])
