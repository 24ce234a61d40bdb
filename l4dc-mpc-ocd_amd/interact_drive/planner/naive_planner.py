"""NaivePlanner on the GPU.  Mirrors interact_drive/planner/naive_planner.py:14-164."""
from typing import List, Optional

import numpy as np

from .car_planner import CarPlanner
from ..tensor import Tensor
from .._describe import describe, engine_for


class NaivePlanner(CarPlanner):
    """MPC planner: K control initialisations x n_iter plain-SGD ascent steps on the predicted
    reward, best initialisation wins (first index on ties)."""

    def __init__(self, world, car, horizon: int, learning_rate: float = 0.1, n_iter: int = 100,
                 leaf_evaluation=None, extra_inits=False):
        super().__init__(world, car)
        if leaf_evaluation is not None and not hasattr(leaf_evaluation, "disc_grid"):
            raise NotImplementedError("leaf_evaluation must come from ValueFeature.interpolate_value "
                                      "(reward_design/value_interpolation.py): arbitrary Python is not compiled")
        self.leaf_evaluation = leaf_evaluation
        self.learning_rate = learning_rate
        self.horizon = horizon
        self.planned_controls = [Tensor([0., 0.]) for _ in range(horizon)]
        self.n_iter = n_iter
        self.extra_inits = extra_inits
        self.last_losses = None
        self.last_best_init = None

    def _engine(self):
        return engine_for(describe(self.world, self.car, self.horizon, self.learning_rate, self.n_iter,
                                   self.extra_inits), leaf=self.leaf_evaluation)

    def _world_state(self, init_state):
        if init_state is None:
            init_state = self.world.state
        ws = np.stack([np.asarray(s, dtype=np.float32) for s in init_state])
        if ws.shape != (len(self.world.cars), 4):
            raise ValueError(f"init_state must hold one (4,) state per car, got {ws.shape}")
        return ws

    def _weights(self, weights):
        if weights is not None:
            return np.asarray(weights, dtype=np.float32)
        if hasattr(self.car, "weights_tf"):
            return self.car.weights
        return None

    def _other_plans(self, other_controls):
        if other_controls is None:
            return None
        n = len(self.world.cars)
        if len(other_controls) != n:
            raise ValueError(f"other_controls must have one entry per car ({n}), got {len(other_controls)}")
        rows = []
        for i, oc in enumerate(other_controls):
            if i == self.car.index:
                continue                                   # placeholder entry for the planning car
            a = np.asarray(oc, dtype=np.float32)
            if a.shape != (self.horizon, 2):
                raise ValueError(f"other_controls[{i}] must have shape ({self.horizon}, 2), got {a.shape}")
            rows.append(a)
        return np.stack(rows) if rows else None

    def reward_func(self, init_state, controls, other_controls=None, weights=None):
        """mpc_reward (naive_planner.py:33-77): the predicted reward of `controls` over the horizon."""
        return Tensor(self.reward_and_gradient(init_state, controls, other_controls, weights)[0])

    def reward_and_gradient(self, init_state, controls, other_controls=None, weights=None):
        """(R, dR/dcontrols [H, 2]): what tf.GradientTape over reward_func gives the reference's optimiser
        (naive_planner.py:124-125,153) and its IOC code (first_order_ioc.py:86)."""
        u = np.stack([np.asarray(c, dtype=np.float32) for c in controls]).reshape(1, self.horizon, 2)
        out = self._engine().mpc_reward_batch(self._world_state(init_state)[None], self._weights(weights), u,
                                              other_plans=self._other_plans(other_controls))
        return out["reward"][0], out["grad"][0]

    def generate_plan(self, init_state=None, weights=None, other_controls: Optional[List] = None,
                      use_lbfgs=False) -> List[Tensor]:
        """naive_planner.py:81-164.  Returns self.planned_controls (list of H Tensors of shape (2,))."""
        if use_lbfgs:
            raise NotImplementedError("use_lbfgs needs tensorflow_probability and is not on the accelerated path")
        # the extra_inits coast at friction * self.car.state[2] ** 2 -- the CAR's speed, whatever init_state says (:114)
        own_speed = None
        if self.extra_inits and init_state is not None:
            own_speed = np.asarray([np.asarray(self.car.state, dtype=np.float32)[2]], dtype=np.float32)
        out = self._engine().plan_batch(self._world_state(init_state)[None], self._weights(weights),
                                        other_plans=self._other_plans(other_controls), want_all=True,
                                        init_speed=own_speed)
        self.last_losses = out["all_losses"][0]
        self.last_best_init = int(out["best_init"][0])
        for control, val in zip(self.planned_controls, out["plans"][0]):
            control.assign(val)
        return self.planned_controls
