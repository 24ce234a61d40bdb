"""Linear-reward cars.  Mirrors interact_drive/car/linear_reward_car.py:12-55."""
from typing import Iterable, Union

import numpy as np

from .car import Car
from ..tensor import Tensor


class LinearRewardCar(Car):
    """A car whose reward is a linear function of features, evaluated on the GPU."""

    def __init__(self, env, init_state, weights, color: str = 'gray', opacity: float = 1.0,
                 friction: float = 0.2, **kwargs):
        super().__init__(env, init_state, color=color, opacity=opacity, friction=friction, **kwargs)
        weights = np.asarray(weights)
        self.weights_tf = Tensor(weights / np.linalg.norm(weights))

    def _reward_engine(self):
        from .._describe import describe, engine_for
        horizon = getattr(self, "horizon", 1)
        return engine_for(describe(self.env, self, horizon))

    def features(self, state, control) -> Tensor:
        ws = np.stack([np.asarray(s, dtype=np.float32) for s in state])
        feats, _ = self._reward_engine().reward_batch(ws[None], self.weights)
        return Tensor(feats[0])

    @property
    def weights(self):
        return self.weights_tf.numpy()

    @weights.setter
    def weights(self, weights):
        weights = np.asarray(weights)
        self.weights_tf.assign(weights / np.linalg.norm(weights))

    def reward_fn(self, state, control, weights=None):
        """reduce_sum(weights * features(state)) (linear_reward_car.py:49-55)."""
        ws = np.stack([np.asarray(s, dtype=np.float32) for s in state])
        w = self.weights if weights is None else np.asarray(weights, dtype=np.float32)
        _, rew = self._reward_engine().reward_batch(ws[None], w)
        return Tensor(rew[0])
