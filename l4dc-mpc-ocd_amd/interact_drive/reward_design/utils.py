"""Mirrors interact_drive/reward_design/utils.py:2-14."""
import numpy as np

from ..tensor import Tensor
from .._describe import describe, engine_for
from ...scenarios import normalize_like_reference


def evaluate_weights(car, agent_weights, world, horizon=15, init_states=None):
    """Designer reward of one `horizon`-step episode planned with `agent_weights`.

    The reference swaps car.weights, re-creates the planner, resets the world and accumulates
    designer_reward_fn(past_state, ctrl[0]) over world.step(); here that is one episode launch."""
    # Quirk kept from the reference: `designer_reward_fn = car.reward_fn` is a bound method that reads
    # car.weights_tf at call time, i.e. AFTER `car.weights = agent_weights` -- the episode is scored
    # with the (normalised) agent weights, not with the designer's (utils.py:3-5,11).
    pa = car.planner_args
    w = np.asarray(agent_weights)
    w32 = normalize_like_reference(w, 1).astype(np.float32)       # the weights setter normalises once
    desc = describe(world, car, car.horizon, pa.get("learning_rate", 0.1), pa.get("n_iter", 100),
                    pa.get("extra_inits", False), episode_len=horizon, n_samples=1,
                    designer_weights=w32)
    out = engine_for(desc).rollout(np.asarray(car.init_state, dtype=np.float32)[None], w32[None])
    return Tensor(out["returns"][0])
