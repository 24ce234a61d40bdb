"""Held-out generalisation sweep.  Mirrors experiments/generalization_data.py:26,64-107: the weights
chosen by each optimisation run are re-evaluated on 32 test inits drawn from the seeds
2**32-1, 2**32-2, ... ; the reference fans this out over Pool(8), here it is one launch."""
import numpy as np

from .run_mpc_ord import envs
from ..reward_design.mpc_ord import MPC_ORD


def test_init_seeds(n=40):
    return [2 ** 32 - i - 1 for i in list(range(n))]


def generalization_table(scenario: str, chosen_weights: dict, n_test_inits: int = 32, test_inits=None):
    """chosen_weights: {key: weight vector} (e.g. key = (n_inits, seed) as in the reference).
    Returns {key: float32 array [n_test_inits]} of designer returns (eval_weights_for_init values)."""
    env_config = envs[scenario]
    car, world, inits = env_config['make_env'](env_seeds=test_init_seeds())
    if test_inits is None:
        test_inits = inits[:n_test_inits]
    bord = MPC_ORD(world, car, [], env_config['eval_horizon'], num_samples=env_config['num_eval_samples'])
    keys = list(chosen_weights)
    table = bord.eval_weights_for_inits(np.stack([np.asarray(chosen_weights[k], dtype=np.float64) for k in keys]),
                                        np.asarray(test_inits))
    return {k: table[i] for i, k in enumerate(keys)}
