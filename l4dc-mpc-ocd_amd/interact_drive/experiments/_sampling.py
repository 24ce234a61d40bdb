"""Init-state sampling shared by the scenario factories (mpc_ord.py:168-181 and twins)."""
import numpy as np
import scipy.stats


def make_get_init_state(x, y, v):
    """x, y, v: (mean, std, (lo, hi)).  Same calls as the reference: np.random.seed(env_seed) then
    three scipy.stats.truncnorm.rvs draws.  (The stream depends on the scipy version, SURVEY.md 7,
    so parity tests pass init states explicitly.)"""

    def get_init_state(env_seed):
        np.random.seed(seed=env_seed)

        def sample(mean, std, rang):
            a, b = (rang[0] - mean) / std, (rang[1] - mean) / std
            return np.squeeze(scipy.stats.truncnorm.rvs(a, b) * std + mean)

        robot_x = sample(*x)
        robot_y = sample(*y)
        robot_init_speed = sample(*v)
        return np.array([robot_x, robot_y, robot_init_speed, np.pi / 2])

    return get_init_state
