"""Scenario 1 at planning horizons 6 and 5.  Mirrors experiments/validate_finite_horizon.py:1-26: the designer
weights are scored on NUM_INITS init states spread over x0 +- 0.1; horizon 6 (n_iter = 200, mpc_ord.py:192)
switches lanes, horizon 5 does not.  The GIFs of the reference are out of scope; the costs are returned.

    python -m l4dc_mpc_ocd_amd.interact_drive.experiments.validate_finite_horizon
"""
import numpy as np

from ..reward_design.mpc_ord import MPC_ORD, finite_horizon_env

NUM_INITS = 3


def init_states_around(car):
    s = np.asarray(car.state, dtype=np.float64)
    off = np.array([0.1, 0.0, 0.0, 0.0])
    return np.linspace(s - off, s + off, NUM_INITS)


def main():
    """{horizon: cost of the designer weights} = MPC_ORD.eval_weights(car.weights) per horizon."""
    car, world, _ = finite_horizon_env(horizon=6, extra_inits=False)
    init_states = init_states_around(car)
    out = {}
    for horizon in (6, 5):
        car, world, _ = finite_horizon_env(horizon=horizon, extra_inits=False)
        bord = MPC_ORD(world, car, init_states, 15, save_path=None)
        out[horizon] = bord.eval_weights(car.weights)
        print(f"finite_horizon, planning horizon {horizon}, n_iter {car.planner_args.get('n_iter', 100)}: "
              f"cost of the designer weights over {NUM_INITS} inits = {out[horizon]:.6f}")
    return out


if __name__ == '__main__':
    main()
