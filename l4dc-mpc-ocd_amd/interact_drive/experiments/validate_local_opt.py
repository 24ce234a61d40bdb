"""Scenario 2 with 3 and with 6 control initialisations.  Mirrors experiments/validate_local_opt.py:1-27: the
designer weights are scored on NUM_INITS init states spread over y0 +- 0.1, first with the planner's three
control initialisations, then with `extra_inits` (six, naive_planner.py:112-116), which finds the lane change.
The GIFs of the reference are out of scope; the costs are returned.

    python -m l4dc_mpc_ocd_amd.interact_drive.experiments.validate_local_opt
"""
import numpy as np

from .local_opt_scenario import local_opt_env
from ..reward_design.mpc_ord import MPC_ORD

NUM_INITS = 3


def main():
    """{extra_inits: cost of the designer weights}."""
    car, world, _ = local_opt_env(extra_inits=False, debug=True)
    s = np.asarray(car.init_state, dtype=np.float64)
    off = np.array([0.0, 0.1, 0.0, 0.0])
    init_states = list(np.linspace(s - off, s + off, NUM_INITS))
    out = {}
    for extra in (False, True):
        car, world, _ = local_opt_env(extra_inits=extra, debug=True)
        bord = MPC_ORD(world, car, init_states, 15, save_path=None)
        out[extra] = bord.eval_weights(car.weights)
        print(f"local_opt, {6 if extra else 3} control initialisations: cost of the designer weights over "
              f"{NUM_INITS} inits = {out[extra]:.6f}")
    return out


if __name__ == '__main__':
    main()
