"""`local_opt_env` of the reference's experiments/local_opt_scenario.py, built from scenarios.local_opt."""
from ._build import world_from_scenario
from ._sampling import make_get_init_state
from ... import scenarios


def local_opt_env(env_seeds=[1], extra_inits=False, debug=True):
    scn = scenarios.local_opt(horizon=5, extra_inits=extra_inits)
    dist = scn.init_dist
    init_states = [make_get_init_state(dist.x, dist.y, dist.v)(s) for s in env_seeds]
    our_car, _, world = world_from_scenario(scn, init_states[0], debug=debug,
                                            visualizer_args=dict(name="Switch Lanes"))
    return our_car, world, init_states
