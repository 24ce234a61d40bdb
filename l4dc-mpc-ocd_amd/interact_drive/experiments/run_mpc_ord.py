"""Command-line driver.  Mirrors experiments/run_mpc_ord.py:19-127 (scenario table, seeds -> init
states, random / cmaes dispatch, history pickle name).  `vis` is out of scope (pyglet / moviepy).

    python -m l4dc_mpc_ocd_amd.interact_drive.experiments.run_mpc_ord finite_horizon cmaes --n_inits 3 --seed 1
"""
from argparse import ArgumentParser

import numpy as np

from .local_opt_scenario import local_opt_env
from .replanning_world import setup_world as replanning_env
from ..reward_design.mpc_ord import MPC_ORD, finite_horizon_env
from ... import scenarios


def fmt(arr):
    s = str(arr).replace("\n", ' ').replace('\t', " ")
    while '  ' in s:
        s = s.replace('  ', ' ')
    return s


def _env_entry(factory, scenario_factory, offset_axis, offset):
    """One row of the reference's `envs` table, with the constants taken from scenarios.py."""
    scn = scenario_factory()
    lo, hi = [0., 0., 0., 0.], [0., 0., 0., 0.]
    lo[offset_axis], hi[offset_axis] = -offset, offset
    return {'make_env': factory, 'eval_horizon': scn.desc.episode_len, 'init_offset_range': [lo, hi],
            'num_eval_samples': scn.desc.n_samples, 'tuned_weights': np.array(scn.tuned_weights)}


envs = {
    'local_opt': _env_entry(local_opt_env, scenarios.local_opt, 1, 0.1),
    'finite_horizon': _env_entry(finite_horizon_env, scenarios.finite_horizon, 0, 0.1),
    'replanning': _env_entry(replanning_env, scenarios.replanning, 0, 0.05),
}


def make_mpc_ord(scenario, horizon=None, n_inits=1, seed=1, save_path=None, **scenario_kwargs):
    """MPC_ORD over `scenario` at an arbitrary planning horizon with `n_inits` synthetic init states
    (scenarios.InitDistribution.sample: the seeded inverse-CDF sampler of the benchmark configs);
    scenario_kwargs go to the scenario factory (e.g. extra_inits=True)."""
    from ._build import world_from_scenario
    from .replanning_world import ReplanningCarWorld
    scn = scenarios.SCENARIOS[scenario](**dict(scenario_kwargs, **({} if horizon is None else {"horizon": horizon})))
    init_states = list(scn.init_dist.sample(n_inits, seed=seed))
    world_cls = ReplanningCarWorld if scenario == "replanning" else None
    car, _, world = world_from_scenario(scn, init_states[0], debug=True, world_cls=world_cls)
    return MPC_ORD(world, car, init_states, scn.desc.episode_len, save_path=save_path,
                   num_samples=scn.desc.n_samples)


def _save_path(car, init_states, args, optimization_seed):
    return (f'{args.optimizer}_{args.scenario}__designer_weights_{fmt(car.weights)}__'
            f'{args.n_inits if not args.one_by_one else fmt(init_states[0])}_init_seed_{args.seed}'
            f'_opt_seed_{optimization_seed}_sigma_{args.sigma}.pkl')


def run_opts_lockstep(env, groups_and_seeds, args):
    """What the reference's Pool(len(groups)).map(run_opt, ...) does (run_mpc_ord.py:83-90): one CMA-ES optimisation per
    (init group, optimisation seed) -- here advanced in lockstep, ONE episode launch per generation for all of them
    (reward_design.mpc_ord.optimize_cmaes_lockstep).  Returns [(mpc_ord, best), ...] in the order given."""
    from ..reward_design.mpc_ord import optimize_cmaes_lockstep
    car, world, _ = env['make_env'](debug=True)
    ords = [MPC_ORD(world, car, group, env['eval_horizon'], num_samples=env['num_eval_samples'],
                    save_path=_save_path(car, group, args, seed) if args.save else None)
            for group, seed in groups_and_seeds]
    res = optimize_cmaes_lockstep(ords, [seed for _, seed in groups_and_seeds], [args.sigma] * len(ords),
                                  popsize=args.popsize, maxiter=args.maxiter, maxfevals=args.maxfevals)
    if res.lockstep and res.generation_wall_seconds:
        print(f'{len(ords)} runs in lockstep: {len(res.generation_seconds)} generations, up to '
              f'{max(res.episodes_per_generation)} episodes per launch, median generation wall-clock '
              f'{np.median(res.generation_wall_seconds) * 1e3:.2f} ms for all runs together')
    return list(zip(res.runs, res.best))


def run_opt(env, init_states, args, optimization_seed):
    car, world, _ = env['make_env'](debug=True)
    save_path = _save_path(car, init_states, args, optimization_seed)
    mpc_ord = MPC_ORD(world, car, init_states, env['eval_horizon'], num_samples=env['num_eval_samples'],
                      save_path=save_path if args.save else None)
    if args.optimizer == 'random':
        best = mpc_ord.optimize_random_search(n_iter=args.n_random, seed=optimization_seed % (2 ** 32))
        return mpc_ord, best[0]
    assert args.optimizer == 'cmaes'
    best = mpc_ord.optimize_cmaes(sigma0=args.sigma, seed=optimization_seed, popsize=args.popsize,
                                  maxiter=args.maxiter, maxfevals=args.maxfevals)
    return mpc_ord, best


def main(argv=None):
    parser = ArgumentParser()
    parser.add_argument('scenario', type=str, choices=['local_opt', 'finite_horizon', 'replanning'])
    parser.add_argument('optimizer', type=str, choices=['random', 'cmaes', 'vis'])
    parser.add_argument('--n_inits', type=int, default=1)
    parser.add_argument('--seed', type=int, default=None)
    parser.add_argument('--one_by_one', action='store_true',
                        help='Runs single init optimization separately for each init.')
    parser.add_argument('--rand_inits', action='store_true')
    parser.add_argument('--sigma', type=float, default=0.05)
    # additions of this build (the reference hard-codes 400 random evaluations and pycma's defaults)
    parser.add_argument('--n_random', type=int, default=400)
    parser.add_argument('--popsize', type=int, default=None)
    parser.add_argument('--maxiter', type=int, default=None)
    parser.add_argument('--maxfevals', type=int, default=85, help='CMA-ES evaluations (85 are used in the paper plots)')
    parser.add_argument('--save', action='store_true', help='write the (weights, reward) history pickle')
    parser.add_argument('--opt_seeds', type=int, nargs='+', default=None,
                        help='several CMA-ES seeds: one optimisation per (init group, seed), all in lockstep')
    parser.add_argument('--sequential', action='store_true',
                        help='run the optimisations one after another instead of in lockstep (same results)')
    args = parser.parse_args(argv)
    assert args.n_inits >= 1
    assert args.seed != 0, 'CMA doesn\'t accept 0 seed'
    if args.optimizer == 'vis':
        raise SystemExit("'vis' renders GIFs / heat maps with pyglet + moviepy: outside the accelerated planner path")
    if args.n_inits == 1:
        args.one_by_one = False
    # under a launcher (python -m torch.distributed.run --nproc-per-node G ... run_mpc_ord ... --one_by_one): one rank per GPU;
    # the optimisations are dealt over the ranks (optimize_cmaes_lockstep), a single optimisation shards its population
    import os
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    own_group = False
    if world_size > 1:
        import torch
        import torch.distributed as dist
        if args.seed is None:
            raise SystemExit("--seed is required under a launcher: every rank must draw the same init states and seeds")
        if not dist.is_initialized():
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1))
            dist.init_process_group(os.environ.get("OCD_DIST_BACKEND", "nccl"))
            own_group = True
    env = envs[args.scenario]
    optimization_seed = np.random.randint(0, 2 ** 31) if world_size == 1 else (args.seed * 7919 + 1) % (2 ** 31)
    try:
        return _run(args, env, envs, optimization_seed)
    finally:
        if own_group:                                               # the group this call created ends with it
            import torch.distributed as dist
            if dist.is_initialized():
                dist.destroy_process_group()


def _run(args, env, envs, optimization_seed):
    if args.seed is None:
        args.seed = optimization_seed
    env_seeds = [(args.seed * 1000000 + i) % (2 ** 32) for i in range(args.n_inits)]
    car, world, init_states = env['make_env'](env_seeds=env_seeds, debug=True)
    init_states_groups = [[s] for s in init_states] if args.one_by_one else [init_states]
    print('init_states:', init_states_groups)
    opt_seeds = args.opt_seeds or [optimization_seed]
    jobs = [(group, seed) for group in init_states_groups for seed in opt_seeds]
    if args.optimizer == 'cmaes' and len(jobs) > 1 and not args.sequential:
        pairs = run_opts_lockstep(env, jobs, args)                  # the reference's Pool over groups, as ONE launch per generation
    else:
        pairs = [run_opt(env, group, args, seed) for group, seed in jobs]
    results = []
    for mpc_ord, best in pairs:
        finite = [h for h in mpc_ord.history if np.isfinite(h[1])]
        top = max(finite or mpc_ord.history, key=lambda a: a[1])
        print(f'evaluations {len(mpc_ord.history)}  designer-weights reward {mpc_ord.history[0][1]:.6f}  '
              f'best reward {top[1]:.6f}  best weights {fmt(top[0])}  non-finite costs {sum(mpc_ord.n_nonfinite)}'
              + (f'  stopped on {getattr(mpc_ord, "stop_reason", None)}' if args.optimizer == 'cmaes' else ''))
        if getattr(mpc_ord, "generation_seconds", None):
            gs = mpc_ord.generation_seconds
            print(f'CMA-ES generations {len(gs)}  median generation wall-clock {np.median(gs) * 1e3:.2f} ms')
        results.append((mpc_ord, best))
    return results


if __name__ == '__main__':
    main()
