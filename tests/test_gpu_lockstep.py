"""Independent optimisation runs batched into one launch per generation.  The reference's only
parallel axis is a process pool over such runs (experiments/run_mpc_ord.py:83-90); here their episodes are rows of one index
(ocd_rollout_indexed) and their CMA-ES states advance in lockstep (ocd_cma_run_many, MPC_ORD.optimize_cmaes_many).
Every episode must be the episode ocd_rollout_episodes / the oracle computes for the same (candidate, init, reset), and
every run's history the history of the run alone, bit for bit."""
import pickle

import numpy as np
import pytest

from l4dc_mpc_ocd_amd import scenarios

pytestmark = pytest.mark.gpu
PI_2 = np.pi / 2


def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and bool(((a == b) | (np.isnan(a) & np.isnan(b))).all())


@pytest.mark.parametrize("name,H,mode", [("finite_horizon", 5, 0), ("replanning", 5, 0), ("replanning", 15, 4), ("merging", 10, 3),
                                         ("local_opt", 10, 2), ("finite_horizon", 7, 1)])
def test_indexed_rollout_equals_each_population_alone(hip, oracle, name, H, mode):
    """Three populations of different sizes over different init groups, one launch, shuffled index: each episode equals
    the oracle's episode of its own (population, inits) evaluation -- returns, trajectories, controls -- including the
    per-run reset numbers that pick the car a replanning world removes."""
    from l4dc_mpc_ocd_amd.engine import Engine
    scn = scenarios.SCENARIOS[name](horizon=H)
    d = scn.desc
    S = d.n_samples
    eng = Engine(scn, "cuda:0")
    eng.set_option("scan_mode", mode)
    shapes = [(3, 2), (1, 3), (4, 1)]                                   # (candidates, inits) per run
    inits = [scn.init_dist.sample(n, seed=40 + r) for r, (_, n) in enumerate(shapes)]
    cands = [np.stack([scenarios.planner_weights_fp32(c) for c in scn.candidate_weights(p, seed=50 + r)])
             for r, (p, _) in enumerate(shapes)]
    rows, ref = [], []
    p0 = n0 = 0
    for r, (P, N) in enumerate(shapes):
        for e in range(P * N * S):
            rows.append((p0 + e // (N * S), n0 + (e // S) % N, e))
        ref.append(oracle.rollout(d, inits[r], cands[r], want_traj=True))
        p0, n0 = p0 + P, n0 + N
    rows = np.array(rows, dtype=np.int32)
    perm = np.random.default_rng(1).permutation(len(rows))
    out = eng.rollout_indexed(np.concatenate(inits), np.concatenate(cands), rows[perm], want_traj=True)
    inv = np.argsort(perm)
    for k in ("returns", "traj", "ctrl"):
        assert same(out[k][inv], np.concatenate([q[k] for q in ref])), (name, H, mode, k)
    # and the same episodes through the flat entry point on the device
    flat = np.concatenate([eng.rollout(inits[r], cands[r])["returns"] for r in range(3)])
    assert same(out["returns"][inv], flat)
    with pytest.raises(ValueError):
        eng.rollout_indexed(np.concatenate(inits), np.concatenate(cands), [[99, 0, 0]])


def _runs(scn, R, n_inits, seed0):
    return [(list(scn.init_dist.sample(n_inits[r % len(n_inits)], seed=seed0 + r)), 5 + 3 * r, [0.05, 0.2, 0.1][r % 3]) for r in range(R)]


@pytest.mark.parametrize("name,groups,save", [("finite_horizon", None, True), ("replanning", None, True), ("finite_horizon", 1, False),
                                              ("finite_horizon", 2, False), ("replanning", 2, False), ("replanning", 3, False)])
def test_lockstep_histories_equal_the_runs_alone(hip, name, groups, save, tmp_path):
    """groups: launches per generation (round 6: the runs dealt to groups on their own streams, one group's tells and asks
    under the other's kernel; None = the automatic choice, four for these shapes: each group planned for a quarter of the chip).  With save paths a native call is one
    generation (the history on disk is complete after every generation); without, 32 generations go through the pipelined loop."""
    from l4dc_mpc_ocd_amd.interact_drive.experiments.run_mpc_ord import make_mpc_ord
    from l4dc_mpc_ocd_amd.interact_drive.reward_design.mpc_ord import MPC_ORD
    scn = scenarios.SCENARIOS[name](horizon=5)
    base = make_mpc_ord(name, horizon=5, n_inits=1, seed=1)
    runs = _runs(scn, 5, [1, 3, 2], 70)
    runs[3] = (runs[3][0], runs[3][1], 1e-13)                        # a vanishing step size: stops after one generation (tolx)
    paths = [str(tmp_path / f"run{r}.pkl") for r in range(5)] if save else None
    res = base.optimize_cmaes_many(runs, maxiter=6, save_paths=paths, groups=groups)
    assert res.lockstep and len(res.runs) == 5 and len(res.generation_seconds) == 6
    assert res.groups == groups if groups else res.groups in (2, 4)   # (five runs: four groups of one or two where the probe finds four queues)
    E = [9 * len(r[0]) * scn.desc.n_samples for r in runs]
    assert res.episodes_per_generation == [sum(E)] + [sum(E) - E[3]] * 5          # run 3 dropped out after generation 0
    for r, (inits, seed, sigma0) in enumerate(runs):
        alone = MPC_ORD(base.world, base.car, inits, base.designer_horizon, num_samples=base.num_samples)
        best = alone.optimize_cmaes(seed=seed, sigma0=sigma0, maxiter=6)
        got = res.runs[r]
        assert got.done and got.stop_reason == alone.stop_reason, (r, got.stop_reason, alone.stop_reason)
        assert ("tolx" in got.stop_reason) == (r == 3)
        assert len(got.history) == len(alone.history) == 1 + 9 * (1 if r == 3 else 6)
        for (wa, ra), (wb, rb) in zip(got.history, alone.history):
            assert np.array_equal(wa, wb) and same(ra, rb)
        assert got.history.seed == alone.history.seed == seed and got.iter == alone.iter
        assert np.array_equal(res.best[r], best) and got.es.best_f == alone.es.best_f
        assert got.n_nonfinite == alone.n_nonfinite and len(got.generation_seconds) == len(alone.generation_seconds)
        if save:
            with open(paths[r], "rb") as f:                          # the pickle of the run, complete
                hist = pickle.load(f)
            assert len(hist) == len(alone.history) and all(np.array_equal(a[0], b[0]) and same(a[1], b[1]) for a, b in zip(hist, alone.history))


@pytest.mark.parametrize("groups", [1, 2, 3])
def test_lockstep_redraws_nan_costs_like_the_run_alone(hip, groups):
    """A runaway init state (speed -6: no speed floor, simulation_utils.py:14) in ONE of three runs: its NaN candidates are
    redrawn by pycma's rule while the other runs go on; all three histories equal the runs alone."""
    from l4dc_mpc_ocd_amd.interact_drive.reward_design.mpc_ord import MPC_ORD, finite_horizon_env
    car, world, _ = finite_horizon_env(horizon=5, env_seeds=[1])
    scn = scenarios.finite_horizon(horizon=5)
    good = list(scn.init_dist.sample(2, seed=3))
    bad = [np.array([0.0, -0.9, 0.8, PI_2]), np.array([0.02, -0.9, -6.0, PI_2])]
    runs = [(good, 3, 0.3), (bad, 3, 0.3), (good[:1], 4, 0.3)]
    base = MPC_ORD(world, car, good, 15)
    res = base.optimize_cmaes_many(runs, popsize=16, maxiter=3, groups=groups)
    assert res.lockstep and res.groups == groups
    for r, (inits, seed, sigma0) in enumerate(runs):
        alone = MPC_ORD(world, car, inits, 15)
        alone.optimize_cmaes(seed=seed, sigma0=sigma0, popsize=16, maxiter=3)
        got = res.runs[r]
        assert got.stop_reason == alone.stop_reason == {"maxiter": 3}
        assert got.n_resampled == alone.n_resampled and (got.n_resampled > 0) == (r == 1)
        assert len(got.history) == len(alone.history) == 1 + 3 * 16 + got.n_resampled
        for (wa, ra), (wb, rb) in zip(got.history, alone.history):
            assert np.array_equal(wa, wb) and same(ra, rb)
        assert got.n_nonfinite == alone.n_nonfinite and got.es.counteval == alone.es.counteval


def test_cli_one_by_one_runs_in_lockstep(hip, capsys):
    """run_mpc_ord.py --one_by_one (the reference's Pool over single-init groups, run_mpc_ord.py:83-90) and several
    optimisation seeds go through the lockstep loop; the result list is what the sequential loop gave."""
    from l4dc_mpc_ocd_amd.interact_drive.experiments import run_mpc_ord
    res = run_mpc_ord.main(["finite_horizon", "cmaes", "--n_inits", "3", "--seed", "3", "--maxiter", "2", "--one_by_one",
                            "--opt_seeds", "11", "12"])
    out = capsys.readouterr().out
    assert len(res) == 6 and all(len(r[0].history) == 1 + 2 * 9 for r in res) and "lockstep" in out
    seq = run_mpc_ord.main(["finite_horizon", "cmaes", "--n_inits", "3", "--seed", "3", "--maxiter", "2", "--one_by_one",
                            "--opt_seeds", "11", "12", "--sequential"])
    for (a, _), (b, _) in zip(res, seq):
        assert all(np.array_equal(x[0], y[0]) and same(x[1], y[1]) for x, y in zip(a.history, b.history))


def test_lockstep_deals_the_runs_over_ranks(hip, tmp_path):
    """Under torch.distributed the RUNS are dealt over the ranks (the reference's Pool of processes, run_mpc_ord.py:83-90,
    as one process per GPU): each rank advances its runs in lockstep, one all_gather_object at the end.  Two gloo ranks on
    the one card: both hold all seven histories afterwards, bit for bit those of the one-process lockstep run."""
    import os
    import subprocess
    import sys
    from l4dc_mpc_ocd_amd.interact_drive.experiments.run_mpc_ord import make_mpc_ord
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import lockstep_ranks_helper as helper
    out = str(tmp_path / "ranks")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(30900 + os.getpid() % 200), helper.__file__, out, "replanning"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    scn = scenarios.replanning(horizon=5)
    base = make_mpc_ord("replanning", horizon=5, n_inits=1, seed=1)
    one = base.optimize_cmaes_many(helper.runs_of(scn), maxiter=4)
    assert one.lockstep and one.ranks == 1
    got = []
    for rank in range(2):
        with open(f"{out}.rank{rank}", "rb") as f:
            got.append(pickle.load(f))
    for g in got:
        assert g["ranks"] == 2 and g["lockstep"] and g["made_here"] == [k % 2 == g["rank"] for k in range(7)]
        for k in range(7):
            ref = one.runs[k]
            assert len(g["histories"][k]) == len(ref.history) == 1 + 4 * 9
            assert all(np.array_equal(a[0], b[0]) and same(a[1], b[1]) for a, b in zip(g["histories"][k], ref.history)), k
            assert g["seeds"][k] == ref.history.seed and g["stops"][k] == ref.stop_reason and g["iters"][k] == ref.iter
            assert np.array_equal(g["best"][k], one.best[k])
    # rank 0 made runs 0, 2, 4, 6 and rank 1 runs 1, 3, 5: their launches hold those runs' episodes only
    E = [9 * len(r[0]) * scn.desc.n_samples for r in helper.runs_of(scn)]
    assert got[0]["episodes"][0][0] == sum(E[0::2]) and got[0]["episodes"][1][0] == sum(E[1::2])


def test_cli_under_a_launcher_deals_the_runs_over_ranks(hip):
    """run_mpc_ord.py started by torch.distributed.run (two gloo ranks on the one card): the three --one_by_one
    optimisations are dealt over the ranks; every rank prints all three results."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(OCD_DIST_BACKEND="gloo", PYTHONPATH=root + os.pathsep + env.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(31200 + os.getpid() % 200), "-m", "l4dc_mpc_ocd_amd.interact_drive.experiments.run_mpc_ord",
                        "finite_horizon", "cmaes", "--n_inits", "3", "--seed", "3", "--maxiter", "2", "--one_by_one"],
                       capture_output=True, text=True, env=env, timeout=600, cwd=root)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert r.stdout.count("evaluations 19 ") == 6                  # 3 runs x (1 + 2 x 9) evaluations, printed by both ranks


def test_side_by_side_streams_and_the_concurrent_launches_option(hip, oracle):
    """The lockstep groups' streams (Engine.side_by_side_streams): four streams whose launches run side by side -- found by
    measurement, because HIP maps streams onto a few hardware queues and two streams on one queue run one after the other
    (profiles/r06_lockstep_streams.txt) -- and the option that plans each launch for its share of the chip.  Results never
    depend on either: the same episodes with the option at 1 and at 4, on the default stream and on a probed one, equal
    the oracle's bit for bit."""
    import time
    import torch
    from l4dc_mpc_ocd_amd.engine import Engine
    scn = scenarios.finite_horizon(horizon=5)
    eng = Engine(scn, "cuda:0")
    inits = scn.init_dist.sample(3, seed=8)
    w = np.stack([scenarios.planner_weights_fp32(c) for c in scn.candidate_weights(63, seed=9)])     # 189 episodes: a quarter of 28 runs
    ref = oracle.rollout(scn.desc, inits, w)["returns"].reshape(-1)
    whole = eng.rollout(inits, w)
    assert eng.last_launch()["mapping"] == "dpp_rows"               # alone: three wavefronts per workgroup, one per compute unit
    eng.set_option("concurrent_launches", 4)
    try:
        quarter = eng.rollout(inits, w)
        ll = eng.last_launch()
        assert ll["mapping"] == "one_wavefront" and ll["build_wavefronts_per_simd"] == 1 and ll["workgroups"] <= 256
        streams = eng.side_by_side_streams(4, inits[0], 189)
        # (four on every box seen so far: HIP's four hardware queues; the lockstep path itself takes what the probe finds)
        assert 2 <= len(streams) <= 4 and len({s.cuda_stream for s in streams}) == len(streams) == eng.side_by_side_probe["found"]
        assert eng.side_by_side_streams(2, inits[0], 189) == streams[:2]          # cached: no second probe
        with torch.cuda.stream(streams[-1]):
            on_stream = eng.rollout(inits, w)
        # the launches together take about as long as one (side by side), far from one after the other
        init_dev, w_dev = eng._to_dev(inits), eng._to_dev(w)
        rets = [torch.empty(189, dtype=torch.float32, device="cuda:0") for _ in streams]

        def timed(sts):
            best = float("inf")
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for s, r in zip(sts, rets):
                    eng._call(eng.lib.ocd_rollout_episodes, eng._h, init_dev.data_ptr(), w_dev.data_ptr(), 63, 3, 0, 189, r.data_ptr(),
                              None, None, s.cuda_stream)
                for s in sts:
                    s.synchronize()
                best = min(best, time.perf_counter() - t0)
            return best
        one, four = timed(streams[:1]), timed(streams)
        assert four < 1.6 * one, (one, four, len(streams))
        for r in rets:
            assert same(r.cpu().numpy(), ref)
    finally:
        eng.set_option("concurrent_launches", 1)
    for out in (whole, quarter, on_stream):
        assert same(out["returns"], ref)
