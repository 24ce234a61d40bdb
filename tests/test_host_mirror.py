"""Host-side mirror of the reference API (l4dc_mpc_ocd_amd.interact_drive): CPU-checkable parts.

Descriptor construction from the object graph, argument validation with the reference's error
behaviour, the numpy Tensor stand-in, the CMA-ES driver, fitness reduction types.
"""
import os
import pickle

import numpy as np
import pytest

from l4dc_mpc_ocd_amd import abi, scenarios, sharding
from l4dc_mpc_ocd_amd.interact_drive import Tensor, simulation_utils
from l4dc_mpc_ocd_amd.interact_drive._describe import describe
from l4dc_mpc_ocd_amd.interact_drive.car import FixedPlanCar, FixedVelocityCar, PlannerCar
from l4dc_mpc_ocd_amd.interact_drive.experiments import local_opt_scenario, merging, replanning_world
from l4dc_mpc_ocd_amd.interact_drive.planner import NaivePlanner
from l4dc_mpc_ocd_amd.interact_drive.reward_design import MPC_ORD, finite_horizon_env
from l4dc_mpc_ocd_amd.interact_drive.reward_design.cmaes import CMAES
from l4dc_mpc_ocd_amd.interact_drive.world import ThreeLaneCarWorld, TwoLaneCarWorld


def _mpc_desc(car, world, inits, T, S=1):
    m = MPC_ORD(world, car, inits, T, num_samples=S)
    pa = car.planner_args
    return describe(world, car, car.horizon, pa.get("learning_rate", 0.1), pa.get("n_iter", 100),
                    pa.get("extra_inits", False), episode_len=T, n_samples=S, designer_weights=m.designer_weights)


def test_factories_build_the_same_descriptors_as_scenarios_py():
    car, world, inits = finite_horizon_env(horizon=5, env_seeds=[1, 2])
    assert bytes(_mpc_desc(car, world, inits, 15)) == bytes(scenarios.finite_horizon(horizon=5).desc)
    car, world, inits = finite_horizon_env(horizon=6, env_seeds=[1], extra_inits=True)
    assert bytes(_mpc_desc(car, world, inits, 15)) == bytes(scenarios.finite_horizon(horizon=6, extra_inits=True).desc)
    car, world, inits = local_opt_scenario.local_opt_env(env_seeds=[3], extra_inits=True)
    assert bytes(_mpc_desc(car, world, inits, 15)) == bytes(scenarios.local_opt(horizon=5, extra_inits=True).desc)
    car, world, inits = replanning_world.setup_world(env_seeds=[1])
    assert bytes(_mpc_desc(car, world, inits, 20, S=2)) == bytes(scenarios.replanning(horizon=5).desc)
    car, o1, o2, world = merging.setup_world()
    assert bytes(_mpc_desc(car, world, [car.init_state], 15)) == bytes(scenarios.merging(horizon=5).desc)


def test_lanes_and_world_bookkeeping():
    w3, w2 = ThreeLaneCarWorld(), TwoLaneCarWorld()
    assert [float(l.p[0]) for l in w3.lanes] == pytest.approx([-0.1, 0.0, 0.1])
    assert [float(l.p[0]) for l in w2.lanes] == pytest.approx([-0.05, 0.05])
    assert all(tuple(l.n) == (-1.0, 0.0) for l in w3.lanes)
    assert float(w3.lanes[0].dist2median((0.0, 3.0))) == pytest.approx(0.01)
    car, world, _ = replanning_world.setup_world(env_seeds=[1])
    assert world.unlucky_car_idx == 2 and world._teleport_cars() == [1, 2, 1, 2]   # after setup's one reset()
    assert [c.index for c in world.cars] == [0, 1, 2]
    assert world.cars[1].control_already_determined_for_current_step


def test_tensor_stand_in():
    t = Tensor([1.0, 2.0])
    assert t.dtype == np.float32 and t.shape == (2,) and isinstance(t.numpy(), np.ndarray)
    assert float(t[0]) == 1.0
    t.assign([3.0, 4.0])
    assert t.tolist() == [3.0, 4.0]
    assert Tensor(2.5).numpy() == np.float32(2.5)


def test_argument_validation_matches_the_reference():
    # interact_drive/tests/test_simulation_utils.py:99-110: wrong shapes raise ValueError before any device work
    with pytest.raises(ValueError):
        simulation_utils.next_car_state(state=[0., 0.], control=[0., 0.], dt=0.1)
    with pytest.raises(ValueError):
        simulation_utils.next_car_state(state=[0., 0., 1., np.pi / 2], control=[0., 0., 0.], dt=0.1)
    with pytest.raises(ValueError):
        simulation_utils.batched_next_car_state(np.zeros((3, 3)), np.zeros((3, 2)), 0.1)
    car, world, _ = finite_horizon_env(horizon=5)
    planner = NaivePlanner(world, car, 5)
    with pytest.raises(NotImplementedError):
        planner.generate_plan(use_lbfgs=True)
    with pytest.raises(NotImplementedError):
        NaivePlanner(world, car, 5, leaf_evaluation=lambda s, c: 0)
    with pytest.raises(ValueError):
        planner._other_plans([None, np.zeros((4, 2))])               # wrong horizon
    with pytest.raises(ValueError):
        planner._world_state([np.zeros(4)])                         # one state for a two-car world


def test_unsupported_worlds_fail_loudly():
    world = ThreeLaneCarWorld()
    plain = PlannerCar(world, np.array([0., 0., 1., np.pi / 2]), horizon=5)
    world.add_car(plain)
    with pytest.raises(NotImplementedError):                       # arbitrary Python reward_fn is not compiled
        describe(world, plain, 5)
    world = ThreeLaneCarWorld()
    other = FixedVelocityCar(world, np.array([0, -0.6, 0.5, np.pi / 2]))
    car = merging.ThreeLaneTestCar(world, np.array([0., 0., 1., np.pi / 2]), horizon=5, weights=np.ones(7))
    world.add_cars([other, car])
    with pytest.raises(NotImplementedError):                       # planning car must be car 0
        describe(world, car, 5)


def test_other_plans_gather_reads_from_index_zero():
    """planner_car.py:66-75 quirk: plan[j] for j < len(plan) else default_control, at every step."""
    car, world, _ = replanning_world.setup_world(env_seeds=[1])
    d = describe(world, car, 6)
    scn = scenarios.Scenario("x", d, None, None)
    op = scn.other_plans()
    assert op.shape == (2, 6, 2)
    np.testing.assert_allclose(op[0, :, 1], [0, 2.7, 0, -2.7, 0, 0], rtol=1e-6)
    np.testing.assert_allclose(op[1, :, 1], [0, -2.7, 0, 2.7, 0, 0], rtol=1e-6)
    np.testing.assert_allclose(op[0, :, 0], [0, 0.7, 0, 0, 0, 0], rtol=1e-6)


def test_fitness_accumulation_types():
    # samples summed in fp32, inits in float64, / S, negated (mpc_ord.py:102,126-151)
    r = np.array([[[1e-8, 1.0], [3.0, 4.0]], [[0.5, 0.25], [1.0, 2.0]]], dtype=np.float32)
    cost = sharding.fitness_from_returns(r.reshape(-1), 2, 2, 2)
    assert cost.dtype == np.float64
    exp0 = -(float(np.float32(np.float32(1e-8) + np.float32(1.0))) + float(np.float32(7.0))) / 2
    assert cost[0] == exp0 and cost[1] == -(0.75 + 3.0) / 2


def test_batched_weight_normalisation_is_bitwise_the_per_row_one():
    rng = np.random.default_rng(3)
    for D in (6, 7, 8):
        W = rng.standard_normal((400, D)) * 10 ** rng.uniform(-3, 3, (400, 1))
        a = scenarios.planner_weights_fp32_batch(W)
        b = np.stack([scenarios.planner_weights_fp32(w) for w in W])
        assert a.dtype == np.float32 and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_planner_weight_normalisation_chain():
    w = np.array([-5, 0., 0., 0., -6., -50, -50])
    w32 = scenarios.planner_weights_fp32(w)
    assert w32.dtype == np.float32 and abs(np.linalg.norm(w32) - 1) < 1e-6
    assert np.array_equal(scenarios.planner_weights_fp32(w[None]), w32)      # (1, D) input, mpc_ord.py:69-70


def test_cmaes_minimises_a_quadratic():
    es = CMAES([1.0] * 7, 0.3, seed=3)
    assert es.lam == 4 + int(3 * np.log(7)) == 9                               # pycma's default popsize for n=7
    for _ in range(150):
        X = es.ask()
        es.tell(X, np.sum((X - 0.25) ** 2, axis=1))
    assert es.best_f < 1e-8 and np.allclose(es.mean, 0.25, atol=1e-3)


def test_cmaes_covariance_stays_bit_symmetric_and_the_run_is_pinned():
    """C is kept bit-symmetric (eigh reads the lower triangle only; any other reader sees the same matrix), the
    deferred part of tell() (finish_tell) changes nothing, and a seeded run is frozen as a regression value."""
    def run(call_finish):
        es = CMAES([0.5] * 7, 0.2, popsize=16, seed=11)
        for _ in range(40):
            X = es.ask()
            es.tell(X, np.sum((X - np.arange(7) * 0.1) ** 2, axis=1))
            assert np.array_equal(es.C, es.C.T)
            if call_finish:
                es.finish_tell()
        return es
    a, b = run(False), run(True)
    assert np.array_equal(a.mean, b.mean) and a.sigma == b.sigma and a.best_f == b.best_f
    assert np.array_equal(a.best_x, b.best_x)
    assert np.allclose(a.mean, np.arange(7) * 0.1, atol=5e-3)
    # frozen on numpy 2.2 / OpenBLAS (a changed update rule or sampling order moves this far beyond 1e-9)
    assert abs(float(np.sum(a.mean)) - 2.1) < 2e-2
    sig = (float(a.sigma), float(a.best_f))
    assert sig == (float(b.sigma), float(b.best_f))


def test_native_cmaes_is_the_numpy_twin():
    """csrc/ocd_cma.c (what MPC_ORD.optimize_cmaes runs) against the numpy CMAES: the normal deviates are
    numpy.random.RandomState's stream bit for bit (first population identical), later populations agree to rounding
    (both sample with the symmetric root of C, independent of the eigenvector order / sign of LAPACK vs Jacobi), the
    float64 fitness reduction is sharding.fitness_from_returns bit for bit, drawing the deviates ahead (prepare) does
    not change the stream."""
    from l4dc_mpc_ocd_amd.interact_drive.reward_design.cmaes import NativeCMAES, fitness_from_returns_native
    f = lambda X: np.sum((X - np.arange(7) * 0.1) ** 2, axis=1) + 0.1 * np.sin(5 * X[:, 0])
    a, b = CMAES([0.5] * 7, 0.2, popsize=16, seed=11), NativeCMAES([0.5] * 7, 0.2, popsize=16, seed=11)
    assert b.lam == a.lam == 16 and NativeCMAES([0.0] * 7, 0.1).lam == 9
    Xa, Xb = a.ask(), b.ask().copy()
    assert np.array_equal(Xa, Xb)
    for g in range(60):
        a.tell(Xa, f(Xa)); b.tell(Xb, f(Xb))
        if g % 3 == 0:
            b.prepare()
        Xa, Xb = a.ask(), b.ask().copy()
        assert np.abs(Xa - Xb).max() < 1e-9 * max(1.0, np.abs(Xa).max()), g
    assert abs(a.sigma - b.sigma) < 1e-9 * a.sigma and np.allclose(a.mean, b.mean, rtol=0, atol=1e-9)
    assert np.allclose(a.C, b.C, rtol=1e-7, atol=1e-12) and np.array_equal(b.C, b.C.T)
    assert b.gen == 60 and b.counteval == 60 * 16 and abs(a.best_f - b.best_f) < 1e-9
    assert np.allclose(a.best_x, b.best_x, atol=1e-9)
    es = NativeCMAES([1.0] * 7, 0.3, seed=3)                       # minimises a quadratic, stops on tolx
    for _ in range(400):
        X = es.ask()
        es.tell(X, np.sum((X - 0.25) ** 2, axis=1))
        if es.stop():
            break
    assert es.best_f < 1e-12 and np.allclose(es.mean, 0.25, atol=1e-5) and es.gen < 400
    rng = np.random.default_rng(0)
    for P, N, S in ((64, 32, 1), (5, 3, 2), (1, 1, 4)):
        r = (rng.standard_normal(P * N * S) * 10).astype(np.float32)
        assert np.array_equal(fitness_from_returns_native(r, P, N, S), sharding.fitness_from_returns(r, P, N, S))
    with pytest.raises(ValueError):
        NativeCMAES([0.0] * 7, -1.0)


def test_cma_library_exports_what_its_header_declares():
    import ctypes
    import re
    from l4dc_mpc_ocd_amd.interact_drive.reward_design.cmaes import load_cma_library
    lib = load_cma_library()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names = re.findall(r"^\w[\w\s\*]*?\b(ocd_\w+)\s*\(", open(os.path.join(root, "include", "ocd_cma.h")).read(), re.M)
    assert set(names) >= {"ocd_cma_create", "ocd_cma_destroy", "ocd_cma_ask", "ocd_cma_tell", "ocd_cma_prepare",
                          "ocd_cma_state", "ocd_cma_popsize", "ocd_fitness_from_returns", "ocd_cma_resample",
                          "ocd_cma_stop_state", "ocd_cma_abi_version", "ocd_normalise_weights", "ocd_cma_stop", "ocd_cma_run",
                          "ocd_eval_generations"}
    for n in names:
        assert isinstance(getattr(lib, n), ctypes._CFuncPtr), n


def test_cmaes_twins_stay_identical_through_non_finite_costs():
    """VERDICT round 3 / ADVICE: one NaN cost used to stay wherever the native insertion sort put it (ranked best).
    Both twins rank NaN last (np.argsort's order), count non-finite costs, and redraw a slot the same way."""
    from l4dc_mpc_ocd_amd.interact_drive.reward_design.cmaes import NativeCMAES
    for fit in ([3, np.nan, 1, 2, 0.5, 4], [np.nan, np.nan, 1, 2, 0.5, 4], [np.inf, 3, np.nan, -np.inf, 0.5, 4],
                [np.nan] * 6):
        a, b = CMAES([0.0] * 4, 0.2, popsize=6, seed=5), NativeCMAES([0.0] * 4, 0.2, popsize=6, seed=5)
        Xa, Xb = a.ask(), b.ask()
        na, nb = a.tell(Xa, fit), b.tell(Xb, fit)
        assert na == nb == int(np.sum(~np.isfinite(fit))) and a.last_nonfinite == b.last_nonfinite == na
        assert np.allclose(a.mean, b.mean, rtol=0, atol=1e-12) and abs(a.sigma - b.sigma) < 1e-12, fit
        if not np.all(np.isnan(fit)):
            assert a.best_f == b.best_f == np.nanmin(fit)
    # the native ranking IS np.argsort(kind='stable') on costs with NaN / inf / ties: a population whose best
    # candidates are unique picks the same mean
    rng = np.random.default_rng(3)
    f = lambda X: np.sum((X - 0.3) ** 2, axis=1)
    a, b = CMAES([0.5] * 7, 0.2, popsize=16, seed=11), NativeCMAES([0.5] * 7, 0.2, popsize=16, seed=11)
    total = 0
    for g in range(40):
        Xa, Xb = a.ask(), b.ask()
        assert np.abs(Xa - Xb).max() < 1e-9, g
        fa = f(Xa)
        bad = rng.random(16) < 0.25
        fa[bad] = rng.choice([np.nan, np.inf, -np.inf, np.nan], size=int(bad.sum()))
        a.prepare(); b.prepare()                                     # (optimize_cmaes draws ahead while the GPU works)
        rows = np.nonzero(np.isnan(fa))[0][:2]                       # pycma-style rejection of (some of) the NaN slots
        if rows.size:
            Ra, Rb = a.resample(rows), b.resample(rows)
            assert np.abs(Ra - Rb).max() < 1e-9 and np.array_equal(Xb[rows], Rb) and np.array_equal(Xa[rows], Ra)
            fa[rows] = f(Ra)
        na, nb = a.tell(Xa, fa), b.tell(Xb, fa)
        total += na
        assert na == nb == int(np.sum(~np.isfinite(fa)))
        assert np.allclose(a.mean, b.mean, rtol=0, atol=1e-9) and abs(a.sigma - b.sigma) < 1e-9 * a.sigma, g
        assert a.stop() == b.stop()
    assert total > 20 and a.nonfinite_total == b.nonfinite_total == total
    with pytest.raises(ValueError):                                  # tell() refuses a population it did not draw
        X = b.ask()
        b.tell(X + 1.0, f(X))
    with pytest.raises(IndexError):
        b.resample([99])


def test_cmaes_termination_follows_pycma_defaults():
    """mpc_ord.py:41 calls fmin2 with pycma's default options: a default iteration cap, tolfunhist on a flat
    function, tolx on a solved one, a dict of the satisfied conditions -- the same answers from both twins."""
    from l4dc_mpc_ocd_amd.interact_drive.reward_design.cmaes import NativeCMAES
    for cls in (CMAES, NativeCMAES):
        es = cls([0.0] * 7, 0.1)                                     # the reference's shape: 7 weights, popsize 9
        assert es.lam == 9 and es.opts["maxiter"] == 100 + 150 * (7 + 3) ** 2 // 9 ** 0.5 == 5100
        assert es.opts["tolstagnation"] == int(100 + 100 * 7 ** 1.5 / 9) and es.stop() == {}
        for g in range(30):                                          # a constant cost: tolfun at once, flat fitness next
            X = es.ask()
            es.tell(X, np.full(9, 2.5))
            why = es.stop()
            if "tolflatfitness" in why:
                break
        assert "tolfun" in why and g == 1, why
        es = cls([1.0] * 7, 0.3, seed=3)
        for g in range(600):
            X = es.ask()
            es.tell(X, np.sum((X - 0.25) ** 2, axis=1))
            why = es.stop()
            if why:
                break
        assert ("tolfun" in why or "tolx" in why or "tolfunhist" in why) and es.best_f < 1e-12 and g < 599, why
        es = cls([1.0] * 7, 0.3, seed=3)
        for g in range(50):
            X = es.ask()
            es.tell(X, np.sum((X - 0.25) ** 2, axis=1))
            if es.stop(maxiter=7):
                break
        assert g == 6 and es.stop(maxiter=7) == {"maxiter": 7} and es.stop(maxfevals=20) == {"maxfevals": 20}
        with pytest.raises(TypeError):
            es.stop(tolfoo=1)
        es = cls([0.0] * 3, 1e-3, seed=2)                            # a step size far too small: tolfacupx
        for g in range(400):
            X = es.ask()
            es.tell(X, -X[:, 0])                                     # unbounded descent direction
            why = es.stop()
            if why:
                break
        assert "tolfacupx" in why, why


def test_tolstagnation_compares_adjacent_windows():
    """pycma's stagnation rule compares the newest l generations with the l just BEFORE them
    (histbest[:l] vs histbest[l:2l], newest first), not with the start of the run.  A run that improves for a
    hundred generations and then sits on a noisy plateau must stop on tolstagnation soon after the two windows
    both lie on the plateau -- against the (worst) costs of the start it never would -- and both twins must stop in
    the same generation."""
    from l4dc_mpc_ocd_amd.interact_drive.reward_design.cmaes import NativeCMAES
    quiet = dict(tolfacupx=np.inf, tolupsigma=np.inf, tolconditioncov=np.inf, tolx=0.0)
    stopped = []
    for cls in (CMAES, NativeCMAES):
        es = cls([0.0] * 7, 0.1, seed=5)
        rng = np.random.default_rng(17)                              # the costs do not depend on X: same for both twins
        for g in range(1200):
            X = es.ask()
            es.tell(X, max(1.0, 100.0 - g) + 1e-3 * rng.random(9))
            why = es.stop(**quiet)
            if why:
                break
        assert set(why) == {"tolstagnation"}, why
        # the rule needs gen > N (5 + 100 / popsize) = 112.8 and 2 l < len(history), l = max(305 / 10, len / 10);
        # both windows reach the plateau (generation 99 on) once 2 l <= len - 99
        assert 160 <= g <= 400, g
        stopped.append(g)
        # while the run still improves every generation the rule stays quiet
        es = cls([0.0] * 7, 0.1, seed=5)
        for g in range(400):
            X = es.ask()
            es.tell(X, 1000.0 - g + 1e-3 * rng.random(9))
            assert "tolstagnation" not in es.stop(**quiet), g
    assert stopped[0] == stopped[1], stopped


def test_noeffect_rules_fire_when_a_step_no_longer_moves_the_mean():
    """pycma's noeffectaxis / noeffectcoord: a mean of 1e6 with sigma 1e-12 cannot be moved by 0.2 sigma in any
    coordinate; an ordinary search never reports either.  Same answers from both twins."""
    from l4dc_mpc_ocd_amd.interact_drive.reward_design.cmaes import NativeCMAES
    for cls in (CMAES, NativeCMAES):
        es = cls([1e6] * 3, 1e-12, seed=4)
        X = es.ask()
        es.tell(X, np.arange(es.lam, dtype=np.float64))
        why = es.stop()
        assert "noeffectcoord" in why and "noeffectaxis" in why and why["noeffectaxis"] is None, why
        es = cls([1.0] * 5, 0.3, seed=4)
        for g in range(40):
            X = es.ask()
            es.tell(X, np.sum((X - 0.25) ** 2, axis=1))
            assert not {"noeffectcoord", "noeffectaxis"} & set(es.stop()), g


def test_a_stale_cma_library_is_refused(tmp_path, monkeypatch):
    """a libocd_cma.so built from an older header must not be bound silently."""
    from l4dc_mpc_ocd_amd.interact_drive.reward_design import cmaes
    lib = cmaes.load_cma_library()
    assert lib.ocd_cma_abi_version() == cmaes._header_abi_version(
        os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "ocd_cma.h"))
    monkeypatch.setattr(cmaes, "_CMA_LIB", None)
    monkeypatch.setattr(cmaes, "_header_abi_version", lambda path: 99)
    with pytest.raises(RuntimeError, match="ABI"):
        cmaes.load_cma_library()


def test_cmaes_minimises_rosenbrock_and_is_deterministic():
    def rosen(X):
        X = np.atleast_2d(X)
        return np.sum(100.0 * (X[:, 1:] - X[:, :-1] ** 2) ** 2 + (1 - X[:, :-1]) ** 2, axis=1)

    def run(seed, gens=600):
        es = CMAES([-1.0, 1.5, 0.5, -0.5], 0.5, seed=seed)
        trace = []
        for _ in range(gens):
            X = es.ask()
            es.tell(X, rosen(X))
            trace.append(es.best_f)
            if es.stop(last_fitness=rosen(X)):
                break
        return es, trace

    es, trace = run(4)          # (seeds 5 and 9 end in the 4-D Rosenbrock function's local minimum at f = 3.70, as CMA-ES may)
    assert es.best_f < 1e-10 and np.allclose(es.best_x, 1.0, atol=1e-4)
    assert all(b <= a for a, b in zip(trace, trace[1:]))                        # best-so-far never gets worse
    es2, trace2 = run(4)
    assert trace == trace2 and np.array_equal(es.mean, es2.mean)                # same seed, same run
    es3, trace3 = run(6)
    assert trace3 != trace                                                       # another seed, another sample path
    # ask() hands back lambda rows and tell() counts them
    es4 = CMAES([0.0] * 7, 0.1, popsize=16, seed=1)
    X = es4.ask()
    assert X.shape == (16, 7)
    es4.tell(X, np.arange(16.0))
    assert es4.counteval == 16 and es4.gen == 1


def test_describe_sets_the_planner_assumptions_like_planner_car():
    """planner_car.py:66-75: beyond its plan a FixedPlanCar is assumed to use its default_control, a
    FixedControlCar (no .plan attribute) is assumed to do (0, 0) whatever its real control is; the teleport
    cycle of ReplanningCarWorld has period 2 and starts with the car the next reset() removes."""
    from l4dc_mpc_ocd_amd.interact_drive.car import FixedControlCar, FixedPlanCar
    car, world, _ = replanning_world.setup_world(env_seeds=[1])
    d = describe(world, car, 6)
    assert d.teleport_period == 2 and d.teleport_step == 4
    first = 3 - world.unlucky_car_idx
    assert list(d.teleport_car[:2]) == [first, 3 - first]
    assert [tuple(d.other_assumed_default[j][:]) for j in range(2)] == [tuple(d.other_default[j][:]) for j in range(2)]
    world = ThreeLaneCarWorld()
    ego = merging.ThreeLaneTestCar(world, np.array([0., 0., 1., np.pi / 2]), horizon=5, weights=np.ones(7), check_plans=True)
    fc = FixedControlCar(world, np.array([0.1, -0.5, 0.5, np.pi / 2]), (0.3, -0.2))
    fp = FixedPlanCar(world, np.array([-0.1, -0.5, 0.5, np.pi / 2]), plan=[(0.1, 0.0), (0.2, 0.1)], default_control=(0.4, 0.0))
    world.add_cars([ego, fc, fp])
    d = describe(world, ego, 5)
    assert tuple(np.float32(v) for v in d.other_default[0][:]) == (np.float32(0.3), np.float32(-0.2))    # real control
    assert tuple(d.other_assumed_default[0][:]) == (0.0, 0.0)                                             # assumed by the planner
    assert tuple(np.float32(v) for v in d.other_assumed_default[1][:]) == (np.float32(0.4), np.float32(0.0))
    op = scenarios.Scenario("x", d, None, None).other_plans()
    np.testing.assert_allclose(op[0], np.zeros((5, 2)))
    np.testing.assert_allclose(op[1][:, 0], [0.1, 0.2, 0.4, 0.4, 0.4], rtol=1e-6)
    assert np.array_equal(op[1], fp.script_row(5)) and ego.assumed_other_controls()[1].shape == (5, 2)


def test_history_pickle_loads_in_the_reference(tmp_path):
    """bar_plot.py:105-130 unpickles interact_drive.reward_design.mpc_ord.list2: the pickle must name that
    path, and load in a process that only has the reference's module tree (simulated by a stub)."""
    import subprocess
    import sys
    from l4dc_mpc_ocd_amd.interact_drive.reward_design.mpc_ord import list2
    h = list2()
    h.seed = 7
    h.append((np.arange(3.0), -2.5))
    p = tmp_path / "hist.pkl"
    with open(p, "wb") as f:
        pickle.dump(h, f)
    raw = p.read_bytes()
    assert b"interact_drive.reward_design.mpc_ord" in raw and b"l4dc" not in raw
    stub = tmp_path / "ref" / "interact_drive" / "reward_design"
    stub.mkdir(parents=True)
    (tmp_path / "ref" / "interact_drive" / "__init__.py").write_text("")
    (stub / "__init__.py").write_text("")
    (stub / "mpc_ord.py").write_text("class list2(list):\n    def __init__(self, *a, **k):\n        super().__init__(*a, **k)\n")
    code = ("import pickle, sys; sys.path.insert(0, sys.argv[1]); h = pickle.load(open(sys.argv[2], 'rb')); "
            "print(type(h).__module__, h.seed, h[0][1], [float(v) for v in h[0][0]])")
    r = subprocess.run([sys.executable, "-c", code, str(tmp_path / "ref"), str(p)], capture_output=True, text=True,
                       cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr
    assert r.stdout.split() == ["interact_drive.reward_design.mpc_ord", "7", "-2.5", "[0.0,", "1.0,", "2.0]"]


def test_pickle_names_the_reference_path_without_touching_sys_modules(tmp_path):
    """Dumping a history imports nothing and registers nothing (the class pickles through copyreg as
    getattr(import_module('interact_drive.reward_design.mpc_ord'), 'list2')); loading one in a process without the
    reference resolves the three names lazily through a finder at the END of sys.meta_path, so a real, importable
    `interact_drive` always wins and `import interact_drive.car` is never shadowed by a stub."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    real = tmp_path / "real" / "interact_drive"
    (real / "reward_design").mkdir(parents=True)
    (real / "__init__.py").write_text("REAL = True\n")
    (real / "car.py").write_text("class Car: pass\n")
    (real / "reward_design" / "__init__.py").write_text("")
    (real / "reward_design" / "mpc_ord.py").write_text("class list2(list):\n    def __init__(self, *a, **k):\n        super().__init__(*a, **k)\n")
    code = """
import pickle, sys, importlib
sys.path.insert(0, sys.argv[1])
import l4dc_mpc_ocd_amd.interact_drive.reward_design.mpc_ord as m
assert not any(k == 'interact_drive' or k.startswith('interact_drive.') for k in sys.modules), 'import registered stubs'
h = m.list2(); h.seed = 3; h.append(([1.0], -1.0))
raw = pickle.dumps(h)
assert not any(k == 'interact_drive' or k.startswith('interact_drive.') for k in sys.modules), 'dump registered stubs'
assert b'interact_drive.reward_design.mpc_ord' in raw and b'l4dc' not in raw
if len(sys.argv) > 2:                       # the reference becomes importable later in the same process
    sys.path.insert(0, sys.argv[2])
    import interact_drive.car               # the REAL package, not a stub
    assert interact_drive.REAL
    g = pickle.loads(raw)
    assert type(g).__module__ == 'interact_drive.reward_design.mpc_ord' and type(g) is not m.list2 and g.seed == 3
else:                                       # no reference anywhere: the lazy finder resolves the path to the mirror's class
    g = pickle.loads(raw)
    assert type(g) is m.list2 and g.seed == 3 and g[0][1] == -1.0
    assert sys.modules['interact_drive']._ocd_pickle_stub
    try:
        importlib.import_module('interact_drive.car')
    except ModuleNotFoundError:
        pass
    else:
        raise AssertionError('the stub package must not resolve other names')
print('ok')
"""
    for extra in ([], [str(tmp_path / "real")]):
        r = subprocess.run([sys.executable, "-c", code, root] + extra, capture_output=True, text=True, cwd=str(tmp_path))
        assert r.returncode == 0 and r.stdout.strip() == "ok", r.stderr


def test_history_pickle_format(tmp_path):
    h = __import__("l4dc_mpc_ocd_amd.interact_drive.reward_design.mpc_ord", fromlist=["list2"]).list2()
    h.seed = 5
    h.append((np.ones(3), -1.5))
    p = tmp_path / "h.pkl"
    with open(p, "wb") as f:
        pickle.dump(h, f)
    with open(p, "rb") as f:
        g = pickle.load(f)
    assert g[0][1] == -1.5 and list(g[0][0]) == [1, 1, 1]


def test_native_weight_normalisation_is_the_numpy_chain():
    """csrc/ocd_cma.c:ocd_normalise_weights (three float64 normalisations + fp32 cast in one call, written straight
    into the pinned rows the kernel reads) must be planner_weights_fp32 bit for bit; which summation order reproduces
    numpy's BLAS dot is found by a self-check, and the numpy forms remain the fallback."""
    from l4dc_mpc_ocd_amd import scenarios as sc
    rng = np.random.default_rng(11)
    for D in (2, 6, 7, 11):
        W = rng.standard_normal((64, D)) * np.exp(rng.uniform(-3, 3, (64, 1)))
        ref = np.stack([sc.planner_weights_fp32(r) for r in W])
        got = sc.planner_weights_fp32_batch(W)
        assert got.dtype == np.float32 and np.array_equal(got.view(np.uint32), ref.view(np.uint32))
        buf = np.zeros((64, D), dtype=np.float32)
        assert sc.planner_weights_fp32_batch(W, out=buf) is buf and np.array_equal(buf, ref)
        assert np.array_equal(W, W.copy())                                   # (the input is only read)
    assert sc._NATIVE_NORMALISE.get(7) in (0, 1, None)
    # the numpy fallback gives the same bits
    keep = dict(sc._NATIVE_NORMALISE)
    try:
        for D in list(sc._NATIVE_NORMALISE):
            sc._NATIVE_NORMALISE[D] = None
        W = rng.standard_normal((16, 7))
        assert np.array_equal(sc.planner_weights_fp32_batch(W), np.stack([sc.planner_weights_fp32(r) for r in W]))
    finally:
        sc._NATIVE_NORMALISE.update(keep)


def test_row_dots_is_the_per_row_blas_dot():
    """The batched norm of the candidate weights must be the per-row call bit for bit (or fall back to it)."""
    from l4dc_mpc_ocd_amd import scenarios as sc
    rng = np.random.default_rng(5)
    for D in (6, 7, 11):
        W = rng.standard_normal((257, D)) * np.exp(rng.uniform(-6, 6, (257, 1)))
        ref = np.array([r.dot(r) for r in W])
        assert np.array_equal(sc.row_dots(W), ref)
        assert np.array_equal(sc._row_dots_loop(W), ref)
        assert D in sc._ROW_DOTS_BATCHED_OK
    # a numpy / BLAS whose batched kernel differed would be caught by the self-check and routed to the loop
    sc._ROW_DOTS_BATCHED_OK[7] = False
    try:
        W = rng.standard_normal((64, 7))
        assert np.array_equal(sc.row_dots(W), np.array([r.dot(r) for r in W]))
    finally:
        del sc._ROW_DOTS_BATCHED_OK[7]
    for row in rng.standard_normal((50, 7)):
        assert np.array_equal(sc.planner_weights_fp32_batch(row[None])[0], sc.planner_weights_fp32(row))


def test_native_generation_loop_is_the_python_loop():
    """csrc/ocd_cma.c:ocd_cma_run (whole generations without returning to the interpreter: what MPC_ORD.optimize_cmaes
    runs in one process) against the same generations driven call by call from Python -- same population, costs,
    history rows, termination and state; a NaN cost hands the generation back untold.  The episode launch is a function
    pointer: here a Python callback that scores the normalised fp32 weights it finds in the "pinned" rows."""
    import ctypes as C
    from l4dc_mpc_ocd_amd.interact_drive.reward_design.cmaes import NativeCMAES, RunArgs, fitness_from_returns_native
    from l4dc_mpc_ocd_amd.scenarios import _native_normalise_variant, planner_weights_fp32_batch
    D, P, N, S = 7, 16, 3, 2
    variant = _native_normalise_variant(D)
    if variant is None:
        pytest.skip("no native summation order reproduces numpy's dot on this machine")
    target = np.linspace(-0.5, 0.4, D).astype(np.float32)
    nan_at = {"calls": 0, "when": 5}

    def score(w32):                                    # returns [P, N, S] fp32 from the fp32 planner weights
        base = -np.sum((w32 - target) ** 2, axis=1, dtype=np.float32)
        r = (base[:, None, None] * (1.0 + 0.1 * np.arange(N, dtype=np.float32)[None, :, None])
             - 0.01 * np.arange(S, dtype=np.float32)[None, None, :]).astype(np.float32)
        return r

    w_pinned = np.zeros((P, D), dtype=np.float32)
    ret_pinned = np.zeros(P * N * S, dtype=np.float32)
    ROLL = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)
    SYNC = C.CFUNCTYPE(C.c_int32, C.c_void_p)

    def rollout(scn, init, w, p, n, e0, e1, ret, traj, ctrl, stream):
        assert (p, n, e0, e1) == (P, N, 0, P * N * S) and w == w_pinned.ctypes.data and ret == ret_pinned.ctypes.data
        nan_at["calls"] += 1
        r = score(w_pinned)
        if nan_at["calls"] == nan_at["when"]:
            r[3, 1, 0] = np.nan
        ret_pinned[:] = r.reshape(-1)
        return 0

    roll_c, sync_c = ROLL(rollout), SYNC(lambda stream: 0)
    x0 = [0.3, -0.2, 0.1, 0.0, 0.2, -0.1, 0.4]
    a = NativeCMAES(x0, 0.2, popsize=P, seed=9)
    chunk = 7
    hist_w, hist_c = np.empty((chunk, P, D)), np.empty((chunk, P))
    secs, nonf = np.zeros((chunk, 8)), np.zeros(chunk, dtype=np.int32)
    args = RunArgs()
    args.N, args.S = N, S
    args.w_pinned, args.ret_pinned = w_pinned.ctypes.data, ret_pinned.ctypes.data
    args.rollout, args.sync = C.cast(roll_c, C.c_void_p).value, C.cast(sync_c, C.c_void_p).value
    args.normalise_variant, args.max_generations = variant, chunk
    args.hist_w, args.hist_cost, args.seconds, args.nonfinite = (hist_w.ctypes.data, hist_c.ctypes.data, secs.ctypes.data,
                                                              nonf.ctypes.data)
    native_hist, gens = [], 0
    overrides = dict(maxiter=23)
    while True:
        done, why, pending = a.run(args, overrides)
        for g in range(done):
            native_hist.extend(zip(hist_w[g].copy(), -hist_c[g]))
        gens += done
        assert np.all(secs[:done, 0] > 0) and np.all(nonf[:done] == 0)
        if pending:                                    # generation `done` came back evaluated but untold
            assert np.isnan(a._f).sum() == 1 and np.isnan(hist_c[done]).sum() == 1
            native_hist.extend(zip(hist_w[done].copy(), -hist_c[done]))
            f = a._f.copy()
            rows = np.nonzero(np.isnan(f))[0]
            Xr = a.resample(rows)
            f[rows] = fitness_from_returns_native(score(planner_weights_fp32_batch(Xr)).reshape(-1), len(rows), N, S)
            a.tell(a._X, f)
            gens += 1
            why = a.stop(**overrides)
        if why:
            break
    assert why == {"maxiter": 23} and gens == 23 and a.gen == 23
    # the same run, call by call
    b = NativeCMAES(x0, 0.2, popsize=P, seed=9)
    py_hist = []
    for g in range(23):
        X = b.ask()
        w32 = planner_weights_fp32_batch(X)
        r = score(w32)
        if g + 1 == nan_at["when"]:
            r[3, 1, 0] = np.nan
        f = fitness_from_returns_native(r.reshape(-1), P, N, S).copy()
        py_hist.extend(zip(X / np.sqrt(scenarios.row_dots(X))[:, None], -f))
        b.prepare()                                    # (the native loop draws the next deviates while the GPU works)
        if np.isnan(f).any():
            rows = np.nonzero(np.isnan(f))[0]
            Xr = b.resample(rows)
            f[rows] = fitness_from_returns_native(score(planner_weights_fp32_batch(Xr)).reshape(-1), len(rows), N, S)
        b.tell(X, f)
        assert (b.stop(**overrides) != {}) == (g == 22)
    assert np.array_equal(a.mean, b.mean) and a.sigma == b.sigma and np.array_equal(a.C, b.C) and a.best_f == b.best_f
    assert len(native_hist) == len(py_hist) == 23 * P
    for (wa, ra), (wb, rb) in zip(native_hist, py_hist):
        assert np.array_equal(wa, wb) and (ra == rb or (np.isnan(ra) and np.isnan(rb)))


def test_lockstep_host_thread_default(monkeypatch):
    """How many host threads a lockstep call asks for (mpc_ord._lockstep_host_threads): at most four, one per four runs, never
    more than this process's share of the cores less one; OCD_CMA_THREADS overrides; nonsense in the environment is ignored."""
    import os
    from l4dc_mpc_ocd_amd.interact_drive.reward_design import mpc_ord
    monkeypatch.delenv("OCD_CMA_THREADS", raising=False)
    monkeypatch.delenv("LOCAL_WORLD_SIZE", raising=False)
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(16)), raising=False)
    assert mpc_ord._lockstep_host_threads(28) == 4 and mpc_ord._lockstep_host_threads(9) == 2 and mpc_ord._lockstep_host_threads(3) == 1
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")                    # eight ranks share the 16 cores: two each, one thread to spare
    assert mpc_ord._lockstep_host_threads(28) == 1
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "not a number")
    assert mpc_ord._lockstep_host_threads(28) == 4
    monkeypatch.setenv("OCD_CMA_THREADS", "6")
    assert mpc_ord._lockstep_host_threads(2) == 6
    monkeypatch.setenv("OCD_CMA_THREADS", "x")
    assert mpc_ord._lockstep_host_threads(28) == 4
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: {0}, raising=False)
    assert mpc_ord._lockstep_host_threads(28) == 1                 # one core: the calling thread only
