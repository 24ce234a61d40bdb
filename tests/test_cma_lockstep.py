"""ocd_cma_run_many (include/ocd_cma.h): R CMA-ES runs in lockstep around ONE indexed episode launch per generation must
give every run exactly the populations, costs and stop generation that ocd_cma_run gives it alone.  No GPU: the episode
launch is a function pointer, here a Python callback that scores (weights, init) pairs on the host -- the native loop
cannot tell the difference.  (The reference's counterpart is a multiprocessing.Pool over init groups,
experiments/run_mpc_ord.py:83-90, around pycma's loop, mpc_ord.py:33-45.)"""
import ctypes as C

import numpy as np
import pytest

from l4dc_mpc_ocd_amd.interact_drive.reward_design.cmaes import (NativeCMAES, RunArgs, RunManyArgs, STOP_NAMES, N_STOP,
                                                                load_cma_library)

D = 7
ROLLOUT = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p,
                      C.c_void_p, C.c_void_p, C.c_void_p)
ROLLOUT_IDX = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
                          C.c_void_p, C.c_void_p, C.c_void_p)
SYNC = C.CFUNCTYPE(C.c_int32, C.c_void_p)


def episode(w, init, reset, nan_below=None):
    """A stand-in for an episode return: fp32 function of the fp32 weights, the init state and the reset number."""
    r = np.float32(-np.sum((w - np.float32(0.2)) ** 2, dtype=np.float32) * (np.float32(1.0) + init[0])
                   + np.float32(0.01) * np.float32(reset % 2) - init[2] * w[0])
    if nan_below is not None and w[1] < nan_below:
        return np.float32(np.nan)
    return r


def as_array(ptr, shape, dtype):
    n = int(np.prod(shape))
    ctype = {np.float32: C.c_float, np.int32: C.c_int32}[dtype]
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(n,)).reshape(shape)


def make_callbacks(S, nan_below=None, log=None, stream_log=None):
    def rollout(scn, init_p, w_p, P, N, e0, e1, ret_p, traj, ctrl, stream):
        init, w, ret = as_array(init_p, (N, 4), np.float32), as_array(w_p, (P, D), np.float32), as_array(ret_p, (e1 - e0,), np.float32)
        for e in range(e0, e1):
            p, n = e // (N * S), (e // S) % N
            ret[e - e0] = episode(w[p], init[n], e, nan_below)
        return 0

    def rollout_idx(scn, init_p, n_rows, w_p, p_rows, idx_p, E, ret_p, traj, ctrl, stream):
        init, w = as_array(init_p, (n_rows, 4), np.float32), as_array(w_p, (p_rows, D), np.float32)
        idx, ret = as_array(idx_p, (E, 3), np.int32), as_array(ret_p, (E,), np.float32)
        if log is not None:
            log.append(int(E))
        if stream_log is not None:
            stream_log.append((stream or 0, int(E), int(idx[0, 0])))
        for e in range(E):
            ret[e] = episode(w[idx[e, 0]], init[idx[e, 1]], int(idx[e, 2]), nan_below)
        return 0

    return ROLLOUT(rollout), ROLLOUT_IDX(rollout_idx), SYNC(lambda s: 0)


def run_alone(lib, x0, sigma0, seed, popsize, inits, S, opts, gens, cbs):
    es = NativeCMAES(x0, sigma0, popsize=popsize, seed=seed)
    lam, N = es.lam, inits.shape[0]
    w = np.zeros((lam, D), dtype=np.float32)
    ret = np.zeros(lam * N * S, dtype=np.float32)
    hist_w, hist_c = np.zeros((gens, lam, D)), np.zeros((gens, lam))
    a = RunArgs()
    a.scn, a.init_dev, a.N, a.S = None, inits.ctypes.data, N, S
    a.w_pinned, a.ret_pinned, a.stream = w.ctypes.data, ret.ctypes.data, None
    a.rollout, a.sync = C.cast(cbs[0], C.c_void_p).value, C.cast(cbs[2], C.c_void_p).value
    a.normalise_variant, a.max_generations = 0, gens
    a.hist_w, a.hist_cost = hist_w.ctypes.data, hist_c.ctypes.data
    done, why, pending = es.run(a, opts)
    return es, done, why, pending, hist_w[:done + int(pending)], hist_c[:done + int(pending)]


def run_many(lib, ess, inits_per_run, S, opts, gens, cbs, active=None, groups=0, threads=0):
    R = len(ess)
    lams = np.array([es.lam for es in ess], dtype=np.int64)
    run_N = np.array([i.shape[0] for i in inits_per_run], dtype=np.int64)
    run_n0 = np.concatenate([[0], np.cumsum(run_N)[:-1]]).astype(np.int64)
    run_p0 = np.concatenate([[0], np.cumsum(lams)[:-1]]).astype(np.int64)
    P_rows, N_rows, E_max = int(lams.sum()), int(run_N.sum()), int((lams * run_N).sum() * S)
    st = dict(inits=np.ascontiguousarray(np.concatenate(inits_per_run), dtype=np.float32), w=np.zeros((P_rows, D), np.float32),
              idx=np.zeros((E_max, 3), np.int32), ret=np.zeros(E_max, np.float32), hist_w=np.zeros((gens, P_rows, D)),
              hist_c=np.zeros((gens, P_rows)), evaluated=np.zeros((gens, R), np.uint8), nonf=np.zeros((gens, R), np.int32),
              launched=np.zeros(gens, np.int64), active=np.ones(R, np.uint8) if active is None else active,
              pending=np.zeros(R, np.uint8), flags=np.zeros((R, N_STOP), np.int32), run_N=run_N, run_n0=run_n0, run_p0=run_p0,
              lams=lams)
    o = ess[0]._stop_opts(opts)
    st["opts"] = np.array([o.get(k, 0.0) for k in STOP_NAMES], dtype=np.float64)
    st["X"] = (C.c_void_p * R)(*[es._X_ptr for es in ess])
    st["f"] = (C.c_void_p * R)(*[es._f_ptr for es in ess])
    st["es"] = (C.c_void_p * R)(*[es._h.value for es in ess])
    a = RunManyArgs()
    a.scn, a.init_dev, a.N_rows, a.P_rows, a.S, a.R, a.normalise_variant = None, st["inits"].ctypes.data, N_rows, P_rows, S, R, 0
    a.run_n0, a.run_N, a.run_p0, a.run_reset_phase = run_n0.ctypes.data, run_N.ctypes.data, run_p0.ctypes.data, None
    a.w_pinned, a.index_pinned, a.ret_pinned, a.stream = st["w"].ctypes.data, st["idx"].ctypes.data, st["ret"].ctypes.data, None
    a.rollout, a.sync = C.cast(cbs[1], C.c_void_p).value, C.cast(cbs[2], C.c_void_p).value
    a.max_generations, a.stop_opts, a.active = gens, st["opts"].ctypes.data, st["active"].ctypes.data
    a.X, a.cost = C.cast(st["X"], C.c_void_p).value, C.cast(st["f"], C.c_void_p).value
    a.hist_w, a.hist_cost, a.evaluated = st["hist_w"].ctypes.data, st["hist_c"].ctypes.data, st["evaluated"].ctypes.data
    a.seconds, a.nonfinite, a.episodes_launched = None, st["nonf"].ctypes.data, st["launched"].ctypes.data
    a.stop_flags, a.pending_nan = st["flags"].ctypes.data, st["pending"].ctypes.data
    if groups:                                                    # (stand-in stream handles: the callbacks only log them)
        st["streams"] = (C.c_void_p * groups)(*[0x1000 + 16 * k for k in range(groups)])
        a.n_groups, a.streams = groups, C.cast(st["streams"], C.c_void_p).value
    a.host_threads = threads
    done = C.c_int64(0)
    assert lib.ocd_cma_run_many(st["es"], C.byref(a), C.byref(done)) == 0
    st["done"] = int(done.value)
    return st


def test_lockstep_runs_equal_the_runs_alone():
    lib = load_cma_library()
    rng = np.random.default_rng(4)
    S, gens = 2, 30
    runs = []
    for r in range(5):
        N = 1 + r % 3
        inits = np.ascontiguousarray(rng.uniform(-0.2, 0.2, (N, 4)), dtype=np.float32)
        runs.append(dict(x0=list(rng.uniform(-1, 1, D)), sigma0=[0.3, 0.05, 1e-13, 0.2, 0.1][r], seed=11 + r, inits=inits))
    opts = dict(maxiter=17)
    log = []
    cbs = make_callbacks(S, log=log)
    alone = [run_alone(lib, q["x0"], q["sigma0"], q["seed"], None, q["inits"], S, opts, gens, cbs) for q in runs]
    stops = [a[1] for a in alone]
    assert stops[2] == 1 and "tolx" in alone[2][2]                # the run with a vanishing step size stops at once ...
    assert max(stops) == 17 and all("maxiter" in a[2] for i, a in enumerate(alone) if i != 2)   # ... the others on the cap
    ess = [NativeCMAES(q["x0"], q["sigma0"], seed=q["seed"]) for q in runs]
    st = run_many(lib, ess, [q["inits"] for q in runs], S, opts, gens, cbs)
    assert st["done"] == 17 and not st["active"].any() and not st["pending"].any()
    for r, (es_a, done_a, why_a, _, hw, hc) in enumerate(alone):
        p0, lam = int(st["run_p0"][r]), int(st["lams"][r])
        took_part = st["evaluated"][:, r].astype(bool)
        assert took_part.sum() == done_a and took_part[:done_a].all()           # dropped out right after it stopped
        assert np.array_equal(st["hist_w"][:done_a, p0:p0 + lam], hw) and np.array_equal(st["hist_c"][:done_a, p0:p0 + lam], hc)
        assert {k for i, k in enumerate(STOP_NAMES) if st["flags"][r, i]} == set(why_a)
        assert np.array_equal(ess[r].mean, es_a.mean) and ess[r].sigma == es_a.sigma and ess[r].best_f == es_a.best_f
        assert ess[r].gen == done_a and ess[r].counteval == es_a.counteval
    # the launch shrinks when a run drops out: all five runs in generation 0, four from generation 1 on
    E_all = int((st["lams"] * st["run_N"]).sum() * S)
    E_wo2 = E_all - int(st["lams"][2] * st["run_N"][2] * S)
    assert log[-17:] == [E_all] + [E_wo2] * 16 and list(st["launched"][:17]) == log[-17:]


def test_a_nan_cost_hands_only_that_run_back():
    """A run whose generation holds a NaN cost is left evaluated-but-untold (pending_nan) for the caller's rejection
    sampling; the other runs of the same generation are told, and everything equals the runs alone."""
    lib = load_cma_library()
    rng = np.random.default_rng(9)
    S, gens = 1, 6
    runs = [dict(x0=[0.5] * D, sigma0=0.3, seed=3 + r, inits=np.ascontiguousarray(rng.uniform(-0.2, 0.2, (2, 4)), dtype=np.float32))
            for r in range(3)]
    cbs = make_callbacks(S, nan_below=0.05)                       # a candidate with w[1] < 0.05 scores NaN (run 0 at once, run 2 later)
    opts = dict(maxiter=gens)
    alone = [run_alone(lib, q["x0"], q["sigma0"], q["seed"], None, q["inits"], S, opts, gens, cbs) for q in runs]
    first_nan = [a[1] if a[3] else None for a in alone]
    assert any(g is not None for g in first_nan) and any(g is None or g > min(x for x in first_nan if x is not None) for g in first_nan)
    g0 = min(g for g in first_nan if g is not None)
    ess = [NativeCMAES(q["x0"], q["sigma0"], seed=q["seed"]) for q in runs]
    st = run_many(lib, ess, [q["inits"] for q in runs], S, opts, gens, cbs)
    assert st["done"] == g0 + 1                                   # returned right after the first generation with a NaN
    for r, a in enumerate(alone):
        p0, lam = int(st["run_p0"][r]), int(st["lams"][r])
        if first_nan[r] == g0:
            assert st["pending"][r] == 1 and ess[r].gen == g0     # evaluated (rows valid), not told
            assert np.isnan(st["hist_c"][g0, p0:p0 + lam]).any()
        else:
            assert st["pending"][r] == 0 and ess[r].gen == g0 + 1
        n = min(g0 + 1, a[4].shape[0])
        assert np.array_equal(st["hist_w"][:n, p0:p0 + lam], a[4][:n])
        assert np.array_equal(st["hist_c"][:n, p0:p0 + lam], a[5][:n], equal_nan=True)


@pytest.mark.parametrize("groups", [2, 3, 8])
def test_groups_on_their_own_streams_equal_the_runs_alone(groups):
    """ABI 7: the runs dealt to G groups, each launching on its own stream and cycling wait -> tell -> ask -> launch by itself
    (one group's host work under the others' kernels).  Every run still makes exactly the calls it makes alone."""
    lib = load_cma_library()
    rng = np.random.default_rng(14)
    S, gens = 1, 25
    runs = []
    for r in range(7):
        inits = np.ascontiguousarray(rng.uniform(-0.2, 0.2, (1 + r % 2, 4)), dtype=np.float32)
        runs.append(dict(x0=list(rng.uniform(-1, 1, D)), sigma0=[0.3, 0.05, 0.2, 1e-13, 0.1, 0.4, 0.25][r], seed=21 + r, inits=inits))
    opts = dict(maxiter=12)
    slog = []
    cbs = make_callbacks(S, stream_log=slog)
    alone = [run_alone(lib, q["x0"], q["sigma0"], q["seed"], None, q["inits"], S, opts, gens, cbs) for q in runs]
    del slog[:]
    ess = [NativeCMAES(q["x0"], q["sigma0"], seed=q["seed"]) for q in runs]
    st = run_many(lib, ess, [q["inits"] for q in runs], S, opts, gens, cbs, groups=groups)
    assert st["done"] == 12 and not st["active"].any() and not st["pending"].any()
    for r, (es_a, done_a, why_a, _, hw, hc) in enumerate(alone):
        p0, lam = int(st["run_p0"][r]), int(st["lams"][r])
        took_part = st["evaluated"][:, r].astype(bool)
        assert took_part.sum() == done_a and took_part[:done_a].all()
        assert np.array_equal(st["hist_w"][:done_a, p0:p0 + lam], hw) and np.array_equal(st["hist_c"][:done_a, p0:p0 + lam], hc)
        assert {k for i, k in enumerate(STOP_NAMES) if st["flags"][r, i]} == set(why_a)
        assert np.array_equal(ess[r].mean, es_a.mean) and ess[r].sigma == es_a.sigma and ess[r].gen == done_a
    # every group launched on its own stream in every generation it had an active run, the groups taking turns
    G = min(groups, len(runs))
    used = sorted({s_ for s_, _, _ in slog})
    assert used == [0x1000 + 16 * k for k in range(G)]
    first_rows = [int(st["run_p0"][len(runs) * k // G]) for k in range(G)]
    assert [row for _, _, row in slog[:G]] == first_rows           # generation 0: one launch per group, in group order
    per_gen = np.array(st["launched"][:12])
    assert per_gen[0] == int((st["lams"] * st["run_N"]).sum() * S) and sum(e for _, e, _ in slog) == per_gen.sum()


def test_a_nan_cost_with_two_groups_hands_that_run_back_and_finishes_what_was_launched():
    lib = load_cma_library()
    rng = np.random.default_rng(9)
    S, gens = 1, 6
    runs = [dict(x0=[0.5] * D, sigma0=0.3, seed=3 + r, inits=np.ascontiguousarray(rng.uniform(-0.2, 0.2, (2, 4)), dtype=np.float32))
            for r in range(4)]
    cbs = make_callbacks(S, nan_below=0.05)
    opts = dict(maxiter=gens)
    alone = [run_alone(lib, q["x0"], q["sigma0"], q["seed"], None, q["inits"], S, opts, gens, cbs) for q in runs]
    ess = [NativeCMAES(q["x0"], q["sigma0"], seed=q["seed"]) for q in runs]
    st = run_many(lib, ess, [q["inits"] for q in runs], S, opts, gens, cbs, groups=2)
    assert st["pending"].any()
    for r, a in enumerate(alone):
        p0, lam = int(st["run_p0"][r]), int(st["lams"][r])
        n_eval = int(st["evaluated"][:st["done"], r].sum())         # generations this run was launched in
        assert st["evaluated"][:n_eval, r].all()
        if st["pending"][r]:
            assert ess[r].gen == n_eval - 1 and np.isnan(st["hist_c"][n_eval - 1, p0:p0 + lam]).any()
            assert a[3] and a[1] == n_eval - 1                      # the run alone met its NaN in the same generation
        else:
            assert ess[r].gen == n_eval
        assert np.array_equal(st["hist_w"][:n_eval, p0:p0 + lam], a[4][:n_eval])
        assert np.array_equal(st["hist_c"][:n_eval, p0:p0 + lam], a[5][:n_eval], equal_nan=True)
    # a run of the OTHER group than the first pending one may be one generation ahead, never more
    gens_done = [int(st["evaluated"][:st["done"], r].sum()) for r in range(4)]
    assert max(gens_done) - min(gens_done) <= 1 and st["done"] == max(gens_done)


@pytest.mark.parametrize("groups,threads", [(0, 2), (2, 4), (3, 16), (0, 64)])
def test_host_threads_change_nothing(groups, threads):
    """ABI 8: the runs' tells (and the work that overlaps the kernel) shared out to host threads.  Each run is told by one
    thread on its own strategy state, so populations, costs, stop generations, NaN hand-backs and final states are bit for
    bit those of the call with one thread -- for every number of threads, with and without launch groups (more threads than
    the library's cap of 16 are clamped, more than runs idle along)."""
    lib = load_cma_library()
    rng = np.random.default_rng(33)
    S, gens = 1, 60
    runs = []
    for r in range(28):                                           # the reference's 28 runs (generalization_data.py:78-84)
        inits = np.ascontiguousarray(rng.uniform(-0.2, 0.2, (1 + r % 3, 4)), dtype=np.float32)
        runs.append(dict(x0=list(rng.uniform(-1, 1, D)), sigma0=[0.3, 0.05, 0.2, 1e-13, 0.1, 0.4, 0.25][r % 7], seed=51 + r, inits=inits))
    opts = dict(maxiter=40)
    cbs = make_callbacks(S, nan_below=-1.2)                       # (a late NaN in some run: the hand-back path runs too)
    out = []
    for t in (1, threads):
        ess = [NativeCMAES(q["x0"], q["sigma0"], seed=q["seed"]) for q in runs]
        st = run_many(lib, ess, [q["inits"] for q in runs], S, opts, gens, cbs, groups=groups, threads=t)
        out.append((st, ess))
    (s1, e1), (s2, e2) = out
    assert s1["done"] == s2["done"] and s1["done"] > 0
    for k in ("hist_w", "hist_c", "evaluated", "nonf", "launched", "active", "pending", "flags"):
        assert np.array_equal(s1[k], s2[k], equal_nan=True), k
    for a_, b_ in zip(e1, e2):
        assert np.array_equal(a_.mean, b_.mean) and a_.sigma == b_.sigma and a_.gen == b_.gen and a_.counteval == b_.counteval
        assert np.array_equal(a_.C, b_.C) and (a_.best_f == b_.best_f or (np.isnan(a_.best_f) and np.isnan(b_.best_f)))


def test_host_threads_many_short_generations():
    """The pool's hand-over (publish, claim, join) a few thousand times in a row with nothing else between: no lost or doubled run."""
    lib = load_cma_library()
    rng = np.random.default_rng(7)
    runs = [dict(x0=list(rng.uniform(-1, 1, D)), sigma0=0.3, seed=5 + r, inits=np.zeros((1, 4), dtype=np.float32)) for r in range(9)]
    opts = dict(maxiter=400, tolfun=0.0, tolfunhist=0.0, tolx=0.0, tolstagnation=10 ** 9)
    cbs = make_callbacks(1)
    res = []
    for t in (1, 8):
        ess = [NativeCMAES(q["x0"], q["sigma0"], popsize=4, seed=q["seed"]) for q in runs]
        st = run_many(lib, ess, [q["inits"] for q in runs], 1, opts, 400, cbs, threads=t)
        res.append((st["done"], st["hist_c"].copy(), [es.gen for es in ess], [es.sigma for es in ess]))
    assert res[0][0] == res[1][0] and res[0][2] == res[1][2] and res[0][3] == res[1][3]
    assert np.array_equal(res[0][1], res[1][1], equal_nan=True)


def test_bad_arguments_are_refused():
    lib = load_cma_library()
    es = NativeCMAES([0.0] * D, 0.1, seed=1)
    cbs = make_callbacks(1)
    inits = np.zeros((2, 4), dtype=np.float32)
    st = run_many(lib, [es], [inits], 1, {}, 0, cbs)              # zero generations: fine, nothing done
    assert st["done"] == 0
    a = RunManyArgs()
    done = C.c_int64(0)
    assert lib.ocd_cma_run_many((C.c_void_p * 1)(es._h.value), C.byref(a), C.byref(done)) == -1
