"""The boundary's threading conventions (INTEGRATION.md "Conventions of the boundary"): re-entrant across
handles, calls asynchronous on the caller's stream, options on the handle, the error string thread-local.  Four host
threads, each with its own scenario handle and its own HIP stream, launch episode batches at the same time (ctypes drops
the interpreter lock around every call); two more share ONE handle.  Every result is the CPU oracle's, bit for bit."""
import ctypes as C
import threading

import numpy as np
import pytest

from l4dc_mpc_ocd_amd import abi, scenarios

pytestmark = pytest.mark.gpu

CASES = [("finite_horizon", 5, 6, 4), ("local_opt", 10, 8, 8), ("replanning", 5, 4, 4), ("merging", 10, 4, 4)]


def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and bool(((a == b) | (np.isnan(a) & np.isnan(b))).all())


def workload(name, H, P, N, seed):
    scn = scenarios.SCENARIOS[name](horizon=H)
    inits = scn.init_dist.sample(N, seed=seed)
    w32 = np.stack([scenarios.planner_weights_fp32(c) for c in scn.candidate_weights(P, seed=seed + 1)])
    return scn, inits, w32


def test_four_threads_four_handles_four_streams(hip, oracle):
    import torch
    from l4dc_mpc_ocd_amd.engine import Engine
    work = [workload(n, H, P, N, 50 + i) for i, (n, H, P, N) in enumerate(CASES)]
    want = [oracle.rollout(scn.desc, inits, w32, want_traj=True) for scn, inits, w32 in work]
    got, errors = [None] * len(work), []
    start = threading.Barrier(len(work))

    def run(i):
        try:
            scn, inits, w32 = work[i]
            stream = torch.cuda.Stream(device="cuda:0")
            with torch.cuda.stream(stream):                     # torch's current stream is per thread
                eng = Engine(scn, "cuda:0")
                eng.set_option("scan_mode", i % 2 * 3)           # options live on the handle: 0, 3, 0, 3
                start.wait()
                outs = [eng.rollout(inits, w32, want_traj=True) for _ in range(3)]
            assert all(same(o["returns"], outs[0]["returns"]) for o in outs)
            got[i] = outs[-1]
        except BaseException as e:                              # noqa: BLE001 -- reported by the main thread
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=run, args=(i,)) for i in range(len(work))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    for i, (g, w) in enumerate(zip(got, want)):
        assert g is not None, f"thread {i} did not finish"
        assert same(g["returns"], w["returns"]) and same(g["traj"], w["traj"]) and same(g["ctrl"], w["ctrl"]), CASES[i]


def test_two_threads_share_one_handle(hip, oracle):
    """The handle's device-side state (per-device plan copy, compute-unit count, launch record) is created under its
    mutex: two threads that make the FIRST calls on a fresh handle at the same time both get the oracle's results."""
    import torch
    from l4dc_mpc_ocd_amd.engine import Engine
    scn, inits, w32 = workload("replanning", 5, 6, 4, 77)
    want = oracle.rollout(scn.desc, inits, w32)["returns"]
    E = want.size
    eng = Engine(scn, "cuda:0")
    halves, errors = [None, None], []
    start = threading.Barrier(2)

    def run(i):
        try:
            with torch.cuda.stream(torch.cuda.Stream(device="cuda:0")):
                start.wait()
                halves[i] = eng.rollout(inits, w32, ep_begin=i * E // 2, ep_end=(i + 1) * E // 2)["returns"]
        except BaseException as e:                              # noqa: BLE001
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=run, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    assert same(np.concatenate(halves), want)


def test_error_string_is_thread_local(hip):
    bad = scenarios.finite_horizon(horizon=abi.OCD_MAX_HORIZON + 1).desc
    seen = {}

    def fails():
        h = C.c_void_p()
        assert hip.ocd_scenario_create(C.byref(bad), C.byref(h)) == abi.OCD_ERR_INVALID_ARG
        seen["failing"] = hip.ocd_last_error()

    def succeeds():
        h = C.c_void_p()
        ok = scenarios.finite_horizon(horizon=5).desc
        assert hip.ocd_scenario_create(C.byref(ok), C.byref(h)) == abi.OCD_OK
        t = threading.Thread(target=fails)
        t.start()
        t.join()
        seen["clean"] = hip.ocd_last_error()                  # the other thread's failure is not visible here
        hip.ocd_scenario_destroy(h)

    t = threading.Thread(target=succeeds)
    t.start()
    t.join()
    assert b"horizon" in seen["failing"] and b"horizon" not in seen["clean"]
