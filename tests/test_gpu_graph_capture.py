"""ocd_rollout_episodes / ocd_plan_batch are plain kernel launches on the caller's stream once a handle is warm (no
allocation, no copy, no synchronisation inside): a caller may capture them into a HIP graph and replay it -- e.g. one
graph per CMA-ES generation shape, the candidate rows rewritten in place between replays.  Replays give the CPU oracle's
results, bit for bit."""
import numpy as np
import pytest

from l4dc_mpc_ocd_amd import scenarios

pytestmark = pytest.mark.gpu


def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and bool(((a == b) | (np.isnan(a) & np.isnan(b))).all())


@pytest.mark.parametrize("name,H,P,N", [("local_opt", 10, 8, 8), ("replanning", 5, 4, 4), ("merging", 25, 4, 4)])
def test_rollout_replays_from_a_hip_graph(hip, oracle, name, H, P, N):
    import torch
    from l4dc_mpc_ocd_amd.engine import Engine, _ptr
    scn = scenarios.SCENARIOS[name](horizon=H)
    inits = scn.init_dist.sample(N, seed=5)
    gens = [np.stack([scenarios.planner_weights_fp32(c) for c in scn.candidate_weights(P, seed=60 + g)]) for g in range(3)]
    E = P * N * scn.desc.n_samples
    eng = Engine(scn, "cuda:0")
    init_dev = torch.as_tensor(inits, dtype=torch.float32).cuda()
    w_dev = torch.as_tensor(gens[0]).cuda()
    ret = torch.zeros(E, dtype=torch.float32, device="cuda")

    def launch():
        eng._call(eng.lib.ocd_rollout_episodes, eng._h, _ptr(init_dev), _ptr(w_dev), P, N, 0, E, _ptr(ret), None, None,
                  eng._stream())

    launch()                                                    # warm: the handle's device-side state exists now
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=torch.cuda.Stream()):    # eng._stream() is the capturing stream in here
        launch()
    for g, w in enumerate(gens):
        w_dev.copy_(torch.as_tensor(w))                          # same pointer, new candidates
        ret.fill_(float("nan"))
        graph.replay()
        torch.cuda.synchronize()
        assert same(ret.cpu().numpy(), oracle.rollout(scn.desc, inits, w)["returns"]), (name, g)


def test_indexed_rollout_replays_from_a_hip_graph(hip, oracle):
    """ocd_rollout_indexed with a DEVICE-memory index (ABI 3: the kernel checks the rows and reports through the handle's
    pinned error word, which the first such call allocates -- so the handle is warmed before the capture): three populations
    of two runs as rows of one index, the candidate rows rewritten between replays; an out-of-range row written into the
    index between replays is caught by the replayed kernel (NaN return, ocd_scenario_index_error)."""
    import torch
    from l4dc_mpc_ocd_amd import abi
    from l4dc_mpc_ocd_amd.engine import Engine, _ptr
    scn = scenarios.finite_horizon(horizon=5)
    d = scn.desc
    pop, R = 4, 2
    runs = [np.asarray(scn.init_dist.sample(3, seed=40 + r), dtype=np.float32) for r in range(R)]
    inits = np.concatenate(runs)
    rows = [(r * pop + p, r * 3 + n, p * 3 + n) for r in range(R) for p in range(pop) for n in range(3)]
    idx = torch.as_tensor(np.asarray(rows, dtype=np.int32)).cuda()
    gens = [np.stack([scenarios.planner_weights_fp32(c) for c in scn.candidate_weights(R * pop, seed=70 + g)]) for g in range(3)]
    eng = Engine(scn, "cuda:0")
    init_dev, w_dev = torch.as_tensor(inits).cuda(), torch.as_tensor(gens[0]).cuda()
    E = len(rows)
    ret = torch.zeros(E, dtype=torch.float32, device="cuda")

    def launch():
        eng._call(eng.lib.ocd_rollout_indexed, eng._h, _ptr(init_dev), inits.shape[0], _ptr(w_dev), R * pop, _ptr(idx), E,
                  _ptr(ret), None, None, eng._stream())

    launch()                                                    # warm: device-side state and the index error word exist now
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=torch.cuda.Stream()):
        launch()
    for g, w in enumerate(gens):
        w_dev.copy_(torch.as_tensor(w))
        ret.fill_(float("nan"))
        graph.replay()
        torch.cuda.synchronize()
        want = np.concatenate([oracle.rollout(d, runs[r], w[r * pop:(r + 1) * pop])["returns"].reshape(-1) for r in range(R)])
        assert same(ret.cpu().numpy(), want), g
        assert hip.ocd_scenario_index_error(eng._h, None) == abi.OCD_OK
    idx[7, 1] = 99                                              # a caller bug between replays: the replayed kernel finds it
    graph.replay()
    torch.cuda.synchronize()
    got = ret.cpu().numpy()
    assert np.isnan(got[7]) and same(np.delete(got, 7), np.delete(want, 7))
    row = abi.C.c_int64(-1)
    assert hip.ocd_scenario_index_error(eng._h, abi.C.byref(row)) == abi.OCD_ERR_INVALID_ARG and row.value == 7


def test_plan_batch_replays_from_a_hip_graph(hip, oracle):
    import torch
    from l4dc_mpc_ocd_amd.engine import Engine
    scn = scenarios.local_opt(horizon=10)
    eng = Engine(scn, "cuda:0")
    rng = np.random.default_rng(3)
    B = 12
    ws = np.tile(np.array([[0.0, -0.9, 0.8, np.pi / 2], [0.0, -0.6, 1.0, np.pi / 2]], dtype=np.float32), (B, 1, 1))
    ws[:, 0, :3] += rng.normal(0, 0.02, (B, 3)).astype(np.float32)
    w = scenarios.planner_weights_fp32(scn.candidate_weights(1, seed=9)[0])
    first = eng.plan_batch(ws, w)                               # warm + the eager result
    ref = oracle.plan_batch(scn.desc, ws, w)
    assert same(first["plans"], ref["plans"])
    ws_dev = torch.as_tensor(ws).cuda()
    w_dev = torch.as_tensor(w).cuda()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=torch.cuda.Stream()):
        out = eng.plan_batch(ws_dev, w_dev, to_numpy=False)     # outputs allocated from the graph's pool
    graph.replay()
    torch.cuda.synchronize()
    assert same(out["plans"].cpu().numpy(), ref["plans"]) and same(out["best_init"].cpu().numpy(), ref["best_init"])
