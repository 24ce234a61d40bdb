"""Committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py).

CPU: the oracle must reproduce them bit for bit (freezes the arithmetic contract).
GPU: the HIP path, through the C ABI, must reproduce them bit for bit.
"""
import glob
import os

import numpy as np
import pytest

from l4dc_mpc_ocd_amd import scenarios

HERE = os.path.dirname(os.path.abspath(__file__))
sys_path_golden = os.path.join(HERE, "golden")
import sys
sys.path.insert(0, sys_path_golden)
from make_golden import CASES  # noqa: E402


def _load(name):
    return np.load(os.path.join(HERE, "golden", name + ".npz"))


def _scn(kw):
    kw = dict(kw)
    kw.pop("leaf", None)
    return scenarios.SCENARIOS[kw.pop("scenario")](**kw)


def _leaf(g):
    if "leaf_values" not in g.files:
        return None
    return [g["leaf_grid0"], g["leaf_grid1"], g["leaf_grid2"]], g["leaf_values"], int(g["leaf_proj_kind"])


def _same(a, b):
    a = np.asarray(a); b = np.asarray(b)
    return a.shape == b.shape and bool(np.all((a == b) | (np.isnan(a) & np.isnan(b))))


def test_every_fixture_has_a_case():
    files = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(HERE, "golden", "*.npz")))
    files = [f for f in files if not f.startswith("torch_")]        # torch_*: tests/test_torch_fixtures.py
    assert files == sorted(c[0] for c in CASES)


@pytest.mark.parametrize("name,kw,P,N", CASES, ids=[c[0] for c in CASES])
def test_oracle_reproduces_golden(oracle, name, kw, P, N):
    g = _load(name)
    scn = _scn(kw)
    w32 = np.stack([scenarios.planner_weights_fp32(c) for c in g["cand_weights_raw"]])
    assert _same(w32, g["cand_weights_fp32"])           # host-side normalisation chain is frozen too
    assert _same(scn.designer_weights, g["designer_weights"])
    leaf = _leaf(g)
    if leaf:
        oracle.set_leaf_value(*leaf)
    try:
        _check_oracle(oracle, scn, g, w32)
    finally:
        oracle.set_leaf_value(None, None)


def _check_oracle(oracle, scn, g, w32):
    ro = oracle.rollout(scn.desc, g["init_states"], w32, want_traj=True)
    assert _same(ro["returns"], g["returns"]) and _same(ro["traj"], g["traj"]) and _same(ro["ctrl"], g["ctrl"])
    pl = oracle.plan_batch(scn.desc, g["plan_world_states"], w32[0], other_plans=scn.other_plans())
    assert _same(pl["all_plans"], g["plan_all_plans"]) and _same(pl["all_losses"], g["plan_all_losses"])
    assert np.array_equal(pl["best_init"], g["plan_best_init"])
    feats, rew = oracle.reward_batch(scn.desc, g["feat_world_states"], scn.designer_weights)
    assert _same(feats, g["feats"]) and _same(rew, g["rewards"])
    for b in range(len(g["plan_world_states"])):
        r, gr, tr = oracle.mpc_reward(scn.desc, g["plan_world_states"][b], w32[0], g["plan_plans"][b], other_plans=scn.other_plans())
        assert _same(r, g["obj_reward"][b]) and _same(gr, g["obj_grad"][b]) and _same(tr, g["obj_traj"][b])


@pytest.mark.gpu
@pytest.mark.parametrize("name,kw,P,N", CASES, ids=[c[0] for c in CASES])
def test_hip_reproduces_golden(hip, name, kw, P, N):
    from l4dc_mpc_ocd_amd.engine import Engine
    g = _load(name)
    scn = _scn(kw)
    eng = Engine(scn, "cuda:0")
    leaf = _leaf(g)
    if leaf:
        eng.set_leaf_value(*leaf)
    ro = eng.rollout(g["init_states"], g["cand_weights_fp32"], want_traj=True)
    assert _same(ro["ctrl"], g["ctrl"]), "applied controls"
    assert _same(ro["traj"], g["traj"]), "trajectories"
    assert _same(ro["returns"], g["returns"]), "returns"
    pl = eng.plan_batch(g["plan_world_states"], g["cand_weights_fp32"][0], want_all=True)
    assert _same(pl["all_plans"], g["plan_all_plans"]) and _same(pl["all_losses"], g["plan_all_losses"])
    assert np.array_equal(pl["best_init"], g["plan_best_init"]) and _same(pl["plans"], g["plan_plans"])
    feats, rew = eng.reward_batch(g["feat_world_states"], scn.designer_weights)
    assert _same(feats, g["feats"]) and _same(rew, g["rewards"])
    ob = eng.mpc_reward_batch(g["plan_world_states"], g["cand_weights_fp32"][0], g["plan_plans"], want_traj=True)
    assert _same(ob["reward"], g["obj_reward"]) and _same(ob["grad"], g["obj_grad"]) and _same(ob["traj"], g["obj_traj"])
