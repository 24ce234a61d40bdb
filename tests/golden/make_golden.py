#!/usr/bin/env python3
"""Generate the committed golden vectors under tests/golden/ (inputs + expected outputs).

Provenance: the reference cannot run here or on the GPU box (TensorFlow 2.1 and pycma are not
installed and there is no network), so these vectors come from the CPU oracle
(oracle/ocd_oracle.c), which is itself pinned by the reference's own known-answer tests
(tests/test_oracle_kat.py) and cross-checked by the torch restatement (tests/test_oracle_vs_torch.py).
They freeze the arithmetic contract: any later change to oracle or kernels that moves a single bit
of a plan, trajectory or return fails tests/test_golden.py.

usage: python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib  # noqa: E402
from l4dc_mpc_ocd_amd import scenarios  # noqa: E402

CASES = [
    # name, factory kwargs, P, N
    ("finite_horizon_h5", dict(scenario="finite_horizon", horizon=5), 2, 3),
    ("finite_horizon_h10", dict(scenario="finite_horizon", horizon=10), 2, 2),
    ("finite_horizon_h6_extra", dict(scenario="finite_horizon", horizon=6, extra_inits=True), 1, 2),
    ("local_opt_h5_extra", dict(scenario="local_opt", horizon=5, extra_inits=True), 2, 2),
    ("replanning_h5", dict(scenario="replanning", horizon=5), 2, 2),
    ("replanning_h15", dict(scenario="replanning", horizon=15), 1, 1),
    ("merging_h5", dict(scenario="merging", horizon=5), 1, 2),
    ("merging_h25", dict(scenario="merging", horizon=25, n_iter=40), 1, 1),
    # terminal value (leaf_evaluation): the table is part of the fixture
    ("finite_horizon_h5_leaf", dict(scenario="finite_horizon", horizon=5, leaf=True), 2, 2),
]


def leaf_table(seed):
    """A synthetic ValueFeature table: (disc_grid, v_grid[t], proj_kind)."""
    rng = np.random.default_rng(seed)
    grid = [np.linspace(-0.3, 0.3, 7).astype(np.float32), np.linspace(-2.0, 2.5, 12).astype(np.float32),
            np.linspace(0.0, 2.5, 6).astype(np.float32)]
    vals = (rng.standard_normal((7, 12, 6)) * 2 - 1).astype(np.float32)
    return grid, vals, 1


def make_case(name, kw, P, N, orc):
    kw = dict(kw)
    leaf = kw.pop("leaf", False)
    scn = scenarios.SCENARIOS[kw.pop("scenario")](**kw)
    seed = sum(map(ord, name))
    extra = {}
    if leaf:
        grid, vals, proj = leaf_table(seed)
        orc.set_leaf_value(grid, vals, proj)
        extra = dict(leaf_grid0=grid[0], leaf_grid1=grid[1], leaf_grid2=grid[2], leaf_values=vals,
                     leaf_proj_kind=np.int32(proj))
    inits = scn.init_dist.sample(N, seed=seed).astype(np.float32)
    cands = scn.candidate_weights(P, seed=seed + 1)
    w32 = np.stack([scenarios.planner_weights_fp32(c) for c in cands])
    ro = orc.rollout(scn.desc, inits, w32, want_traj=True)
    # plan-level vectors: the world states visited by the first episode, planned with candidate 0
    ws = ro["traj"][0, :4]
    pl = orc.plan_batch(scn.desc, ws, w32[0], other_plans=scn.other_plans())
    feats, rew = orc.reward_batch(scn.desc, ro["traj"][0], scn.designer_weights)
    # the planner's objective and gradient at the selected plans (NaivePlanner.reward_func, naive_planner.py:33-77)
    obj = [orc.mpc_reward(scn.desc, ws[b], w32[0], pl["plans"][b], other_plans=scn.other_plans()) for b in range(len(ws))]
    extra.update(obj_reward=np.array([o[0] for o in obj], dtype=np.float32), obj_grad=np.stack([o[1] for o in obj]),
                 obj_traj=np.stack([o[2] for o in obj]))
    if leaf:
        orc.set_leaf_value(None, None)
    return dict(**extra, init_states=inits, cand_weights_raw=cands, cand_weights_fp32=w32,
                returns=ro["returns"], traj=ro["traj"], ctrl=ro["ctrl"],
                plan_world_states=ws, plan_all_plans=pl["all_plans"], plan_all_losses=pl["all_losses"],
                plan_best_init=pl["best_init"], plan_plans=pl["plans"],
                feat_world_states=ro["traj"][0], feats=feats, rewards=rew,
                designer_weights=scn.designer_weights)


def main():
    orc = oracle_lib.load()
    assert not orc.lib.ocd_oracle_uses_libm()
    for name, kw, P, N in CASES:
        out = make_case(name, kw, P, N, orc)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, {k: v.shape for k, v in out.items() if k in ("returns", "traj", "plan_all_plans")},
              "returns", out["returns"])


if __name__ == "__main__":
    main()
