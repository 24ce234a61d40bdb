import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from l4dc_mpc_ocd_amd import scenarios
from l4dc_mpc_ocd_amd.engine import Engine
scn, inits, cands = scenarios.baseline_config(2)
w32 = np.stack([scenarios.planner_weights_fp32(c) for c in cands])
eng = Engine(scn, "cuda:0")
out = eng.rollout(inits, w32, want_traj=True)
dbg = out["ctrl"].reshape(-1)[: 8 * 3 * 8].reshape(8, 3, 8)
names = ["fwd scan1", "own step+sincos+scan2", "features fwd+bwd", "bwd scan1", "lane ops+bwd scan2", "update", "iters"]
for e in range(3):
    for k in range(3):
        a = dbg[e, k]
        it = a[6]
        print(f"episode {e} init {k}: iters {it:.0f} cycles/iter:", {n: round(float(a[i] / it), 1) for i, n in enumerate(names[:6])}, "sum", round(float(a[:6].sum() / it), 1))
