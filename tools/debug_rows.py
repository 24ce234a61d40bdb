import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import oracle_lib
from l4dc_mpc_ocd_amd import scenarios
from l4dc_mpc_ocd_amd.engine import Engine
orc = oracle_lib.load()
for n_iter in (0, 1, 2, 5):
    scn = scenarios.finite_horizon(horizon=10, n_iter=n_iter)
    eng = Engine(scn, "cuda:0")
    ws = np.array([[[0.02, -0.9, 0.8, np.pi / 2], [0.0, -0.6, 0.5, np.pi / 2]]], dtype=np.float32)
    w = scenarios.planner_weights_fp32(scn.designer_weights)
    ref = orc.plan_batch(scn.desc, ws, w)
    for mode in (1, 2):
        eng.lib.ocd_set_option(b"scan_mode", mode)
        out = eng.plan_batch(ws, w, want_all=True)
        dl = out["all_losses"][0] - ref["all_losses"][0]
        dp = np.abs(out["all_plans"][0] - ref["all_plans"][0]).max(axis=(0, 2))
        print(f"n_iter={n_iter} mode={mode}: loss diff {dl}  plan max-abs diff per horizon lane {dp}")
eng.lib.ocd_set_option(b"scan_mode", 0)
