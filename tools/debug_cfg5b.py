import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import oracle_lib
from l4dc_mpc_ocd_amd import scenarios
from l4dc_mpc_ocd_amd.engine import Engine
scn, inits, cands = scenarios.baseline_config(5)
w32 = np.stack([scenarios.planner_weights_fp32(c) for c in cands])
eng = Engine(scn, "cuda:0")
orc = oracle_lib.load()
e = 12711
ref = orc.rollout(scn.desc, inits, w32, ep_begin=e, ep_end=e + 1, want_traj=True)
for segs, noskip in [(1, 0), (1, 1), (2, 0)]:
    eng.lib.ocd_set_option(b"segs_per_wave", segs); eng.lib.ocd_set_option(b"no_feature_skips", noskip)
    got = eng.rollout(inits, w32, ep_begin=e, ep_end=e + 1, want_traj=True)
    bad = np.nonzero(np.any(got["ctrl"][0] != ref["ctrl"][0], axis=1))[0]
    print(f"segs={segs} noskip={noskip}: return {got['returns'][0]!r} (oracle {ref['returns'][0]!r}); first differing control step: {bad[:1]}")
    if len(bad):
        st = int(bad[0])
        ws = ref["traj"][0, st]
        p, n = e // inits.shape[0], e % inits.shape[0]
        pl = eng.plan_batch(ws[None], w32[p], want_all=True)
        po = orc.plan_batch(scn.desc, ws, w32[p])
        print("  step", st, "world state", ws.tolist())
        print("  gpu losses", pl["all_losses"][0], "oracle", po["all_losses"][0], "best", pl["best_init"], po["best_init"])
        for k in range(3):
            d = np.nonzero(np.any(pl["all_plans"][0, k] != po["all_plans"][0, k], axis=1))[0]
            print("   init", k, "differing horizon lanes", d[:8], "nan?", np.isnan(pl["all_plans"][0,k]).any())
eng.lib.ocd_set_option(b"segs_per_wave", 0); eng.lib.ocd_set_option(b"no_feature_skips", 0)
