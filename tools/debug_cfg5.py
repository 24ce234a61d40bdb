import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import oracle_lib
from l4dc_mpc_ocd_amd import scenarios
from l4dc_mpc_ocd_amd.engine import Engine
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 5
scn, inits, cands = scenarios.baseline_config(cfg)
w32 = np.stack([scenarios.planner_weights_fp32(c) for c in cands])
eng = Engine(scn, "cuda:0")
res = {}
for segs in (1, 1, 2, 2):
    eng.lib.ocd_set_option(b"segs_per_wave", segs)
    r = eng.rollout(inits, w32)["returns"]
    res.setdefault(segs, []).append(r)
eng.lib.ocd_set_option(b"segs_per_wave", 0)
for segs in (1, 2):
    a, b = res[segs]
    print(f"segs={segs}: run-to-run differing episodes: {(a != b).sum()}")
a, b = res[1][0], res[2][0]
diff = np.nonzero(a != b)[0]
print("segs 1 vs 2 differing episodes:", len(diff), diff[:20])
orc = oracle_lib.load()
for e in diff[:6]:
    ref = orc.rollout(scn.desc, inits, w32, ep_begin=int(e), ep_end=int(e) + 1)["returns"][0]
    print(f"episode {e}: segs1 {a[e]!r} segs2 {b[e]!r} oracle {ref!r}")
