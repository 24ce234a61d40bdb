#!/usr/bin/env python3
"""How many (trajectory, initialisation, horizon step) states of a gradient pass have an ACTIVE fence / collision feature?
Measured on the CPU oracle (test infrastructure: this is a measurement tool, not the product): a sample of episodes of a
BASELINE config is rolled out, every planning problem's SGD is replayed pass by pass through ocd_oracle_mpc_reward, and the
conservative activity tests of the kernels (needs_fence / needs_collision1, csrc/ocd_device.h) are applied to every state.
The answer (about 30 % on configs 3 / 4 / 5) is why the shared-SIMD builds of the chunked kernel evaluate the active
features as work items.   usage: python tools/feature_density.py <config 3|4|5> <episodes>"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import oracle_lib
from l4dc_mpc_ocd_amd import scenarios
cfg = int(sys.argv[1]); n_ep = int(sys.argv[2])
orc = oracle_lib.load()
scn, inits, cands = scenarios.baseline_config(cfg)
d = scn.desc
w32 = np.stack([scenarios.planner_weights_fp32(c) for c in cands])
rng = np.random.default_rng(0)
P, N, S = w32.shape[0], inits.shape[0], d.n_samples
E = P * N * S
eps = rng.choice(E, n_ep, replace=False)
H, T, NO = d.horizon, d.episode_len, d.n_cars - 1
dt = np.float32(d.dt)
lo = np.float32(d.fence_lo); hx = np.float32(d.bump_half_x); hy = np.float32(d.bump_half_y)
tot = 0; act_f = 0; act_c = 0; act_any = 0; act_multi = 0
per_pass = []
t0 = time.time()
for e in eps:
    r = orc.rollout(d, inits, w32, ep_begin=int(e), ep_end=int(e) + 1, want_traj=True)
    p_ = int(e) // (N * S)
    w = w32[p_]
    traj = r['traj'][0]
    for t in range(T):
        ws = traj[t]
        # predicted others (constant velocity)
        oxy = np.zeros((H, NO, 2), np.float32)
        for j in range(NO):
            x, y, v, th = ws[j + 1]
            s, c = np.float32(orc.sinf(th)), np.float32(orc.cosf(th))
            for tt in range(H):
                x = np.float32(x + np.float32(np.float32(c * v) * dt)); y = np.float32(y + np.float32(np.float32(s * v) * dt))
                oxy[tt, j] = (x, y)
        for k in range(3):
            u = np.zeros((H, 2), np.float32); u[:, 1] = [0.0, -0.65, 0.65][k]
            for it in range(d.n_iter):
                R, g, tr = orc.mpc_reward(d, ws, w, u)
                x, y = tr[:, 0], tr[:, 1]
                nf = np.abs(x) > lo
                nc = np.zeros(H, int)
                for j in range(NO):
                    nc += ((np.abs(x - oxy[:, j, 0]) < hx * np.float32(1.001)) & (np.abs(y - oxy[:, j, 1]) < hy * np.float32(1.001))).astype(int)
                a = nf | (nc > 0)
                tot += H; act_f += nf.sum(); act_c += (nc > 0).sum(); act_any += a.sum(); act_multi += ((nf.astype(int) + nc) > 1).sum()
                per_pass.append(a.sum())
                u = (u + np.float32(d.learning_rate) * g).astype(np.float32)
    print(f"ep {e}: {time.time()-t0:.0f}s  density any {act_any/tot:.3f} fence {act_f/tot:.3f} col {act_c/tot:.3f} multi {act_multi/tot:.4f}", flush=True)
pp = np.array(per_pass)
print("cfg", cfg, "H", H, "lane-steps", tot, "density any", act_any / tot, "fence", act_f / tot, "col", act_c / tot, "multi", act_multi / tot)
print("per-segment-pass active count histogram:", np.bincount(pp, minlength=H + 1) / len(pp))
# wave-level: cfg-specific segments per wavefront
for segs in (12, 6, 4):
    m = len(pp) // segs * segs
    wsum = rng.permutation(pp)[:m].reshape(-1, segs).sum(1)
    print(f" random {segs} segments per wavefront: mean active {wsum.mean():.1f} of {segs*H}, p50 {np.percentile(wsum,50)}, p90 {np.percentile(wsum,90)}, p99 {np.percentile(wsum,99)}, max {wsum.max()}")
