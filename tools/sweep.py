#!/usr/bin/env python3
"""Kernel-time sweep on the GPU box: ms per launch of the episode kernel for BASELINE configs and
packing choices.  usage: python tools/sweep.py [--configs 2,3] [--segs 0,1,2,6] [--reps 5]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def launch_tag(eng):
    ll = eng.last_launch()
    return (f"[{ll['mapping']}" + (f" S={ll['chunk']}" if ll["chunk"] else "") + f" x{ll['trajectories_per_wavefront']}"
            f" wg={ll['workgroups']}x{ll['wavefronts_per_workgroup']} build={ll['build_wavefronts_per_simd']}]")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="2")
    ap.add_argument("--segs", default="0")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--pop-scale", type=int, default=1, help="multiply the population (bigger batches)")
    ap.add_argument("--pop", type=int, default=0, help="override the population (candidates)")
    ap.add_argument("--chunk", type=int, default=0, help="chunk_size option (scan mode 4)")
    ap.add_argument("--no-skips", type=int, default=0)
    ap.add_argument("--no-unify", type=int, default=0)
    ap.add_argument("--no-lat", type=int, default=0)
    ap.add_argument("--scan-mode", default="0", help="comma list of scan modes: 0 auto, 1 LDS windows, 2 DPP rows, 3 all inits in one wavefront")
    a = ap.parse_args()
    import torch
    from l4dc_mpc_ocd_amd import scenarios
    from l4dc_mpc_ocd_amd.engine import Engine
    for cfg in [int(c) for c in a.configs.split(",")]:
        c = scenarios.BASELINE_CONFIGS[cfg]
        scn = scenarios.SCENARIOS[c["scenario"]](horizon=c["horizon"])
        P, N, S = (a.pop or c["pop"]) * a.pop_scale, c["n_inits"], scn.desc.n_samples
        inits = scn.init_dist.sample(N, seed=1000 + cfg)
        w32 = np.stack([scenarios.planner_weights_fp32(x) for x in scn.candidate_weights(P, seed=2000 + cfg)])
        eng = Engine(scn, "cuda:0")
        init_dev = torch.as_tensor(inits, dtype=torch.float32).cuda()
        w_dev = torch.as_tensor(w32).cuda()
        E = P * N * S
        ret = torch.empty(E, dtype=torch.float32, device="cuda")
        eng.set_option("no_feature_skips", a.no_skips)
        eng.set_option("chunk_size", a.chunk)
        eng.set_option("no_unified_features", a.no_unify)
        eng.set_option("no_latency_build", a.no_lat)
        for mode in [int(m) for m in a.scan_mode.split(",")]:
            if (mode == 2 and c["horizon"] > 16) or (mode == 3 and scn.desc.n_ctrl_inits * c["horizon"] > 64) or \
                    (mode == 4 and not a.chunk and c["horizon"] % 5):
                continue
            eng.set_option("scan_mode", mode)
            for segs in [int(s) for s in a.segs.split(",")]:
                if (mode == 2 and segs > 4) or (mode == 3 and segs > 64 // (scn.desc.n_ctrl_inits * c["horizon"])) or \
                        (mode == 4 and segs > 64 // (scn.desc.n_ctrl_inits * -(-c["horizon"] // (a.chunk or 5)))):
                    continue
                eng.set_option("segs_per_wave", segs)
                eng.time_rollout(init_dev, w_dev, 0, E, ret, 1)
                ms = eng.time_rollout(init_dev, w_dev, 0, E, ret, a.reps)
                print(f"cfg{cfg} {c['scenario']} H={c['horizon']} E={E} scan_mode={mode} segs={segs}: {ms:.3f} ms/launch "
                      f"-> {E / ms * 1e3:.0f} episodes/s  checksum {float(ret.sum()):.6f}  {launch_tag(eng)}", flush=True)


if __name__ == "__main__":
    main()
