#!/usr/bin/env python3
"""Randomised differential stress of the specialised kernels (the shapes with V_ROW / V_SEG / V_CHUNK builds):
random descriptors as in tests/test_gpu_random_scenarios.py, plus the rare paths on purpose -- coinciding
lane centres (the reduce_min tie in every pass), two scripted cars on top of each other (multi-feature
lanes: the complete evaluation), scan modes 0 / 2 / 3 / 4 with the latency builds on and off.  HIP path vs
the CPU oracle (checker), bit for bit.   usage (GPU box): python tools/stress_parity.py --cases 300"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def same(a, b):
    return a.shape == b.shape and bool(np.all((a == b) | (np.isnan(a) & np.isnan(b))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=7000)
    a = ap.parse_args()
    import oracle_lib
    from test_gpu_random_scenarios import random_scenario
    from l4dc_mpc_ocd_amd.engine import Engine
    orc = oracle_lib.load()
    shapes = [(5, 1, 3), (6, 1, 3), (10, 1, 3), (15, 1, 3), (5, 2, 2), (10, 2, 2), (15, 2, 2), (5, 2, 3), (10, 2, 3),
              (25, 1, 3), (25, 2, 3)]
    bad = 0
    for case in range(a.cases):
        rng = np.random.default_rng(a.seed + case)
        H, NO, L = shapes[case % len(shapes)]
        scn = random_scenario(rng, H, NO, L)
        d = scn.desc
        d.n_iter = int(rng.integers(20, 60))
        kind = case % 4
        if kind == 1 and L >= 2:                       # coinciding lane centres: lane ties in every pass
            d.lane_center[1] = d.lane_center[0]
        if kind == 2 and NO >= 2:                      # two scripted cars at the same place: multi-feature lanes
            for k in range(4):
                d.other_init[1][k] = d.other_init[0][k]
            d.other_plan_len[1] = d.other_plan_len[0]
            for t in range(int(d.other_plan_len[0])):
                d.other_plan[1][t][0], d.other_plan[1][t][1] = d.other_plan[0][t][0], d.other_plan[0][t][1]
        eng = Engine(scn, "cuda:0")
        mode = int(rng.choice([0, 2, 3, 4]))
        eng.set_option("scan_mode", mode)
        eng.set_option("no_latency_build", int(rng.integers(0, 2)))
        # every compiled chunk size of the shape in turn (sizes that do not divide H included); 0 = the launcher's choice
        chunk = int(rng.choice({10: [0, 2, 5], 15: [0, 2, 3, 5], 25: [0, 2, 3, 5]}.get(H, [0]))) if mode == 4 else 0
        eng.set_option("chunk_size", chunk)
        B = int(rng.integers(1, 40))
        ws = np.zeros((B, NO + 1, 4), dtype=np.float32)
        ws[:, 0, 0] = rng.uniform(-0.2, 0.2, B); ws[:, 0, 1] = rng.uniform(-1.4, -0.5, B)
        ws[:, 0, 2] = rng.uniform(0.2, 1.3, B); ws[:, 0, 3] = np.pi / 2 + rng.uniform(-0.4, 0.4, B)
        for j in range(NO):
            ws[:, j + 1] = np.array(d.other_init[j][:]) + rng.uniform(-0.05, 0.05, (B, 4))
        if kind == 3:                                  # the ego right on a scripted car and next to the fence
            ws[:, 0, 0] = ws[:, 1, 0] + rng.uniform(-0.02, 0.02, B)
            ws[:, 0, 1] = ws[:, 1, 1] - rng.uniform(0.0, 0.2, B)
        wts = rng.standard_normal((B, L + 4))
        wts[:, L + 1:] = -np.abs(wts[:, L + 1:]) * 2
        w32 = (wts / np.linalg.norm(wts, axis=1, keepdims=True)).astype(np.float32)
        ref = orc.plan_batch(d, ws, w32, other_plans=scn.other_plans())
        out = eng.plan_batch(ws, w32, want_all=True)
        ok = all(same(out[k], ref[k]) for k in ("all_losses", "all_plans", "plans", "best_loss")) and \
            np.array_equal(out["best_init"], ref["best_init"])
        inits = ws[: min(B, 3), 0]
        ro = eng.rollout(inits, w32[:2], want_traj=True)
        rr = orc.rollout(d, inits, w32[:2], want_traj=True)
        ok = ok and all(same(ro[k], rr[k]) for k in ("ctrl", "traj", "returns"))
        if not ok:
            bad += 1
            print(f"MISMATCH case {case}: H={H} NO={NO} L={L} kind={kind} scan_mode={mode} chunk={chunk} B={B}", flush=True)
        if case % 25 == 24:
            print(f"{case + 1} cases, {bad} mismatches", flush=True)
    print(f"done: {a.cases} cases, {bad} mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
