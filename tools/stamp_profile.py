#!/usr/bin/env python3
"""Where a wavefront's cycles go: runs one generation of a BASELINE config on the diagnostic build
(`make -C l4dc-mpc-ocd_amd/csrc stamps`: in-kernel s_memtime stamps, ~10 % slower, never shipped) and
prints the per-section cycle totals of the slowest, the median and the fastest wavefront.
usage: python tools/stamp_profile.py --config 2 [--scan-mode 0] [--segs 0]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SECTIONS = ["outside the passes", "v/heading recurrence", "own step + sincos", "x/y recurrence",
            "evaluation choice (ballots)", "features: everything", "features: one per lane", "features: none active",
            "x/y adjoint recurrence", "Jacobian + v/heading adjoint rec.", "control update", "#one-per-lane passes with a collision lane",
            "#passes everything", "#passes one-per-lane", "#passes none", "#one-per-lane passes with a fence lane"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--scan-mode", type=int, default=0)
    ap.add_argument("--segs", type=int, default=0)
    ap.add_argument("--no-skips", type=int, default=0)
    ap.add_argument("--no-lat", type=int, default=0)
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--pop", type=int, default=0, help="override the population (e.g. the share of one of 8 GPUs)")
    ap.add_argument("--light", action="store_true", help="the build with two stamps per wavefront (make stamps_light): totals and placement only")
    ap.add_argument("--json", action="store_true", help="one JSON line with the placement summary instead of the tables (tests/test_gpu_placement.py)")
    ap.add_argument("--warm", type=int, default=0, help="untimed launches before the stamped one")
    a = ap.parse_args()
    os.environ["OCD_HIP_LIB"] = os.environ.get("OCD_STAMPS_LIB") or os.path.join(      # (OCD_STAMPS_LIB: an experimental stamps build)
        ROOT, "l4dc-mpc-ocd_amd", "csrc", "libocd_hip_stamps_light.so" if a.light else "libocd_hip_stamps.so")
    import torch
    from l4dc_mpc_ocd_amd import scenarios
    from l4dc_mpc_ocd_amd.engine import Engine
    scn, inits, cands = scenarios.baseline_config(a.config)
    if a.pop:
        cands = scn.candidate_weights(a.pop, seed=2000 + a.config)
    w32 = np.stack([scenarios.planner_weights_fp32(c) for c in cands])
    eng = Engine(scn, "cuda:0")
    eng.set_option("scan_mode", a.scan_mode)
    eng.set_option("segs_per_wave", a.segs)
    eng.set_option("no_feature_skips", a.no_skips)
    eng.set_option("no_latency_build", a.no_lat)
    eng.set_option("chunk_size", a.chunk)
    E = w32.shape[0] * inits.shape[0] * scn.desc.n_samples
    nw = E * scn.desc.n_ctrl_inits + 64
    buf = torch.zeros(nw * 16, dtype=torch.int64, device="cuda:0")
    eng.lib.ocd_debug_set_stamp_buffer.argtypes = [C.c_void_p]
    eng.lib.ocd_debug_set_stamp_buffer(buf.data_ptr())
    for _ in range(a.warm):
        eng.rollout(inits, w32)
    buf.zero_()
    eng.rollout(inits, w32)
    st = buf.cpu().numpy().reshape(nw, 16)
    st = st[st[:, :11].sum(1) > 0]
    tot = st[:, :11].sum(1)
    order = np.argsort(tot)
    if a.json:
        import json
        hw = st[:, 14]
        out = dict(config=a.config, episodes=int(E), wavefronts=int(len(st)), launch=eng.last_launch(),
                   cycles_median=float(np.median(tot)), cycles_max=float(tot.max()), cycles_min=float(tot.min()),
                   cycles_p99=float(np.percentile(tot, 99)), cycles_p01=float(np.percentile(tot, 1)))
        if (hw >> 40).all():
            key = [((int(v) >> 32) & 0xf, (int(v) >> 13) & 7, (int(v) >> 12) & 1, (int(v) >> 8) & 0xf, (int(v) >> 4) & 3) for v in hw]
            from collections import Counter
            per_simd = Counter(key)
            out.update(simds_used=len(per_simd), max_wavefronts_on_a_simd=max(per_simd.values()),
                       cus_used=len({k[:4] for k in key}), xccs_used=len({k[0] for k in key}))
        print(json.dumps(out))
        return
    picks = [("slowest", order[-1]), ("median", order[len(order) // 2]), ("fastest", order[0])]
    print(f"config {a.config} scan_mode {a.scan_mode} segs {a.segs}: {len(st)} wavefronts, "
          f"T*(n_iter+1) = {scn.desc.episode_len * (scn.desc.n_iter + 1)} passes each")
    print(f"{'section':<36}" + "".join(f"{n:>22}" for n, _ in picks))
    npass = scn.desc.episode_len * (scn.desc.n_iter + 1)
    for i, name in enumerate(SECTIONS):
        if name == "-":
            continue
        row = f"{name:<36}"
        for _, w in picks:
            v = int(st[w, i])
            row += f"{v:>12d}" + (f" ({v / npass:7.1f})" if i < 11 else " " * 10)
        print(row)
    print(f"{'total cycles (per pass)':<36}" + "".join(f"{int(tot[w]):>12d} ({tot[w] / npass:7.1f})" for _, w in picks))
    hw = st[:, 14]
    if (hw >> 40).all():                   # latency build of the chunked kernel: slot 14 = HW_ID | XCC_ID << 32
        key = [((int(v) >> 32) & 0xf, (int(v) >> 13) & 7, (int(v) >> 12) & 1, (int(v) >> 8) & 0xf, (int(v) >> 4) & 3) for v in hw]
        from collections import Counter
        per_simd = Counter(key)
        shared = [k for k, n in per_simd.items() if n > 1]
        print(f"placement: {len(per_simd)} SIMDs used by {len(key)} wavefronts; SIMDs with more than one wavefront: {len(shared)}")
        pct = np.percentile(tot, [50, 90, 99, 99.9, 100])
        print("  wavefront cycles: median / 90 % / 99 % / 99.9 % / max = " + " / ".join(f"{q:.0f}" for q in pct))
        slow = order[-8:][::-1]
        for wv in slow:
            k = key[wv]
            print(f"  wavefront {wv}: {int(tot[wv])} cycles, (xcc, se, sh, cu, simd) = {k}, wavefronts on that SIMD: {per_simd[k]}, "
                  f"on that CU: {sum(n for kk, n in per_simd.items() if kk[:4] == k[:4])}")
        per_cu = Counter(k[:4] for k in key)
        cu_tot = {}
        for i, k in enumerate(key):
            cu_tot.setdefault(k[:4], []).append(float(tot[i]))
        cu_mean = sorted(((np.mean(v), np.max(v) - np.min(v), k) for k, v in cu_tot.items()), reverse=True)
        qs = np.percentile([m for m, _, _ in cu_mean], [0, 10, 50, 90, 99, 100])
        print("  per-CU mean cycles: min / 10 % / median / 90 % / 99 % / max = " + " / ".join(f"{q:.0f}" for q in qs))
        print("  slowest CUs (mean cycles, spread inside the CU, (xcc, se, sh, cu)): " +
              "; ".join(f"{m:.0f} +-{sp:.0f} {k}" for m, sp, k in cu_mean[:6]))
        sec = st[:, :11].astype(np.float64)
        slow_cu = [i for i, k in enumerate(key) if k[:4] in {c for _, _, c in cu_mean[:8]}]
        rest = [i for i in range(len(key)) if i not in set(slow_cu)]
        ratio = sec[slow_cu].mean(0) / np.maximum(sec[rest].mean(0), 1.0)
        print("  section time on the 8 slowest CUs / on the others: " + " ".join(f"{r:.2f}" for r in ratio))
        print(f"  wavefronts per CU: min {min(per_cu.values())} max {max(per_cu.values())} over {len(per_cu)} CUs")
        # mean total cycles of the wavefronts by how many share their SIMD
        for n in sorted(set(per_simd.values())):
            sel = [i for i, k in enumerate(key) if per_simd[k] == n]
            print(f"  {len(sel)} wavefronts on SIMDs holding {n}: mean {tot[sel].mean():.0f} cycles, max {tot[sel].max():.0f}")


if __name__ == "__main__":
    main()
