#!/usr/bin/env python3
"""Host-side latency of one generation around the episode kernel (config 3), for the ways the candidate weights can
reach the kernel and the returns can reach the host:
    weights: copied into a device buffer (pinned staging, async copy)  |  read by the kernel from pinned host memory
    returns: device buffer + async copy into pinned host memory        |  written by the kernel into pinned host memory
    wait:    event.synchronize()  |  polling event.query()
Prints the median wall time of (stage weights, launch, wait, returns readable) minus the kernel's own duration."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from l4dc_mpc_ocd_amd import abi, scenarios
    from l4dc_mpc_ocd_amd.engine import Engine
    cfg = scenarios.BASELINE_CONFIGS[3]
    scn = scenarios.SCENARIOS[cfg["scenario"]](horizon=cfg["horizon"])
    P, N, S = cfg["pop"], cfg["n_inits"], scn.desc.n_samples
    inits = scn.init_dist.sample(N, seed=1003)
    w32 = np.stack([scenarios.planner_weights_fp32(c) for c in scn.candidate_weights(P, seed=2003)])
    eng = Engine(scn, "cuda:0")
    E = P * N * S
    init_dev = torch.as_tensor(inits, dtype=torch.float32).cuda()
    w_dev = torch.as_tensor(w32).cuda()
    w_pin = torch.as_tensor(w32).pin_memory()
    ret_dev = torch.empty(E, dtype=torch.float32, device="cuda")
    ret_pin = torch.empty(E, dtype=torch.float32).pin_memory()
    kern_ms = eng.time_rollout(init_dev, w_dev, 0, E, ret_dev, 20)
    print(f"kernel {kern_ms:.4f} ms")
    ev = torch.cuda.Event()
    stream = torch.cuda.current_stream()
    lib, h = eng.lib, eng._h
    for w_mode in ("copy", "pinned"):
        for r_mode in ("copy", "pinned"):
            for wait in ("sync", "poll"):
                ts = []
                for i in range(120):
                    t0 = time.perf_counter()
                    if w_mode == "copy":
                        w_dev.copy_(w_pin, non_blocking=True)
                        wp = w_dev.data_ptr()
                    else:
                        wp = w_pin.data_ptr()
                    rp = ret_dev.data_ptr() if r_mode == "copy" else ret_pin.data_ptr()
                    abi.check(lib, lib.ocd_rollout_episodes(h, init_dev.data_ptr(), wp, P, N, 0, E, rp, None, None, stream.cuda_stream))
                    if r_mode == "copy":
                        ret_pin.copy_(ret_dev, non_blocking=True)
                    ev.record(stream)
                    if wait == "sync":
                        ev.synchronize()
                    else:
                        while not ev.query():
                            pass
                    s = float(ret_pin[0])
                    ts.append(time.perf_counter() - t0)
                ts = np.array(ts[20:]) * 1e3
                print(f"weights {w_mode:6s} returns {r_mode:6s} wait {wait}: median {np.median(ts):.4f} ms, min {ts.min():.4f}  "
                      f"-> overhead over the kernel {1e3 * (np.median(ts) - kern_ms):.1f} us", flush=True)


if __name__ == "__main__":
    main()
