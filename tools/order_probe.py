#!/usr/bin/env python3
"""Does the ORDER in which a launch's episodes are dealt to wavefront slots matter?  (GPU box.)  A launch of at most one
wavefront per SIMD lasts as long as its slowest wavefront, and a wavefront takes the multi-feature evaluation in every step
in which ANY of its trajectories needs it.  The flat order puts consecutive inits of one candidate into a wavefront; this
times the same episodes through ocd_rollout_indexed in candidate-major (flat), init-major and shuffled order.
usage: order_probe.py <config> <pop>"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from l4dc_mpc_ocd_amd import abi, scenarios
    from l4dc_mpc_ocd_amd.engine import Engine
    cfg, pop = int(sys.argv[1]), int(sys.argv[2])
    c = scenarios.BASELINE_CONFIGS[cfg]
    scn = scenarios.SCENARIOS[c["scenario"]](horizon=c["horizon"])
    N, S = c["n_inits"], scn.desc.n_samples
    inits = scn.init_dist.sample(N, seed=1000 + cfg)
    w32 = np.stack([scenarios.planner_weights_fp32(x) for x in scn.candidate_weights(c["pop"], seed=2000 + cfg)])[:pop]
    eng = Engine(scn, "cuda:0")
    init_dev = torch.as_tensor(np.asarray(inits, dtype=np.float32)).cuda()
    w_dev = torch.as_tensor(w32).cuda()
    E = pop * N * S
    flat = np.array([(p, n, (p * N + n) * S + s) for p in range(pop) for n in range(N) for s in range(S)], dtype=np.int32)
    orders = {"candidate-major (flat)": np.arange(E),
              "init-major": np.array(sorted(range(E), key=lambda e: (flat[e, 1], flat[e, 2] % S, flat[e, 0]))),
              "init-major, samples apart": np.array(sorted(range(E), key=lambda e: (flat[e, 2] % S, flat[e, 1], flat[e, 0]))),
              "shuffled": np.random.default_rng(0).permutation(E)}
    ref = None
    for name, perm in orders.items():
        idx = torch.as_tensor(flat[perm]).cuda()
        ret = torch.empty(E, dtype=torch.float32, device="cuda")

        def launch():
            abi.check(eng.lib, eng.lib.ocd_rollout_indexed(eng._h, init_dev.data_ptr(), N, w_dev.data_ptr(), pop, idx.data_ptr(), E,
                                                           ret.data_ptr(), None, None, eng._stream()))
        for _ in range(3):
            launch()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(8):
            launch()
        ev1.record()
        ev1.synchronize()
        out = np.empty(E, dtype=np.float32)
        out[perm] = ret.cpu().numpy()
        ref = out if ref is None else ref
        print(f"cfg{cfg} pop {pop} ({E} episodes) {name}: {ev0.elapsed_time(ev1) / 8:.3f} ms/launch {eng.last_launch()['mapping']} "
              f"S={eng.last_launch()['chunk']}; same returns: {np.array_equal(out, ref, equal_nan=True)}")


if __name__ == "__main__":
    main()
