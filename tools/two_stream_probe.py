#!/usr/bin/env python3
"""Do two launches of the episode kernel on two streams run side by side?  (GPU box.)  The lockstep path of round 6 splits a
generation's runs into groups on their own streams; this times, for R runs of the reference's shape (pop 9 x 3 inits, H = 5):
one launch of all R x 27 episodes, two launches of half each on two streams, and the same two on ONE stream."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from l4dc_mpc_ocd_amd import abi, scenarios
    from l4dc_mpc_ocd_amd.engine import Engine
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 28
    scn = scenarios.finite_horizon(horizon=5)
    eng = Engine(scn, "cuda:0")
    pop, N = 9, 3
    rows, inits, ws = [], [], []
    for r in range(R):
        rows += [(r * pop + e // N, r * N + e % N, e) for e in range(pop * N)]
        inits.append(np.asarray(scn.init_dist.sample(N, seed=300 + r), dtype=np.float32))
        ws.append(np.stack([scenarios.planner_weights_fp32(c) for c in scn.candidate_weights(pop, seed=400 + r)]))
    idx = torch.as_tensor(np.asarray(rows, dtype=np.int32)).pin_memory()
    init_dev = torch.as_tensor(np.concatenate(inits)).cuda()
    w_dev = torch.as_tensor(np.concatenate(ws)).cuda()
    E = len(rows)
    ret = torch.empty(E, dtype=torch.float32).pin_memory()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    half = (R // 2) * pop * N

    def launch(e0, e1, stream):
        abi.check(eng.lib, eng.lib.ocd_rollout_indexed(eng._h, init_dev.data_ptr(), init_dev.shape[0], w_dev.data_ptr(), w_dev.shape[0],
                                                       idx.data_ptr() + 12 * e0, e1 - e0, ret.data_ptr() + 4 * e0, None, None, stream.cuda_stream))

    def timed(fn, reps=30):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    def one():
        launch(0, E, s1); s1.synchronize()

    def two_streams():
        launch(0, half, s1); launch(half, E, s2); s1.synchronize(); s2.synchronize()

    def two_same():
        launch(0, half, s1); launch(half, E, s1); s1.synchronize()

    def half_only():
        launch(0, half, s1); s1.synchronize()

    print(f"{R} runs, {E} episodes: one launch {timed(one):.3f} ms ({eng.last_launch()}); half alone {timed(half_only):.3f} ms; "
          f"two halves on two streams {timed(two_streams):.3f} ms; two halves on one stream {timed(two_same):.3f} ms")


if __name__ == "__main__":
    main()
