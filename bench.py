#!/usr/bin/env python3
"""Headline benchmark: full receding-horizon MPC episode rollouts per second at planning horizon
H=10 (BASELINE.json metric), i.e. the fitness evaluation of one CMA-ES generation.

One "step" = one generation: launch the episode kernel over this rank's candidate block, all-gather
the fp32 returns (the only collective), copy them to the host and reduce them to per-candidate
costs in float64 (mpc_ord.py:126-151).  Workload at N GPUs: BASELINE config 2 (finite_horizon,
pop 16 x 8 inits, H=10, 128 episodes) PER GPU -- weak scaling, the population grows with N.

Prints ONE JSON line (rank 0).  `roofline` follows the contract's hbm/mfma vocabulary although the
path is bound by fp32 VALU issue and dependent-op latency: the HBM fraction on
algorithmic bytes is reported as it is (tiny), and `valu` gives the fp32-vector fraction next to it.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
VALU_PEAK_TFLOPS = 157.3       # MI355X_MICROARCH.md: peak fp32 vector


def pmc_traffic_bytes(cfg, n_episodes):
    """HBM bytes per launch of the episode kernel from the committed rocprofv3 PMC passes
    (profiles/pmc_traffic.json: FETCH_SIZE + WRITE_SIZE, separate --pmc runs of this same command,
    KiB -> bytes; dword-granular accesses, so the guide's 2x FETCH correction for wide streaming
    reads does not apply).  None when no profile of this workload is committed."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            rec = json.load(f).get(f"cfg{cfg}")
    except (OSError, ValueError):
        return None
    if not rec or rec.get("episodes_per_launch") != n_episodes:
        return None
    return (rec["fetch_kib"] + rec["write_kib"]) * 1024.0


def algorithmic_per_episode(desc):
    """compulsory HBM bytes and flops of one episode."""
    T, H, I = desc.episode_len, desc.horizon, desc.n_iter
    K = desc.n_ctrl_inits
    C = desc.n_cars
    f_step = 450 if C == 2 else 540
    flops = T * K * (I + 1) * H * f_step
    nbytes = 16 + 4 * desc.n_features + 4      # init state + weights (not amortised) + return
    return nbytes, flops


def usable_cores():
    """CPU threads this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:            # cgroup v2: "<quota> <period>" or "max <period>"
            q, per = f.read().split()
            if q != "max":
                n = min(n, max(1, int(int(q) / int(per))))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                per = int(f.read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(scn, inits, w32, budget_s=20.0):
    """The CPU oracle (kind "port") timed on this host's cores on a bounded sample of the workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    orc = oracle_lib.load()
    cores = min(usable_cores(), int(os.environ.get("OCD_CPU_THREADS", "64")))
    P, N = w32.shape[0], inits.shape[0]
    E = P * N * scn.desc.n_samples
    # calibrate on a few episodes, then size the sample for ~budget_s of wall time
    n0 = min(E, 2 * cores)
    t0 = time.perf_counter()
    orc.rollout(scn.desc, inits, w32, ep_begin=0, ep_end=n0, n_threads=cores)
    dt0 = time.perf_counter() - t0
    rate0 = n0 / max(dt0, 1e-6)
    reps = max(1, int(budget_s * rate0 / E))
    t0 = time.perf_counter()
    for _ in range(reps):
        orc.rollout(scn.desc, inits, w32, n_threads=cores)
    dt = time.perf_counter() - t0
    return {"value": reps * E / dt, "unit": "episodes/s", "cores": cores, "kind": "port",
            "sample": f"{reps} x the {E}-episode workload, OpenMP over episodes on {cores} threads, "
                      f"{dt:.1f} s of CPU work (oracle/ocd_oracle.c)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=2, help="BASELINE.json config index (2..5)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (default); gloo only to rehearse N>1 on a one-GPU box")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from l4dc_mpc_ocd_amd import scenarios, sharding
    from l4dc_mpc_ocd_amd.engine import Engine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs an MI355X: the planner has no CPU fallback")
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    device = f"cuda:{dev_index}"
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(device))
        else:
            dist.init_process_group("gloo")

    cfg = scenarios.BASELINE_CONFIGS[args.config]
    scn = scenarios.SCENARIOS[cfg["scenario"]](horizon=cfg["horizon"])
    d = scn.desc
    N, S = cfg["n_inits"], d.n_samples
    P = cfg["pop"] * world                              # weak scaling: population grows with the GPUs
    inits = scn.init_dist.sample(N, seed=1000 + args.config)
    cands = scn.candidate_weights(P, seed=2000 + args.config)
    w32 = np.stack([scenarios.planner_weights_fp32(c) for c in cands])

    eng = Engine(scn, device)
    init_dev = torch.as_tensor(inits, dtype=torch.float32).to(device)
    w_dev = torch.as_tensor(w32).to(device)                 # inputs resident in HBM before timing
    e0, e1 = sharding.episode_range(P, N, S, world, rank)
    ret_dev = torch.empty(e1 - e0, dtype=torch.float32, device=device)

    def generation():
        eng._call(eng.lib.ocd_rollout_episodes, eng._h, init_dev.data_ptr(), w_dev.data_ptr(), P, N, e0, e1,
                  ret_dev.data_ptr(), None, None, eng._stream())
        local = ret_dev if (world == 1 or args.backend == "nccl") else ret_dev.cpu()
        full = sharding.gather_returns(local, P, N, S)
        return sharding.fitness_from_returns(full.cpu().numpy(), P, N, S)

    for _ in range(args.warmup):
        generation()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fit = generation()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # dominant kernel: average launch duration from HIP events on the launch stream
    kern_ms = eng.time_rollout(init_dev, w_dev, e0, e1, ret_dev, reps=max(3, min(args.steps, 10)))

    if rank == 0:
        E = P * N * S
        n_local = e1 - e0
        nbytes, flops = algorithmic_per_episode(d)
        ach_gbs = n_local * nbytes / (kern_ms * 1e-3) / 1e9
        ach_tf = n_local * flops / (kern_ms * 1e-3) / 1e12
        out = {
            "metric": f"MPC episode rollouts/sec at H={d.horizon} (one CMA-ES generation's fitness evaluation)",
            "value": E * args.steps / dt,
            "unit": "episodes/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"BASELINE config {args.config}: {cfg['scenario']}, CMA-ES pop {cfg['pop']} per GPU x "
                                   f"{N} inits x {S} samples, planning horizon H={d.horizon}, n_iter={d.n_iter}, "
                                   f"K={d.n_ctrl_inits} control inits, episode length T={d.episode_len}",
                       "episodes_per_generation": E, "episodes_per_gpu": n_local,
                       "sharding": f"candidate blocks over {world} rank(s); one all_gather of fp32 returns per generation"},
            "roofline": {"bound": "hbm", "achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach_gbs / HBM_PEAK_GBS, "traffic": pmc_traffic_bytes(args.config, n_local),
                         "kernel": "ocd::mpc_kernel", "kernel_ms": kern_ms,
                         "algorithmic_bytes_per_episode": nbytes,
                         "note": "path is fp32-VALU/latency bound, not HBM bound (SURVEY.md 8d); see valu"},
            "valu": {"achieved": ach_tf, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach_tf / VALU_PEAK_TFLOPS,
                     "algorithmic_flops_per_episode": flops},
            "generation_cost_checksum": float(np.sum(fit)),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scn, inits, w32)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
