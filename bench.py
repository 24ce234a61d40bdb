#!/usr/bin/env python3
"""Headline benchmark: full receding-horizon MPC episode rollouts per second at planning horizon
H=10 (BASELINE.json metric), i.e. the fitness evaluation of one CMA-ES generation, plus the CMA-ES
generation wall-clock through the reference-shaped API.

One "step" = one generation: launch the episode kernel over this rank's candidate block, all-gather the
fp32 returns (the only collective), copy them to the host and reduce them to per-candidate costs in
float64 (mpc_ord.py:126-151).  Workload at N GPUs: BASELINE config 3 (local_opt, CMA-ES pop 64 x 32
inits, H=10, 2 048 episodes -- the largest single-GPU H=10 configuration) PER GPU: weak scaling, the
population grows with N.  BASELINE config 2 (pop 16 x 8 inits, 128 episodes: the small-batch latency
case) is timed as a second block of the same JSON line on rank 0.

`python bench.py --gpus N` starts its N ranks itself (a torch.distributed.run child, spawned before
anything touches the GPU) unless it already runs under a launcher (WORLD_SIZE set); the world size
must equal --gpus or the run fails.

Prints ONE JSON line (rank 0).  `roofline` follows the contract's hbm/mfma vocabulary although the
path is bound by fp32 VALU issue: the HBM fraction on algorithmic bytes is reported as
it is (tiny), and `valu` gives the fp32-vector fraction and the measured VALU issue utilisation.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
VALU_PEAK_TFLOPS = 157.3       # MI355X_MICROARCH.md: peak fp32 vector


def pmc_record(cfg, n_episodes):
    """Counters of the episode kernel from the committed rocprofv3 PMC passes (profiles/pmc_counters.json:
    separate --pmc runs of this same command; FETCH_SIZE / WRITE_SIZE in KiB per launch -- dword-granular
    accesses, so the guide's 2x FETCH correction for wide streaming reads does not apply -- and the SQ
    counters per launch).  None when no profile of this workload is committed."""
    path = os.path.join(ROOT, "profiles", "pmc_counters.json")
    try:
        with open(path) as f:
            rec = json.load(f).get(f"cfg{cfg}")
    except (OSError, ValueError):
        return None
    if not rec or rec.get("episodes_per_launch") != n_episodes:
        return None
    return rec


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
    rate0 = n0 / max(time.perf_counter() - t0, 1e-6)
    n = int(min(E, max(cores, budget_s * rate0)))
    reps = max(1, int(budget_s * rate0 / n)) if n == E else 1
    t0 = time.perf_counter()
    for _ in range(reps):
        orc.rollout(scn.desc, inits, w32, ep_begin=0, ep_end=n, n_threads=cores)
    dt = time.perf_counter() - t0
    return {"value": reps * n / dt, "unit": "episodes/s", "cores": cores, "kind": "port",
            "sample": f"{reps} x the first {n} of the {E} episodes of the workload, OpenMP over episodes on {cores} "
                      f"threads, {dt:.1f} s of CPU work (oracle/ocd_oracle.c)"}


def spawn_ranks(args):
    """--gpus N without a launcher: start N ranks as a torch.distributed.run child.  Nothing in this
    process has touched the GPU; it only waits for the child and passes its exit code on."""
    port = args.master_port or (29000 + os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


def plumbing_check(args, world, rank):
    """Launcher rehearsal without a GPU (tests/test_bench_launcher.py): the ranks meet over gloo, check
    the world size and gather one value each."""
    import torch
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo")
        got = [None] * world
        dist.all_gather_object(got, rank)
        assert got == list(range(world)), got
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"plumbing": True, "n_gpus": world, "torch": torch.__version__}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", type=int, default=3, help="BASELINE.json config index of the headline workload (2..5)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the config-2 block and the CMA-ES generation timing")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (default); gloo only to rehearse N>1 on a one-GPU box")
    ap.add_argument("--master-port", type=int, default=0)
    ap.add_argument("--plumbing-check", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: world size {world} (WORLD_SIZE) != --gpus {args.gpus}")
    if args.plumbing_check:
        return plumbing_check(args, world, rank)

    import torch
    import torch.distributed as dist
    from l4dc_mpc_ocd_amd import scenarios, sharding
    from l4dc_mpc_ocd_amd.engine import Engine

    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs an MI355X: the planner has no CPU fallback")
    if args.backend == "nccl" and world > ndev:
        raise SystemExit(f"bench.py: {world} ranks but {ndev} GPU(s); one rank per GPU (use --backend gloo to rehearse)")
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    device = f"cuda:{dev_index}"
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(device))
        else:
            dist.init_process_group("gloo")
        assert dist.get_world_size() == args.gpus

    def workload(cfg_index, pop_scale):
        cfg = scenarios.BASELINE_CONFIGS[cfg_index]
        scn = scenarios.SCENARIOS[cfg["scenario"]](horizon=cfg["horizon"])
        N, S = cfg["n_inits"], scn.desc.n_samples
        P = cfg["pop"] * pop_scale
        inits = scn.init_dist.sample(N, seed=1000 + cfg_index)
        cands = scn.candidate_weights(P, seed=2000 + cfg_index)
        w32 = np.stack([scenarios.planner_weights_fp32(c) for c in cands])
        return cfg, scn, inits, w32, P, N, S

    def timed_generations(cfg_index, pop_scale, ranks, rnk, steps, warmup):
        """(seconds for `steps` generations, kernel ms, fitness, context) of one workload."""
        cfg, scn, inits, w32, P, N, S = workload(cfg_index, pop_scale)
        eng = Engine(scn, device)
        init_dev = torch.as_tensor(inits, dtype=torch.float32).to(device)
        w_dev = torch.as_tensor(w32).to(device)                 # inputs resident in HBM before timing
        e0, e1 = sharding.episode_range(P, N, S, ranks, rnk)
        ret_dev = torch.empty(e1 - e0, dtype=torch.float32, device=device)
        sharded = ranks > 1

        def generation():
            eng._call(eng.lib.ocd_rollout_episodes, eng._h, init_dev.data_ptr(), w_dev.data_ptr(), P, N, e0, e1,
                      ret_dev.data_ptr(), None, None, eng._stream())
            if sharded:
                local = ret_dev if args.backend == "nccl" else ret_dev.cpu()
                full = sharding.gather_returns(local, P, N, S)
            else:
                full = ret_dev
            return sharding.fitness_from_returns(full.cpu().numpy(), P, N, S)

        for _ in range(warmup):
            generation()
        if sharded:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fit = generation()
        torch.cuda.synchronize()
        if sharded:
            dist.barrier()
        dt = time.perf_counter() - t0
        if sharded:
            tt = torch.tensor([dt], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        # dominant kernel: average launch duration from HIP events on the launch stream
        kern_ms = eng.time_rollout(init_dev, w_dev, e0, e1, ret_dev, reps=max(3, min(steps, 20)))
        return dt, kern_ms, fit, (cfg, scn, inits, w32, P, N, S, e1 - e0)

    def block(cfg_index, dt, kern_ms, ctx, steps):
        cfg, scn, inits, w32, P, N, S, n_local = ctx
        d = scn.desc
        nbytes, flops = algorithmic_per_episode(d)
        ach_gbs = n_local * nbytes / (kern_ms * 1e-3) / 1e9
        ach_tf = n_local * flops / (kern_ms * 1e-3) / 1e12
        pmc = pmc_record(cfg_index, n_local)
        traffic = (pmc["fetch_kib"] + pmc["write_kib"]) * 1024.0 if pmc else None
        valu = {"achieved": ach_tf, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach_tf / VALU_PEAK_TFLOPS,
                "algorithmic_flops_per_episode": flops,
                "note": "at one wavefront per SIMD (2 048 episodes at H=10) a lone wavefront issues one instruction per "
                        "~4.4 cycles: one instruction per algorithmic flop would give ~21 % of the vector peak at this batch size (DESIGN.md section 4)"}
        if pmc and pmc.get("sq_wave_cycles"):
            # quad-cycles in which a wavefront issued a VALU instruction / quad-cycles wavefronts were resident
            valu["issue_utilisation"] = pmc["sq_active_inst_valu"] / pmc["sq_wave_cycles"]
            valu["wait_fraction"] = pmc["sq_wait_any"] / pmc["sq_wave_cycles"]
            valu["valu_instructions_per_launch"] = pmc["sq_insts_valu"]
        return {
            "workload": f"BASELINE config {cfg_index}: {cfg['scenario']}, CMA-ES pop {cfg['pop']} per GPU x {N} inits x "
                        f"{S} samples, planning horizon H={d.horizon}, n_iter={d.n_iter}, K={d.n_ctrl_inits} control "
                        f"inits, episode length T={d.episode_len}",
            "episodes_per_generation": P * N * S, "episodes_per_gpu": n_local,
            "value": P * N * S * steps / dt, "unit": "episodes/s", "ms_per_step": dt / steps * 1e3,
            "roofline": {"bound": "hbm", "achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach_gbs / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "ocd::mpc_kernel", "kernel_ms": kern_ms, "algorithmic_bytes_per_episode": nbytes,
                         "note": "path is fp32-VALU issue bound, not HBM bound (SURVEY.md 8d); see valu"},
            "valu": valu,
        }

    # ---- headline: weak scaling of the chosen config over the ranks ----
    dt, kern_ms, fit, ctx = timed_generations(args.config, world, world, rank, args.steps, args.warmup)

    # ---- extras: config 2 (small-batch latency) on rank 0, CMA-ES generation wall-clock on all ranks ----
    extra2 = None
    cma = None
    if not args.no_extras:
        if rank == 0 and args.config != 2:
            dt2, k2, _, ctx2 = timed_generations(2, 1, 1, 0, args.steps, args.warmup)
            extra2 = block(2, dt2, k2, ctx2, args.steps)
        # ask -> host normalisation -> H2D -> launch -> (gather) -> D2H -> float64 reduction -> tell, through
        # MPC_ORD.optimize_cmaes (mpc_ord.py:33-45); every rank runs the same deterministic strategy
        from l4dc_mpc_ocd_amd.interact_drive.experiments import run_mpc_ord as rmo
        cfg = scenarios.BASELINE_CONFIGS[args.config]
        m = rmo.make_mpc_ord(cfg["scenario"], horizon=cfg["horizon"], n_inits=cfg["n_inits"], seed=1)
        gens = 12
        m.optimize_cmaes(seed=1, sigma0=0.05, popsize=cfg["pop"] * world, maxiter=gens)
        gs = np.array(m.generation_seconds[1:]) * 1e3           # the first generation pays one-off setup
        fs = np.array(m.fitness_seconds[1:]) * 1e3              # eval_population alone (no ask / tell)
        if world > 1:
            tt = torch.tensor([float(np.median(gs))], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            med = float(tt.item())
        else:
            med = float(np.median(gs))
        cma = {"cma_generation_ms": med, "fitness_ms": float(np.median(fs)), "generations_timed": int(len(gs)),
               "popsize": cfg["pop"] * world,
               "n_inits": cfg["n_inits"], "path": "MPC_ORD.optimize_cmaes: ask, normalise, H2D, launch, gather, D2H, "
                                                   "float64 reduction, tell (own CMA-ES; pycma is not installed)"}

    if rank == 0:
        cfg, scn, inits, w32, P, N, S, n_local = ctx
        d = scn.desc
        hb = block(args.config, dt, kern_ms, ctx, args.steps)
        out = {
            "metric": f"MPC episode rollouts/sec at H={d.horizon} (one CMA-ES generation's fitness evaluation); "
                      f"CMA-ES generation wall-clock",
            "value": hb["value"],
            "unit": "episodes/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": hb["ms_per_step"],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": hb["workload"], "episodes_per_generation": hb["episodes_per_generation"],
                       "episodes_per_gpu": hb["episodes_per_gpu"],
                       "sharding": f"candidate blocks over {world} rank(s); one all_gather of fp32 returns per generation"},
            "roofline": hb["roofline"],
            "valu": hb["valu"],
            "generation_cost_checksum": float(np.sum(fit)),
        }
        if cma:
            out["cma_generation_ms"] = cma["cma_generation_ms"]
            out["cma"] = cma
        if extra2:
            out["config2"] = extra2
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scn, inits, w32)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
