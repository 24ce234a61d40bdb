#!/usr/bin/env python3
"""Headline benchmark: full receding-horizon MPC episode rollouts per second at planning horizon
H=10 (BASELINE.json metric), i.e. the fitness evaluation of one CMA-ES generation, plus the CMA-ES
generation wall-clock through the reference-shaped API.

One "step" = one generation: launch the episode kernel over this rank's candidate block, all-gather the
fp32 returns (the only collective), copy them to the host and reduce them to per-candidate costs in
float64 (mpc_ord.py:126-151).  Workload at N GPUs: BASELINE config 3 (local_opt, CMA-ES pop 64 x 32
inits, H=10, 2 048 episodes -- the largest single-GPU H=10 configuration) PER GPU: weak scaling, the
population grows with N.  `--scaling strong` keeps the population fixed and splits its candidate blocks
over the ranks instead (the reference's own parallel axis splits a fixed batch, run_mpc_ord.py:83-90,
mpc_ord.py:128-137): `--config 4 --gpus 8 --scaling strong` / `--config 5 ...` are BASELINE configs 4 / 5 as
BASELINE.json states them.  `--emulate-rank R/W` runs, on ONE GPU, exactly the block rank R of a W-way strong
split would run (config 4: 2 048 episodes at H=15, config 5: 4 096 at H=25, for W=8).
Extra blocks of the same JSON line on rank 0 (single-GPU runs): BASELINE config 2 (pop 16 x 8 inits, 128
episodes: the small-batch latency case), the rank-0-of-8 shares of configs 4 and 5 and the two configs whole on this one
GPU (`config4_whole`, `config5_whole`: 16 384 / 32 768 episodes per launch), and the REFERENCE's own shape
(`reference_h5`: finite_horizon H=5, pycma's default population 9 x 3 inits, K=3 -- what a user of
`run_mpc_ord.py finite_horizon cmaes --n_inits 3` gets per generation; `reference_h6_extra`: H=6, n_iter 200,
extra_inits K=6, the validation scripts' planner), each with its own CMA-ES generation wall-clock and a bounded CPU
baseline beside it.
`collective`: the one collective of a generation (all_gather_into_tensor of the fp32 returns over RCCL) timed by
itself on device memory, with the ranks the process group saw -- at N=1 a one-rank RCCL group is created for it
after the timed region (the headline stays the plain single-GPU step); `--force-collective` puts it INSIDE the timed
step at N=1 as well.

`python bench.py --gpus N` starts its N ranks itself (a torch.distributed.run child, spawned before
anything touches the GPU) unless it already runs under a launcher (WORLD_SIZE set); the world size
must equal --gpus or the run fails.

At N > 1 the line also carries `config4` and `config5`: BASELINE configs 4 / 5 AS STATED (replanning pop 128 x 64 inits x 2
samples at H=15; merging pop 256 x 128 inits at H=25), the fixed population split into N candidate blocks (strong
scaling), the RCCL all-gather of the returns inside every timed step, max-over-ranks wall-clock, every rank's kernel ms.
Every block that names a BASELINE config carries `parity`: an UNTIMED check of this run's HIP results against the CPU
oracle (the oracle as checker, outside every timed region) on a bounded sample of episodes spread over every rank's block
-- returns, world trajectories and applied controls bit for bit, returns within 1e-4 relative, and the kept control
initialisation of every control step (`argmin_flips`); counts are summed over the ranks.

Prints ONE JSON line (rank 0) of at most 8 KB (asserted before printing; round 5's had grown to 32 KB and the driver's
record of it did not parse): the contract's keys, a compact `roofline` (bound / achieved / peak / unit / frac / traffic /
kernel / kernel_ms / binding / binding_frac ...), `cpu_baseline`, `parity`, `collective`, per extra block six numbers
(episodes, ms_per_step, value, kernel_ms, binding_frac, parity_ok), and `predicted_strong_scaling` (N = 1: every rank's
block of a 2 / 4 / 8-way split of configs 4 / 5 timed on this one GPU + the measured gather -- a prediction, flagged as one).
Everything else -- launch records, host splits, rocprof replay objects, what was sampled and against what -- is written
whole to `bench_detail.json` beside this file (or $OCD_BENCH_DETAIL); the line names it (`detail`).
`roofline` follows the contract's hbm/mfma vocabulary although the path is bound by fp32 VALU issue: the HBM
fraction on algorithmic bytes is reported as it is (tiny), and `binding` / `binding_frac` name the bound that binds.
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


def pmc_record(key, n_episodes, kernel_name):
    """Counters of the episode kernel from the committed rocprofv3 PMC passes (profiles/pmc_counters.json:
    separate --pmc runs of this same command; FETCH_SIZE / WRITE_SIZE in KiB per launch -- dword-granular
    accesses, so the guide's 2x FETCH correction for wide streaming reads does not apply -- and the SQ
    counters per launch).  Returns (record, None), or (None, why): a record is replayed only beside THE kernel it was
    taken from -- same demangled name as the launch it annotates, same episodes per launch, and kernels built from the
    same sources (abi.kernel_source_sha) as the profiled ones."""
    from l4dc_mpc_ocd_amd import abi
    path = os.path.join(ROOT, "profiles", "pmc_counters.json")
    try:
        with open(path) as f:
            allrec = json.load(f)
    except (OSError, ValueError):
        return None, "no profiles/pmc_counters.json"
    rec = allrec.get(key)
    if not rec:
        return None, f"no profile of {key} is committed"
    if rec.get("episodes_per_launch") != n_episodes:
        return None, f"the profile of {key} has {rec.get('episodes_per_launch')} episodes per launch, this launch {n_episodes}"
    if rec.get("kernel_name") != kernel_name:
        return None, f"the profile of {key} is of {rec.get('kernel_name')}, this launch runs {kernel_name}"
    if allrec.get("_kernel_source_sha") != abi.kernel_source_sha():
        return None, (f"stale: the profile was taken on kernel sources {allrec.get('_kernel_source_sha')} (commit "
                      f"{allrec.get('_git_commit')}), this tree has {abi.kernel_source_sha()}")
    rec = dict(rec, git_commit=allrec.get("_git_commit"), kernel_source_sha=allrec.get("_kernel_source_sha"))
    return rec, None


PMC_SOURCE = ("profiles/pmc_counters.json: rocprofv3 --pmc passes of this command on the builder's box "
              "(tools/profile_round.sh), NOT observed in this run")


def parse_emulate(text):
    """'R/W' -> (R, W)."""
    try:
        r, w = (int(v) for v in text.split("/"))
    except ValueError:
        raise SystemExit(f"--emulate-rank wants R/W, got {text!r}")
    if not (w >= 1 and 0 <= r < w):
        raise SystemExit(f"--emulate-rank {text}: need 0 <= R < W")
    return r, w


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


def cpu_baseline(scn, inits, w32, budget_s=16.0):
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
            "sample_short": f"{reps} x first {n} of {E} episodes, {cores} OpenMP threads, {dt:.1f} s",
            "sample": f"{reps} x the first {n} of the {E} episodes of the workload, OpenMP over episodes on {cores} "
                      f"threads, {dt:.1f} s of CPU work (oracle/ocd_oracle.c)"}


def _same_bits(a, b):
    """Element-wise: equal, or both NaN."""
    return (a == b) | (np.isnan(a) & np.isnan(b))


def parity_sample(scn, eng, inits, w32, e0, e1, n_want, n_threads):
    """Untimed parity of the HIP path against the CPU oracle (the checker) on episodes of [e0, e1): `n_want` episodes in
    up to 8 contiguous chunks spread evenly over the range.  Returns counts (numpy int64 [6]):
    episodes_checked, bitwise_equal (returns AND every world state AND every applied control of the episode),
    within_1e-4_rel (returns), plan_steps_checked, argmin_flips (control steps at which ocd_plan_batch, run on the HIP
    episode's own world state, keeps another control initialisation than the oracle's planner does), nonfinite_returns."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    orc = oracle_lib.load()
    d = scn.desc
    N, S, T = inits.shape[0], d.n_samples, d.episode_len
    n_local = e1 - e0
    n = int(min(n_local, n_want))
    chunks = 8 if n >= 64 else 1
    per = n // chunks
    counts = np.zeros(6, dtype=np.int64)
    if per < 1:
        return counts
    other = scn.other_plans()
    for c in range(chunks):
        b = e0 + (c * (n_local - per)) // max(chunks - 1, 1)
        got = eng.rollout(inits, w32, ep_begin=b, ep_end=b + per, want_traj=True)
        ref = orc.rollout(d, inits, w32, ep_begin=b, ep_end=b + per, want_traj=True, n_threads=n_threads)
        same = _same_bits(got["returns"], ref["returns"]) \
            & _same_bits(got["traj"], ref["traj"]).reshape(per, -1).all(axis=1) \
            & _same_bits(got["ctrl"], ref["ctrl"]).reshape(per, -1).all(axis=1)
        with np.errstate(invalid="ignore"):
            close = np.abs(got["returns"].astype(np.float64) - ref["returns"]) <= 1e-4 * np.maximum(1e-2, np.abs(ref["returns"]))
        close |= np.isnan(got["returns"]) & np.isnan(ref["returns"])
        # the state each control step plans from: the state after the previous step, after the teleport if this is its step
        ws = got["traj"][:, :T].copy()                                          # [per, T, C, 4]
        if d.teleport_step > 0 and d.teleport_step <= T:
            ep = np.arange(b, b + per)
            k = (ep % d.teleport_period) if d.teleport_period > 0 else (ep % S)
            car = np.array([d.teleport_car[i] for i in range(max(d.teleport_period, S))], dtype=np.int64)[k]
            ws[np.arange(per), d.teleport_step - 1, car] = np.array(d.teleport_state[:], dtype=np.float32)
        rows = np.repeat(w32[np.arange(b, b + per) // (N * S)], T, axis=0)
        ws = ws.reshape(per * T, d.n_cars, 4)
        finite = np.isfinite(ws).reshape(per * T, -1).all(axis=1)               # (a runaway episode plans from inf / NaN)
        k_hip = eng.plan_batch(ws[finite], rows[finite])["best_init"]
        k_ref = orc.plan_batch(d, ws[finite], rows[finite], other_plans=other, n_threads=n_threads)["best_init"]
        counts += np.array([per, int(same.sum()), int(close.sum()), int(finite.sum()), int((k_hip != k_ref).sum()),
                            int((~np.isfinite(got["returns"])).sum())], dtype=np.int64)
    return counts


PARITY_KEYS = ("episodes_checked", "bitwise_equal", "within_1e-4_rel", "plan_steps_checked", "argmin_flips",
               "nonfinite_returns")


# ---- the printed line and its side file ---------------------------------------------------------------------------
# stdout carries ONE line of at most LINE_CAP bytes: the contract's keys, a compact `roofline`, `cpu_baseline`, `parity`,
# `collective`, and per extra block six numbers.  Everything else (launch records, host splits, rocprof replay objects,
# the sentences that say what was sampled and against what) is written whole to DETAIL_NAME beside this file (or to
# $OCD_BENCH_DETAIL); the line names that file.  Round 5's line had grown to 32 KB and the driver could not parse it.
LINE_CAP = 8192
DETAIL_NAME = "bench_detail.json"


def _sig(x, digits=7):
    """Floats of the extra blocks to 7 significant digits (the contract's own keys are printed in full)."""
    if isinstance(x, float) and np.isfinite(x):
        return float(f"{x:.{digits}g}")
    return x


def parity_ok(par):
    """One flag of a `parity` object: something was checked, every checked episode is bit for bit the oracle's (returns,
    world states, applied controls), every return is within 1e-4 relative and no kept control initialisation differs."""
    if not par:
        return None
    if "episodes_checked" in par and "bitwise_equal" in par:
        n = par["episodes_checked"]
        return bool(n > 0 and par["bitwise_equal"] == n and par.get("within_1e-4_rel", n) == n
                    and par.get("argmin_flips", 0) == 0)
    flags = [v for k, v in par.items() if k.endswith("bitwise_equal")]
    return bool(flags and all(flags))


def compact_parity(par):
    if not par:
        return None
    return {k: par[k] for k in ("episodes_checked", "bitwise_equal", "within_1e-4_rel", "argmin_flips") if k in par}


def compact_roofline(rf):
    keep = ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms", "binding", "binding_frac",
            "binding_achieved", "binding_peak", "binding_unit")
    out = {k: rf[k] for k in keep if k in rf}
    out["kernel"] = rf.get("kernel_symbol") or rf.get("kernel")
    rp = rf.get("rocprof") or {}
    if rp.get("replayed"):                                   # the committed rocprofv3 summary this kernel_ms must agree with
        out["rocprof_avg_ms"] = rp.get("kernel_steady_avg_ms")
        out["rocprof_file"] = rp.get("profile")
    return out


def compact_block(b):
    """The six numbers of an extra block (+ its CMA-ES generation wall-clock and CPU figure where it has them)."""
    rf = b.get("roofline") or {}
    out = {"episodes": b.get("episodes_per_gpu", b.get("episodes", b.get("episodes_per_generation"))),
           "ms_per_step": b.get("ms_per_step"), "value": b.get("value"),
           "kernel_ms": rf.get("kernel_ms", b.get("kernel_ms")), "binding_frac": rf.get("binding_frac"),
           "parity_ok": parity_ok(b.get("parity"))}
    for k in ("cma_generation_ms", "eval_weights_ms", "world_step_ms", "runs", "launch_groups", "generation_ms_ratio_to_one_run"):
        if k in b:
            out[k] = b[k]
    if b.get("cpu_baseline"):
        out["cpu_value"] = b["cpu_baseline"]["value"]
    return {k: _sig(v) for k, v in out.items() if v is not None or k == "parity_ok"}


CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data")


def compact_line(out, detail_path):
    """The printed line from the full record `out`."""
    line = {k: out[k] for k in CONTRACT_KEYS}
    cfg = out["config"]
    line["config"] = {"workload": cfg["workload"], "episodes_per_generation": cfg["episodes_per_generation"],
                      "episodes_per_gpu": cfg["episodes_per_gpu"], "sharding": cfg["sharding"]}
    line["roofline"] = compact_roofline(out["roofline"])
    cb = out.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {"value": cb["value"], "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                                "sample": cb.get("sample_short") or cb.get("sample", "")[:120]}
    if out.get("parity"):
        line["parity"] = compact_parity(out["parity"])
    co = out.get("collective")
    if co:
        line["collective"] = ({"error": co["error"][:120]} if "error" in co else
                              {"backend": co["backend"], "ranks_seen": co["ranks_seen"],
                               "all_gather_us": _sig(co["all_gather_us"]), "in_timed_step": co.get("in_timed_step")})
    if "cma_generation_ms" in out:
        line["cma_generation_ms"] = out["cma_generation_ms"]
    if "generation_cost_checksum" in out:
        line["generation_cost_checksum"] = out["generation_cost_checksum"]
    for k, v in out.items():
        if isinstance(v, dict) and k not in line and k not in ("valu", "cma", "profiled", "predicted_strong_scaling") \
                and ("ms_per_step" in v or "kernel_ms" in v):
            line[k] = compact_block(v)
    pss = out.get("predicted_strong_scaling")
    if pss:                                                   # per config: [N, predicted ms, speed-up, efficiency] rows
        line["predicted_strong_scaling"] = {
            "measured_on": "every rank's block timed on ONE GPU + the one-rank all-gather; NOT an N-GPU run",
            **{c: [[r["n_gpus"], _sig(r["generation_ms"], 4), _sig(r["speedup"], 3), _sig(r["efficiency"], 3)]
                   for r in t["rows"]] for c, t in pss["configs"].items()}}
    line["detail"] = detail_path
    return line


def write_detail(out):
    """The full record, whole, beside bench.py (or at $OCD_BENCH_DETAIL); returns the path the line names (None if
    nowhere was writable: the line is still printed)."""
    want = os.environ.get("OCD_BENCH_DETAIL") or os.path.join(ROOT, DETAIL_NAME)
    for path in (want, os.path.join(os.environ.get("TMPDIR", "/tmp"), DETAIL_NAME)):
        try:
            tmp = f"{path}.tmp.{os.getpid()}"
            with open(tmp, "w") as f:
                json.dump(out, f, indent=1)
            os.replace(tmp, path)
            return os.path.relpath(path, ROOT) if path.startswith(ROOT + os.sep) else path
        except OSError:
            continue
    return None


def print_line(out, line_out):
    line = compact_line(out, write_detail(out))
    text = json.dumps(line, separators=(",", ":"))
    if len(text) >= LINE_CAP:
        raise SystemExit(f"bench.py: the line is {len(text)} bytes (cap {LINE_CAP}); move keys to {DETAIL_NAME}")
    print(text, file=line_out, flush=True)


def spawn_ranks(args):
    """--gpus N without a launcher: start N ranks as a torch.distributed.run child.  Nothing in this
    process has touched the GPU; it only waits for the child and passes its exit code on."""
    port = args.master_port or (29000 + os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


def claim_stdout():
    """stdout carries ONE line.  Everything else that anything writes to file descriptor 1 -- RCCL prints a version banner
    there when its first communicator comes up -- goes to stderr from here on; the returned file is the real stdout."""
    sys.stdout.flush()
    line_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    return line_out


def plumbing_check(args, world, rank, line_out):
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
        print(json.dumps({"plumbing": True, "n_gpus": world, "torch": torch.__version__, "scaling": args.scaling,
                          "emulate_rank": args.emulate_rank or None}), file=line_out, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="3",
                    help="headline workload: BASELINE.json config index 2..5, or reference_h5 / reference_h6_extra "
                         "(the reference's own shapes; profiling runs)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: the population grows with the ranks (config's pop per GPU); strong: the config's "
                         "population is split over the ranks (BASELINE configs 4 / 5 on 8 GPUs)")
    ap.add_argument("--emulate-rank", default="", metavar="R/W",
                    help="single GPU: run the block that rank R of a W-way strong split would run")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the untimed oracle parity samples of the BASELINE blocks")
    ap.add_argument("--parity-episodes", type=int, default=256,
                    help="episodes of every BASELINE-config block checked against the CPU oracle, over all ranks")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the config-2 block, the config-4/5 share blocks and the CMA-ES generation timing")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (default); gloo only to rehearse N>1 on a one-GPU box")
    ap.add_argument("--master-port", type=int, default=0)
    ap.add_argument("--force-collective", action="store_true",
                    help="N=1: create a one-rank RCCL group and run the all-gather of the returns inside every timed "
                         "step, exactly as N ranks do")
    ap.add_argument("--python-steps", action="store_true",
                    help="one process: drive the timed steps from Python (one C-ABI call, one event wait, one native "
                         "reduction per step) instead of the native step loop")
    ap.add_argument("--plumbing-check", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    args.config = int(args.config) if args.config.isdigit() else args.config

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: world size {world} (WORLD_SIZE) != --gpus {args.gpus}")
    emulate = parse_emulate(args.emulate_rank) if args.emulate_rank else None
    if emulate and world != 1:
        raise SystemExit("--emulate-rank runs on one GPU (--gpus 1)")
    line_out = claim_stdout()
    if args.plumbing_check:
        return plumbing_check(args, world, rank, line_out)

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
    def init_group():
        """The process group of this run; a one-rank run rendezvous with itself on 127.0.0.1."""
        kw = {}
        if "MASTER_ADDR" not in os.environ or "WORLD_SIZE" not in os.environ:
            port = args.master_port or (31000 + os.getpid() % 2000)
            import datetime
            kw = dict(init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                      timeout=datetime.timedelta(seconds=120))       # a stuck rendezvous must raise, not hang the line
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(device), **kw)
        else:
            dist.init_process_group("gloo", **kw)
        assert dist.get_world_size() == args.gpus

    if world > 1 or args.force_collective:
        init_group()

    # the reference's own shapes (run_mpc_ord.py:19-44 with --n_inits 3; naive_planner.py:20,107-116; mpc_ord.py:192;
    # pycma's default population for 7 weights is 9): not BASELINE configs, measured beside them
    REFERENCE_SHAPES = {
        "reference_h5": dict(scenario="finite_horizon", horizon=5, pop=9, n_inits=3, kwargs={}, seed=11,
                             label="the reference's shape (run_mpc_ord.py finite_horizon cmaes --n_inits 3)"),
        "reference_h6_extra": dict(scenario="finite_horizon", horizon=6, pop=9, n_inits=3, kwargs={"extra_inits": True},
                                   seed=12, label="the reference's validation planner (H=6 -> n_iter 200, extra_inits)"),
    }

    def workload(cfg_index, P):
        if cfg_index in REFERENCE_SHAPES:
            cfg = REFERENCE_SHAPES[cfg_index]
            scn = scenarios.SCENARIOS[cfg["scenario"]](horizon=cfg["horizon"], **cfg["kwargs"])
            seed = cfg["seed"]
        else:
            cfg = scenarios.BASELINE_CONFIGS[cfg_index]
            scn = scenarios.SCENARIOS[cfg["scenario"]](horizon=cfg["horizon"])
            seed = cfg_index
        N, S = cfg["n_inits"], scn.desc.n_samples
        inits = scn.init_dist.sample(N, seed=1000 + seed)
        cands = scn.candidate_weights(P, seed=2000 + seed)
        w32 = np.stack([scenarios.planner_weights_fp32(c) for c in cands])
        return cfg, scn, inits, w32, P, N, S

    def time_all_gather(ret_dev, P, N, S, reps=20):
        """Mean microseconds of the all-gather of the returns alone (device memory in, device memory out) and the
        ranks the group saw.  Every rank calls it at the same point."""
        for _ in range(3):
            sharding.gather_returns(ret_dev, P, N, S, force=True)
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            sharding.gather_returns(ret_dev, P, N, S, force=True)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / reps * 1e6
        return {"backend": ("nccl (RCCL)" if dist.get_backend() == "nccl" else dist.get_backend()),
                "ranks_seen": dist.get_world_size(), "all_gather_us": us,
                "bytes_per_rank": int(ret_dev.numel() * 4), "calls_timed": reps}

    def timed_generations(cfg_index, P, ranks, rnk, steps, warmup, collective=True):
        """(seconds for `steps` generations, kernel ms, fitness, context) of one workload: the population P split
        into `ranks` candidate blocks, this process running block `rnk` (collective=False: the other blocks are
        nobody's -- rank emulation on one GPU -- and the fitness covers this block's candidates only)."""
        cfg, scn, inits, w32, P, N, S = workload(cfg_index, P)
        eng = Engine(scn, device)
        init_dev = torch.as_tensor(inits, dtype=torch.float32).to(device)
        w_dev = torch.as_tensor(w32).to(device)                 # inputs resident in HBM before timing
        e0, e1 = sharding.episode_range(P, N, S, ranks, rnk)
        ret_dev = torch.empty(e1 - e0, dtype=torch.float32, device=device)
        sharded = (ranks > 1 or args.force_collective) and collective and dist.is_initialized()
        lo, hi = sharding.candidate_block(P, ranks, rnk)
        host = torch.empty(P * N * S, dtype=torch.float32).pin_memory()     # one pinned buffer, one event per step
        host_np = host.numpy()
        done = torch.cuda.Event()

        out_ptr = ret_dev.data_ptr() if sharded else host.data_ptr()   # one process: the kernel writes its returns
        n_out = e1 - e0                                                #   straight into the pinned host buffer

        # the host side of a step as the product runs it (MPC_ORD._returns / eval_population): one C-ABI call on the
        # current stream, one event wait, the native float64 reduction (csrc/ocd_cma.c, bit for bit
        # sharding.fitness_from_returns)
        from l4dc_mpc_ocd_amd import abi
        from l4dc_mpc_ocd_amd.interact_drive.reward_design.cmaes import fitness_from_returns_native
        rollout_fn, handle = eng.lib.ocd_rollout_episodes, eng._h
        init_ptr, w_ptr, stream_ptr = init_dev.data_ptr(), w_dev.data_ptr(), eng._stream()
        cost_buf = np.empty(hi - lo if not sharded else P, dtype=np.float64)

        def generation():
            abi.check(eng.lib, rollout_fn(handle, init_ptr, w_ptr, P, N, e0, e1, out_ptr, None, None, stream_ptr))
            if not sharded:
                done.record()
                done.synchronize()
                return fitness_from_returns_native(host_np[:n_out], hi - lo, N, S, out=cost_buf)
            src = sharding.gather_returns(ret_dev, P, N, S, force=args.force_collective)
            if src.is_cuda:
                host.copy_(src, non_blocking=True)
                done.record()
                done.synchronize()
                arr = host_np
            else:                                                   # gloo rehearsal: gathered on the host already
                arr = src.numpy()
            return sharding.fitness_from_returns(arr, P, N, S)

        # bring the GPU to its sustained clocks first (untimed; a cold process sees the same kernel take 1.73 ms and
        # ten generations later 1.61 ms): >= 40 ms of back-to-back launches, then the caller's W warm-up steps
        one = eng.time_rollout(init_dev, w_dev, e0, e1, ret_dev, reps=1)
        eng.time_rollout(init_dev, w_dev, e0, e1, ret_dev, reps=int(min(64, max(3, 40.0 / max(one, 1e-3)))))
        for _ in range(warmup):
            generation()
        if sharded:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if sharded or args.python_steps or e0 != 0:          # (a rank's block of a larger population: Python steps)
            for _ in range(steps):
                fit = generation()
        else:
            # one process: the K steps run as the product runs a generation's fitness inside its native CMA-ES loop
            # (csrc/ocd_cma.c: ocd_eval_generations -- the same launch, stream wait and float64 reduction through the
            # same entry points, without the interpreter between steps)
            import ctypes as C
            from l4dc_mpc_ocd_amd.interact_drive.reward_design.cmaes import RunArgs, load_cma_library
            ra = RunArgs()
            ra.scn, ra.init_dev, ra.N, ra.S = handle.value, init_ptr, N, S
            ra.w_pinned, ra.ret_pinned, ra.stream = w_ptr, out_ptr, stream_ptr
            ra.rollout = C.cast(eng.lib.ocd_rollout_episodes, C.c_void_p).value
            ra.sync = C.cast(eng.lib.ocd_stream_synchronize, C.c_void_p).value
            st_ = load_cma_library().ocd_eval_generations(C.byref(ra), hi - lo, steps, cost_buf.ctypes.data, None)
            if st_ != 0:
                raise SystemExit(f"ocd_eval_generations -> {st_}: {eng.lib.ocd_last_error().decode()}")
            fit = cost_buf
        torch.cuda.synchronize()
        if sharded:
            dist.barrier()
        dt = time.perf_counter() - t0
        if sharded:
            tt = torch.tensor([dt], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        # dominant kernel: average launch duration from HIP events on the launch stream, writing where the timed
        # generations write (one process: the pinned host buffer the kernel stores its returns into directly;
        # sharded: the device buffer the all-gather reads) -- kernel_ms and ms_per_step describe the same launch
        # (the MEDIAN of single-launch timings: one launch on a clock transition is +10 % and would move a mean over six)
        kt = [eng.time_rollout(init_dev, w_dev, e0, e1, ret_dev if sharded else host[:n_out], reps=1)
              for _ in range(max(7, min(steps, 21)))]
        kern_ms = float(np.median(kt))
        launch = dict(eng.last_launch(), returns_written_to="device memory" if sharded else "pinned host memory (zero-copy)")
        coll = None
        if sharded:                                                # the generation's one collective by itself
            coll = time_all_gather(ret_dev, P, N, S)
        if sharded:                                                # every rank's kernel time, in rank order
            got = [None] * dist.get_world_size()
            dist.all_gather_object(got, float(kern_ms))
            launch["kernel_ms_per_rank"] = got
        return dt, kern_ms, fit, (cfg, scn, inits, w32, P, N, S, e1 - e0, launch, coll, eng, e0, sharded)

    def parity_of(ctx):
        """The `parity` object of a block (None with --no-parity): this rank checks its own slice, counts are summed
        over the ranks when the block was sharded."""
        if args.no_parity:
            return None
        scn, inits, w32, eng, e0, sharded = ctx[1], ctx[2], ctx[3], ctx[10], ctx[11], ctx[12]
        ranks = dist.get_world_size() if sharded else 1
        want = max(32, -(-args.parity_episodes // ranks))
        threads = max(1, min(usable_cores(), int(os.environ.get("OCD_CPU_THREADS", "64"))) // ranks)
        t0 = time.perf_counter()
        counts = parity_sample(scn, eng, inits, w32, e0, e0 + ctx[7], want, threads)
        if sharded:
            tt = torch.as_tensor(counts, device=device if args.backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.SUM)
            counts = tt.cpu().numpy()
        out = {k: int(v) for k, v in zip(PARITY_KEYS, counts)}
        out["against"] = ("oracle/ocd_oracle.c (CPU restatement, the checker), untimed: returns + every world state + every "
                          "applied control bit for bit; argmin_flips = control steps whose kept control initialisation "
                          "differs (ocd_plan_batch at the episode's own world states)")
        out["ranks"] = ranks
        out["seconds"] = time.perf_counter() - t0
        return out

    def share_block(cfg_index, r, w, steps, warmup, light=False):
        """Rank r's block of a w-way strong split of BASELINE config cfg_index, on this GPU alone (light: the timing
        only, no parity sample and no CPU figure -- the rows of `predicted_strong_scaling`)."""
        P = scenarios.BASELINE_CONFIGS[cfg_index]["pop"]
        dt_, k_, fit_, ctx_ = timed_generations(cfg_index, P, w, r, steps, warmup, collective=False)
        if light:
            return {"emulated_rank": f"{r}/{w}", "episodes_per_gpu": ctx_[7], "ms_per_step": dt_ / steps * 1e3,
                    "kernel_ms": k_, "generation_cost_checksum": float(np.sum(fit_))}
        b = block(cfg_index, dt_, k_, ctx_, steps, per_gpu_only=True)
        b["parity"] = parity_of(ctx_)
        if w > 1:
            b["emulated_rank"] = f"{r}/{w}"
        else:
            b["sharding"] = "the whole population on this one GPU (no split, no collective)"
        b["generation_cost_checksum"] = float(np.sum(fit_))
        if not args.no_cpu_baseline:                                # this rank's candidate block on the host's cores
            lo_, hi_ = sharding.candidate_block(P, w, r)
            b["cpu_baseline"] = cpu_baseline(ctx_[1], ctx_[2], ctx_[3][lo_:hi_], budget_s=2.0)
        return b

    def cma_generations(cfg, popsize, reduce_over_ranks, gens=48):
        """ask -> host normalisation -> launch -> (gather) -> returns in pinned host memory -> float64 reduction ->
        tell, through MPC_ORD.optimize_cmaes (mpc_ord.py:33-45); every rank runs the same deterministic strategy.
        The first generations run on a GPU whose clocks are still rising (the kernel itself takes 1.73 -> 1.61 ms
        over the first ten generations of a cold process): the last 32 of `gens` are timed, like kernel_ms after its
        warm-up."""
        from l4dc_mpc_ocd_amd.interact_drive.experiments import run_mpc_ord as rmo
        m = rmo.make_mpc_ord(cfg["scenario"], horizon=cfg["horizon"], n_inits=cfg["n_inits"], seed=1,
                             **cfg.get("kwargs", {}))
        # exactly `gens` generations: the step-size rules that would end this run earlier are switched off (the cost
        # is invariant to the scale of the weights, so sigma drifts upwards: pycma's tolfacupx fires after ~38
        # generations at pop 64)
        m.native_chunk = 16                                     # (the timed 32 generations = two whole native calls)
        m.optimize_cmaes(seed=1, sigma0=0.05, popsize=popsize, maxiter=gens,
                         termination={"tolfacupx": float("inf"), "tolupsigma": float("inf")})
        # wall-clock per generation INCLUDING the interpreter's bookkeeping between native calls (history rows,
        # counters): ADVICE round 4 -- the C timers (ask ... termination test) alone are reported beside it
        gs = np.array(m.generation_wall_seconds[-32:]) * 1e3
        gs_native = np.array(m.generation_seconds[-32:]) * 1e3
        fs = np.array(m.fitness_seconds[-32:]) * 1e3            # eval_population alone (no ask / tell)
        med = float(np.median(gs))
        if reduce_over_ranks and world > 1:
            tt = torch.tensor([med], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            med = float(tt.item())
        return {"cma_generation_ms": med, "cma_generation_native_timers_ms": float(np.median(gs_native)),
                "fitness_ms": float(np.median(fs)),
                "sampler_parity": "random stream unpinned (own CMA-ES, pycma is not installed); strategy parameters and the "
                                  "active update pinned to arXiv 1604.00772 Table 1 / eqs. 46-58 "
                                  "(tests/test_cma_paper_constants.py); the fitness values it is fed are bit-exact",
                "generations_run": int(len(m.generation_seconds)), "generations_timed": int(len(gs)),
                "popsize": int(m.es.lam), "n_inits": cfg["n_inits"], "stop_reason": {k: (None if v is None else float(v)) for k, v in m.stop_reason.items()},
                "n_nonfinite": int(sum(m.n_nonfinite)), "n_resampled": int(m.n_resampled),
                "host_split_ms": m.host_split_ms(),
                "path": "MPC_ORD.optimize_cmaes: ask (native, csrc/ocd_cma.c), host normalisation into pinned memory "
                        "the kernel reads, launch, (gather,) returns written to / copied into pinned host memory, "
                        "float64 reduction (native), NaN costs redrawn as pycma does, tell (native), pycma's "
                        "termination rules; own CMA-ES, pycma is not installed"}

    def lockstep_block(name, spec, R=28, gens=48):
        """R independent CMA-ES runs of a reference shape in LOCKSTEP, one indexed launch per generation
        (MPC_ORD.optimize_cmaes_many; the reference's Pool over init groups, run_mpc_ord.py:83-90 -- 28 = the chosen
        weight vectors of generalization_data.py:78-84): generation wall-clock for all R runs together."""
        from l4dc_mpc_ocd_amd.interact_drive.experiments import run_mpc_ord as rmo
        m = rmo.make_mpc_ord(spec["scenario"], horizon=spec["horizon"], n_inits=spec["n_inits"], seed=1, **spec.get("kwargs", {}))
        scn = scenarios.SCENARIOS[spec["scenario"]](horizon=spec["horizon"], **spec.get("kwargs", {}))
        runs = [(list(scn.init_dist.sample(spec["n_inits"], seed=300 + r)), 1 + r, 0.05) for r in range(R)]
        res = m.optimize_cmaes_many(runs, maxiter=gens, termination={"tolfacupx": float("inf"), "tolupsigma": float("inf")})
        wall = np.array(res.generation_wall_seconds[-32:]) * 1e3       # (N > 1: the slowest rank's)
        nat = np.array(res.generation_seconds[-32:]) * 1e3
        E = int(res.episodes_per_generation[-1])
        if world > 1:
            # the runs were dealt over the ranks (run r on rank r mod N), each rank in lockstep on its own GPU, one
            # all_gather_object at the end: the reference's Pool of processes as one process per GPU
            E_all = sum(int(st["episodes_per_generation"][-1]) for st in res.per_rank if st)
            return {"workload": f"{R} independent CMA-ES runs of {spec['label']} dealt over {world} ranks, each rank's runs in "
                                f"lockstep: one launch per generation and rank",
                    "runs": R, "ranks": world, "episodes_per_generation": E_all, "lockstep": bool(res.lockstep),
                    "episodes_per_generation_per_rank": [int(st["episodes_per_generation"][-1]) if st else 0 for st in res.per_rank],
                    "cma_generation_ms": float(np.median(wall)), "cma_generation_native_timers_ms": float(np.median(nat)),
                    "cma_generation_ms_per_rank": [float(np.median(np.array(st["generation_wall_seconds"][-32:]) * 1e3)) if st else None
                                                   for st in res.per_rank],
                    "value": E_all / (float(np.median(wall)) * 1e-3), "unit": "episodes/s",
                    "generations_run": len(res.generation_seconds), "launch": res.launch,
                    "stop_reason": sorted({k for o in res.runs for k in o.stop_reason}),
                    "collective": "none while the runs advance; one all_gather_object of the histories at the end",
                    "sampler_parity": "unpinned (own CMA-ES, pycma absent); every run's history is bit for bit the run alone"}
        # the launch by itself: HIP events on the launch stream (torch's current stream is the one the launches go to)
        eng = m._engine()
        pop = res.runs[0].es.lam
        rows, p0, n0 = [], 0, 0
        for init_states, _, _ in runs:
            N_r = len(init_states)
            rows += [(p0 + e // (N_r * scn.desc.n_samples), n0 + (e // scn.desc.n_samples) % N_r, e) for e in range(pop * N_r * scn.desc.n_samples)]
            p0, n0 = p0 + pop, n0 + N_r
        idx = torch.as_tensor(np.asarray(rows, dtype=np.int32)).to(device)
        init_all = np.concatenate([np.asarray(r[0], dtype=np.float32).reshape(-1, 4) for r in runs])
        w_all = np.concatenate([np.stack([scenarios.planner_weights_fp32(c) for c in scn.candidate_weights(pop, seed=400 + r)]) for r in range(R)])
        init_dev, w_dev = torch.as_tensor(init_all).to(device), torch.as_tensor(w_all).to(device)
        ret = torch.empty(len(rows), dtype=torch.float32, device=device)
        from l4dc_mpc_ocd_amd import abi

        def launch():
            abi.check(eng.lib, eng.lib.ocd_rollout_indexed(eng._h, init_dev.data_ptr(), init_dev.shape[0], w_dev.data_ptr(), w_dev.shape[0],
                                                           idx.data_ptr(), len(rows), ret.data_ptr(), None, None, eng._stream()))
        for _ in range(5):
            launch()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(10):
            launch()
        ev1.record()
        ev1.synchronize()
        out = {"workload": f"{R} independent CMA-ES runs of {spec['label']} in lockstep: one launch of {E} episodes per generation",
               "runs": R, "episodes_per_generation": E, "lockstep": bool(res.lockstep), "launch_groups": int(res.groups),
               "host_threads": int(res.host_threads), "cma_generation_ms": float(np.median(wall)), "cma_generation_native_timers_ms": float(np.median(nat)),
               "kernel_ms": ev0.elapsed_time(ev1) / 10, "launch": eng.last_launch(), "generations_run": len(res.generation_seconds),
               "generations_timed": int(len(wall)), "host_split_ms": res.host_split_ms(),
               "value": E / (float(np.median(wall)) * 1e-3), "unit": "episodes/s",
               "stop_reason": sorted({k for o in res.runs for k in o.stop_reason}),
               "sampler_parity": "unpinned (own CMA-ES, pycma absent); every run's history is bit for bit the run alone (tests/test_gpu_lockstep.py)"}
        if not args.no_parity:                                      # the indexed launch's returns against the oracle, run by run
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib
            orc = oracle_lib.load()
            got = ret.cpu().numpy()
            want = np.concatenate([orc.rollout(scn.desc, runs[r][0], w_all[r * pop:(r + 1) * pop],
                                               n_threads=usable_cores())["returns"].reshape(-1) for r in range(R)])
            with np.errstate(invalid="ignore"):
                close = np.abs(got.astype(np.float64) - want) <= 1e-4 * np.maximum(1e-2, np.abs(want))
            out["parity"] = {"episodes_checked": int(got.size), "bitwise_equal": int(_same_bits(got, want).sum()),
                             "within_1e-4_rel": int(close.sum()),
                             "against": "oracle/ocd_oracle.c, untimed: the returns of one indexed launch over all runs"}
        if not args.no_cpu_baseline:                                # the same R populations on the host's cores
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib
            orc = oracle_lib.load()
            cores = min(usable_cores(), int(os.environ.get("OCD_CPU_THREADS", "64")))
            t0, n_ep, reps = time.perf_counter(), 0, 0
            while time.perf_counter() - t0 < 2.0:
                for r in range(R):
                    orc.rollout(scn.desc, runs[r][0], w_all[r * pop:(r + 1) * pop], n_threads=cores)
                n_ep += E
                reps += 1
            dt_c = time.perf_counter() - t0
            out["cpu_baseline"] = {"value": n_ep / dt_c, "unit": "episodes/s", "cores": cores, "kind": "port",
                                   "sample": f"{reps} x the {R} populations ({E} episodes), one oracle call per run with OpenMP over its "
                                             f"episodes on {cores} threads, {dt_c:.1f} s of CPU work (oracle/ocd_oracle.c)"}
        return out

    def predicted_strong_scaling(whole, gather_us, steps=3):
        """What the ONE-GPU measurements predict for BASELINE configs 4 / 5 split over N = 1, 2, 4, 8 GPUs (strong scaling:
        run_mpc_ord.py:83-90, mpc_ord.py:128-137 split a fixed batch): for every N, EVERY rank's candidate block run on
        this GPU alone (the step as a rank runs it: launch, wait, float64 reduction); the generation takes the slowest
        block + the all-gather of the returns (measured here on a one-rank RCCL group: the N-rank gather is not
        measured).  Unmeasured on N GPUs -- a prediction, flagged as one."""
        table = {}
        for c in (4, 5):
            rows = []
            base = whole[c]["ms_per_step"]
            for n in (1, 2, 4, 8):
                if n == 1:
                    per_rank = [base]
                else:
                    per_rank = [share_block(c, r, n, steps, 1, light=True)["ms_per_step"] for r in range(n)]
                gen = max(per_rank) + (gather_us * 1e-3 if n > 1 else 0.0)
                rows.append({"n_gpus": n, "generation_ms": gen, "slowest_block_ms": max(per_rank),
                             "fastest_block_ms": min(per_rank), "block_ms_per_rank": per_rank,
                             "speedup": base / gen, "efficiency": base / gen / n})
            table[f"config{c}"] = {"episodes_per_generation": whole[c]["episodes_per_generation"], "rows": rows}
        return {"status": "PREDICTED from one-GPU runs of every rank's block; unmeasured on N GPUs",
                "all_gather_us_assumed": gather_us, "configs": table}

    def config1_block(reps=40):
        """BASELINE config 1 -- finite_horizon, 3 inits, the designer's ("true") weights, H = 5: the README's `vis` path
        (run_mpc_ord.py:107-119: MPC_ORD.eval_weights(designer_weights)) through the SCALAR drop-in API, and the
        object-by-object path under it (world.reset(); 15 x world.step(): PlannerCar._get_next_control ->
        NaivePlanner.generate_plan -> Car.step, world.py:79-109, planner_car.py:54-85) -- plumbing latency, not
        throughput; the CPU oracle on the same three episodes beside it."""
        from l4dc_mpc_ocd_amd.interact_drive.reward_design.mpc_ord import MPC_ORD, finite_horizon_env
        car, world, init_states = finite_horizon_env(horizon=5, env_seeds=[1000001, 1000002, 1000003])
        m = MPC_ORD(world, car, init_states, 15)
        cost = m.eval_weights(m.designer_weights)                  # (first call: handle, buffers)
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            c2 = m.eval_weights(m.designer_weights)
            ts.append(time.perf_counter() - t0)
        assert c2 == cost
        # object by object: one episode = reset + 15 world.step() calls, each a plan launch + the real dynamics of every car
        step_ts, ret_obj = [], []
        w_designer = m.designer_weights
        wn = w_designer / np.linalg.norm(w_designer)                # mpc_ord.py:120,71, then the car's setter: the
        car.weights = wn / np.linalg.norm(wn)                      #   planner weights eval_weights would install
        for init in init_states:
            car.init_state = type(car.state)(init)
            world.reset()
            G = np.float32(0)
            for _ in range(15):
                t0 = time.perf_counter()
                past, controls, _state = world.step()
                step_ts.append(time.perf_counter() - t0)
                G = np.float32(G + car.reward_fn(past, controls[car.index], weights=w_designer))
            ret_obj.append(float(G))
        scn = scenarios.finite_horizon(horizon=5)
        eng = Engine(scn, device)
        inits = np.asarray(init_states, dtype=np.float32)
        w32 = scenarios.planner_weights_fp32(np.asarray(m.designer_weights, dtype=np.float64))[None]
        init_dev, w_dev = torch.as_tensor(inits).to(device), torch.as_tensor(w32).to(device)
        ret_dev = torch.empty(3, dtype=torch.float32, device=device)
        kern = eng.time_rollout(init_dev, w_dev, 0, 3, ret_dev, reps=10)
        out = {"workload": "BASELINE config 1: finite_horizon, 3 inits, designer weights, H=5, n_iter=100, K=3, T=15 -- "
                           "MPC_ORD.eval_weights through the scalar drop-in API, and world.step() object by object",
               "episodes": 3, "eval_weights_ms": float(np.median(ts) * 1e3), "kernel_ms": kern,
               "world_step_ms": float(np.median(step_ts) * 1e3), "world_steps_timed": len(step_ts),
               "episode_via_world_step_ms": float(np.sum(step_ts) / 3 * 1e3), "launch": eng.last_launch(),
               "cost": float(cost)}
        if not args.no_parity or not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib
            orc = oracle_lib.load()
            t0 = time.perf_counter()
            n_rep = 0
            while time.perf_counter() - t0 < 1.0:
                ref = orc.rollout(scn.desc, inits, w32, n_threads=1)["returns"]
                n_rep += 1
            dt_c = time.perf_counter() - t0
            out["cpu_baseline"] = {"value": 3 * n_rep / dt_c, "unit": "episodes/s", "cores": 1, "kind": "port",
                                   "ms_per_eval_weights": dt_c / n_rep * 1e3,
                                   "sample": f"{n_rep} x the 3 episodes, one thread (oracle/ocd_oracle.c)"}
            want = float(sharding.fitness_from_returns(ref, 1, 3, 1)[0])
            out["parity"] = {"episodes_checked": 3, "eval_weights_cost_bitwise_equal": bool(want == cost),
                             "world_step_returns_bitwise_equal": bool(np.array_equal(np.asarray(ret_obj, dtype=np.float32), ref)),
                             "against": "oracle/ocd_oracle.c, untimed"}
        return out

    def block(cfg_index, dt, kern_ms, ctx, steps, per_gpu_only=False):
        cfg, scn, inits, w32, P, N, S, n_local, launch, _coll = ctx[:10]
        kernel_name = ("ocd::mpc_chunk_kernel" if launch["scan_mode"] == 4 else "ocd::mpc_kernel") + \
            f" ({launch['mapping']}" + (f", {launch['chunk']} steps per lane" if launch["chunk"] else "") + \
            f", {launch['trajectories_per_wavefront']} trajectories per wavefront, {launch['workgroups']} workgroups x " \
            f"{launch['wavefronts_per_workgroup']} wavefronts" + \
            (", latency build" if launch["build_wavefronts_per_simd"] == 1 else "") + ")"
        n_done = n_local if per_gpu_only else P * N * S           # episodes this timing covers per step
        d = scn.desc
        nbytes, flops = algorithmic_per_episode(d)
        ach_gbs = n_local * nbytes / (kern_ms * 1e-3) / 1e9
        ach_tf = n_local * flops / (kern_ms * 1e-3) / 1e12
        n_split = max(1, round(P * N * S / max(n_local, 1)))
        from l4dc_mpc_ocd_amd import abi as _abi
        mangled = _abi.planner_kernel_name(d, launch)
        pmc, pmc_why = pmc_record(f"cfg{cfg_index}_share{n_split}" if per_gpu_only else
                                  (cfg_index if isinstance(cfg_index, str) else f"cfg{cfg_index}"), n_local, mangled)
        traffic = (pmc["fetch_kib"] + pmc["write_kib"]) * 1024.0 if pmc else None
        rocprof = {"replayed": False, "why": pmc_why}
        if pmc:
            rocprof = {"replayed": True, "profile": pmc.get("profile"), "git_commit": pmc["git_commit"],
                       "kernel_source_sha": pmc["kernel_source_sha"], "kernel_name": pmc["kernel_name"],
                       "kernel_avg_ms": pmc["kernel_avg_us"] / 1e3, "kernel_calls": pmc["kernel_calls"],
                       "kernel_steady_avg_ms": (pmc.get("kernel_steady_avg_us") or pmc["kernel_avg_us"]) / 1e3,
                       "rule": "the profile's steady-state kernel average (later half of its dispatches) must not exceed this "
                               "line's ms_per_step beyond box-to-box variation (3 %): tests/test_gpu_bench_contract.py"}
        profiled = None
        valu = {"achieved": ach_tf, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach_tf / VALU_PEAK_TFLOPS,
                "algorithmic_flops_per_episode": flops,
                "note": "at one wavefront per SIMD (2 048 episodes at H=10) a lone wavefront issues one instruction per "
                        "~4.4 cycles: one instruction per algorithmic flop would give ~21 % of the vector peak at this batch size (DESIGN.md section 4)"}
        if pmc:
            # counters REPLAYED from the committed profile of this same command, not measured in this run
            profiled = {"source": PMC_SOURCE, "hbm_bytes_per_launch": traffic}
            if pmc.get("sq_wave_cycles"):
                # quad-cycles in which a wavefront issued a VALU instruction / quad-cycles wavefronts were resident
                profiled["valu_issue_utilisation"] = pmc["sq_active_inst_valu"] / pmc["sq_wave_cycles"]
                profiled["wait_fraction"] = pmc["sq_wait_any"] / pmc["sq_wave_cycles"]
                profiled["valu_instructions_per_launch"] = pmc["sq_insts_valu"]
        pop_text = (f"pop {P} split over {n_split} ranks" if n_split > 1 else f"pop {P}")
        what = cfg["label"] if isinstance(cfg_index, str) else f"BASELINE config {cfg_index}"
        out = {
            "workload": f"{what}: {cfg['scenario']}, CMA-ES {pop_text} x {N} inits x "
                        f"{S} samples, planning horizon H={d.horizon}, n_iter={d.n_iter}, K={d.n_ctrl_inits} control "
                        f"inits, episode length T={d.episode_len}",
            "episodes_per_generation": P * N * S, "episodes_per_gpu": n_local,
            "value": n_done * steps / dt, "unit": "episodes/s", "ms_per_step": dt / steps * 1e3,
            "roofline": {"bound": "hbm", "achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach_gbs / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": PMC_SOURCE if pmc else None,
                         "kernel": kernel_name, "kernel_symbol": mangled, "kernel_ms": kern_ms, "launch": launch,
                         "rocprof": rocprof,
                         "algorithmic_bytes_per_episode": nbytes,
                         # what actually bounds the kernel: fp32 vector issue -- algorithmic flops per
                         # launch / kernel_ms against the fp32 vector peak (the same numbers as `valu`)
                         "binding": "valu", "binding_frac": ach_tf / VALU_PEAK_TFLOPS,
                         "binding_achieved": ach_tf, "binding_peak": VALU_PEAK_TFLOPS, "binding_unit": "TFLOP/s",
                         "note": "the contract's hbm figures are kept as measured; the path is fp32-VALU issue bound, "
                                 "not HBM bound (SURVEY.md 8d): `binding` / `binding_frac` name the bound that binds"},
            "valu": valu,
        }
        if profiled:
            out["profiled"] = profiled
        return out

    # ---- headline: the chosen config over the ranks (weak: pop per GPU; strong: pop split) ----
    def spec_of(cfg_index):
        if cfg_index in REFERENCE_SHAPES:
            return REFERENCE_SHAPES[cfg_index]
        if cfg_index not in scenarios.BASELINE_CONFIGS or cfg_index == 1:
            raise SystemExit(f"bench.py: --config {cfg_index}: not one of 2..5, {', '.join(REFERENCE_SHAPES)}")
        return scenarios.BASELINE_CONFIGS[cfg_index]

    pop = spec_of(args.config)["pop"]
    if emulate:
        dt, kern_ms, fit, ctx = timed_generations(args.config, pop, emulate[1], emulate[0], args.steps, args.warmup,
                                                  collective=False)
    else:
        P_total = pop * world if args.scaling == "weak" else pop
        if P_total < world:
            raise SystemExit(f"bench.py: population {P_total} < {world} ranks")
        dt, kern_ms, fit, ctx = timed_generations(args.config, P_total, world, rank, args.steps, args.warmup)
    coll = ctx[9]
    head_parity = parity_of(ctx)

    # ---- extras: config 2 (small-batch latency) on rank 0, CMA-ES generation wall-clock on all ranks ----
    extra2 = None
    strong_blocks = {}
    shares = {}
    reference_blocks = {}
    cma = None
    if not args.no_extras:
        if rank == 0 and args.config != 2:
            dt2, k2, _, ctx2 = timed_generations(2, scenarios.BASELINE_CONFIGS[2]["pop"], 1, 0, args.steps, args.warmup)
            extra2 = block(2, dt2, k2, ctx2, args.steps)
            extra2["parity"] = parity_of(ctx2)
            if world == 1 and not args.no_cpu_baseline:
                extra2["cpu_baseline"] = cpu_baseline(ctx2[1], ctx2[2], ctx2[3], budget_s=2.0)
        if rank == 0 and world == 1 and not emulate:
            # what ONE of 8 GPUs runs of BASELINE configs 4 / 5 (strong split): the shapes the 8-GPU lines are made of
            n_sh = max(3, min(args.steps, 20))
            shares = {f"config{c}_share8": share_block(c, 0, 8, n_sh, 2) for c in (4, 5)}
            # ... and the two configs WHOLE on this one GPU (16 384 / 32 768 episodes per launch: three / eight wavefronts per
            # SIMD, the shared-SIMD builds of the chunked kernel with their work-item lists)
            shares.update({f"config{c}_whole": share_block(c, 0, 1, 3, 1) for c in (4, 5)})
        if world > 1 and not emulate:
            # BASELINE configs 4 / 5 as BASELINE.json states them: the fixed population split over the ranks (strong
            # scaling; mpc_ord.py:128-137, run_mpc_ord.py:83-90), the all-gather of the returns inside every timed step
            n_st = max(3, min(args.steps, 20))
            for c in (4, 5):
                if c == args.config and args.scaling == "strong":
                    continue                                       # (it is the headline of this run already)
                P_c = scenarios.BASELINE_CONFIGS[c]["pop"]
                if P_c < world:
                    continue
                dt_c, k_c, fit_c, ctx_c = timed_generations(c, P_c, world, rank, n_st, 2)
                par_c = parity_of(ctx_c)
                if rank == 0:
                    b = block(c, dt_c, k_c, ctx_c, n_st)
                    b.update(scaling="strong", steps=n_st, warmup=2, parity=par_c, collective=dict(ctx_c[9], in_timed_step=True),
                             generation_cost_checksum=float(np.sum(fit_c)),
                             sharding=f"population {P_c} split into {world} candidate blocks, one all_gather of fp32 "
                                      f"returns per generation inside the timed step, max-over-ranks wall-clock")
                    strong_blocks[f"config{c}"] = b
        if world > 1 and not emulate:
            lb = lockstep_block("reference_h5", REFERENCE_SHAPES["reference_h5"])   # every rank: its share of the 28 runs
            if rank == 0:
                reference_blocks["reference_h5_x28"] = lb
        cfg = spec_of(args.config)
        cma_pop = cfg["pop"] * world if args.scaling == "weak" else cfg["pop"]
        cma = cma_generations(cfg, cma_pop, reduce_over_ranks=True)
        if rank == 0 and world == 1 and not emulate:
            n_rf = max(3, min(args.steps, 20))
            for name, spec in REFERENCE_SHAPES.items():
                if name == args.config:
                    continue
                dt_r, k_r, fit_r, ctx_r = timed_generations(name, spec["pop"], 1, 0, n_rf, 2, collective=False)
                b = block(name, dt_r, k_r, ctx_r, n_rf)
                b["parity"] = parity_of(ctx_r)
                b["generation_cost_checksum"] = float(np.sum(fit_r))
                c = cma_generations(spec, None, reduce_over_ranks=False)      # popsize None: pycma's default 9
                b["cma_generation_ms"], b["cma"] = c["cma_generation_ms"], c
                if not args.no_cpu_baseline:
                    b["cpu_baseline"] = cpu_baseline(ctx_r[1], ctx_r[2], ctx_r[3], budget_s=2.0)
                reference_blocks[name] = b
            reference_blocks["config1"] = config1_block()
            # the reference's only parallel axis (a Pool over independent runs) as ONE launch per generation
            lb = lockstep_block("reference_h5", REFERENCE_SHAPES["reference_h5"])
            if "reference_h5" in reference_blocks:
                lb["one_run_generation_ms"] = reference_blocks["reference_h5"]["cma_generation_ms"]
                lb["generation_ms_ratio_to_one_run"] = lb["cma_generation_ms"] / lb["one_run_generation_ms"]
            reference_blocks["reference_h5_x28"] = lb

    if rank == 0:
        cfg, scn, inits, w32, P, N, S, n_local = ctx[:8]
        d = scn.desc
        hb = block(args.config, dt, kern_ms, ctx, args.steps, per_gpu_only=bool(emulate))
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
            "scaling": "strong" if (emulate or args.scaling == "strong") else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": hb["workload"], "episodes_per_generation": hb["episodes_per_generation"],
                       "episodes_per_gpu": hb["episodes_per_gpu"],
                       "sharding": (f"rank {emulate[0]} of a {emulate[1]}-way strong split, emulated on one GPU (no collective)"
                                    if emulate else
                                    f"candidate blocks over {world} rank(s); one all_gather of fp32 returns per generation")},
            "roofline": hb["roofline"],
            "valu": hb["valu"],
            "generation_cost_checksum": float(np.sum(fit)),
        }
        if head_parity is not None:
            out["parity"] = head_parity
        if cma:
            out["cma_generation_ms"] = cma["cma_generation_ms"]
            out["cma"] = cma
        if hb.get("profiled"):
            out["profiled"] = hb["profiled"]
        if extra2:
            out["config2"] = extra2
        out.update(shares)
        out.update(strong_blocks)
        out.update(reference_blocks)
        if world == 1 and not args.no_cpu_baseline and not emulate:
            out["cpu_baseline"] = cpu_baseline(scn, inits, w32)
    # ---- the generation's one collective, by itself: RCCL all-gather of the returns on device memory ----
    if coll is None and world == 1 and not emulate and not args.no_extras:
        # a one-GPU run has no collective in its timed step; run the N-rank code path once anyway (one-rank RCCL
        # group, sharding.gather_returns on device tensors) so that it has executed before any multi-GPU run
        try:
            if not dist.is_initialized():
                init_group()
            n_ret = ctx[7]
            coll = time_all_gather(torch.zeros(n_ret, dtype=torch.float32, device=device), ctx[4], ctx[5], ctx[6])
            coll["in_timed_step"] = False
        except Exception as exc:                                  # the headline must not depend on it
            coll = {"error": f"{type(exc).__name__}: {exc}"[:300], "in_timed_step": False}
    elif coll is not None:
        coll["in_timed_step"] = True
    if rank == 0:
        out["collective"] = coll
        if world == 1 and "config4_whole" in shares and "config5_whole" in shares:
            g_us = coll["all_gather_us"] if coll and "all_gather_us" in coll else 0.0
            out["predicted_strong_scaling"] = predicted_strong_scaling({4: shares["config4_whole"], 5: shares["config5_whole"]}, g_us)
        print_line(out, line_out)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
