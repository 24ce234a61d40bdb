#!/usr/bin/env python3
"""Condense rocprofv3 rocpd SQLite outputs (gpurun_out/...) into small text summaries for profiles/.

usage: rocprof_summary.py <results.db> [...]   -> prints per-kernel stats and per-kernel PMC averages
"""
import sqlite3
import sys


def summarize(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    print(f"## {path}")
    try:
        rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
        if rows:
            print("kernel-trace stats (durations in us):")
            print(f"{'calls':>6} {'total_us':>12} {'avg_us':>12} {'pct':>7}  name")
            for name, calls, tot, avg, pct in rows:
                print(f"{calls:>6} {tot:>12.3f} {avg:>12.3f} {pct:>7.2f}  {name}")
    except sqlite3.Error as e:
        print("no kernel stats:", e)
    try:
        rows = list(cur.execute(
            "select kernel_name, counter_name, count(*), avg(value), min(value), max(value), avg(duration), "
            "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_block_size), max(scratch_size), "
            "max(workgroup_size), max(grid_size) "
            "from counters_collection group by kernel_name, counter_name"))
        if rows:
            print("PMC counters (per dispatch):")
            for r in rows:
                print(f"  {r[0]}\n    {r[1]}: n={r[2]} avg={r[3]:.4f} min={r[4]:.4f} max={r[5]:.4f}  avg_dur_ns={r[6]:.0f} "
                      f"vgpr={r[7]} agpr={r[8]} sgpr={r[9]} lds={r[10]} scratch={r[11]} wg={r[12]} grid={r[13]}")
    except sqlite3.Error as e:
        print("no counters:", e)
    print()


def counters_json(round_name, out_dir):
    """profiles/pmc_counters.json: per config, the per-launch averages of the planner kernel's counters."""
    import glob
    import json
    import os
    import re
    rec = {"_source": f"profiles/{round_name}_cfg<N>_rocprofv3.txt: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_* "
                      "(separate passes, --kernel-trace only) of `bench.py --config N [--emulate-rank 0/8] --no-extras`, kernels ocd::mpc_kernel / ocd::mpc_chunk_kernel (cfgN_share8: rank 0's block of an 8-way split); "
                      "FETCH/WRITE in KiB per launch as counted (dword-granular accesses), SQ_* per launch (quad-cycles)"}
    # which tree the profiled kernels were built from: the commit (given by the caller: the GPU box holds no .git) and
    # the hash of the kernel sources, which bench.py recomputes before it replays any of these numbers
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from l4dc_mpc_ocd_amd import abi
    rec["_git_commit"] = os.environ.get("OCD_GIT_COMMIT", "unknown")
    rec["_kernel_source_sha"] = abi.kernel_source_sha()
    for d in sorted(glob.glob(os.path.join(out_dir, "c*_sq"))):
        cfg = re.search(r"c(\d+s?|reference_\w+?)_sq$", d).group(1)   # "4" = the whole config, "4s" = one of 8 GPUs' share
        vals = {}
        # THE planner kernel of this workload: the one the run spent most of its time in (a bench run may launch others --
        # e.g. the small plan / trajectory launches of a parity sample -- which must not be mixed into its averages)
        top = None
        for path in glob.glob(os.path.join(out_dir, f"c{cfg}_stats", "**", "*.db"), recursive=True):
            rows = list(sqlite3.connect(path).cursor().execute(
                "select name from top_kernels where name like '%mpc_%kernel%' order by total_duration desc limit 1"))
            top = rows[0][0] if rows else None
        for kind in ("fetch", "write", "sq", "stats"):
            for path in glob.glob(os.path.join(out_dir, f"c{cfg}_{kind}", "**", "*.db"), recursive=True):
                cur = sqlite3.connect(path).cursor()
                try:
                    for name, cnt, avg, grid, wg in cur.execute(
                            "select counter_name, count(*), avg(value), max(grid_size), max(workgroup_size) "
                            "from counters_collection where kernel_name = ? group by counter_name", (top,)):
                        vals[name] = avg
                        vals["grid_size"], vals["workgroup_size"] = grid, wg
                except sqlite3.Error:
                    pass
                if kind == "stats":
                    try:
                        for name, calls, avg in cur.execute("select name, total_calls, average from top_kernels where name = ?", (top,)):
                            vals["kernel_avg_us"], vals["kernel_calls"], vals["kernel_name"] = avg, calls, name
                        # steady state: the later half of the dispatches (the first launches of a cold process run on
                        # clocks that are still rising -- bench.py's own kernel_ms is taken after its warm-up as well)
                        durs = [r[0] for r in cur.execute("select duration from kernels where name = ? order by start", (top,))]
                        if durs:
                            tail = durs[len(durs) // 2:]
                            vals["kernel_steady_avg_us"] = sum(tail) / len(tail) / 1e3
                    except sqlite3.Error:
                        pass
        log = os.path.join(out_dir, f"c{cfg}_stats.log")
        eps = None
        if os.path.exists(log):
            m = re.search(r'"episodes_per_gpu":\s*(\d+)', open(log).read())
            eps = int(m.group(1)) if m else None
        key = cfg if cfg.startswith("reference_") else (f"cfg{cfg[:-1]}_share8" if cfg.endswith("s") else f"cfg{cfg}")
        rec[key] = {
            "episodes_per_launch": eps, "fetch_kib": vals.get("FETCH_SIZE"), "write_kib": vals.get("WRITE_SIZE"),
            "sq_waves": vals.get("SQ_WAVES"), "sq_wave_cycles": vals.get("SQ_WAVE_CYCLES"), "sq_busy_cycles": vals.get("SQ_BUSY_CYCLES"),
            "sq_insts_valu": vals.get("SQ_INSTS_VALU"), "sq_active_inst_valu": vals.get("SQ_ACTIVE_INST_VALU"),
            "sq_active_inst_any": vals.get("SQ_ACTIVE_INST_ANY"), "sq_wait_any": vals.get("SQ_WAIT_ANY"),
            "sq_insts_lds": vals.get("SQ_INSTS_LDS"), "kernel_avg_us": vals.get("kernel_avg_us"),
            "kernel_steady_avg_us": vals.get("kernel_steady_avg_us"),
            "kernel_calls": vals.get("kernel_calls"), "kernel_name": vals.get("kernel_name"),
            "profile": f"profiles/{round_name}_{key if key.startswith('reference_') else key}_rocprofv3.txt",
            "grid_size": vals.get("grid_size"), "workgroup_size": vals.get("workgroup_size")}
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--json":
        counters_json(sys.argv[2], sys.argv[3])
    else:
        for p in sys.argv[1:]:
            summarize(p)
