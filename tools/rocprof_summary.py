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


if __name__ == "__main__":
    for p in sys.argv[1:]:
        summarize(p)
