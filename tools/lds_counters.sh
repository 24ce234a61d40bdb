#!/bin/bash
# LDS counters of the whole-config launches (the work-item lists of the chunked kernel's shared-SIMD builds), GPU box:
#   bash tools/lds_counters.sh   -> gpurun_out/lds_counters.txt
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT="$ROOT/gpurun_out/lds_prof"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for c in 4 5; do
  rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES --kernel-trace -d "$OUT/c$c" -o res -- python3 "$ROOT"/bench.py --config $c --steps 4 --warmup 1 --no-extras --no-cpu-baseline --no-parity > "$OUT/c$c.log" 2>&1
done
cd "$ROOT"
python3 - "$OUT" <<'PY' > "$ROOT/gpurun_out/lds_counters.txt"
import glob, sqlite3, sys
for db in sorted(glob.glob(sys.argv[1] + "/*/**/*.db", recursive=True)):
    con = sqlite3.connect(db)
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
    print("##", db.split("gpurun_out/")[-1])
    try:
        view = [t for t in tabs if t.startswith("counters_collection")] or ["counters_collection"]
        for r in con.execute(f"select kernel_name, counter_name, count(*), avg(value) from {view[0]} where kernel_name like '%mpc%' group by kernel_name, counter_name"):
            print(f"  {r[0][:64]:64s} {r[1]:24s} n={r[2]} avg={r[3]:.6g}")
    except Exception as e:
        print("  no counters:", e, tabs[:8])
PY
cat "$ROOT/gpurun_out/lds_counters.txt"
