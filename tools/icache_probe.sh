#!/bin/bash
# Instruction-cache counters of the planner kernel at one shape (GPU box).  usage: bash tools/icache_probe.sh <cfg> <pop> [extra sweep.py args]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT="$ROOT/gpurun_out/icache_c$1_p$2"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --kernel-trace -d "$OUT/ic" -o res -- python3 "$ROOT"/tools/sweep.py --configs $1 --pop $2 --reps 2 "${@:3}" > "$OUT/ic.log" 2>&1
rocprofv3 --pmc SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace -d "$OUT/sq" -o res -- python3 "$ROOT"/tools/sweep.py --configs $1 --pop $2 --reps 2 "${@:3}" > "$OUT/sq.log" 2>&1
cd "$ROOT"
python3 - "$OUT" <<'PY'
import glob, sqlite3, sys
for db in sorted(glob.glob(sys.argv[1] + "/*/**/*.db", recursive=True)):
    con = sqlite3.connect(db)
    try:
        for r in con.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                             "where kernel_name like '%mpc%' group by kernel_name, counter_name"):
            print(r[0][:70], r[1], r[2], f"{r[3]:.4g}")
    except Exception as e:
        print("no counters in", db, e)
PY
