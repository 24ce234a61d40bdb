#!/bin/bash
# Instruction / stall counters of the planner kernel for two builds of the library at one shape (GPU box):
#   bash tools/pmc_ab.sh <cfg> <pop> libA.so libB.so
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
CFG=$1; POP=$2; shift 2
export TMPDIR=/tmp
for LIB in "$@"; do
  TAG=$(basename $LIB .so)
  OUT="$ROOT/gpurun_out/pmc_ab_c${CFG}_p${POP}_$TAG"
  mkdir -p "$OUT"
  export OCD_HIP_LIB=$ROOT/$LIB
  cd /tmp
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH --kernel-trace -d "$OUT/sq" -o res -- python3 "$ROOT"/tools/sweep.py --configs $CFG --pop $POP --reps 2 > "$OUT/sq.log" 2>&1
  rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_IFETCH SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM --kernel-trace -d "$OUT/ic" -o res -- python3 "$ROOT"/tools/sweep.py --configs $CFG --pop $POP --reps 2 > "$OUT/ic.log" 2>&1
  cd "$ROOT"
  echo "== $LIB"
  python3 - "$OUT" <<'PY'
import glob, sqlite3, sys
for db in sorted(glob.glob(sys.argv[1] + "/*/**/*.db", recursive=True)):
    con = sqlite3.connect(db)
    try:
        for r in con.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                             "where kernel_name like '%mpc%' group by kernel_name, counter_name"):
            print(r[0][:60], r[1], r[2], f"{r[3]:.5g}")
    except Exception as e:
        print("no counters in", db, e)
PY
done
