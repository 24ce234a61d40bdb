#!/bin/bash
# rocprofv3 evidence for profiles/: per BASELINE config, one kernel-trace/stats run and three PMC passes
# (FETCH_SIZE, WRITE_SIZE, SQ_*) of the SAME bench.py command, each in its own run (gpurun refuses
# --pmc combined with the trace domains other than --kernel-trace).  Run ON THE GPU BOX from the repo root:
#     OCD_GIT_COMMIT=<hash of the tree being profiled> bash tools/profile_round.sh r05 "2 3 4 5 4s 5s reference_h5 reference_h6_extra"
# (the GPU box holds no .git: the commit travels in the environment; profiles/pmc_counters.json records it together with
#  the hash of the kernel sources, and bench.py replays the counters only beside kernels built from those sources)
# A workload token is a BASELINE config index, or <index>s = the share of rank 0 of an 8-way strong split of that
# config run on this one GPU (bench.py --emulate-rank 0/8): 4s = 2 048 episodes at H=15, 5s = 4 096 at H=25.
# Writes rocpd databases under gpurun_out/prof_<round>/ (scratch) and the condensed summaries
# profiles/<round>_cfg<N>[_share8]_rocprofv3.txt + profiles/pmc_counters.json (tracked).
set -e -o pipefail
ROUND=${1:-r02}
CONFIGS=${2:-"2 3 4 5"}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$ROUND
mkdir -p "$OUT"
export TMPDIR=/tmp
SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_ANY"
for c in $CONFIGS; do
    case "$c" in
      reference_*)   # the reference's own shapes (bench.py REFERENCE_SHAPES): 27 episodes per launch
        n=$c; steps=30; extra=""; tag=$c;;
      *)
        n=${c%s}
        steps=30; [ "$n" -ge 4 ] && steps=6
        extra=""; tag=$n; [ "$c" != "$n" ] && { extra="--emulate-rank 0/8"; steps=20; tag="${n}_share8"; };;
    esac
    args="bench.py --config $n $extra --steps $steps --warmup 2 --no-extras --no-cpu-baseline --no-parity"
    cd /tmp
    rocprofv3 --kernel-trace --stats -d "$OUT/c${c}_stats" -o res -- python3 "$ROOT"/$args > "$OUT/c${c}_stats.log" 2>&1
    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/c${c}_fetch" -o res -- python3 "$ROOT"/$args > "$OUT/c${c}_fetch.log" 2>&1
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/c${c}_write" -o res -- python3 "$ROOT"/$args > "$OUT/c${c}_write.log" 2>&1
    rocprofv3 --pmc $SQ --kernel-trace -d "$OUT/c${c}_sq" -o res -- python3 "$ROOT"/$args > "$OUT/c${c}_sq.log" 2>&1
    cd "$ROOT"
    python3 tools/rocprof_summary.py $(find "$OUT/c${c}_stats" "$OUT/c${c}_fetch" "$OUT/c${c}_write" "$OUT/c${c}_sq" -name "*.db" | sort) \
        > "$OUT/${ROUND}_$( [[ "$tag" == reference_* ]] && echo "$tag" || echo "cfg${tag}" )_rocprofv3.txt"
    echo "workload $c profiled"
done
python3 tools/rocprof_summary.py --json "$ROUND" "$OUT" > "$OUT/pmc_counters.json"
# the tracked copies (bench.py replays profiles/pmc_counters.json; the judge reads profiles/)
mkdir -p "$ROOT/profiles"
cp "$OUT"/${ROUND}_*_rocprofv3.txt "$ROOT/profiles/"
cp "$OUT/pmc_counters.json" "$ROOT/profiles/pmc_counters.json"
echo "done: profiles/${ROUND}_cfg*_rocprofv3.txt and profiles/pmc_counters.json written (gpurun merges gpurun_out/ only: copy them back from $OUT)"
