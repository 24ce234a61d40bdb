#!/bin/bash
# Ablation of the round-2 kernel techniques on ONE GPU box (same clocks for every arm):
#   build the arms HERE (no GPU needed):   bash tools/ablation.sh build
#   time them on the GPU box:              bash tools/ablation.sh run     (prints ms per launch, configs 2 and 3)
#                                          bash tools/ablation.sh run_items   (configs 5 / 4 whole, 16 x config 3: with / without work items)
# Arms: shipped library; -DOCD_NO_PACKED (scalar division / exp cores); -DOCD_NO_ASM_CHAINS (compiler-scheduled
# DPP recurrences); both; and the shipped library with the diagnostics knobs no_unified_features / no_feature_skips /
# no_latency_build (the knobs other than no_latency_build also select the non-LAT builds).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CSRC=$ROOT/l4dc-mpc-ocd_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-function"
if [ "$1" = "build" ]; then
    make -C "$CSRC" -j4 all > /dev/null
    for arm in "nopk:-DOCD_NO_PACKED" "noasm:-DOCD_NO_ASM_CHAINS" "neither:-DOCD_NO_PACKED -DOCD_NO_ASM_CHAINS"; do
        name=${arm%%:*}; defs=${arm#*:}
        /opt/rocm/bin/hipcc $FLAGS $defs -c "$CSRC/ocd_kernels.hip" -o /tmp/abl_$name.o
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-z,defs -o "$CSRC/libocd_hip_abl_$name.so" /tmp/abl_$name.o "$CSRC/ocd_chunk_kernel.o" "$CSRC/ocd_api.o" "$CSRC/ocd_debug_kernels.o"
        echo built $name
    done
    # the shared-SIMD builds of the chunked kernel without their work-item lists (round 5): every (lane, step) pair through reward_one
    /opt/rocm/bin/hipcc $FLAGS -DOCD_NO_ITEMS -c "$CSRC/ocd_chunk_kernel.hip" -o /tmp/abl_noitems.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$CSRC/libocd_hip_abl_noitems.so" "$CSRC/ocd_kernels.o" /tmp/abl_noitems.o "$CSRC/ocd_api.o" "$CSRC/ocd_debug_kernels.o"
    echo built noitems
elif [ "$1" = "run_items" ]; then
    cd "$ROOT"
    for lib in libocd_hip.so libocd_hip_abl_noitems.so libocd_hip.so libocd_hip_abl_noitems.so; do
        echo "== $lib"
        OCD_HIP_LIB=$CSRC/$lib python tools/sweep.py --configs 5,4 --reps 4 2>&1 | grep cfg
        OCD_HIP_LIB=$CSRC/$lib python tools/sweep.py --configs 3 --pop 1024 --reps 6 2>&1 | grep cfg
        OCD_HIP_LIB=$CSRC/$lib python tools/sweep.py --configs 3 --pop 960 --reps 6 2>&1 | grep cfg
    done
else
    cd "$ROOT"
    for lib in libocd_hip.so libocd_hip_abl_nopk.so libocd_hip_abl_noasm.so libocd_hip_abl_neither.so; do
        echo "== $lib"
        OCD_HIP_LIB=$CSRC/$lib python tools/sweep.py --configs 2,3 --scan-mode 0 --segs 0 --reps 10 2>&1 | grep cfg
    done
    echo "== libocd_hip.so, no_unified_features"
    python tools/sweep.py --configs 2,3 --scan-mode 0 --segs 0 --reps 10 --no-unify 1 2>&1 | grep cfg
    echo "== libocd_hip.so, no_feature_skips"
    python tools/sweep.py --configs 2,3 --scan-mode 0 --segs 0 --reps 10 --no-skips 1 2>&1 | grep cfg
    echo "== libocd_hip.so, no_latency_build (V_ROW / V_SEG with the none-active path and the has_col / has_f skips)"
    python tools/sweep.py --configs 2,3 --scan-mode 0 --segs 0 --reps 10 --no-lat 1 2>&1 | grep cfg
    echo "== libocd_hip.so again"
    python tools/sweep.py --configs 2,3 --scan-mode 0 --segs 0 --reps 10 2>&1 | grep cfg
fi
