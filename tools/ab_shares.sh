#!/bin/bash
# A/B of two builds of libocd_hip.so on the per-GPU shares of BASELINE configs 4 / 5 (1/8 of the population: one wavefront per
# SIMD, the latency builds of the chunked kernel) and on config 3 / config 2, alternating on ONE box:
#   bash tools/ab_shares.sh build/ab/base.so build/ab/new.so [rounds]
A=$1; B=$2; N=${3:-2}
for i in $(seq 1 $N); do
  for L in $A $B; do
    echo "== $L"
    OCD_HIP_LIB=$(pwd)/$L python tools/sweep.py --configs 5,4 --pop 16 --reps 8 2>&1 | grep -v amdgpu.ids
    OCD_HIP_LIB=$(pwd)/$L python tools/sweep.py --configs 3,2 --reps 8 2>&1 | grep -v amdgpu.ids
  done
done
