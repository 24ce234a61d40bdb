#!/bin/bash
# A/B of two builds of libocd_hip.so on the throughput workloads (kernel ms per launch), alternating on ONE box:
#   bash tools/ab_throughput.sh build/ab/base.so build/ab/new.so [rounds]
# workloads: BASELINE configs 5 and 4 whole on one GPU (32 768 / 16 384 episodes), 16 x config 3 (32 768 episodes, S=2),
# and 10 240 / 20 480 / 30 720 episodes of config 3's shape (one / two / three wavefronts per SIMD at S=5)
A=$1; B=$2; N=${3:-2}
for i in $(seq 1 $N); do
  for L in $A $B; do
    echo "== $L"
    OCD_HIP_LIB=$(pwd)/$L python tools/sweep.py --configs 5,4 --reps 4 2>&1 | grep -v amdgpu.ids
    OCD_HIP_LIB=$(pwd)/$L python tools/sweep.py --configs 3 --pop 1024 --reps 6 2>&1 | grep -v amdgpu.ids
    for P in 320 640 960; do OCD_HIP_LIB=$(pwd)/$L python tools/sweep.py --configs 3 --pop $P --reps 6 2>&1 | grep -v amdgpu.ids; done
  done
done
