#!/bin/bash
# A/B on ONE box: alternate two builds of the library on the same workloads (kernel ms per launch).
#   bash tools/ab.sh <libA.so> <libB.so> "<sweep args>" [rounds]
A=$1; B=$2; ARGS=$3; N=${4:-3}
for i in $(seq 1 $N); do
  for L in $A $B; do
    echo "== $L"
    OCD_HIP_LIB=$(pwd)/$L python tools/sweep.py $ARGS 2>&1 | grep -v amdgpu.ids
  done
done
