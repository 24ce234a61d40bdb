#!/bin/bash
# A/B of two builds of the library on the H = 10 headline shapes (config 3, config 2) and the reference's shape, alternating:
#   bash tools/ab_headline_r6.sh build/ab/base.so build/ab/new.so [rounds]
A=$1; B=$2; N=${3:-3}
for i in $(seq 1 $N); do
  for L in $A $B; do
    echo "== $L"
    OCD_HIP_LIB=$(pwd)/$L python tools/sweep.py --configs 3,2 --reps 12 2>&1 | grep cfg
    OCD_HIP_LIB=$(pwd)/$L python tools/small_shapes.py --reps 8 --modes 0 2>&1 | grep "E="
  done
done
