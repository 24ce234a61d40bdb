#!/bin/bash
# usage: ab_r6.sh A.so B.so rounds
A=$1; B=$2; N=${3:-2}
for i in $(seq 1 $N); do
  for L in $A $B; do
    echo "== $L"
    OCD_HIP_LIB=$(pwd)/$L python tools/sweep.py --configs 4 --pop 16 --reps 8 2>&1 | grep cfg
    OCD_HIP_LIB=$(pwd)/$L python tools/sweep.py --configs 5 --pop 32 --reps 8 2>&1 | grep cfg
  done
done
