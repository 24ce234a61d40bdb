#!/bin/bash
# same-box A/B/C...: alternate builds of the library on one workload (kernel ms per launch)
#   bash tools/ab3.sh "<lib1.so lib2.so ...>" "<sweep args>" [rounds]
LIBS=$1; ARGS=${2:---configs 3 --reps 60}; N=${3:-3}
for i in $(seq 1 $N); do
  for L in $LIBS; do
    echo -n "$L "
    OCD_HIP_LIB=$(pwd)/l4dc-mpc-ocd_amd/csrc/$L python tools/sweep.py $ARGS 2>&1 | grep "ms/launch" | awk '{printf "%s ", $7}'
    echo
  done
done
