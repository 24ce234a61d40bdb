#!/bin/bash
# Kernel time of every lane mapping / chunk size over batch sizes (the data the launch heuristics of
# launch_mpc / launch_chunk_dispatch are fitted on).  usage (GPU box): bash tools/sweep_sizes.sh > gpurun_out/sweep_sizes.log
python tools/sweep.py --configs 3 --reps 20 > /dev/null          # GPU clock warm-up
for pop in 64 96 128 192 256 384 512 1024; do
    python tools/sweep.py --configs 3 --pop $pop --scan-mode 0,1,3 --reps 3
    for ch in 2 5; do python tools/sweep.py --configs 3 --pop $pop --scan-mode 4 --chunk $ch --reps 3; done
done
for pop in 8 16 24 32 48 64 128; do
    python tools/sweep.py --configs 4 --pop $pop --scan-mode 0,1,2,3 --reps 3
    for ch in 2 3 5; do python tools/sweep.py --configs 4 --pop $pop --scan-mode 4 --chunk $ch --reps 3; done
done
for pop in 1 4 8 16 32 64 128 256; do
    python tools/sweep.py --configs 5 --pop $pop --scan-mode 0,1 --reps 3
    for ch in 2 3 5; do python tools/sweep.py --configs 5 --pop $pop --scan-mode 4 --chunk $ch --reps 3; done
done
