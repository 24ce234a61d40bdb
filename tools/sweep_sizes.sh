for ps in 2 4 8 16; do python tools/sweep.py --configs 3 --pop-scale $ps --scan-mode 0,1,3,4 --reps 3; done
for p in 32 64 128; do python tools/sweep.py --configs 4 --pop $p --scan-mode 0,1,2,3 --reps 3; python tools/sweep.py --configs 4 --pop $p --scan-mode 4 --chunk 3 --reps 3; python tools/sweep.py --configs 4 --pop $p --scan-mode 4 --chunk 5 --reps 3; done
for p in 16 64 128 256; do python tools/sweep.py --configs 5 --pop $p --scan-mode 0,1,4 --reps 3; done
