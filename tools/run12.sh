#!/bin/bash
set -o pipefail
O=gpurun_out/r03_run12; mkdir -p $O
echo "== soak settle"; timeout -k 10 600 python tools/soak_settle.py 300 1 > $O/soak_settle.txt 2>&1; tail -8 $O/soak_settle.txt
echo "== soak warp"; timeout -k 10 500 python tools/soak_warp.py 500 3 > $O/soak_warp.txt 2>&1; tail -4 $O/soak_warp.txt
echo "== soak warp rgba"; CH=4 timeout -k 10 300 python tools/soak_warp.py 200 4 > $O/soak_warp4.txt 2>&1; tail -3 $O/soak_warp4.txt
echo "== soak stitch"; timeout -k 10 400 python tools/soak_stitch.py 150 5 > $O/soak_stitch.txt 2>&1; tail -4 $O/soak_stitch.txt
echo "== soak ransac"; timeout -k 10 400 python tools/soak_ransac.py 200 6 > $O/soak_ransac.txt 2>&1; tail -3 $O/soak_ransac.txt
