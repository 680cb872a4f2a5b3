#!/bin/bash
# where do the multi-frame lab kernels win?  frame size x batch, product (MF 0) vs block window 3 / 4 frames (103 / 104) vs wave windows 3 (3); two passes per process, read the second
cd "$(dirname "$0")/.."
run() { echo "== $1 x $2"; N=${3:-80} FRAMES=$2 SRC=$1 MF=0,103,104,3,0,103,104,3 timeout -k 10 200 python tools/warp_quick.py 0 2>&1 | grep kind | sed 's/.*kind  0 //' | cut -c1-110; }
run 3840x2160 8 200; run 3840x2160 12 150; run 3840x2160 16 120; run 3840x2160 24 100; run 3840x2160 64 50
run 1920x1080 32 200; run 1920x1080 128 100; run 1280x720 128 150; run 7680x4320 16 50; run 2560x1440 48 100
