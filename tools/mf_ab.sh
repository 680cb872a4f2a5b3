#!/bin/bash
# one-frame product kernel vs the multi-frame lab kernels (MF = frames per block; 100 + n = one window per block), same box, alternating
cd "$(dirname "$0")/.."
echo "== 32 x 4K"
N=100 MF=0,3,103,0,3,103,4,104 timeout -k 10 200 python tools/warp_quick.py 0 2>&1 | grep kind | cut -c40-
echo "== 8 x 8K"
N=100 FRAMES=8 SRC=7680x4320 MF=0,2,3,102,0,2,3 timeout -k 10 200 python tools/warp_quick.py 0 2>&1 | grep kind | cut -c40-
echo "== 512 x 1080p"
N=30 FRAMES=512 SRC=1920x1080 MF=0,3,4,104,0,3,4 timeout -k 10 200 python tools/warp_quick.py 0 2>&1 | grep kind | cut -c40-
echo "== 8 x 4K"
N=200 FRAMES=8 MF=0,2,3,4,0,2 timeout -k 10 200 python tools/warp_quick.py 0 2>&1 | grep kind | cut -c40-
