#!/bin/bash
# tools/occ_sweep.sh: the staged kernels at 2..8 resident blocks per CU (lab builds with -DRWH_LDS_PAD, tools/build_variant.sh occN) -- uint8 and
# float32 output, 16 x 4K (f32_out_probe), and the uint8 kernel on 32 x 4K and 8 x 8K (warp_quick)
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for L in occ2 occ3 occ4 occ5 occ6 product; do
    if [ $L = product ]; then unset RWH_LIB; else export RWH_LIB=tools/labbuild/librwh_$L.so; fi
    echo "== $L (pass $rep)"
    timeout -k 10 100 python tools/f32_out_probe.py 2>&1 | grep "out:"
    N=100 timeout -k 10 100 python tools/warp_quick.py 0 2>&1 | grep "kind"
    N=100 FRAMES=8 SRC=7680x4320 timeout -k 10 100 python tools/warp_quick.py 0 2>&1 | grep "kind"
  done
done
