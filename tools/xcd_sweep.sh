#!/bin/bash
# block -> tile mapping: XCD k takes chunks of 2^LOG consecutive tiles round-robin (lab builds xcdLOG) vs the product (one contiguous band per XCD)
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for L in product xcd0 xcd5 xcd8 xcd10 xcd12; do
    if [ $L = product ]; then unset RWH_LIB; else export RWH_LIB=tools/labbuild/librwh_$L.so; fi
    echo "== $L (pass $rep)"
    N=100 SHA=1 timeout -k 10 100 python tools/warp_quick.py 0 2>&1 | grep "kind" | cut -c40-
    N=100 FRAMES=8 SRC=7680x4320 timeout -k 10 100 python tools/warp_quick.py 0 2>&1 | grep "kind" | cut -c40-
  done
done
