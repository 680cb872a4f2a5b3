#!/bin/bash
# minification A/B: round 3's halves kernel (tools/labbuild/librwh_f32old.so = the header of the previous commit) vs this build
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for S in 0.7692 0.6667 0.5714; do
    for L in f32old product; do
      if [ $L = product ]; then unset RWH_LIB; else export RWH_LIB=tools/labbuild/librwh_$L.so; fi
      echo -n "scale $S $L: "
      N=100 BOUNDS=1 SCALE=$S SHA=1 timeout -k 10 100 python tools/warp_quick.py 0 2>&1 | grep "kind"
    done
  done
done
