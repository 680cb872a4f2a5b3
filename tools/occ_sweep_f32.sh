#!/bin/bash
# float32-output kernel only, 4 passes, alternating builds (the figure moves by +-4 % from process to process: medians)
cd "$(dirname "$0")/.."
for rep in 1 2 3 4; do
  for L in f32old f32w5 occ4 occ5 occ6 product; do
    if [ $L = product ]; then unset RWH_LIB; else export RWH_LIB=tools/labbuild/librwh_$L.so; fi
    echo -n "$L  "
    timeout -k 10 100 python tools/f32_out_probe.py 2>&1 | grep "float32 out:" | sed 's/.*out: //'
  done
done
