#!/bin/bash
set -o pipefail
O=gpurun_out/r03_run2; mkdir -p $O
echo "== warp power 4K"; POWER=1 N=400 timeout -k 10 200 python tools/warp_quick.py 6 > $O/wq_power_4k.txt 2>&1; tail -1 $O/wq_power_4k.txt
echo "== warp power 8K"; SRC=7680x4320 FRAMES=8 POWER=1 N=400 timeout -k 10 200 python tools/warp_quick.py 6 > $O/wq_power_8k.txt 2>&1; tail -1 $O/wq_power_8k.txt
for v in NOLOAD NOSTORE; do echo "== $v"; RWH_LIB=tools/labbuild/librwh_$v.so POWER=1 N=400 timeout -k 10 200 python tools/warp_quick.py 6 > $O/wq_power_$v.txt 2>&1; tail -1 $O/wq_power_$v.txt; done
echo "== energy_probe"; timeout -k 10 400 ./tools/energy_probe > $O/energy_probe.txt 2>&1; cat $O/energy_probe.txt
