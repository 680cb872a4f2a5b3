#!/bin/bash
set -o pipefail
O=gpurun_out/r03_run3; mkdir -p $O
echo "== gpu tests"; timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -15 $O/pytest_gpu.txt
for rep in 1 2; do
for v in NOMIX PRODUCT; do
  if [ $v = PRODUCT ]; then L=""; else L=tools/labbuild/librwh_$v.so; fi
  echo "== $v 4K"; RWH_LIB=$L SHA=1 POWER=1 N=400 timeout -k 10 200 python tools/warp_quick.py 6 > $O/wq_${v}_4k_$rep.txt 2>&1; tail -1 $O/wq_${v}_4k_$rep.txt
  echo "== $v 8K"; RWH_LIB=$L SHA=1 SRC=7680x4320 FRAMES=8 N=400 timeout -k 10 200 python tools/warp_quick.py 6 > $O/wq_${v}_8k_$rep.txt 2>&1; tail -1 $O/wq_${v}_8k_$rep.txt
done; done
