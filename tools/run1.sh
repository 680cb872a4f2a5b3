#!/bin/bash
# GPU call 1 of round 3: baseline on this box, clocks, ablations
set -o pipefail
O=gpurun_out/r03_run1; mkdir -p $O
rocm-smi --showpower --showclocks --showmaxpower > $O/smi_idle.txt 2>&1
echo "== clock_probe" ; timeout -k 10 300 ./tools/clock_probe > $O/clock_probe.txt 2>&1; tail -12 $O/clock_probe.txt
echo "== warp_quick base 4K" ; N=400 timeout -k 10 200 python tools/warp_quick.py 6 6 > $O/wq_base_4k.txt 2>&1 & 
PID=$!
sleep 45; for i in 1 2 3 4 5 6; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk|fclk" | tr '\n' ' '; echo; sleep 0.3; done > $O/smi_during_warp.txt 2>&1
wait $PID; cat $O/wq_base_4k.txt | tail -3; cat $O/smi_during_warp.txt | tail -3
echo "== 8K" ; SRC=7680x4320 FRAMES=8 N=400 timeout -k 10 200 python tools/warp_quick.py 6 > $O/wq_base_8k.txt 2>&1; tail -1 $O/wq_base_8k.txt
for v in NOLOAD NOSTORE NOLDS; do echo "== $v"; RWH_LIB=tools/labbuild/librwh_$v.so N=400 timeout -k 10 200 python tools/warp_quick.py 6 > $O/wq_$v.txt 2>&1; tail -1 $O/wq_$v.txt; done
echo "== base again"; N=400 timeout -k 10 200 python tools/warp_quick.py 6 > $O/wq_base_4k_b.txt 2>&1; tail -1 $O/wq_base_4k_b.txt
