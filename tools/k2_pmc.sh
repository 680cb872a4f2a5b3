#!/bin/bash
# PMC pass over tools/k2_only.py in exact (1) and filter (0) mode.  usage: tools/k2_pmc.sh <outdir>
set -e
out=$1
mkdir -p "$out"
export TMPDIR=/tmp
for mode in 1 0; do
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY --output-format csv -d "$out/mode$mode" -- python3 tools/k2_only.py $mode > "$out/mode$mode.log" 2>&1 || echo "mode $mode failed" >> "$out/mode$mode.log"
done
