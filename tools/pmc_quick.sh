#!/bin/bash
# tools/pmc_quick.sh <outdir> <MF value> "<counter set 1>" ...: rocprofv3 --pmc passes over tools/warp_quick.py (kind 0, N=20) -- the headline kernel only,
# seconds per pass instead of the minute a bench.py pass takes.  Counters only with --kernel-trace (gpurun refuses other trace domains beside --pmc).
out=$1; mf=$2; shift; shift
mkdir -p "$out"
export TMPDIR=/tmp
i=0
for set in "$@"; do
  i=$((i+1))
  MF=$mf N=20 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$out/pass$i" -- python3 tools/warp_quick.py 0 > "$out/pass$i.log" 2>&1 || echo "pass $i failed" >> "$out/pass$i.log"
done
python3 tools/pmc_summary.py "$out" > "$out/summary.txt" 2>&1 || true
