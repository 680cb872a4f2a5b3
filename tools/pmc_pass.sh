#!/bin/bash
# PMC passes over `python3 bench.py --steps 3 --warmup 1 --no-cpu` (one counter set per pass; counters only with --kernel-trace).
# usage: tools/pmc_pass.sh <outdir> "<counters of pass 1>" "<counters of pass 2>" ...
set -e
out=$1; shift
mkdir -p "$out"
export TMPDIR=/tmp
i=0
for set in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$out/pass$i" -- python3 bench.py --steps 3 --warmup 1 --no-cpu > "$out/pass$i.log" 2>&1 || echo "pass $i failed" >> "$out/pass$i.log"
done
python3 tools/pmc_summary.py "$out" > "$out/summary.txt" 2>&1 || true
