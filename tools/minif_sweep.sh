#!/bin/bash
# minification sweep: the library's own choice (kind 0) against forced shapes (6 = 64 x 8 whole patches, 14 = 64 x 8 by halves, 13 = 32 x 16 by halves, 5 = 32 x 16 whole)
cd "$(dirname "$0")/.."
for S in ${SCALES:-0.8333 0.8 0.7692 0.7143 0.5556 0.5 0.4545 0.4}; do
  echo "== scale $S (minification $(python3 -c "print(round(1/$S,3))"))"
  N=60 BOUNDS=1 SCALE=$S timeout -k 10 200 python tools/warp_quick.py ${KINDS:-0 6 14 13 5} 2>&1 | grep "kind"
done
