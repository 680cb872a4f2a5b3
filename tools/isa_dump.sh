#!/bin/bash
# tools/isa_dump.sh [extra hipcc flags]: device assembly of rwh_warp.hip -> /tmp/isa/w.s, then resource lines of the kernels named in $KERNELS
set -e
mkdir -p /tmp/isa
cd "$(dirname "$0")/../ransac_with_homography_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -fvisibility=hidden --cuda-device-only -S "$@" -o /tmp/isa/w.s rwh_warp.hip 2>/dev/null
for k in ${KERNELS:-_ZN3rwh16warp_rgb8_fast8mILi6ELb1EEEvNS_8FastArgsE _ZN3rwh16warp_rgb8_fast8mILi6ELb0EEEvNS_8FastArgsE _ZN3rwh15warp_rgb8_fast8IhLi6EEEvNS_8FastArgsE}; do
  start=$(grep -n "^$k:" /tmp/isa/w.s | cut -d: -f1)
  [ -z "$start" ] && { echo "$k: not found"; continue; }
  end=$(awk -v s=$start 'NR>s && /^\.Lfunc_end/{print NR; exit}' /tmp/isa/w.s)
  sed -n "${start},${end}p" /tmp/isa/w.s > /tmp/isa/$k.s
  echo "$k: $((end-start)) lines;" $(sed -n "${end},$((end+45))p" /tmp/isa/w.s | grep -E "TotalNumSgprs|NumVgprs:|Occupancy|LDSByteSize|ScratchSize" | tr -d ';' | tr '\n' ' ')
done
