#!/bin/bash
# tools/build_variant.sh NAME "<extra hipcc flags>": a lab build of librwh_hip.so -> tools/labbuild/librwh_NAME.so (rwh_warp.hip recompiled
# with the flags, the other objects taken from the product build); load it with RWH_LIB=... tools/warp_quick.py
set -e
cd "$(dirname "$0")/../ransac_with_homography_amd/csrc"
NAME=$1; shift
mkdir -p /tmp/lab_$NAME ../../tools/labbuild
cp ../lib/rwh_api.o ../lib/rwh_ransac.o ../lib/rwh_stitch.o ../lib/rwh_host.o ../lib/rwh_run.o /tmp/lab_$NAME/
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -fvisibility=hidden "$@" -c rwh_warp.hip -o /tmp/lab_$NAME/rwh_warp.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/labbuild/librwh_$NAME.so /tmp/lab_$NAME/*.o -Wl,-rpath,/opt/rocm/lib -lpthread
echo built tools/labbuild/librwh_$NAME.so
