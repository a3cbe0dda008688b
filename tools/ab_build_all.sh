#!/bin/bash
# Tuning aid: a complete variant build of libodt_hip.so with extra -D flags for EVERY translation unit.
#   tools/ab_build_all.sh NAME [-DFOO=1 ...]  ->  ab/NAME.so   (ab/ is git-ignored, travels with gpurun)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p ab/obj_$name
C=object_detection_tracking_amd/csrc
srcs=$(python -c "from object_detection_tracking_amd.build import SOURCES; print(' '.join(SOURCES))")
for s in $srcs; do
  ( x=""; case $s in *.cpp) x="";; *) x="-x hip";; esac
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -I $C "$@" -c $x $C/$s -o ab/obj_$name/$s.o 2>/dev/null ) &
  while [ $(jobs -r | wc -l) -ge 8 ]; do sleep 0.2; done
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/$name.so ab/obj_$name/*.o
rm -rf ab/obj_$name
echo ab/$name.so
