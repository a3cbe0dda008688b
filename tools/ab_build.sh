#!/bin/bash
# Tuning aid: build a variant of libodt_hip.so with extra -D flags for conv_igemm.hip only.
#   tools/ab_build.sh NAME [-DFOO=1 ...]  ->  ab/NAME.so   (ab/ is git-ignored, travels with gpurun)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p ab
C=object_detection_tracking_amd/csrc
O=object_detection_tracking_amd/build/hip
python -m object_detection_tracking_amd.build hip > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -I $C "$@" -c $C/conv_igemm.hip -o ab/conv_$name.o
objs=$(ls $O/*.o | grep -v conv_igemm)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/$name.so $objs ab/conv_$name.o
echo ab/$name.so
