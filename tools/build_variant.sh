#!/bin/bash
# Scratch: build a variant of the library with extra -D flags for ONE source file (A/B runs on the GPU box).
#   tools/build_variant.sh NAME "-DGPS_PB_PRIO=0 ..." [file.hip]   ->  gpflow-slim_amd/lib_NAME/libgpflowslim_hip.so   (default file: potrf_base.hip)
set -e
R=$(cd $(dirname $0)/.. && pwd)
N=$1; FLAGS=$2; F=${3:-potrf_base.hip}; O=${F%.hip}.o
mkdir -p $R/gpflow-slim_amd/lib_$N
cd $R/gpflow-slim_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-result $FLAGS -c $F -o ../lib_$N/$O
OBJS=$(ls ../lib/*.o | grep -v "/$O")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib_$N/libgpflowslim_hip.so $OBJS ../lib_$N/$O -ldl
echo built lib_$N
