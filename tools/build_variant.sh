#!/bin/bash
# Scratch: build a variant of the library with extra -D flags for potrf_base.hip only (A/B runs on the GPU box).
#   tools/build_variant.sh NAME "-DGPS_PB_PRIO=0 ..."   ->  gpflow-slim_amd/lib_NAME/libgpflowslim_hip.so
set -e
R=$(cd $(dirname $0)/.. && pwd)
N=$1; shift
mkdir -p $R/gpflow-slim_amd/lib_$N
cd $R/gpflow-slim_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-result $@ -c potrf_base.hip -o ../lib_$N/potrf_base.o
OBJS=$(ls ../lib/*.o | grep -v potrf_base.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib_$N/libgpflowslim_hip.so $OBJS ../lib_$N/potrf_base.o -ldl
echo built lib_$N
