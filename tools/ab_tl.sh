#!/bin/bash
for lib in lib_base lib; do
  echo "== $lib"
  GPFLOWSLIM_HIP_LIB=$PWD/gpflow-slim_amd/$lib/libgpflowslim_hip.so timeout 200 python tools/gemm_timeline.py "$@" 2>&1 | grep -E "shape|prologue"
done
