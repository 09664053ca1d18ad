#!/bin/bash
# Scratch: same-box A/B of two builds of the library (lib_base = a build of an earlier commit).  usage: ab_lib.sh N...
for rep in 1 2; do
  for lib in lib_base lib; do
    echo "== $lib (rep $rep)"
    GPFLOWSLIM_HIP_LIB=$PWD/gpflow-slim_amd/$lib/libgpflowslim_hip.so timeout 200 python tools/first_light.py "$@" 2>&1 | grep "N=" | awk 'NR%3==0'
  done
done
