#!/bin/bash
# fabric traffic vs sustained clock.  For each library variant (STRIP = tile columns per strip of the GEMM's
# workgroup -> tile order): power / sclk sampled at 10 ms during N = 32768 evaluations, then FETCH_SIZE of one evaluation.
# Run on the GPU box from the repo root; writes gpurun_out/pvt_*.json(l); tools/power_vs_traffic.py merges them into profiles/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for v in ${VARIANTS:-lib lib_strip4 lib_strip16}; do
  export GPFLOWSLIM_HIP_LIB=$R/gpflow-slim_amd/$v/libgpflowslim_hip.so
  python3 $R/tools/power_trace.py 32768 12 2>/dev/null | tail -1 > $OUT/pvt_power_$v.json || exit 1
  echo "power $v done"
  GPS_LOOKAHEAD=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pvt_fetch_$v -- python3 $R/tools/one_eval.py 32768 1 > $OUT/pvt_fetch_$v.log 2>&1 || exit 1
  echo "fetch $v done"
done
