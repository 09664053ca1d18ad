#!/bin/bash
# Collects the rocprofv3 evidence judged under profiles/ (run on the GPU box through gpurun from the repo root):
#   kernel-trace statistics of bench.py and of BASELINE configs 2, 4, 5; PMC passes (FETCH_SIZE, WRITE_SIZE, MFMA busy)
#   of one N = 32768 evaluation.  Outputs under gpurun_out/prof_*; tools/summarise_profiles.py turns them into profiles/${TAG}_* (TAG: round tag, default r03).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
export TMPDIR=/tmp
cd /tmp
what=${1:-all}
TAG=${TAG:-r06}
if [ "$what" = all ] || [ "$what" = pmc ]; then
  # counter collection serialises the dispatches: a look-ahead hand-over could only time out (and the evaluation would
  # be re-run without it) -- switch it off up front
  export GPS_LOOKAHEAD=0
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/prof_pmc_$ctr -- python3 $R/tools/one_eval.py 32768 1 > $OUT/prof_pmc_$ctr.log 2>&1 || exit 1
    echo "pmc $ctr done"
  done
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/prof_pmc_mfma -- python3 $R/tools/one_eval.py 32768 1 > $OUT/prof_pmc_mfma.log 2>&1 || exit 1
  echo "pmc mfma done"
  # the bench below reports roofline.traffic only from a PMC summary of the same kernel sources: write it now (on this box)
  (cd $R && python3 tools/summarise_profiles.py $TAG > $OUT/summarise_on_box.log 2>&1) || true
  unset GPS_LOOKAHEAD
fi
if [ "$what" = all ] || [ "$what" = stats ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -- python3 $R/bench.py --steps 5 --no-cpu-baseline > $OUT/prof_bench.log 2>&1 || exit 1
  echo "bench stats done"
  # the reference's own workload size (examples/gpr.py: N ~ 455): per-step latency table, kernel statistics
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_small -- python3 $R/tools/small_n.py 512 2048 > $OUT/prof_small.log 2>&1 || exit 1
  echo "small-N stats done"
  # potrf_base phase stamps (with and without the transposed inverse) and the DPP / SIMD-sharing probe behind its design
  (cd $R && python3 tools/pb_stamps.py 2>&1 | grep -v amdgpu.ids; GPS_PB_NO_T=1 python3 tools/pb_stamps.py 2>&1 | grep -v amdgpu.ids) > $OUT/potrf_base_stamps.txt || true
  # predict_f on few test points (round 6): kernel trace of a warm call at N* = 1024 (tools/trace_tail.py: from its cross kernel matrix
  # on, and the build of the 2048-column inverse blocks), the latency table with and without the wide blocks
  rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_pred -- python3 $R/tools/predict_once.py 32768 1024 > $OUT/prof_pred.log 2>&1 || exit 1
  python3 $R/tools/trace_tail.py $(find $OUT/prof_pred -name "*kernel_trace.csv" | head -1) kmat_prep > $OUT/predict_timeline_1024.txt || true
  python3 $R/tools/trace_tail.py $(find $OUT/prof_pred -name "*kernel_trace.csv" | head -1) blocks_to_diag 2> /dev/null | sed -n "1,14p" > $OUT/predict_wide_build.txt || true
  (cd $R && echo "wide inverse blocks (default)"; python3 tools/predict_once.py 32768 64 256 1024 2048 2>&1 | grep n_new | cut -c1-90; echo "predict_inverse_blocks=0 (the recursive solve down to 512-column launches)"; GPS_OPTS=predict_inverse_blocks=0 python3 tools/predict_once.py 32768 64 256 1024 2048 2>&1 | grep n_new | cut -c1-90; echo "GPS_GEMM_PAIR=0 (one tile per workgroup in the triangular products)"; GPS_GEMM_PAIR=0 python3 tools/predict_once.py 32768 64 256 1024 2>&1 | grep n_new | cut -c1-90) > $OUT/predict_ab.txt || true
  echo "predict done"
  # timelines of one evaluation (kernel trace -> tools/eval_timeline.py): where the time of the chain and of the bulk goes
  for n in 8192 32768; do
    rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_tl$n -- python3 $R/tools/one_eval.py $n 1 > $OUT/prof_tl$n.log 2>&1 || exit 1
    python3 $R/tools/eval_timeline.py $(find $OUT/prof_tl$n -name "*kernel_trace.csv" | head -1) $([ $n = 8192 ] && echo 60 || echo 400) > $OUT/timeline_$n.txt || true
  done
  # the per-step budget of a sweep (what the chain of 128-column steps spends where): N = 4096 (one sweep, nothing beside it) and N = 8192
  rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_tl4096 -- python3 $R/tools/one_eval.py 4096 2 > $OUT/prof_tl4096.log 2>&1 || exit 1
  (python3 $R/tools/sweep_budget.py $(find $OUT/prof_tl4096 -name "*kernel_trace.csv" | head -1); python3 $R/tools/sweep_budget.py $(find $OUT/prof_tl8192 -name "*kernel_trace.csv" | head -1) | sed -n "1,2p;\$p") > $OUT/sweep_step_budget.txt || true
  echo "timelines done"
  # the 512-column triangular solve: one launch against launch by launch, 128 .. 262144 rows
  (cd $R && python3 tools/trsm512.py 128 4096 16384 65536 262144 1048576 2>&1 | grep -v amdgpu.ids) > $OUT/trsm512.txt || true
  # ... and where a row block of it spends its time (phase stamps of every workgroup; round 5): 32 rows per workgroup and two
  # workgroups per CU / 64 rows, persistent / 64 rows per workgroup
  (cd $R && for f in 32 64 65; do python3 tools/tp_stamps.py 1048576 0 $f 2>&1 | grep -v amdgpu.ids; done) > $OUT/trsm512_stamps.txt || true
  (cd $R && python3 tools/kmat_ab.py 2>&1 | grep -v amdgpu.ids) > $OUT/kmat_ab.txt || true
  # the bench line without a profiler around it (kernel tracing costs ~3 %)
  (cd $R && python3 bench.py > $OUT/bench_unprofiled.log 2> $OUT/bench_unprofiled.err) || exit 1
  echo "un-profiled bench done"
  for c in 2 4 5; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_cfg$c -- python3 $R/tools/configs.py $c > $OUT/prof_cfg$c.log 2>&1 || exit 1
    echo "cfg$c stats done"
  done
fi
if [ "$what" = all ] || [ "$what" = kmatpmc ]; then
  # SQ counters of the kernel-matrix kernel of config 4 (five passes)
  for c in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
    n=$(echo $c | tr " " "_" | cut -c1-40)
    GPS_LOOKAHEAD=0 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/kpmc_$n -- python3 $R/tools/kmat_once.py 16384 > $OUT/kpmc_$n.log 2>&1 || exit 1
  done
  echo "kmat counters done"
fi
if [ "$what" = all ] || [ "$what" = soak ]; then
  # long mixed run of every entry point on the final sources (summary -> gpurun_out/soak.json, copied by summarise_profiles.py)
  (cd $R && python3 tools/soak.py ${SOAK_STEPS:-20000} > $OUT/soak.log 2>&1) || exit 1
  tail -1 $OUT/soak.log
fi
