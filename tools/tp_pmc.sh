cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out
for rows in 64 33; do
for c in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE"; do
  n=$(echo $c | tr " " "_" | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/tppmc_${rows}_$n -- python3 $R/tools/tp_once.py 262144 $rows > $OUT/tppmc_${rows}_$n.log 2>&1 || exit 1
done; done
echo done
