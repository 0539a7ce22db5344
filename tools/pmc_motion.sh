export TMPDIR=/tmp; W=/tmp/pmcm; rm -rf $W; mkdir -p $W; R=$(pwd); cd /tmp
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA"; do
  rocprofv3 --pmc $C --output-format csv -d $W/p -o x -- python3 $R/tools/bench_motion.py fp16 > /dev/null 2>&1
  python3 $R/tools/pmc_sum.py $W/p motion_c320 | head -8
  rm -rf $W/p
done
