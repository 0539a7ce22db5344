# round 5: SQ / LDS counters of the three slab-streaming fused kernels (chain tail, chain front, motion module) -- separate rocprofv3 --pmc passes
export TMPDIR=/tmp; W=/tmp/pmcf; rm -rf $W; mkdir -p $W; R=$(pwd); O=$R/gpurun_out/r5_pmc_fused; mkdir -p $O; cd /tmp
python3 $R/tools/bench_chain.py fp16 > $O/bench_chain.txt 2>&1
python3 $R/tools/bench_motion.py fp16 > $O/bench_motion.txt 2>&1
for B in bench_chain bench_motion; do
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_IFETCH SQ_WAIT_IFETCH SQ_INSTS_SMEM SQ_LDS_ADDR_CONFLICT"; do
  rocprofv3 --pmc $C --output-format csv -d $W/p -o x -- python3 $R/tools/$B.py fp16 > /dev/null 2>&1
  python3 $R/tools/pmc_sum.py $W/p c320 >> $O/$B.pmc.txt
  rm -rf $W/p
done
done
cat $O/bench_chain.txt $O/bench_motion.txt; cat $O/*.pmc.txt
