# SQ counters of the fused chain tail for one build of the library: bash tools/jobs/r5_pmc_chain_lib.sh <lib path relative to csrc> <tag>
export TMPDIR=/tmp; W=/tmp/pmcf; rm -rf $W; mkdir -p $W; R=$(pwd); O=$R/gpurun_out/r5_pmc_chain_$2; mkdir -p $O; cd /tmp
export VV_LIB_PATH=$R/videovanish_amd/csrc/$1
python3 $R/tools/bench_chain.py fp16 > $O/bench.txt 2>&1
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY"; do
  rocprofv3 --pmc $C --output-format csv -d $W/p -o x -- python3 $R/tools/bench_chain.py fp16 > /dev/null 2>&1
  python3 $R/tools/pmc_sum.py $W/p c320_kernel >> $O/pmc.txt
  rm -rf $W/p
done
grep -v amdgpu.ids $O/bench.txt; cat $O/pmc.txt
