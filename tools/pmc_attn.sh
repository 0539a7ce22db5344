# SQ / LDS counters of a spatial attention kernel (separate rocprofv3 --pmc passes): bash tools/pmc_attn.sh [kernel-name filter] [frames] [D] [N]
# (d = 40: "attn40q2_kernel 32"; d = 80: "attn80_kernel 32 80 3600")
export TMPDIR=/tmp; W=/tmp/pmca; rm -rf $W; mkdir -p $W; R=$(pwd); F=${1:-attn40_kernel}; B=${2:-32}; D=${3:-40}; N=${4:-14400}; cd /tmp
python3 $R/tools/bench_attn_d40.py $B fp16 $D $N
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY"; do
  rocprofv3 --pmc $C --output-format csv -d $W/p -o x -- python3 $R/tools/bench_attn_d40.py $B fp16 $D $N > /dev/null 2>&1
  python3 $R/tools/pmc_sum.py $W/p $F | head -8
  rm -rf $W/p
done
