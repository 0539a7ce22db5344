# effective clock + matrix-pipe / LDS counters of the GEMM kernels on ONE shape of tools/bench_gemm256.py:
#   bash tools/pmc_gemm.sh "<VV_BENCH_ONLY substring>" [fp16|bf16]
# Per kernel: GRBM_GUI_ACTIVE (cycles of the launch; / kernel-trace duration = effective clock), MFMA busy, VALU, LDS bank conflicts.
export TMPDIR=/tmp; W=/tmp/pmcg; rm -rf $W; mkdir -p $W; R=$(pwd); export VV_BENCH_ONLY="$1"; D=${2:-fp16}; cd /tmp
python3 $R/tools/bench_gemm256.py $D
for C in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $W/p -o x -- python3 $R/tools/bench_gemm256.py $D > /dev/null 2>&1
  python3 $R/tools/pmc_sum.py $W/p gemm
  python3 - <<PY
import csv,glob,collections
for f in glob.glob("$W/p/**/*kernel_trace.csv", recursive=True):
    d=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "gemm" in r["Kernel_Name"]: d[r["Kernel_Name"][:90]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
    for k,v in d.items(): print("   kernel-trace:", k, "launches", len(v), "avg ms", sum(v)/len(v))
PY
  rm -rf $W/p
done
