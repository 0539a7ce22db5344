# effective clock + matrix-pipe / VALU busy of one attention variant: bash tools/pmc_clock.sh <variant> <kernel-name filter> [frames]
export TMPDIR=/tmp; W=/tmp/pmcc; rm -rf $W; mkdir -p $W; R=$(pwd); V=$1; F=$2; B=${3:-32}; cd /tmp
export VV_ATTN_VARIANT=$V
python3 $R/tools/bench_attn_d40.py $B fp16
for C in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $W/p -o x -- python3 $R/tools/bench_attn_d40.py $B fp16 > /dev/null 2>&1
  python3 $R/tools/pmc_sum.py $W/p $F | head -8
  python3 - <<PY
import csv,glob
for f in glob.glob("$W/p/**/*kernel_trace.csv", recursive=True):
    rows=[r for r in csv.DictReader(open(f)) if "$F" in r["Kernel_Name"]]
    if rows:
        d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6 for r in rows]
        print("   kernel-trace: launches", len(d), "avg ms", sum(d)/len(d))
PY
  rm -rf $W/p
done
