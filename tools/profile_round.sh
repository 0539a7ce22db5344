#!/bin/bash
# Collect the round's profiles on the GPU box (run from the repo root):  bash tools/profile_round.sh r2
#   1. rocprofv3 --kernel-trace --stats of the bench command (--one-stream: the schedule the pricing pass runs, kernels do not overlap)            -> gpurun_out/prof_<tag>/kernel_stats.csv + bench_under_rocprof.json
#   2. rocprofv3 --pmc FETCH_SIZE, then WRITE_SIZE (separate passes)     -> traffic.json (tools/pmc_summary.py; gfx950 x2 FETCH correction)
#   3. rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE          -> mfma_busy.txt (tools/pmc_mfma_busy.py)
# Raw traces stay in /tmp on the box (hundreds of MB); only the summaries are copied into gpurun_out/.
TAG=${1:-r2}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
W=/tmp/vvprof_$TAG
rm -rf $W; mkdir -p $W
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $W/stats -o bench -- python3 $ROOT/bench.py --steps 2 --warmup 1 --one-stream --no-cpu-baseline --no-power-trace > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
find $W/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
P2="--steps 1 --warmup 0 --denoise-steps 2 --no-cpu-baseline --no-kernel-events --no-power-trace"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $W/fetch -o f -- python3 $ROOT/bench.py $P2 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $W/write -o w -- python3 $ROOT/bench.py $P2 > $OUT/pmc_write.log 2>&1
F=$(find $W/fetch -name "*counter_collection.csv" | head -1); WR=$(find $W/write -name "*counter_collection.csv" | head -1)
python3 $ROOT/tools/pmc_summary.py $F $WR $OUT/traffic.json > $OUT/traffic_top.txt 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $W/mfma -o m -- python3 $ROOT/bench.py $P2 > $OUT/pmc_mfma.log 2>&1
python3 $ROOT/tools/pmc_mfma_busy.py $W/mfma $OUT/mfma_busy.txt > /dev/null 2>&1
ls -la $OUT
