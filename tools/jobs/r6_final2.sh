# round 6, closing pass on the final tree: attention tests, PMC traffic again (the scratch array of the attention prologue is gone), the staged-fp32 pipeline A/B, a second 20-step line
O=gpurun_out/r6final2; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py tests/test_chain_gpu.py -m gpu -x -q -k "attention or head_major or chain" 2>&1 | tail -3 | tee $O/pytest.txt
export TMPDIR=/tmp; W=/tmp/vvprof_r6b; rm -rf $W; mkdir -p $W; ROOT=$(pwd); cd /tmp
P2="--steps 1 --warmup 0 --denoise-steps 2 --no-cpu-baseline --no-kernel-events --no-power-trace"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $W/fetch -o f -- python3 $ROOT/bench.py $P2 > $ROOT/$O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $W/write -o w -- python3 $ROOT/bench.py $P2 > $ROOT/$O/pmc_write.log 2>&1
F=$(find $W/fetch -name "*counter_collection.csv" | head -1); WR=$(find $W/write -name "*counter_collection.csv" | head -1)
python3 $ROOT/tools/pmc_summary.py $F $WR $ROOT/$O/traffic.json > $ROOT/$O/traffic_top.txt 2>&1
cd $ROOT
bash tools/jobs/r6_stage32_ab.sh
python bench.py --steps 20 --warmup 5 > $O/bench_20steps_b.json 2> $O/bench_20steps_b.err
python -c "
import json; d=json.load(open('$O/bench_20steps_b.json')); print({k: d[k] for k in ('value','ms_per_step','temporal_block','job_tflops')}, d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
