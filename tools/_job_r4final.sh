# round-4 evidence pass on the final tree (one box): GPU tests of the touched kernels, driver-style bench, parity report, rocprof / PMC summaries
mkdir -p gpurun_out/r4final; O=gpurun_out/r4final
python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py -q -m gpu -k "attention or upconv or conv_gemm" 2>&1 | tail -3 > $O/pytest_kernels.txt; cat $O/pytest_kernels.txt
python bench.py --steps 20 --warmup 5 > $O/bench_20steps.json 2> $O/bench_20steps.err; python -c "
import json; d=json.load(open('$O/bench_20steps.json')); print({k: d[k] for k in ('value','ms_per_step','roofline','temporal_block','job_tflops')})"
VV_PARITY_REPORT=$O/parity.txt python -m pytest tests/test_configs_gpu.py -q -m gpu 2>&1 | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $O/parity.txt
bash tools/profile_round.sh r4final > $O/profile_round.log 2>&1; tail -3 $O/profile_round.log
