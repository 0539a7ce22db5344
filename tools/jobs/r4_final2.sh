# second evidence pass of round 4 (final tree): full GPU suite, driver-style bench, parity report, smoke
mkdir -p gpurun_out/r4final2; O=gpurun_out/r4final2
VV_PARITY_REPORT=$O/parity.txt python -m pytest tests -q -m gpu 2>&1 | tail -4 > $O/pytest.txt; cat $O/pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $O/parity.txt
python bench.py --steps 20 --warmup 5 > $O/bench_20steps.json 2> $O/bench_20steps.err; python -c "
import json; d=json.load(open('$O/bench_20steps.json')); print({k: d[k] for k in ('value','ms_per_step','roofline','temporal_block','job_tflops','power')})"
