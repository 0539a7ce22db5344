# round 5: default bench line (host_to_host beside value, row-split chain tail in the pipeline) and the new c5 mode (--prior raft --dilate 8)
O=gpurun_out/r5_bench_modes; mkdir -p $O
python bench.py --steps 4 --warmup 2 > $O/bench_default.json 2> $O/bench_default.err; tail -3 $O/bench_default.err
python -c "
import json; d=json.load(open('$O/bench_default.json')); print({k: d[k] for k in ('value','ms_per_step','host_to_host','roofline','temporal_block','job_tflops')}); print({k:v for k,v in d['kernel_times_s'].items() if 'fused' in k})"
python bench.py --prior raft --dilate 8 --steps 2 --warmup 1 > $O/bench_c5.json 2> $O/bench_c5.err; tail -3 $O/bench_c5.err
python -c "
import json; d=json.load(open('$O/bench_c5.json')); print({k: d[k] for k in ('value','ms_per_step','roofline','job_tflops')}); print(json.dumps(d['prior'], indent=1))"
