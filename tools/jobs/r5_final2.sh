# round 5, second session, last evidence pass on the final tree (reduced: the c5 / reference-defaults / 2-step lines stay those of tools/jobs/r5_final.sh run earlier in the session):
# full GPU suite with the parity report + smoke, rocprofv3 summaries (kernel stats, PMC traffic, MFMA busy), the driver-style bench line
bash tools/jobs/r5_gpu_tests.sh
mkdir -p gpurun_out/r5final2; O=gpurun_out/r5final2
bash tools/profile_round.sh r5b > $O/profile_round.log 2>&1
python bench.py --steps 20 --warmup 5 > $O/bench_20steps.json 2> $O/bench_20steps.err
python -c "
import json; d=json.load(open('$O/bench_20steps.json')); print({k: d[k] for k in ('value','ms_per_step','host_to_host','roofline','temporal_block','job_tflops','power')})"
