# round 6 evidence pass on the final tree: rocprofv3 summaries (kernel stats, PMC traffic, MFMA busy), driver-style bench, c5 line, reference-defaults line, 2-step line
mkdir -p gpurun_out/r6final; O=gpurun_out/r6final
bash tools/profile_round.sh r6 > $O/profile_round.log 2>&1
python bench.py --steps 20 --warmup 5 > $O/bench_20steps.json 2> $O/bench_20steps.err
python -c "
import json; d=json.load(open('$O/bench_20steps.json')); print({k: d[k] for k in ('value','ms_per_step','host_to_host','roofline','temporal_block','job_tflops','power')})"
python bench.py --prior raft --dilate 8 --steps 4 --warmup 1 > $O/bench_c5.json 2> $O/bench_c5.err
python -c "
import json; d=json.load(open('$O/bench_c5.json')); print({k: d[k] for k in ('value','ms_per_step','job_tflops')}, d['prior']['seconds_per_32_frames'])"
python bench.py --prior raft --reference-defaults --steps 2 --warmup 1 --no-kernel-events > $O/bench_refdefaults.json 2> $O/bench_refdefaults.err
python -c "
import json; d=json.load(open('$O/bench_refdefaults.json')); print({k: d[k] for k in ('value','ms_per_step')})"; tail -3 $O/bench_refdefaults.err
python bench.py --steps 4 --warmup 2 --denoise-steps 2 --no-cpu-baseline > $O/bench_2step.json 2> $O/bench_2step.err
python -c "
import json; d=json.load(open('$O/bench_2step.json')); print({k: d[k] for k in ('value','ms_per_step')})"
