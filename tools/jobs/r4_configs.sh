mkdir -p gpurun_out/r4cfg; O=gpurun_out/r4cfg
python bench.py --steps 2 --warmup 1 --denoise-steps 2 --no-cpu-baseline > $O/bench_2step.json 2> $O/e1.txt
python bench.py --steps 1 --warmup 0 --height 1080 --width 1920 --no-cpu-baseline > $O/bench_c4_chunk.json 2> $O/e2.txt
python bench.py --frames 64 --height 480 --width 848 --dtype bf16 --no-cpu-baseline > $O/bench_c2_64f.json 2> $O/e3.txt
python bench.py --frames 256 --no-cpu-baseline --denoise-steps 2 > $O/bench_c3_256f_2step.json 2> $O/e4.txt
for f in $O/*.json; do python -c "
import json,sys; d=json.load(open('$f')); print('$f', d.get('metric'), d.get('value'), d.get('ms_per_step'), (d.get('roofline') or {}).get('kernel'), (d.get('roofline') or {}).get('achieved'))"; done
