# round 6: bench lines of the other BASELINE configurations on the final tree (one 1080p chunk, c2 in bf16, c3 as a fixed 256-frame clip host to host)
mkdir -p gpurun_out/r6cfg; O=gpurun_out/r6cfg
python bench.py --steps 1 --warmup 0 --height 1080 --width 1920 --no-cpu-baseline > $O/bench_c4_chunk.json 2> $O/e2.txt
python bench.py --frames 64 --height 480 --width 848 --dtype bf16 --no-cpu-baseline > $O/bench_c2_64f.json 2> $O/e3.txt
python bench.py --frames 256 --no-cpu-baseline > $O/bench_c3_256f.json 2> $O/e4.txt
for f in $O/*.json; do python -c "
import json,sys; d=json.load(open('$f')); print('$f', d.get('metric'), d.get('value'), d.get('ms_per_step'), (d.get('roofline') or {}).get('kernel'), (d.get('roofline') or {}).get('achieved'), d.get('per_rank_seconds'))"; done
