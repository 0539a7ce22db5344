# Round 6, first measurement pass (one box): per-shape kernel table of a denoise step (2- and 6-step runs subtracted), then the in-pipeline A/B of three builds of the library
# under videovanish_amd/csrc/ab/ (base = the tree; stage_h16 = -DVV_STAGE_H16; nopin = the staged epilogue without the round-6 order pins), interleaved, two rounds.
O=gpurun_out/r6_survey; mkdir -p $O
python bench.py --steps 1 --warmup 1 --denoise-steps 2 --no-cpu-baseline --no-power-trace --profile-shapes --dump-kernels $O/k2.json > $O/b2.json 2> $O/b2.err
python bench.py --steps 1 --warmup 1 --denoise-steps 6 --no-cpu-baseline --no-power-trace --profile-shapes --dump-kernels $O/k6.json > $O/b6.json 2> $O/b6.err
python tools/shape_table.py $O/k2.json $O/k6.json 2 6 > $O/shape_table.txt 2>&1
for r in 1 2; do
  for v in base stage_h16 nopin; do
    echo -n "round $r $v: "; VV_LIB_PATH=videovanish_amd/csrc/ab/$v.so python tools/bench_with_lib.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events --no-power-trace 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
  done
done | tee $O/ab.txt
