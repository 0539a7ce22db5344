mkdir -p gpurun_out/r4u; O=gpurun_out/r4u
python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "conv_gemm or gemm256" 2>&1 | tail -2
python bench.py --steps 1 --warmup 0 --denoise-steps 2 --no-cpu-baseline --profile-shapes --dump-kernels $O/k2.json > $O/bench2.json 2> $O/bench2.err
python bench.py --steps 1 --warmup 0 --denoise-steps 6 --no-cpu-baseline --profile-shapes --dump-kernels $O/k6.json > $O/bench6.json 2> $O/bench6.err
python tools/shape_table.py $O/k2.json $O/k6.json 2 6 | head -42
python tools/ab_schedule.py 3 2s > $O/ab.txt 2>&1; grep -E "s/chunk|Error|error" $O/ab.txt
