mkdir -p gpurun_out/r4p; O=gpurun_out/r4p
python bench.py --steps 1 --warmup 0 --denoise-steps 2 --no-cpu-baseline --profile-shapes --dump-kernels $O/k2.json > $O/bench2.json 2> $O/bench2.err
python tools/shape_table.py $O/k2.json | head -40
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "upconv" 2>&1 | tail -3
