mkdir -p gpurun_out/r4n; O=gpurun_out/r4n
F2=$PWD/videovanish_amd/csrc/ab/libvvhip_f2.so
VV_BENCH_CHECK=1 python tools/bench_attn_d40.py 4 fp16 80 3600 > $O/check.txt 2>&1
VV_BENCH_CHECK=1 VV_LIB_PATH=$F2 python tools/bench_attn_d40.py 4 fp16 80 3600 >> $O/check.txt 2>&1
VV_BENCH_CHECK=1 VV_LIB_PATH=$F2 python tools/bench_attn_d40.py 2 bf16 80 4096 >> $O/check.txt 2>&1
bash tools/ab_libs.sh $PWD/videovanish_amd/csrc/libvvhip.so $F2 3 tools/bench_attn_d40.py 32 fp16 80 3600 > $O/ab80.txt 2>&1
bash tools/ab_libs.sh $PWD/videovanish_amd/csrc/libvvhip.so $F2 2 tools/bench_attn_d40.py 8 fp16 80 4096 >> $O/ab80.txt 2>&1
# power / clock while the d40 kernel loops
(for i in $(seq 1 14); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Package Power|sclk"; sleep 0.3; done) > $O/power_d40.txt &
VV_BENCH_ITERS=600 python tools/bench_attn_d40.py 32 fp16 40 14400 > $O/d40_long.txt 2>&1
wait
(for i in $(seq 1 14); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Package Power|sclk"; sleep 0.3; done) > $O/power_d80.txt &
VV_BENCH_ITERS=4000 python tools/bench_attn_d40.py 32 fp16 80 3600 > $O/d80_long.txt 2>&1
wait
python bench.py --steps 1 --warmup 0 --denoise-steps 2 --no-cpu-baseline --profile-shapes --dump-kernels $O/k2.json > $O/bench2.json 2> $O/bench2.err
python bench.py --steps 1 --warmup 0 --denoise-steps 6 --no-cpu-baseline --profile-shapes --dump-kernels $O/k6.json > $O/bench6.json 2> $O/bench6.err
cat $O/check.txt $O/ab80.txt | grep -v amdgpu.ids
