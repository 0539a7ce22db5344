# Round 6: block shapes of the d = 160 spatial attention (level 2) in a lab build (ab/attn160lab.so, VV_ATTN160_FORM=0..3), interleaved
O=gpurun_out/r6_attn160; mkdir -p $O
for r in 1 2 3; do
  for f in 0 1 2 3; do
    echo -n "round $r form $f: "; VV_ATTN160_FORM=$f VV_BENCH_ITERS=40 VV_BENCH_WARM=5 VV_BENCH_CHECK=1 VV_LIB_PATH=videovanish_amd/csrc/ab/attn160lab.so python tools/bench_attn_d40.py 32 fp16 160 920 2>&1 | grep -v amdgpu.ids | tail -1
  done
done | tee $O/ab.txt
