# Round 6: persistent form of the 2-phase 256 x 256 GEMM (VV_GEMM256_PERSIST=1; plain linears, even number of k tiles): correctness against the one-tile-per-block form, then the
# GEGLU / 256 x 256 shapes in steady loops, interleaved
O=gpurun_out/r6_persist; mkdir -p $O
VV_GEMM256_PERSIST=1 timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "gemm or geglu or linear or tile" 2>&1 | tail -3 | tee $O/pytest.txt
for r in 1 2 3; do
  for v in 0 1; do
    echo "== round $r VV_GEMM256_PERSIST=$v"; VV_GEMM256_PERSIST=$v VV_BENCH_SECONDS=0.3 VV_BENCH_HINTS=0 VV_BENCH_ONLY=geglu timeout 300 python tools/bench_gemm256.py fp16 2>&1 | grep -v amdgpu.ids | grep -i "geglu L1\|geglu L2"
  done
done | tee $O/geglu_ab.txt
