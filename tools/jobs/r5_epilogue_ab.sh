# GEMM epilogue A/B (round 5, second session): every build of the library under videovanish_amd/csrc/ab/ on the GEMM shapes of a denoise step, product heuristic (hint 0),
# steady-state loops, interleaved rounds:  bash tools/jobs/r5_epilogue_ab.sh <tag> [rounds]
O=gpurun_out/r5_epi_ab_$1; mkdir -p $O
L=videovanish_amd/csrc
python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py -x -q -k "gemm or conv or precise or upconv or fused" 2>&1 | tail -3 | tee $O/pytest.txt      # (the in-tree build)
for r in $(seq 1 ${2:-2}); do
  for v in $(ls $L/ab/*.so); do
    echo "== round $r $(basename $v)"; VV_BENCH_SECONDS=0.3 VV_BENCH_HINTS=0 VV_LIB_PATH=$v python tools/bench_gemm256.py fp16 2>&1 | grep -v amdgpu.ids
  done
done | tee $O/ab.txt
