# Round 6: polynomial GELU (gelu_poly2, packed fp32, no transcendentals) in the GEGLU epilogue of the GEMM kernels against the scalar A&S form (ab/gelu_scalar.so = the tree with
# -DVV_GELU_SCALAR and the A&S gelu2 in the fused kernels = everything as in round 5; ab/fused_as.so = polynomial in the GEMM epilogue, A&S in the fused kernels): correctness, the GEGLU shapes in steady loops (product tile heuristic), then the pipeline; interleaved, one box.
O=gpurun_out/r6_gelu; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py tests/test_motion_gpu.py tests/test_chain_gpu.py -m gpu -x -q -k "geglu or gemm or motion or chain or linear" 2>&1 | tail -3 | tee $O/pytest.txt
for r in 1 2; do
  for v in gelu_scalar tree; do
    L=videovanish_amd/csrc/ab/$v.so; [ $v == tree ] && L=videovanish_amd/csrc/libvvhip.so
    echo "== round $r $v"; VV_BENCH_SECONDS=0.3 VV_BENCH_HINTS=0 VV_BENCH_ONLY=geglu VV_LIB_PATH=$L python tools/bench_gemm256.py fp16 2>&1 | grep -v amdgpu.ids | grep -i geglu
  done
done | tee $O/geglu_ab.txt
for r in 1 2; do
  for v in gelu_scalar fused_as tree; do
    L=videovanish_amd/csrc/ab/$v.so; [ $v == tree ] && L=videovanish_amd/csrc/libvvhip.so
    echo -n "round $r $v: "; VV_LIB_PATH=$L python tools/bench_with_lib.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events --no-power-trace 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
  done
done | tee $O/pipeline_ab.txt
for r in 1 2; do
  for v in fused_as tree; do
    L=videovanish_amd/csrc/ab/$v.so; [ $v == tree ] && L=videovanish_amd/csrc/libvvhip.so
    echo -n "round $r $v: "; VV_LIB_PATH=$L python tools/bench_chain.py fp16 2>&1 | grep -E "\(fused\)" | sed 's/fp16 spatial chain level 0//; s/of the MFMA peak//' | tr '\n' ' '; VV_LIB_PATH=$L python tools/bench_motion.py fp16 2>&1 | grep -i "fused" | head -1
  done
done | tee $O/fused_ab.txt
