# round 5: the fused chain tail at 8 waves x 16 tokens (two per SIMD, staggered) against the round-3 form (4 waves x 32 tokens): correctness, then interleaved timing on one device
O=gpurun_out/r5_chain_tt; mkdir -p $O
python -m pytest tests/test_chain_gpu.py -x -q 2>&1 | tail -3 | tee $O/pytest.txt
L=videovanish_amd/csrc
for r in 1 2 3; do
  for v in ab/libvvhip_tt2.so ab/libvvhip_lag0.so ab/libvvhip_lag1.so libvvhip.so; do
    echo -n "round $r $v: "; VV_LIB_PATH=$L/$v python tools/bench_chain.py fp16 2>&1 | grep -E "\(fused\)|rel max" | tr '\n' ' '; echo
  done
done | tee $O/ab.txt
