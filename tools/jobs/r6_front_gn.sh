# Round 6: (1) store forms of the block front (videovanish_amd/csrc/ab/fs{0,1,2,3}.so = -DVV_FRONT_STORE=0..3), (2) GroupNorm with four rows in flight per thread
# (ab/gnold.so = the tree with the previous vv_norm.hip), steady loops, interleaved, one box; correctness first.
O=gpurun_out/r6_front_gn; mkdir -p $O
python -m pytest tests/test_chain_gpu.py tests/test_kernels_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q -k "chain or groupnorm or fused or norm" 2>&1 | tail -4 | tee $O/pytest.txt
for v in fs1 fs2 fs3; do echo -n "$v correctness: "; VV_LIB_PATH=videovanish_amd/csrc/ab/$v.so python tools/pytest_with_lib.py tests/test_chain_gpu.py -m gpu -x -q 2>&1 | tail -1; done | tee -a $O/pytest.txt
for r in 1 2 3; do
  for v in fs0 fs1 fs2 fs3; do
    echo -n "round $r $v: "; VV_LIB_PATH=videovanish_amd/csrc/ab/$v.so python tools/bench_chain.py fp16 2>&1 | grep -E "\(front\)" | sed 's/fp16 spatial chain front level 0//; s/of the MFMA peak//' | tr '\n' ' '; echo
  done
done | tee $O/front_ab.txt
for r in 1 2 3; do
  for v in gnold tree; do
    L=videovanish_amd/csrc/ab/$v.so; [ $v == tree ] && L=videovanish_amd/csrc/libvvhip.so
    echo "round $r $v:"; VV_LIB_PATH=$L python tools/bench_gn.py 2>&1 | grep groupnorm | sed 's/of algorithmic traffic.*//'
  done
done | tee $O/gn_ab.txt
