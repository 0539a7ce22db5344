# Round 6: head-major attention output + one-burst QKV store of the block front -- correctness, then timing against the library of the round's start (ab/base.so), one box.
O=gpurun_out/r6_stores; mkdir -p $O
python -m pytest tests/test_chain_gpu.py tests/test_kernels_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q -k "chain or attention or fused or head_major" 2>&1 | tail -4 | tee $O/pytest.txt
python tools/bench_attn_o_layout.py 32 3 2>&1 | grep -v amdgpu.ids | tee $O/attn_o_layout.txt
for r in 1 2 3; do
  for v in base tree; do
    L=videovanish_amd/csrc/ab/$v.so; [ $v == tree ] && L=videovanish_amd/csrc/libvvhip.so
    echo -n "round $r $v: "; VV_LIB_PATH=$L python tools/bench_chain.py fp16 2>&1 | grep -E "\(fused\)|\(front\)" | sed 's/fp16 spatial chain level 0//; s/fp16 spatial chain front level 0//; s/of the MFMA peak//' | tr '\n' ' '; echo
  done
done | tee $O/chain_ab.txt
