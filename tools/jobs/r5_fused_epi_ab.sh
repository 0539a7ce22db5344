# fused chain tail / motion module: block residual requested before the first store (round 5, second session): bash tools/jobs/r5_fused_epi_ab.sh <tag> [rounds]
O=gpurun_out/r5_fused_epi_$1; mkdir -p $O
python -m pytest tests/test_chain_gpu.py tests/test_motion_gpu.py -x -q 2>&1 | tail -3 | tee $O/pytest.txt      # (the in-tree build)
L=videovanish_amd/csrc
for r in $(seq 1 ${2:-3}); do
  for v in $(ls $L/ab/*.so); do
    echo -n "round $r $(basename $v): "; VV_LIB_PATH=$v python tools/bench_chain.py fp16 2>&1 | grep -E "\(fused\)|\(front\)" | sed 's/fp16 spatial chain level 0//; s/fp16 spatial chain front level 0//; s/of the MFMA peak//' | tr '\n' ' '
    VV_LIB_PATH=$v python tools/bench_motion.py fp16 2>&1 | grep -i "fused" | head -1
  done
done | tee $O/ab.txt
