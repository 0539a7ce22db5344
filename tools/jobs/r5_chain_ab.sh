# interleaved timing of every build of the library under videovanish_amd/csrc/ab/ plus the product build, fused chain tail: bash tools/jobs/r5_chain_ab.sh <tag> [rounds]
O=gpurun_out/r5_chain_ab_$1; mkdir -p $O
VV_LIB_PATH= python -m pytest tests/test_chain_gpu.py -x -q 2>&1 | tail -3 | tee $O/pytest.txt
L=videovanish_amd/csrc
for r in $(seq 1 ${2:-2}); do
  for v in $(ls $L/ab/*.so) $L/libvvhip.so; do
    echo -n "round $r $(basename $v): "; case $v in *_tok*) LAY=tokens;; *_cs*) LAY=columns;; *) LAY=rowsplit;; esac; VV_CHAIN_LAYOUT=$LAY VV_LIB_PATH=$v python tools/bench_chain.py fp16 2>&1 | grep -E "\(fused\)|rel max|\(front\)" | sed 's/fp16 spatial chain level 0//; s/fp16 spatial chain front level 0//; s/of the MFMA peak//' | tr '\n' ' '; echo
  done
done | tee $O/ab.txt
