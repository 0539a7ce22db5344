# motion module: product build vs builds under csrc/ab (interleaved): bash tools/jobs/r5_motion_ab.sh <tag> [rounds]
O=gpurun_out/r5_motion_ab_$1; mkdir -p $O
L=videovanish_amd/csrc
for r in $(seq 1 ${2:-2}); do
  for v in $(ls $L/ab/*.so) $L/libvvhip.so; do
    echo -n "round $r $(basename $v): "; case $v in *_rs*) LAY=rowsplit;; *) LAY=tokens;; esac; VV_MOTION_LAYOUT=$LAY VV_LIB_PATH=$v python tools/bench_motion.py fp16 2>&1 | grep -E "\(fused\)|rel max" | sed 's/fp16 motion module level 0//; s/of the MFMA peak//' | tr '\n' ' '; echo
  done
done | tee $O/ab.txt
VV_LIB_PATH= python -m pytest tests/test_motion_gpu.py -x -q 2>&1 | tail -2
