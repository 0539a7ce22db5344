#!/bin/bash
# A/B of the d = 40 spatial attention variants (lab build: VV_AB=1 videovanish_amd/csrc/build.sh), interleaved rounds on ONE device.
# usage: tools/attn_ab.sh "50 51 52" [frames] [rounds]
VARS=${1:-"50 51"}; B=${2:-8}; R=${3:-3}
for r in $(seq 1 $R); do
  for v in $VARS; do
    echo -n "round $r variant $v: "; VV_ATTN_VARIANT=$v python tools/bench_attn_d40.py $B fp16 2>&1 | tail -1
  done
done
