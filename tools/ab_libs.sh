#!/bin/bash
# interleaved A/B of two builds of libvvhip.so with one benchmark script on ONE device: tools/ab_libs.sh <libA.so> <libB.so> <rounds> <script> [args...]
A=$1; B=$2; R=$3; shift 3
for r in $(seq 1 $R); do
  echo -n "round $r A: "; VV_LIB_PATH=$A python "$@" 2>&1 | tail -1
  echo -n "round $r B: "; VV_LIB_PATH=$B python "$@" 2>&1 | tail -1
done
