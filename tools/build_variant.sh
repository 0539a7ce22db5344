#!/bin/bash
# Lab: build a VARIANT of libvvhip.so into videovanish_amd/csrc/ab/<name>.so (git-ignored; travels with gpurun) without touching the product build:
#   tools/build_variant.sh <name> [--src DIR] [extra hipcc flags, e.g. -DVV_STAGE_H16]
# --src DIR: take the sources from DIR (a copy of videovanish_amd/csrc at another revision: `git worktree` / `git archive`), default: the tree's own.
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/videovanish_amd/csrc
if [ "$1" == "--src" ]; then SRC=$2; shift 2; fi
OUT=$ROOT/videovanish_amd/csrc/ab; B=/tmp/vv_variant_$NAME
mkdir -p $OUT $B
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -I$ROOT/include $*"
cd $SRC
pids=()
cc() { local src=$1 obj=$2; shift 2; hipcc $FLAGS "$@" -c $src.hip -o $B/$obj.o & pids+=($!); }
for f in vv_gemm vv_gemm256; do cc $f ${f}_bf16 -DVV_DT_ONLY=0; cc $f ${f}_f16 -DVV_DT_ONLY=1; done
for f in vv_api vv_motion vv_chain vv_norm vv_elem vv_image vv_flow vv_deform vv_sam2; do cc $f $f; done
cc vv_attn vv_attn_small -DVV_ATTN_PART=0 -mllvm -amdgpu-mfma-vgpr-form
cc vv_attn32 vv_attn32 -mllvm -amdgpu-mfma-vgpr-form
cc vv_attn vv_attn_large -DVV_ATTN_PART=1
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/$NAME.so $B/*.o
echo "built $OUT/$NAME.so"
