#!/bin/bash
# Build libvvhip.so for gfx950 (MI355X).
set -e
cd "$(dirname "$0")"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
mkdir -p build
pids=()
compile() {   # compile <src> <obj> [extra flags]
  local src=$1 obj=$2; shift 2
  if [ ! -f build/$obj.o ] || [ $src.hip -nt build/$obj.o ] || [ vv_common.h -nt build/$obj.o ] || [ vv_gemm_epilogue.h -nt build/$obj.o ] || [ ../../include/vvhip.h -nt build/$obj.o ] || [ build.sh -nt build/$obj.o ]; then
    hipcc $FLAGS "$@" -c $src.hip -o build/$obj.o &
    pids+=($!)
  fi
}
for f in vv_api vv_gemm vv_conv3 vv_norm vv_elem vv_image vv_flow; do
  [ -f $f.hip ] && compile $f $f
done
# attention, small head dims: MFMA results feed VALU code (softmax) every tile -> keep accumulators in arch VGPRs
# (no v_accvgpr_read/write traffic); large head dims need the AGPR half of the register file
compile vv_attn vv_attn_small -DVV_ATTN_PART=0 -mllvm -amdgpu-mfma-vgpr-form
compile vv_attn vv_attn_large -DVV_ATTN_PART=1
for p in "${pids[@]}"; do wait $p; done
rm -f build/vv_attn.o
hipcc --offload-arch=gfx950 -shared -fPIC -o libvvhip.so build/*.o
echo "built $(pwd)/libvvhip.so"
