#!/bin/bash
# Build libvvhip.so for gfx950 (MI355X).  Usage: ./build.sh [-j N]
set -e
cd "$(dirname "$0")"
JOBS=${JOBS:-8}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
mkdir -p build
pids=()
for f in vv_api vv_gemm vv_norm vv_attn vv_elem vv_image vv_flow; do
  [ -f $f.hip ] || continue
  if [ ! -f build/$f.o ] || [ $f.hip -nt build/$f.o ] || [ vv_common.h -nt build/$f.o ] || [ ../../include/vvhip.h -nt build/$f.o ]; then
    hipcc $FLAGS -c $f.hip -o build/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o libvvhip.so build/*.o
echo "built $(pwd)/libvvhip.so"
