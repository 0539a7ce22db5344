#!/bin/bash
# Build libvvhip.so for gfx950 (MI355X).
set -e
cd "$(dirname "$0")"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
mkdir -p build
pids=()
compile() {   # compile <src> <obj> [extra flags]
  local src=$1 obj=$2; shift 2
  if [ ! -f build/$obj.o ] || [ $src.hip -nt build/$obj.o ] || [ vv_common.h -nt build/$obj.o ] || [ vv_attn_common.h -nt build/$obj.o ] || [ vv_gemm_epilogue.h -nt build/$obj.o ] || [ vv_chain_lab.h -nt build/$obj.o ] || [ vv_motion_lab.h -nt build/$obj.o ] || [ ../../include/vvhip.h -nt build/$obj.o ] || [ build.sh -nt build/$obj.o ]; then
    hipcc $FLAGS "$@" -c $src.hip -o build/$obj.o &
    pids+=($!)
  fi
}
# (vv_motion_lab.h: row-split lab form of the motion module, -DVV_MOTION_FORM=1; vv_chain_lab.h: lab forms of the fused chain tail, compiled only by hand with -DVV_CHAIN_FORM=0 / 2 -- tools/jobs/r5_chain_ab.sh; vv_conv3.hip, vv_attn_lab.hip: VV_AB=1 only)
# VV_AB=1 ./build.sh builds the lab variant: environment-selected A/B kernels + the opt-in vv_conv3 kernel
AB=""
SRCS="vv_api vv_motion vv_chain vv_norm vv_elem vv_image vv_flow vv_deform vv_sam2"
if [ -n "$VV_AB" ]; then AB="-DVV_AB"; SRCS="$SRCS vv_conv3"; else rm -f build/vv_conv3.o; fi
if [ "$(cat build/.ab 2>/dev/null)" != "$AB" ]; then rm -f build/*.o; echo "$AB" > build/.ab; fi
# the two GEMM sources hold every tile form x loader mode x operand type: one translation unit per operand type (BF16 / F16) halves the longest pole
rm -f build/vv_gemm.o build/vv_gemm256.o
for f in vv_gemm vv_gemm256; do
  compile $f ${f}_bf16 -DVV_DT_ONLY=0 $AB
  compile $f ${f}_f16 -DVV_DT_ONLY=1 $AB
done
for f in $SRCS; do
  [ -f $f.hip ] && compile $f $f $AB
done
# attention, small head dims: MFMA results feed VALU code (softmax) every tile -> keep accumulators in arch VGPRs
# (no v_accvgpr_read/write traffic); large head dims need the AGPR half of the register file
compile vv_attn vv_attn_small -DVV_ATTN_PART=0 -mllvm -amdgpu-mfma-vgpr-form $AB
compile vv_attn32 vv_attn32 -mllvm -amdgpu-mfma-vgpr-form $AB      # d = 40 / 80 on the 32x32x16 MFMA (the dominant kernels)
compile vv_attn vv_attn_large -DVV_ATTN_PART=1 $AB
if [ -n "$VV_AB" ]; then      # lab build: every attention A/B variant and timing probe (VV_ATTN_VARIANT), kept out of the product sources
  compile vv_attn_lab vv_attn_lab_small -DVV_ATTN_PART=0 -mllvm -amdgpu-mfma-vgpr-form $AB
  compile vv_attn_lab vv_attn_lab_large -DVV_ATTN_PART=1 $AB
else
  rm -f build/vv_attn_lab_small.o build/vv_attn_lab_large.o
fi
for p in "${pids[@]}"; do wait $p; done
rm -f build/vv_attn.o build/vv_attn_lab.o
hipcc --offload-arch=gfx950 -shared -fPIC -o libvvhip.so build/*.o
echo "built $(pwd)/libvvhip.so"
# host-side frame I/O codec (FFV1, plain C, no GPU): libvvio.so
if [ ! -f libvvio.so ] || [ vv_ffv1.c -nt libvvio.so ]; then
  gcc -O2 -std=c99 -fPIC -shared -Wall -o libvvio.so vv_ffv1.c
fi
