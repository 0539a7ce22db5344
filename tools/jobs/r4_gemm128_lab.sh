# steady-state A/B of the 128-row GEMM family's switches (lab build of the library, VV_AB) on the shapes that matter
mkdir -p gpurun_out/r4v; O=gpurun_out/r4v/gemm128_lab.txt; : > $O
export VV_LIB_PATH=$PWD/videovanish_amd/csrc/ab/libvvhip_lab.so VV_BENCH_SECONDS=0.4 VV_BENCH_HINTS=1
export VV_BENCH_ONLY="out   L1|out   L2|qkv   L1|ff2   L1|out   L0|conv3 L0 320->320 +res|conv3 L0 640->320|conv3 L1 640->640|conv3 L3 1280|vae conv3 256|vae conv3 512->512 180"
for cfg in "" "VV_GEMM_PREF128=1" "VV_GEMM_NO_OCC4=1" "VV_GEMM_NO_LIN=1" "VV_GEMM_NO_HALO=1" "VV_GEMM_N320=1" "VV_GEMM_SPLIT=1" "VV_GEMM_NO_FAST9=1"; do
  echo "=== ${cfg:-default}" >> $O
  env $cfg python tools/bench_gemm256.py fp16 2>&1 | grep -v amdgpu.ids >> $O
done
cat $O | cut -c1-120
