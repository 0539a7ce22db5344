mkdir -p gpurun_out/r4gn; O=gpurun_out/r4gn
A=$PWD/videovanish_amd/csrc/libvvhip.so; B1=$PWD/videovanish_amd/csrc/ab/libvvhip_gnrev1.so; B2=$PWD/videovanish_amd/csrc/ab/libvvhip_gnrev2.so
for r in 1 2; do for v in A B1 B2; do
  eval L=\$$v
  VV_LIB_PATH=$L python tools/bench_with_lib.py --steps 1 --warmup 0 --denoise-steps 4 --no-cpu-baseline --dump-kernels $O/k_${v}_$r.json > $O/b_${v}_$r.json 2> $O/e.txt
  python - <<PY
import json
k=json.load(open("$O/k_${v}_$r.json")); b=json.load(open("$O/b_${v}_$r.json"))
print("$v round $r: groupnorm %.4f s (%d launches), motion:groupnorm %.4f s, all kernels %.3f s, chunk %.3f s" % (k["groupnorm"][1], k["groupnorm"][0], k["motion:groupnorm"][1], sum(v[1] for v in k.values()), b["ms_per_step"] / 1e3))
PY
done; done
