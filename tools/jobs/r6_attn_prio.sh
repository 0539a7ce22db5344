# Round 6: static issue priority for every second block in attn40q2_kernel (ab/attn_prio1.so = -DVV_ATTN_PRIO=1) against the product, 32 frames x 8 heads x 14400 tokens, interleaved
O=gpurun_out/r6_attn_prio; mkdir -p $O
for r in 1 2 3; do
  for v in tree attn_prio1; do
    L=videovanish_amd/csrc/ab/$v.so; [ $v == tree ] && L=videovanish_amd/csrc/libvvhip.so
    echo -n "round $r $v: "; VV_LIB_PATH=$L python tools/bench_attn_o_layout.py 32 2 2>&1 | grep "head-major" | tail -1
  done
done | tee $O/ab.txt
