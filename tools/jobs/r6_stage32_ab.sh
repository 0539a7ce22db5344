# Round 6: in-pipeline A/B of the STAGED fp32 epilogue of round 5 (out-projections, zero convolutions, 3x3 + residual): the tree's library against the same tree
# built with -DVV_NO_STAGE_F32 (fp32 strips stored from the accumulator layout), GroupNorm partials off in both arms (they need the staged tile), interleaved, one box.
O=gpurun_out/r6_stage32; mkdir -p $O
for r in 1 2 3; do
  for v in tree nostage32; do
    L=videovanish_amd/csrc/ab/$v.so; [ $v == tree ] && L=videovanish_amd/csrc/libvvhip.so
    echo -n "round $r $v: "; VV_LIB_PATH=$L python tools/bench_with.py ResBlock.GN_FROM_EPILOGUE=0 -- --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events --no-power-trace 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
  done
done | tee $O/pipeline_ab.txt
