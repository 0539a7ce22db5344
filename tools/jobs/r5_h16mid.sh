# round 5: conv1 output of the UNet / BrushNet ResBlocks in h16 (ResBlock.H16_MID): parity on the GPU, then an interleaved A/B of the bench line on one box
O=gpurun_out/r5_h16mid; mkdir -p $O; rm -f $O/parity.txt
VV_PARITY_REPORT=$O/parity.txt python -m pytest tests/test_configs_gpu.py tests/test_model_gpu.py tests/test_kernels_gpu.py -q -m gpu -x 2>&1 | tail -5 | tee $O/pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $O/parity.txt
for r in 1 2; do
  for v in 0 1; do
    python tools/bench_with.py ResBlock.H16_MID=$v -- --steps 4 --warmup 2 --no-cpu-baseline > $O/bench_${v}_${r}.json 2> $O/err.txt
    python -c "
import json; d=json.load(open('$O/bench_${v}_${r}.json')); k=d['kernel_times_s']; print('round $r H16_MID=$v', d['value'], d['ms_per_step'], 'groupnorm', k['groupnorm'], 'conv k3', k.get('conv_gemm[128x160,h16in,k3]'))" | tee -a $O/ab.txt
  done
done
