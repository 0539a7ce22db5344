# Round 6: GroupNorm partials from conv1 (-> norm2) AND conv2 (-> the spatial transformer's GroupNorm): correctness, then the pipeline A/B (ResBlock.GN_FROM_EPILOGUE 0 / 1), interleaved, one box
O=gpurun_out/r6_gn_partials2; mkdir -p $O
python -m pytest tests/test_chain_gpu.py tests/test_model_gpu.py tests/test_configs_gpu.py -m gpu -x -q -k "partials or denoiser_one_step or brushnet or full_architecture_one_step or parity_50_steps or two_stream or concurrent" 2>&1 | tail -4 | tee $O/pytest.txt
for r in 1 2 3; do
  for v in 0 1; do
    echo -n "round $r GN_FROM_EPILOGUE=$v: "; python tools/bench_with.py ResBlock.GN_FROM_EPILOGUE=$v -- --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events --no-power-trace 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
  done
done | tee $O/pipeline_ab.txt
