# Round 6: GroupNorm statistics out of the producing convolution's staged epilogue (vv_conv_params.gn_partials; ResBlock.GN_FROM_EPILOGUE) -- correctness,
# then the A/B: the conv2 shapes with / without partials in steady loops, the priced pass's groupnorm / conv totals, and the pipeline, interleaved on one box.  Also: the
# staged contiguous O store of attn40q2 (head-major output) against the library of the round's start.
O=gpurun_out/r6_gn_partials; mkdir -p $O
python -m pytest tests/test_chain_gpu.py tests/test_kernels_gpu.py tests/test_fullsize_gpu.py tests/test_model_gpu.py -m gpu -x -q -k "chain or groupnorm or partials or head_major or attention or fused or denoiser_one_step or brushnet" 2>&1 | tail -4 | tee $O/pytest.txt
python tools/bench_attn_o_layout.py 32 3 2>&1 | grep -v amdgpu.ids | tee $O/attn_o_layout.txt
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $O/conv2_ab.txt
import sys, torch
sys.path.insert(0, ".")
from videovanish_amd import hip, nn as vnn
ctx = vnn.Ctx("cuda:0", "fp16", 0)
for (Fr, H, W, C) in ((32, 90, 160, 320), (32, 45, 80, 640)):
    conv = vnn.Conv(ctx, "unet.down_blocks.0.resnets.0.conv2", C, C)
    x = torch.randn(Fr * H * W, C, device="cuda").to(ctx.h16); res = torch.randn(Fr * H * W, C, device="cuda")
    g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    def run(flag):
        out, _, _ = conv(x, Fr, H, W, res0=res, gn_partials=flag)
        return hip.groupnorm(ctx.dt, out, g, b, 32, 1e-6, F=Fr, HW=H * W, partials=getattr(out, "vv_gn", None))
    for r in range(3):
        for flag in (False, True):
            for _ in range(3): run(flag)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): run(flag)
            e1.record(); torch.cuda.synchronize()
            print(f"round {r} C={C} {H}x{W}: conv2 (3x3 + residual) + GroupNorm, partials={flag}: {e0.elapsed_time(e1) / 20:.3f} ms")
PY
for r in 1 2; do
  for v in 0 1; do
    echo -n "round $r GN_FROM_EPILOGUE=$v: "; python tools/bench_with.py ResBlock.GN_FROM_EPILOGUE=$v -- --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events --no-power-trace 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
  done
done | tee $O/pipeline_ab.txt
