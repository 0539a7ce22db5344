# Round 6: GroupNorm partials out of the fused chain tail (-> the fused motion module's pooled GroupNorm): correctness, the tail kernel with / without, pipeline A/B
O=gpurun_out/r6_tail_partials; mkdir -p $O
python -m pytest tests/test_chain_gpu.py tests/test_motion_gpu.py tests/test_model_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q -k "partials or chain or motion or denoiser_one_step or fused or two_stream or concurrent" 2>&1 | tail -4 | tee $O/pytest.txt
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $O/tail_ab.txt
import sys, torch
sys.path.insert(0, ".")
from videovanish_amd import hip, nn as vnn
from videovanish_amd.config import UNetConfig
from videovanish_amd.unet import sinusoidal_pos_emb
cfg = UNetConfig(); ctx = vnn.Ctx("cuda:0", "fp16", 0)
text = ctx.dev(torch.randn(77, 768), ctx.h16)
st = vnn.SpatialTransformer(ctx, "unet.down_blocks.0.attentions.0", 320, cfg, text)
mm = vnn.MotionModule(ctx, "unet.down_blocks.0.motion_modules.0", 320, cfg, ctx.dev(sinusoidal_pos_emb(cfg.motion_max_seq, 320)))
Fr, H, W = 32, 90, 160
x = torch.randn(Fr * H * W, 320, device="cuda")
def run(flag):
    return mm(st(x, Fr, H, W, want_gn=flag), Fr, H, W)
for r in range(3):
    for flag in (False, True):
        for _ in range(2): run(flag)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run(flag)
        e1.record(); torch.cuda.synchronize()
        print(f"round {r}: level-0 spatial transformer + motion module (32 x 90 x 160), tail partials={flag}: {e0.elapsed_time(e1) / 10:.3f} ms")
PY
