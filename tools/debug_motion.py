import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from videovanish_amd import nn as vnn
from videovanish_amd.config import UNetConfig
from videovanish_amd.unet import sinusoidal_pos_emb
from videovanish_amd.weights import SyntheticWeights

class Z(SyntheticWeights):
    def __init__(self, zero): super().__init__(0); self.zero = zero
    def linear(self, name, cin, cout, gain=1.0, bias=True):
        w, b = super().linear(name, cin, cout, gain, bias)
        if any(z in name for z in self.zero):
            w = w * 0
            if b is not None: b = b * 0
        return w, b

H, W, Fr, C = 6, 8, 32, 320
cfg = UNetConfig()
x = torch.randn(Fr * H * W, C, generator=torch.Generator().manual_seed(1)).cuda()
for zero in ([], ["ff.net"], ["attn1", "attn2"], ["attn1", "attn2", "ff.net"], ["attn2", "ff.net"], ["attn1", "ff.net"]):
    ctx = vnn.Ctx("cuda:0", "fp16", 0, weights=Z(zero))
    mod = vnn.MotionModule(ctx, "unet.down_blocks.0.motion_modules.0", C, cfg, ctx.dev(sinusoidal_pos_emb(32, C)))
    vnn.MotionModule.FUSED = True; a = mod(x, Fr, H, W)
    vnn.MotionModule.FUSED = False; b = mod(x, Fr, H, W)
    d = (a - b).abs()
    print(f"zeroed {zero}: rel max {float(d.max() / b.abs().max()):.2e}; worst channel {int(d.max(0).values.argmax())}, frac elements > 1e-3*max: {float((d > 1e-3 * b.abs().max()).float().mean()):.4f}")
