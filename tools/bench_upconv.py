#!/usr/bin/env python3
"""nn.UpConv2x (four 2x2 parity convolutions with a scattered store) at the 720p shapes: UNet 640 ch 45x80 -> 90x160 (32 frames), VAE decoder 512 ch 180x320 -> 360x640 and
256 ch 360x640 -> 720x1280 (4 frames).  VV_LIB_PATH: another build of the library."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from videovanish_amd import hip, nn as vnn
if os.environ.get("VV_LIB_PATH"):
    hip._LIB_PATH = os.environ["VV_LIB_PATH"]
ctx = vnn.Ctx("cuda:0", sys.argv[1] if len(sys.argv) > 1 else "fp16", 0)
for name, F, H, W, C in (("unet 640 45x80", 32, 45, 80, 640), ("vae 512 180x320", 4, 180, 320, 512), ("vae 256 360x640", 4, 360, 640, 256)):
    up = vnn.UpConv2x(ctx, f"bench.up.{C}", C, C)
    x = torch.randn(F * H * W, C, device=ctx.device).to(ctx.h16)
    fn = lambda: up(x, F, H, W)
    fn(); torch.cuda.synchronize()
    t0 = time.time(); n = 0
    while time.time() - t0 < 0.3:
        fn(); n += 1
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / n * 1e-3
    print(f"upconv2x {name}: {t * 1e3:.3f} ms = {2.0 * F * H * W * 4 * C * 4 * C / t / 1e12:.1f} TFLOP/s (4 C^2 per source pixel x 4 parities)")
