#!/usr/bin/env python3
"""Fused motion module (vv_motion.hip) against the layer-by-layer path at the level-0 shape of a 720p chunk (C = 320, 32 frames, 90x160)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from videovanish_amd import hip, nn as vnn
if os.environ.get("VV_LIB_PATH"):
    hip._LIB_PATH = os.environ["VV_LIB_PATH"]          # A/B of two builds of the library on one device
from videovanish_amd import packing
if os.environ.get("VV_MOTION_LAYOUT"):
    packing.MOTION_LAYOUT = os.environ["VV_MOTION_LAYOUT"]     # lab build of the rounds 2-4 kernel (-DVV_MOTION_FORM=0) reads the "tokens" stream order
from videovanish_amd.config import UNetConfig
from videovanish_amd.unet import sinusoidal_pos_emb

dname = sys.argv[1] if len(sys.argv) > 1 else "fp16"
H, W, Fr, C = 90, 160, 32, 320
ctx = vnn.Ctx("cuda:0", dname, 0)
cfg = UNetConfig()
mod = vnn.MotionModule(ctx, "unet.down_blocks.0.motion_modules.0", C, cfg, ctx.dev(sinusoidal_pos_emb(32, C)))
x = torch.randn(Fr * H * W, C, device=ctx.device)


def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


fl = (2.0 * 22 * C * C + 8.0 * Fr * C) * Fr * H * W
for fused in (False, True):
    vnn.MotionModule.FUSED = fused
    t = timeit(lambda: mod(x, Fr, H, W))
    print(f"{dname} motion module level 0 ({'fused' if fused else 'layer-by-layer'}): {t*1e3:7.3f} ms  {fl/t/1e12:7.1f} TFLOP/s = {fl/t/2.5e15*100:5.1f} % of the MFMA peak")
a = mod(x, Fr, H, W)
vnn.MotionModule.FUSED = False
b = mod(x, Fr, H, W)
print("fused vs layer-by-layer rel max:", float((a - b).abs().max() / b.abs().max()))
