#!/usr/bin/env python3
"""GroupNorm + SiLU at the level-0 shape of a 720p chunk (32 frames x 14400 pixels x 320 channels, fp32 in, h16 out) and at the VAE decoder's
largest shape (4 frames x 921600 x 128): algorithmic bytes 2 reads + 1 write.  Lab build: VV_GN_GROUP_MB=0 disables the frame-group loop."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from videovanish_amd import hip
if os.environ.get("VV_LIB_PATH"):
    hip._LIB_PATH = os.environ["VV_LIB_PATH"]          # A/B of two builds of the library on one device

dt = hip.F16
for F, HW, C in ((32, 14400, 320), (32, 14400, 640), (32, 3600, 640), (4, 921600, 128)):
    x = torch.randn(F * HW, C, device="cuda")
    g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    fn = lambda: hip.groupnorm(dt, x, g, b, 32, 1e-5, F=F, HW=HW, silu=True)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = int(os.environ.get("VV_BENCH_N", "20"))
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / n * 1e-3
    by = F * HW * C * (4 + 4 + 2)
    print(f"groupnorm F={F} HW={HW} C={C}: {t * 1e3:.3f} ms = {by / t / 1e9:.0f} GB/s of algorithmic traffic (VV_GN_GROUP_MB={os.environ.get('VV_GN_GROUP_MB', 'default')})")
