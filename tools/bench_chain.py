#!/usr/bin/env python3
"""Fused tail of the level-0 spatial transformer block (vv_chain.hip) at the shape of a 720p chunk (C = 320, 32 frames x 90x160 tokens), against the
layer-by-layer tail on the same weights."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from videovanish_amd import hip, nn as vnn
if os.environ.get("VV_LIB_PATH"):
    hip._LIB_PATH = os.environ["VV_LIB_PATH"]          # A/B of two builds of the library on one device
from videovanish_amd import packing
if os.environ.get("VV_CHAIN_LAYOUT"):
    packing.CHAIN_LAYOUT = os.environ["VV_CHAIN_LAYOUT"]     # lab builds of the token-split kernels (-DVV_CHAIN_FORM=0) read the "tokens" stream order
from videovanish_amd.config import UNetConfig

dname = sys.argv[1] if len(sys.argv) > 1 else "fp16"
H, W, Fr, C = 90, 160, 32, 320
ctx = vnn.Ctx("cuda:0", dname, 0)
cfg = UNetConfig()
text = ctx.dev(torch.randn(77, 768), ctx.h16)
mod = vnn.SpatialTransformer(ctx, "unet.down_blocks.0.attentions.0", C, cfg, text)
M = Fr * H * W
o = torch.randn(M, C, device=ctx.device).to(ctx.h16)
t = torch.randn(M, C, device=ctx.device)
x = torch.randn(M, C, device=ctx.device)


def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


def unfused():
    tt = mod.attn1.out(o, res0=t)
    tt = mod.attn2(mod.n2(tt), tt, Fr, H * W)
    tt = mod.ff(mod.n3(tt), tt)
    return mod.proj_out(tt, Fr, H, W, res0=x)[0]


fl = 2.0 * M * C * C * 16 + 4.0 * M * 77 * C
for name, fn in (("layer-by-layer", unfused), ("fused", lambda: hip.spatial_chain_c320(ctx.dt, o, t, x, mod.fused[0], mod.fused[1]))):
    s = timeit(fn)
    print(f"{dname} spatial chain level 0 ({name}): {s*1e3:7.3f} ms  {fl/s/1e12:7.1f} TFLOP/s = {fl/s/2.5e15*100:5.1f} % of the MFMA peak")
a, b = hip.spatial_chain_c320(ctx.dt, o, t, x, mod.fused[0], mod.fused[1]), unfused()
print("fused vs layer-by-layer rel max:", float((a - b).abs().max() / b.abs().max()))

# the block front (GroupNorm statistics + apply, proj_in, LN1, fused q | k | v projection): hip.PROFILE gives the fused kernel's own time
hip.PROFILE = []
for _ in range(3):
    hip.spatial_chain_front_c320(ctx.dt, x, mod.norm.g, mod.norm.b, mod.norm.groups, mod.norm.eps, mod.front[0], mod.front[1], F=Fr, HW=H * W)
hip.PROFILE = []
for _ in range(20):
    hip.spatial_chain_front_c320(ctx.dt, x, mod.norm.g, mod.norm.b, mod.norm.groups, mod.norm.eps, mod.front[0], mod.front[1], F=Fr, HW=H * W)
torch.cuda.synchronize()
recs = [r for r in hip.PROFILE if r[0].startswith("spatial_chain_front")]
hip.PROFILE = None
sf = sum(e0.elapsed_time(e1) for _, _, _, e0, e1 in recs) / len(recs) * 1e-3
ff = 2.0 * M * C * C * 4
print(f"{dname} spatial chain front level 0 (front): {sf*1e3:7.3f} ms  {ff/sf/1e12:7.1f} TFLOP/s = {ff/sf/2.5e15*100:5.1f} % of the MFMA peak")
