#!/usr/bin/env python3
"""In-pipeline A/B of the GEMM tile choice on ONE device, ONE model build, the product schedule (two chunks in flight):
    python tools/ab_tiles.py [chunks per config] [config ...]
config = "base" (the library's own heuristic) or a '+'-joined set of rules; a rule forces tile_hint = 2 (the 2-phase 256-row kernel of vv_gemm256.hip)
on one shape class of the level-1 / level-2 linears:
    qkv1  K 640  N 1920          out1  K 640  N 640  (+ fp32 residual)      ff21  K 2560 N 640 (+ residual)      geglu1  K 640 N 5120 (GEGLU)
    out2  K 1280 N 1280 (+ residual)
Prints seconds per 32-frame 720p / 50-step chunk per config in the order given (repeat a config to see the box's drift) and whether the output equals
the first config's bit for bit (the tile forms sum in the same k order: it should)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from videovanish_amd import hip
from videovanish_amd.config import RunConfig
from videovanish_amd.pipeline import DiffuEraserHIP, chunk_plan

RULES = {          # name -> (K, N, geglu)
    "qkv1": (640, 1920, False), "out1": (640, 640, False), "ff21": (2560, 640, False), "geglu1": (640, 5120, True), "out2": (1280, 1280, False),
}
active = set()
_orig = hip.conv_gemm


def conv_gemm(dtype, x0, weight, N, K, **kw):
    if active and kw.get("ksize", 1) == 1 and kw.get("tile_hint", 0) == 0 and x0.dtype != torch.float32:
        g = kw.get("epilogue", hip.EPI_NONE) == hip.EPI_GEGLU
        for r in active:
            if RULES[r] == (K, N, g):
                kw["tile_hint"] = 2
                break
    return _orig(dtype, x0, weight, N, K, **kw)


hip.conv_gemm = conv_gemm
import videovanish_amd.nn as _nn
if getattr(_nn, "conv_gemm", None) is _orig:
    _nn.conv_gemm = conv_gemm

K_ = int(sys.argv[1]) if len(sys.argv) > 1 else 2
configs = sys.argv[2:] or ["base", "qkv1+out1+ff21+geglu1+out2", "base", "qkv1+out1+ff21+geglu1+out2"]
steps = int(os.environ.get("VV_AB_DENOISE_STEPS", "50"))
H, W, chunk, overlap = 720, 1280, 32, 8
run = RunConfig(steps=steps, chunk=chunk, overlap=overlap, seed=42, dtype="fp16")
model = DiffuEraserHIP(run, "cuda:0")
dev = model.ctx.device
T = (chunk - overlap) * K_ + overlap
assert len(chunk_plan(T, chunk, overlap)) == K_
fr, mk, pr = bench.synth_clip(T, H, W)
fr, mk, pr = torch.from_numpy(fr).to(dev), torch.from_numpy(mk).to(dev), torch.from_numpy(pr).to(dev)
model.forward_device(fr[:chunk], pr[:chunk], mk[:chunk], chunk, 0, steps=min(steps, 4), scheduler="ddim")      # warm-up
torch.cuda.synchronize()
ref = None
for c in configs:
    active.clear()
    if c != "base":
        active.update(c.split("+"))
    torch.cuda.synchronize(); t0 = time.time()
    out, _ = model.forward_device(fr, pr, mk, T, 0, steps=steps, scheduler="ddim")
    torch.cuda.synchronize(); dt = time.time() - t0
    same = "" if ref is None else f" identical={bool(torch.equal(out, ref))}"
    ref = out if ref is None else ref
    print(f"{c}: {dt / K_:8.3f} s/chunk  {(chunk - overlap) * K_ / dt:.4f} frames/s  ({K_} chunks){same}", flush=True)
