#!/usr/bin/env python3
"""CPU-only prediction (oracle/emulate.py) of what storing a ResBlock's conv1 output in h16 instead of fp32 would cost in parity: the input of norm2 of every
UNet / BrushNet ResnetBlock2D rounded to fp16 on top of the product plan (fp16 operands, cheap set exact = split-precision decoder + precise_io).
    python tools/parity_h16_conv1.py [T H W steps]      (default 4 64 64 10, FULL width: ~20 s per plan on 8 cores)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import emulate as E                      # noqa: E402
from oracle import model_ref as M                    # noqa: E402
from oracle import pipeline_ref as R                 # noqa: E402
from videovanish_amd.config import UNetConfig, VAEConfig   # noqa: E402
from tools.parity_emulate import clip                # noqa: E402

T, H, W, steps = (int(a) for a in (sys.argv[1:5] + ["4", "64", "64", "10"][len(sys.argv) - 1:]))
ucfg, vcfg = UNetConfig(), VAEConfig()
frames, m2d, prior = clip(T, H, W)
torch.set_num_threads(os.cpu_count() or 8)
dec = lambda n: n.startswith("vae.decoder") or n.startswith("vae.post_quant")
cheap = lambda n: dec(n) or "time_emb" in n or ((n.startswith("unet.") or n.startswith("brushnet.")) and (n.endswith("conv_in") or n.endswith("conv_out")))
kw = dict(steps=steps, chunk=T, overlap=0, seed=7, ucfg=ucfg, vcfg=vcfg, return_float=True)
t0 = time.time()
ref = R.diffueraser_forward(frames, m2d, prior, **kw)
print(f"# full width, T={T} {W}x{H}, {steps} DDIM steps; oracle {time.time() - t0:.0f} s per run", flush=True)
orig_gn = M.group_norm


def gn_h16(which):
    def group_norm(P, name, x, groups, eps):
        if ".resnets." in name and name.endswith(which) and (name.startswith("unet.") or name.startswith("brushnet.")):
            x = x.to(torch.float16).to(torch.float32)
        return orig_gn(P, name, x, groups, eps)
    return group_norm


for label, patch in (("product plan", None), ("+ conv1 output (norm2 input) stored in fp16", gn_h16(".norm2"))):
    if patch is not None:
        M.group_norm = patch
    try:
        with E.emulate(dtype=torch.float16, exact=cheap):
            got = R.diffueraser_forward(frames, m2d, prior, **kw)
    finally:
        M.group_norm = orig_gn
    e = np.abs(got - ref)
    print(f"{label:50s} max_abs={e.max():.3e} mean_abs={e.mean():.3e} rms={np.sqrt((e ** 2).mean()):.3e}", flush=True)
