#!/usr/bin/env python3
"""CPU-only error attribution for the DiffuEraser path: plain fp32 oracle vs the oracle with the HIP path's operand
roundings emulated (oracle/emulate.py).  Prints per-pixel max-abs in [0,1] after N DDIM steps for several precision plans."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import emulate as E                      # noqa: E402
from oracle import model_ref as M                    # noqa: E402
from oracle import pipeline_ref as R                 # noqa: E402
from videovanish_amd.config import SMALL_UNET, SMALL_VAE, TINY_UNET, TINY_VAE   # noqa: E402


def clip(T, H, W, seed=1234):
    rng = np.random.default_rng(seed)
    frames = [rng.integers(0, 256, (H, W, 3), dtype=np.uint8) for _ in range(T)]
    m2d = []
    for t in range(T):
        m = np.zeros((H, W), np.uint8)
        m[H // 4: H // 2, W // 4 + 2 * t: W // 2 + 2 * t] = 255
        m2d.append(m)
    prior = []
    for f, m in zip(frames, m2d):
        p = f.copy()
        p[m > 0] = f.reshape(-1, 3).mean(0).astype(np.uint8)
        prior.append(p)
    return frames, m2d, prior


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="tiny")
    ap.add_argument("--steps", type=int, nargs="+", default=[3, 10, 50])
    ap.add_argument("--T", type=int, default=4)
    ap.add_argument("--H", type=int, default=32)
    ap.add_argument("--W", type=int, default=40)
    args = ap.parse_args()
    ucfg, vcfg = (TINY_UNET, TINY_VAE) if args.arch == "tiny" else (SMALL_UNET, SMALL_VAE)
    frames, m2d, prior = clip(args.T, args.H, args.W)
    torch.set_num_threads(8)
    vae = lambda n: n.startswith("vae.")
    io = lambda n: n.endswith("conv_in") or n.endswith("conv_out") or "time_emb" in n
    plans = [
        ("fp16 all operands", dict(dtype=torch.float16)),
        ("bf16 all operands", dict(dtype=torch.bfloat16)),
        ("fp16, VAE exact", dict(dtype=torch.float16, exact=vae)),
        ("fp16, denoiser exact (VAE only rounded)", dict(dtype=torch.float16, exact=lambda n: not vae(n))),
        ("fp16, conv_in/conv_out/time-emb exact", dict(dtype=torch.float16, exact=io)),
        ("fp16, VAE + conv_in/out/time-emb exact", dict(dtype=torch.float16, exact=lambda n: vae(n) or io(n))),
        ("fp16 weights only", dict(dtype=torch.float16, classes=("w",))),
        ("fp16 activations only", dict(dtype=torch.float16, classes=("a",))),
        ("fp16 attention operands + P only", dict(dtype=torch.float16, classes=("qkv", "p"))),
        ("bf16 weights only", dict(dtype=torch.bfloat16, classes=("w",))),
        ("bf16 activations only", dict(dtype=torch.bfloat16, classes=("a",))),
    ]
    dec = lambda n: n.startswith("vae.decoder") or n.startswith("vae.post_quant")
    enc = lambda n: n.startswith("vae.encoder") or n.startswith("vae.quant")
    L = len(vcfg.block_out)
    tail = lambda n: n.startswith("vae.decoder.conv_out") or n.startswith(f"vae.decoder.up_blocks.{L - 1}")
    plans += [      # which part of the VAE, and how much of a split-precision layer, is needed
        ("fp16, VAE decoder exact (3-pass split)", dict(dtype=torch.float16, exact=dec)),
        ("fp16, VAE encoder exact", dict(dtype=torch.float16, exact=enc)),
        ("fp16, decoder last block + conv_out exact", dict(dtype=torch.float16, exact=tail)),
        ("fp16, decoder activations exact (2-pass)", dict(dtype=torch.float16, exact=lambda n: {"a"} if dec(n) else False)),
        ("fp16, decoder weights exact (2-pass)", dict(dtype=torch.float16, exact=lambda n: {"w"} if dec(n) else False)),
        ("fp16, decoder exact + conv_in/out exact", dict(dtype=torch.float16, exact=lambda n: dec(n) or n.endswith("conv_in") or n.endswith("conv_out"))),
        ("fp16, decoder exact + ResBlock conv acts exact", dict(dtype=torch.float16, exact=lambda n: True if dec(n) else ({"a"} if ".resnets." in n else False))),
    ]
    print(f"# arch={args.arch} T={args.T} {args.W}x{args.H}; per-pixel max-abs / mean-abs in [0,1] vs the fp32 oracle")
    for steps in args.steps:
        kw = dict(steps=steps, chunk=args.T, overlap=0, seed=7, ucfg=ucfg, vcfg=vcfg, return_float=True)
        ref = R.diffueraser_forward(frames, m2d, prior, **kw)
        for name, plan in plans:
            with E.emulate(**plan):
                got = R.diffueraser_forward(frames, m2d, prior, **kw)
            err = np.abs(got - ref)
            print(f"steps={steps:3d}  {name:45s} max_abs={err.max():.3e} mean_abs={err.mean():.3e}", flush=True)


if __name__ == "__main__":
    main()
