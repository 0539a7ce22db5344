#!/usr/bin/env python3
"""A/B of the 256-row tile GEMM kernel (vv_gemm256.hip, tile_hint=2) against the 128-row kernels (tile_hint=1) on the
shapes of one 720p / F=32 denoise step + VAE, in one process; also checks that both give the same numbers.
   python tools/bench_gemm256.py [bf16|fp16]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from videovanish_amd import hip, packing
if os.environ.get("VV_LIB_PATH"):
    hip._LIB_PATH = os.environ["VV_LIB_PATH"]          # lab: another build of the library (e.g. the VV_AB build with its VV_GEMM_* switches)

dev = torch.device("cuda:0")
dname = sys.argv[1] if len(sys.argv) > 1 else "bf16"
DT = hip.dtype_id(dname)
td = hip.h16(DT)


SECONDS = float(os.environ.get("VV_BENCH_SECONDS", "0"))      # > 0: every timing is a back-to-back loop of about this many seconds (the board is power capped:
                                                               #      a 6-launch loop runs while the clock is still ramping and misranks kernels, profiles/r4_attn80_ab.txt)


def timeit(fn, n=6, warm=2):
    if SECONDS > 0:
        import time
        fn(); torch.cuda.synchronize()
        t0 = time.time(); k = 0
        while time.time() - t0 < 0.25 * SECONDS:          # warm-up under load, and an estimate of the launch time
            fn(); k += 1
            if k % 8 == 0:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        n, warm = max(8, int(SECONDS / max((time.time() - t0) / k, 1e-6))), 0
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


F = 32
CASES = [   # name, F, H, W, cin (or (c0,c1)), cout, k, stride, flags, calls per denoise step
    ("qkv   L0 K320  N960 split", F, 90, 160, 320, 960, 1, 1, "split", 20),
    ("geglu L0 K320  N2560", F, 90, 160, 320, 2560, 1, 1, "geglu", 15),
    ("ff2   L0 K1280 N320 +res h16out", F, 90, 160, 1280, 320, 1, 1, "res h16out", 15),
    ("out   L0 K320  N320 +res", F, 90, 160, 320, 320, 1, 1, "res", 70),
    ("qkv   L1 K640  N1920", F, 45, 80, 640, 1920, 1, 1, "split80", 20),
    ("geglu L1 K640  N5120", F, 45, 80, 640, 5120, 1, 1, "geglu", 15),
    ("ff2   L1 K2560 N640 +res", F, 45, 80, 2560, 640, 1, 1, "res h16out", 15),
    ("out   L1 K640  N640 +res", F, 45, 80, 640, 640, 1, 1, "res", 70),
    ("qkv   L2 K1280 N3840", F, 23, 40, 1280, 3840, 1, 1, "", 20),
    ("geglu L2 K1280 N10240", F, 23, 40, 1280, 10240, 1, 1, "geglu", 15),
    ("ff2   L2 K5120 N1280 +res", F, 23, 40, 5120, 1280, 1, 1, "res h16out", 15),
    ("out   L2 K1280 N1280 +res", F, 23, 40, 1280, 1280, 1, 1, "res", 70),
    ("zero  L0 K320  N320 f32in +res", F, 90, 160, 320, 320, 1, 1, "res f32in", 8),          # BrushNet zero convolutions / conv_shortcuts: fp32 trunk in, fp32 out
    ("zero  L1 K640  N640 f32in +res", F, 45, 80, 640, 640, 1, 1, "res f32in", 8),
    ("zero  L2 K1280 N1280 f32in +res", F, 23, 40, 1280, 1280, 1, 1, "res f32in", 8),
    ("conv3 L0 320->320 +res", F, 90, 160, 320, 320, 3, 1, "res", 14),
    ("conv3 L0 640->320 (cat)", F, 90, 160, (320, 320), 320, 3, 1, "", 4),
    ("conv3 L0 960->320 (cat)", F, 90, 160, (640, 320), 320, 3, 1, "", 2),
    ("conv3 L0 320->320 s2 down", F, 90, 160, 320, 320, 3, 2, "", 2),
    ("conv3 L1 1920->640 (cat)", F, 45, 80, (1280, 640), 640, 3, 1, "", 2),
    ("conv3 L1 960->640 (cat)", F, 45, 80, (640, 320), 640, 3, 1, "", 2),
    ("conv3 L2 1920->1280 (cat)", F, 23, 40, (1280, 640), 1280, 3, 1, "", 2),
    ("conv3 L1 640->640", F, 45, 80, 640, 640, 3, 1, "res", 12),
    ("conv3 L1 1280->640 (cat)", F, 45, 80, (640, 640), 640, 3, 1, "", 2),
    ("conv3 L1 320->640", F, 45, 80, 320, 640, 3, 1, "", 2),
    ("conv3 L2 1280->1280", F, 23, 40, 1280, 1280, 3, 1, "res", 14),
    ("conv3 L2 2560->1280 (cat)", F, 23, 40, (1280, 1280), 1280, 3, 1, "", 4),
    ("conv3 L3 1280->1280 12x20", F, 12, 20, 1280, 1280, 3, 1, "res", 22),
    ("conv3 L3 2560->1280 12x20", F, 12, 20, (1280, 1280), 1280, 3, 1, "", 6),
    ("vae conv3 512->512 180x320 x4f", 4, 180, 320, 512, 512, 3, 1, "res", 0),
    ("vae conv3 256->256 360x640 x4f", 4, 360, 640, 256, 256, 3, 1, "res", 0),
    ("vae conv3 512->512 90x160 x4f", 4, 90, 160, 512, 512, 3, 1, "res", 0),
    ("vae conv3 128->128 720x1280 x4f", 4, 720, 1280, 128, 128, 3, 1, "res", 0),
    ("vae conv3 256->128 720x1280 x4f", 4, 720, 1280, 256, 128, 3, 1, "", 0),
    ("unet conv_out 320->4 90x160", F, 90, 160, 320, 4, 3, 1, "", 1),
    ("vae conv_out 128->3 720x1280 x4f", 4, 720, 1280, 128, 3, 3, 1, "", 0),
    ("big   K4096 N4096 M8192", 1, 8192, 1, 4096, 4096, 1, 1, "", 0),
    ("big   K8192 N8192 M8192", 1, 8192, 1, 8192, 8192, 1, 1, "", 0),
]


def main():
    only = os.environ.get("VV_BENCH_ONLY")
    tot = {1: 0.0, 2: 0.0, "best": 0.0}
    for name, Fr, H, W, cin, cout, k, stride, flags, calls in CASES:
        if only and not any(o in name for o in only.split("|")):
            continue
        c0, c1 = cin if isinstance(cin, tuple) else (cin, 0)
        Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
        M = Fr * Ho * Wo
        g = torch.Generator(device="cpu").manual_seed(1)
        x0 = torch.randn(Fr * H * W, c0, generator=g).to(torch.float32 if "f32in" in flags else td).to(dev)
        x1 = torch.randn(Fr * H * W, c1, generator=g).to(td).to(dev) if c1 else None
        wt = torch.randn(cout, c0 + c1, k, k, generator=g) / ((c0 + c1) * k * k) ** 0.5
        bias = torch.randn(cout, generator=g)
        geglu = "geglu" in flags
        if geglu:
            wi, bi = packing.geglu_interleave(wt.reshape(cout, -1), bias)
            wp, K = packing.pack_matrix(wi, td, geglu=True).to(dev), c0
            bias = bi
        else:
            wp, K = packing.pack_conv(wt, td)
            wp = wp.to(dev)
        bias = bias.to(dev)
        res = torch.randn(M, cout, generator=g).to(dev) if "res" in flags else None
        od = td if ("h16out" in flags or geglu or "split" in flags or res is None) else torch.float32
        skw = {}
        if flags.startswith("split"):
            D = 80 if flags == "split80" else 40
            skw = dict(split_heads=cout // 3 // D, split_dim=D, split_tokens=Ho * Wo)
        outs, times = {}, {}
        hints = (1, 2, 3, 4) if (cout % 256 == 0) else (1, 2)
        if SECONDS > 0:
            hints = (0,) + hints                       # 0 = what the product's heuristic picks
        if os.environ.get("VV_BENCH_HINTS"):
            hints = tuple(int(h) for h in os.environ["VV_BENCH_HINTS"].split(","))
        for hint in hints:
            out = torch.zeros(M, cout // 2 if geglu else cout, dtype=od, device=dev)
            fn = lambda: hip.conv_gemm(DT, x0, wp, cout, K, x1=x1, F=Fr, Hin=H, Win=W, Hout=Ho, Wout=Wo, ksize=k, stride=stride, pad_t=k // 2,
                                       pad_l=k // 2, bias=bias, res0=res, out=out, epilogue=hip.EPI_GEGLU if geglu else hip.EPI_NONE,
                                       tile_hint=hint, **skw)
            times[hint] = timeit(fn)
            outs[hint] = out.float()
        fl = 2.0 * M * cout * K
        if 1 not in times or 2 not in times:
            print(f"{name:34s} M={M:7d} " + " | ".join(f"hint{h} {times[h]*1e3:7.3f} ms {fl/times[h]/1e12:7.1f} TF/s" for h in hints), flush=True)
            continue
        diff = max((outs[1] - outs[h]).abs().max().item() for h in hints)
        scale = outs[1].abs().max().item()
        t1, t2 = times[1], times[2]
        for h in (1, 2):
            tot[h] += times[h] * calls
        tot["best"] += min(times.values()) * calls
        extra = "".join(f" | hint{h} {times[h]*1e3:7.3f} ms {fl/times[h]/1e12:7.1f} TF/s x{t1/times[h]:4.2f}" for h in hints if h > 2 or h == 0)
        print(f"{name:34s} M={M:7d} 128-row {t1*1e3:7.3f} ms {fl/t1/1e12:7.1f} TF/s | 256-row {t2*1e3:7.3f} ms {fl/t2/1e12:7.1f} TF/s | x{t1/t2:5.2f}{extra} | "
              f"maxdiff {diff:.2e} (|out| {scale:.1f})", flush=True)
    print(f"per denoise step (calls-weighted): 128-row {tot[1]*1e3:.1f} ms, 256-row {tot[2]*1e3:.1f} ms, best-of {tot['best']*1e3:.1f} ms")


if __name__ == "__main__":
    main()
