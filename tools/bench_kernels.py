#!/usr/bin/env python3
"""Micro-benchmarks of the hot kernels at the 720p / F=32 shapes (run on the GPU box):
   python tools/bench_kernels.py [gemm|attn|norm|all]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from videovanish_amd import hip, packing

dev = torch.device("cuda:0")
DT = hip.BF16
td = torch.bfloat16


def timeit(fn, n=5, warm=2):
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


def gemm_cases():
    F, h, w = 32, 90, 160
    M = F * h * w
    cases = [
        ("qkv  L0 K320 N960", dict(F=F, H=h, W=w, cin=320, cout=960, k=1)),
        ("qkv  L0 K320 N960 head-major store", dict(F=F, H=h, W=w, cin=320, cout=960, k=1, split=(8, 40, h * w))),
        ("out  L0 K320 N320 +res f32", dict(F=F, H=h, W=w, cin=320, cout=320, k=1, res=True)),
        ("geglu L0 K320 N2560", dict(F=F, H=h, W=w, cin=320, cout=2560, k=1, geglu=True)),
        ("ffout L0 K1280 N320 +res", dict(F=F, H=h, W=w, cin=1280, cout=320, k=1, res=True)),
        ("qkv  L1 K640 N1920", dict(F=F, H=45, W=80, cin=640, cout=1920, k=1)),
        ("out  L1 K640 N640 +res f32", dict(F=F, H=45, W=80, cin=640, cout=640, k=1, res=True)),
        ("ffout L1 K2560 N640 +res", dict(F=F, H=45, W=80, cin=2560, cout=640, k=1, res=True)),
        ("out  L2 K1280 N1280 +res f32", dict(F=F, H=23, W=40, cin=1280, cout=1280, k=1, res=True)),
        ("conv3 L0 320->320 +res", dict(F=F, H=h, W=w, cin=320, cout=320, k=3, res=True)),
        ("conv3 L0 640->320 (cat) f32out", dict(F=F, H=h, W=w, cin=640, cout=320, k=3)),
        ("conv3 L1 640->640", dict(F=F, H=45, W=80, cin=640, cout=640, k=3)),
        ("conv3 L2 1280->1280", dict(F=F, H=23, W=40, cin=1280, cout=1280, k=3)),
        ("conv3 L3 2560->1280", dict(F=F, H=12, W=20, cin=2560, cout=1280, k=3)),
        ("projout L0 f32in K320 N320", dict(F=F, H=h, W=w, cin=320, cout=320, k=1, f32in=True, res=True)),
        ("shortcut L0 f32in K640 N320", dict(F=F, H=h, W=w, cin=640, cout=320, k=1, f32in=True)),
        ("shortcut L2 f32in K2560 N1280", dict(F=F, H=23, W=40, cin=2560, cout=1280, k=1, f32in=True)),
        ("vae conv3 256->256 f32in 720p x1f", dict(F=1, H=720, W=1280, cin=256, cout=256, k=3, f32in=True)),
        ("stem conv3 8->320 (K72, H16 mode)", dict(F=F, H=h, W=w, cin=8, cout=320, k=3)),
        ("vae conv3 128->128 720p x2f", dict(F=2, H=720, W=1280, cin=128, cout=128, k=3)),
        ("vae conv3 512->512 360p x2f", dict(F=2, H=180, W=320, cin=512, cout=512, k=3)),
    ]
    only = os.environ.get("VV_BENCH_ONLY")
    for name, c in cases:
        if only and only not in name:
            continue
        Fr, H, W, cin, cout, k = c["F"], c["H"], c["W"], c["cin"], c["cout"], c["k"]
        M = Fr * H * W
        x = torch.randn(M, cin, device=dev)
        xin = x if c.get("f32in") else x.to(td)
        wt = torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5
        bias = torch.randn(cout, device=dev)
        geglu = c.get("geglu", False)
        if geglu:
            wi, bi = packing.geglu_interleave(wt.reshape(cout, -1), bias.cpu())
            wp, K = packing.pack_matrix(wi, td, geglu=True).to(dev), cin
            bias = bi.to(dev)
        else:
            wp, K = packing.pack_conv(wt, td)
            wp = wp.to(dev)
        res = torch.randn(M, cout, device=dev) if c.get("res") else None
        od = torch.float32 if (c.get("res") or "f32out" in name) else td
        out = torch.empty(M, cout // 2 if geglu else cout, dtype=od, device=dev)
        sp = c.get("split")
        skw = dict(split_heads=sp[0], split_dim=sp[1], split_tokens=sp[2]) if sp else {}
        fn = lambda: hip.conv_gemm(DT, xin, wp, cout, K, F=Fr, Hin=H, Win=W, ksize=k, pad_t=k // 2, pad_l=k // 2, bias=bias, res0=res, out=out,
                                   epilogue=hip.EPI_GEGLU if geglu else hip.EPI_NONE, **skw)
        t = timeit(fn)
        fl = 2.0 * M * cout * K
        by = xin.numel() * xin.element_size() + out.numel() * out.element_size() + (res.numel() * 4 if res is not None else 0)
        print(f"gemm {name:34s} {t*1e3:8.3f} ms  {fl/t/1e12:7.1f} TF/s  {by/t/1e9:7.0f} GB/s(alg)")


def attn_cases():
    for name, B, heads, N, D in [("spatial L0 d40 N14400 x8f", 8, 8, 14400, 40), ("spatial L1 d80 N3600 x32f", 32, 8, 3600, 80),
                                 ("spatial L2 d160 N920 x32f", 32, 8, 920, 160)]:
        C = heads * D
        qkv = torch.randn(B * N, 3 * C, device=dev).to(td)
        out = torch.empty(B * N, C, dtype=td, device=dev)
        fn = lambda: hip.attention(DT, qkv, qkv, qkv, out, B=B, heads=heads, Nq=N, Nkv=N, D=D, q_bs=N * 3 * C, k_bs=N * 3 * C, v_bs=N * 3 * C,
                                   o_bs=N * C, q_rs=3 * C, k_rs=3 * C, v_rs=3 * C, o_rs=C, k_off=C, v_off=2 * C)
        t = timeit(fn, n=3, warm=1)
        print(f"attn {name:34s} {t*1e3:8.3f} ms  {4.0*B*heads*N*N*D/t/1e12:7.1f} TF/s")
    for name, Fr, HW, heads, D in [("temporal L0 d40", 32, 14400, 8, 40), ("temporal L1 d80", 32, 3600, 8, 80)]:
        C = heads * D
        qkv = torch.randn(Fr * HW, 3 * C, device=dev).to(td)
        out = torch.empty(Fr * HW, C, dtype=td, device=dev)
        fn = lambda: hip.attention(DT, qkv, qkv, qkv, out, B=HW, heads=heads, Nq=Fr, Nkv=Fr, D=D, q_bs=3 * C, k_bs=3 * C, v_bs=3 * C, o_bs=C,
                                   q_rs=HW * 3 * C, k_rs=HW * 3 * C, v_rs=HW * 3 * C, o_rs=HW * C, k_off=C, v_off=2 * C)
        t = timeit(fn)
        print(f"attn {name:34s} {t*1e3:8.3f} ms  {Fr*HW*C*2*4/t/1e9:7.0f} GB/s(alg)")


def norm_cases():
    F, HW = 32, 14400
    for C, f32 in [(320, True), (640, True), (320, False)]:
        x = torch.randn(F * HW, C, device=dev)
        x = x if f32 else x.to(td)
        g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        t = timeit(lambda: hip.groupnorm(DT, x, g, b, 32, 1e-5, F=F, HW=HW, silu=True))
        by = x.numel() * x.element_size() * 2 + x.numel() * 2
        print(f"groupnorm C={C} f32in={f32}: {t*1e3:.3f} ms {by/t/1e9:.0f} GB/s")
        t = timeit(lambda: hip.groupnorm(DT, x, g, b, 32, 1e-5, F=F, HW=HW, pool_frames=True))
        print(f"groupnorm(pooled) C={C}: {t*1e3:.3f} ms {by/t/1e9:.0f} GB/s")
    x = torch.randn(F * HW, 320, device=dev)
    g, b = torch.ones(320, device=dev), torch.zeros(320, device=dev)
    t = timeit(lambda: hip.layernorm(DT, x, g, b))
    print(f"layernorm C=320: {t*1e3:.3f} ms {x.numel()*6/t/1e9:.0f} GB/s")


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("gemm", "all"): gemm_cases()
    if what in ("attn", "all"): attn_cases()
    if what in ("norm", "all"): norm_cases()
