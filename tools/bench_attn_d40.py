#!/usr/bin/env python3
"""Spatial attention at the pipeline's shapes (8 heads, B frames; head-major QKV, q_prescaled): python tools/bench_attn_d40.py [B] [dtype] [D] [N]
defaults: level 0 (D = 40, N = 14400); level 1 = "32 fp16 80 3600"."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from videovanish_amd import hip
if os.environ.get("VV_LIB_PATH"):
    hip._LIB_PATH = os.environ["VV_LIB_PATH"]          # lab: A/B of two builds of the library on one device

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dname = sys.argv[2] if len(sys.argv) > 2 else "fp16"
dev = torch.device("cuda:0")
DT = hip.dtype_id(dname); td = hip.h16(DT)
heads = 8
D = int(sys.argv[3]) if len(sys.argv) > 3 else 40
N = int(sys.argv[4]) if len(sys.argv) > 4 else 14400
C = heads * D
g = torch.Generator().manual_seed(0)
qkv = torch.randn(B, 3, heads, N, D, generator=g)
qkv[:, 0] *= hip.attention_q_scale(D)
qkv = qkv.to(td).to(dev)
out = torch.empty(B * N, C, dtype=td, device=dev)
fn = lambda: hip.attention(DT, qkv, qkv, qkv, out, B=B, heads=heads, Nq=N, Nkv=N, D=D, q_bs=N * 3 * C, k_bs=N * 3 * C, v_bs=N * 3 * C, o_bs=N * C,
                           q_rs=D, k_rs=D, v_rs=D, o_rs=C, k_off=N * C, v_off=2 * N * C, q_hs=N * D, k_hs=N * D, v_hs=N * D, q_prescaled=True)
for _ in range(int(os.environ.get("VV_BENCH_WARM", "2"))):
    fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
n = int(os.environ.get("VV_BENCH_ITERS", "5"))
for _ in range(n):
    fn()
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / n * 1e-3
if os.environ.get("VV_BENCH_CHECK"):               # one (frame, head) against a plain fp32 softmax on the device
    b, h = B - 1, 5
    q, k, v = (qkv[b, i, h].float() for i in range(3))
    ref = torch.softmax((q @ k.t()) * 0.6931471805599453, -1) @ v
    err = (out.view(B, N, heads, D)[b, :, h].float() - ref).abs().max().item()
    print(f"check (b={b}, h={h}): max-abs {err:.3e} (output range {ref.abs().max().item():.2f})")
print(f"{dname} spatial attention d{D} N{N} x{B} frames: {t * 1e3:.3f} ms = {4.0 * B * heads * N * N * D / t / 1e12:.1f} TFLOP/s")
