#!/usr/bin/env python3
"""SAM 2 memory attention core at the published size: one head of d = 256, 4096 queries, self (4096 keys) and cross (7 x 4096 + 64 keys).
   python tools/bench_attn_d256.py      (VV_LIB_PATH: another build of the library; VV_ATTN_VARIANT: lab variants 71-78)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from videovanish_amd import hip
if os.environ.get("VV_LIB_PATH"):
    hip._LIB_PATH = os.environ["VV_LIB_PATH"]
dev = torch.device("cuda:0")
DT = hip.F16
D, Nq = 256, 4096
g = torch.Generator().manual_seed(0)
for Nk in (4096, 7 * 4096 + 64):
    q = (torch.randn(Nq, D, generator=g)).half().to(dev)
    k = (torch.randn(Nk, D, generator=g)).half().to(dev)
    v = (torch.randn(Nk, D, generator=g)).half().to(dev)
    o = torch.empty(Nq, D, dtype=torch.float16, device=dev)
    fn = lambda: hip.attention(DT, q, k, v, o, B=1, heads=1, Nq=Nq, Nkv=Nk, D=D, q_bs=0, k_bs=0, v_bs=0, o_bs=0, q_rs=D, k_rs=D, v_rs=D, o_rs=D, q_hs=D, k_hs=D, v_hs=D)
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fn()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 5 * 1e-3
    ref = torch.nn.functional.scaled_dot_product_attention(q[None, None, :256].float(), k[None, None].float(), v[None, None].float())[0, 0]
    err = float((o[:256].float() - ref).abs().max() / ref.abs().max())
    print(f"variant {os.environ.get('VV_ATTN_VARIANT', '-')}: d256 Nq {Nq} Nkv {Nk}: {t * 1e3:.3f} ms = {4.0 * Nq * Nk * D / t / 1e12:.1f} TFLOP/s (rel err {err:.1e})", flush=True)
