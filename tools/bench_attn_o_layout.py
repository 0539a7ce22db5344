#!/usr/bin/env python3
"""Lab: the level-0 spatial self-attention (32 frames x 8 heads x 14400 tokens, d = 40, head-major QKV) with its output row-major [M][320] against head-major
[frame][head][token][40] (vv_attn_params.o_hs, round 6), interleaved:  python tools/bench_attn_o_layout.py [frames] [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from videovanish_amd import hip, nn as vnn
if os.environ.get("VV_LIB_PATH"):
    hip._LIB_PATH = os.environ["VV_LIB_PATH"]      # (lab) another build of the library
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
R = int(sys.argv[2]) if len(sys.argv) > 2 else 3
heads, N, D = 8, 14400, 40
C = heads * D
ctx = vnn.Ctx("cuda:0", "fp16", 0)
qkv = (torch.randn(B, 3, heads, N, D, device="cuda") * 0.5).to(ctx.h16)
outs = {False: torch.empty(B * N, C, dtype=ctx.h16, device="cuda"), True: torch.empty(B, heads, N, D, dtype=ctx.h16, device="cuda")}


def run(hm):
    hip.attention(ctx.dt, qkv, qkv, qkv, outs[hm], B=B, heads=heads, Nq=N, Nkv=N, D=D, q_bs=N * 3 * C, k_bs=N * 3 * C, v_bs=N * 3 * C, o_bs=N * C, q_rs=D, k_rs=D,
                  v_rs=D, o_rs=D if hm else C, k_off=N * C, v_off=2 * N * C, q_hs=N * D, k_hs=N * D, v_hs=N * D, q_prescaled=True, o_hs=N * D if hm else 0)


fl = 4.0 * B * heads * N * N * D
for hm in (False, True):
    run(hm)
torch.cuda.synchronize()
assert torch.equal(outs[True].permute(0, 2, 1, 3).reshape(B * N, C), outs[False])
for r in range(R):
    for hm in (False, True):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            run(hm)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"round {r} o {'head-major' if hm else 'row-major '}: {ms:7.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s")
