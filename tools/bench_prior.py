#!/usr/bin/env python3
"""Time the prior (row a4 / n1) on a synthetic 720p clip: python tools/bench_prior.py [T] [H] [W] [iters] [stages]
stages: "prop" = RAFT + flow-guided propagation (default); "fc" = + recurrent flow completion; "full" = + inpainting generator (the complete
ProPainter pipeline, sliding windows of 10 frames, reference stride 10)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from videovanish_amd import flowprop

T = int(sys.argv[1]) if len(sys.argv) > 1 else 32
H = int(sys.argv[2]) if len(sys.argv) > 2 else 720
W = int(sys.argv[3]) if len(sys.argv) > 3 else 1280
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 20
stages = sys.argv[5] if len(sys.argv) > 5 else "prop"
rng = np.random.default_rng(0)
base = rng.integers(0, 256, (H + 2 * T, W + 2 * T, 3), dtype=np.uint8)
frames = [np.ascontiguousarray(base[t: t + H, 2 * t: 2 * t + W]) for t in range(T)]
masks = []
for t in range(T):
    m = np.zeros((H, W), np.uint8); m[H // 4: H // 2, W // 4 + 2 * t: W // 2 + 2 * t] = 255; masks.append(m)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    stamps = []
    out = flowprop.flow_propagation_prior(frames, masks, device="cuda:0", dtype="fp16", iters=iters, subvideo_length=50,
                                          flow_completion=stages in ("fc", "full"), generator=stages == "full",
                                          progress=lambda pct, msg: (torch.cuda.synchronize(), stamps.append((time.time() - t0, msg))))
    torch.cuda.synchronize(); dt = time.time() - t0
    print(f"prior[{stages}] {T} frames {W}x{H}, {iters} RAFT iterations: {dt:.2f} s = {T / dt:.2f} frames/s (rep {rep}); stage starts: "
          + ", ".join(f"{m.split('(')[-1].rstrip(')')} @{s_:.2f}s" for s_, m in stamps))
