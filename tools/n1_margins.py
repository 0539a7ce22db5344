#!/usr/bin/env python3
"""Print the measured parity numbers of the row-n1 networks (the quantities tests/test_flowcomplete_gpu.py and tests/test_inpaintgen_gpu.py bound)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import flowcomplete_ref as FC, inpaintgen_ref as G
from oracle.model_ref import Params
from videovanish_amd import nn
from videovanish_amd.flowcomplete import FlowCompleteNet
from videovanish_amd.inpaintgen import InpaintGenerator, inpaint_clip
from tests.test_flowcomplete_gpu import _case as fc_case
from tests.test_inpaintgen_gpu import _case as gen_case

gpu = torch.device("cuda:0")
for dname in ("fp16", "bf16"):
    T, H, W, width, dg = 5, 32, 48, (16, 32, 64), 8
    fw, bw, m = fc_case(T, H, W, 3)
    P = Params(11)
    mf = (m > 0).float()[None, :, None]
    to5 = lambda f: f.permute(0, 3, 1, 2)[None]
    with torch.no_grad():
        rf, rb = FC.forward_bidirect_flow(P, to5(fw), to5(bw), mf, width=width, deform_groups=dg)
    net = FlowCompleteNet(nn.Ctx("cuda:0", dname, 11), width=width, deform_groups=dg)
    pf, pb = net.forward_bidirect_flow(fw.to(gpu), bw.to(gpu), m.to(gpu))
    rel = max(((g.cpu() - r[0].permute(0, 2, 3, 1)).abs().max() / r.abs().max()).item() for g, r in ((pf, rf), (pb, rb)))
    print(f"flow completion [{dname}]: rel. max error of the predicted flows {rel:.2e}")
    t, lt, H, W, depths = 5, 3, 80, 144, 2
    frames, m_in, m_up, ff, fb = gen_case(t, lt, H, W, 4)
    P = Params(21)
    fr = torch.from_numpy(frames).float().permute(0, 3, 1, 2)[None] / 127.5 - 1.0
    mi = torch.from_numpy(m_in > 0).float()[None, :, None]
    mu = torch.from_numpy(m_up > 0).float()[None, :, None]
    with torch.no_grad():
        ref = G.generator(P, fr, to5(ff), to5(fb), mi, mu, lt, depths=depths, t_dilation=2)[0].permute(0, 2, 3, 1)
    gen = InpaintGenerator(nn.Ctx("cuda:0", dname, 21), depths=depths, t_dilation=2)
    raw = gen.forward(torch.from_numpy(frames).to(gpu), ff.to(gpu), fb.to(gpu), torch.from_numpy(m_in).to(gpu), torch.from_numpy(m_up).to(gpu), lt)
    err = (torch.tanh(raw.cpu()).reshape(lt, H, W, 3) - ref).abs()
    print(f"inpainting generator [{dname}, {depths} transformer blocks]: max abs error of the tanh output {err.max():.2e}, mean {err.mean():.2e} (output std {ref.std():.2f})")
T, H, W, depths = 7, 48, 80, 2
frames, m_in, m_up, _, _ = gen_case(T, T, H, W, 9)
g = torch.Generator().manual_seed(2)
ff = torch.randn(T - 1, H, W, 2, generator=g); fb = -ff + 0.05 * torch.randn(T - 1, H, W, 2, generator=g)
updated = frames.copy(); updated[m_up > 0] = 127
ref = G.inpaint_clip(Params(23), updated, frames, ff.permute(0, 3, 1, 2), fb.permute(0, 3, 1, 2), m_in, m_up, neighbor_length=4, ref_stride=3, depths=depths)
gen = InpaintGenerator(nn.Ctx("cuda:0", "fp16", 23), depths=depths)
dev = lambda a: torch.from_numpy(a).to(gpu)
got = inpaint_clip(gen, dev(updated), dev(frames), ff.to(gpu), fb.to(gpu), dev(m_in), dev(m_up), neighbor_length=4, ref_stride=3).cpu().numpy()
d = np.abs(got.astype(np.int32) - ref.astype(np.int32))
print(f"sliding-window clip [fp16]: uint8 max difference {d.max()}, pixels off by more than 1 level {100 * (d > 1).mean():.3f} %, by at least 1 {100 * (d > 0).mean():.2f} %")
