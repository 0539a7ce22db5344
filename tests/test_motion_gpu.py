"""The fused motion module (csrc/vv_motion.hip: one kernel for the whole AnimateDiff temporal transformer at C = 320, F = 32) against
the fp32 oracle (oracle/model_ref.py::motion_module) and against the layer-by-layer HIP path on the same weights."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from videovanish_amd.config import UNetConfig


def _rel(a, b):
    return ((a - b).abs().max() / b.abs().max()).item()


@pytest.mark.parametrize("dname,tol", [("fp16", 3e-3), ("bf16", 2.5e-2)])
@pytest.mark.parametrize("H,W,with_res1,Fr", [(6, 8, False, 32), (5, 4, True, 32), (4, 4, False, 22), (3, 4, True, 17)])      # 22 = the reference's window
def test_fused_motion_module_vs_oracle_and_unfused(gpu, dname, tol, H, W, with_res1, Fr):
    from oracle import model_ref as M
    from videovanish_amd import nn as vnn
    from videovanish_amd.unet import sinusoidal_pos_emb
    cfg = UNetConfig()
    C = 320
    name = "unet.down_blocks.0.motion_modules.0"
    g = torch.Generator().manual_seed(17)
    x = torch.randn(Fr, C, H, W, generator=g) * 1.5 + 0.2
    res1 = torch.randn(Fr, C, H, W, generator=g) if with_res1 else None
    with torch.no_grad():
        ref = M.motion_module(M.Params(0), name, x, cfg)
        if res1 is not None:
            ref = ref + res1
    ctx = vnn.Ctx("cuda:0", dname, 0)
    mod = vnn.MotionModule(ctx, name, C, cfg, ctx.dev(sinusoidal_pos_emb(cfg.motion_max_seq, C)))
    assert mod.fused is not None and mod.fused[0].shape == (670, 64, 64)
    nhwc = lambda t: t.permute(0, 2, 3, 1).reshape(Fr * H * W, C).contiguous().to(gpu)
    back = lambda t: t.float().cpu().reshape(Fr, H, W, C).permute(0, 3, 1, 2)
    xin, rin = nhwc(x), (nhwc(res1) if res1 is not None else None)
    vnn.MotionModule.FUSED = True
    fused = back(mod(xin, Fr, H, W, res1=rin))
    vnn.MotionModule.FUSED = False
    try:
        plain = back(mod(xin, Fr, H, W, res1=rin))
    finally:
        vnn.MotionModule.FUSED = True
    e_f, e_p, e_fp = _rel(fused, ref), _rel(plain, ref), _rel(fused, plain)
    print(f"motion module [{dname}, {H}x{W}]: fused vs oracle {e_f:.2e}, layer-by-layer vs oracle {e_p:.2e}, fused vs layer-by-layer {e_fp:.2e}")
    assert torch.isfinite(fused).all()
    assert e_f <= tol and e_f <= 2.0 * e_p + 1e-4
    out16 = mod(xin, Fr, H, W, res1=rin, out_dtype=ctx.h16)
    assert out16.dtype == ctx.h16 and _rel(back(out16), fused) <= (2e-3 if dname == "fp16" else 1.6e-2)
