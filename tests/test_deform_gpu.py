"""Parity of the deformable-convolution path (vv_deform_im2col + vv_conv_gemm, videovanish_amd/deform.py) against oracle/deform_ref.py
(SURVEY 8f row n1), through the C ABI on the GPU."""
import numpy as np
import pytest
import torch

from oracle import deform_ref as D

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _rand(shape, seed, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def _rows(t):          # NCHW -> [B*H*W, C]
    return t.permute(0, 2, 3, 1).reshape(-1, t.shape[1]).contiguous()


@pytest.mark.parametrize("dname,td,ulp", [("bf16", torch.bfloat16, 2 ** -8), ("fp16", torch.float16, 2 ** -11)])
@pytest.mark.parametrize("B,C,H,W,dg,k,stride,pad,dil,x32", [(2, 32, 9, 13, 4, 3, 1, 1, 1, False), (1, 128, 12, 10, 16, 3, 1, 1, 1, True),
                                                            (1, 16, 11, 9, 1, 3, 2, 1, 1, False), (1, 16, 8, 8, 2, 3, 1, 2, 2, True),
                                                            (1, 8, 6, 7, 1, 1, 1, 0, 1, False)])
def test_deform_im2col_matches_oracle(gpu, dname, td, ulp, B, C, H, W, dg, k, stride, pad, dil, x32):
    from videovanish_amd import hip
    dt = hip.dtype_id(dname)
    K = k * k
    Ho = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1
    Wo = (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
    x = _rand((B, C, H, W), 1)
    if not x32:
        x = x.to(td).float()
    off = _rand((B, 2 * dg * K, Ho, Wo), 2, 2.5)                    # |offset| up to ~8 px: many samples leave the image
    off[:, :, 0, 0] = 0.0                                           # ... some sit exactly on the grid
    off[:, 0, 1, 1] = -50.0                                         # ... and one far outside
    msk = torch.sigmoid(_rand((B, dg * K, Ho, Wo), 3))
    ref = _rows(D.deform_columns(x, off, msk, k, k, stride, pad, dil, dg))
    xr = _rows(x).to(gpu) if x32 else _rows(x).to(td).to(gpu)
    col, ho, wo = hip.deform_im2col(dt, xr, B=B, H=H, W=W, kh=k, kw=k, stride=stride, pad=pad, dil=dil, deform_groups=dg,
                                    offset=_rows(off).to(gpu), mask=_rows(msk).to(gpu))
    assert (ho, wo) == (Ho, Wo) and col.shape == ref.shape and col.dtype == td
    err = (col.float().cpu() - ref).abs().max().item()
    assert err <= 1.5 * ulp * max(1.0, ref.abs().max().item()), err           # one rounding to h16 + fp32 blend order
    nomask, _, _ = hip.deform_im2col(dt, xr, B=B, H=H, W=W, kh=k, kw=k, stride=stride, pad=pad, dil=dil, deform_groups=dg, offset=_rows(off).to(gpu))
    ref2 = _rows(D.deform_columns(x, off, None, k, k, stride, pad, dil, dg))
    assert (nomask.float().cpu() - ref2).abs().max().item() <= 1.5 * ulp * max(1.0, ref2.abs().max().item())


def test_zero_offsets_reproduce_the_plain_convolution_kernel(gpu):
    """offset = 0, no mask: the gathered columns are exactly the im2col of the input, so the deformable path must agree with the
    implicit-GEMM 3x3 convolution of the same packed weights up to accumulation order."""
    from videovanish_amd import hip, nn
    from videovanish_amd.deform import DeformConv2d
    ctx = nn.Ctx("cuda:0", "fp16", 0)
    B, C, H, W, Co = 2, 64, 10, 14, 48
    x = _rows(_rand((B, C, H, W), 5)).to(torch.float16).to(gpu)
    dcn = DeformConv2d(ctx, "t.dcn", C, Co, 3, 1, 1, 1, deform_groups=8)
    conv = nn.Conv(ctx, "t.dcn", C, Co, k=3)
    off = torch.zeros(B * H * W, 2 * 8 * 9, device=gpu)
    got, ho, wo = dcn(x, B, H, W, offset=off)
    ref, _, _ = conv(x, B, H, W)
    assert (ho, wo) == (H, W)
    assert (got - ref).abs().max().item() <= 2e-3 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("dname,tol", [("fp16", 2e-3), ("bf16", 1.6e-2)])
def test_deform_conv_and_alignment_match_oracle(gpu, dname, tol):
    from oracle.model_ref import Params
    from videovanish_amd import nn
    from videovanish_amd.deform import DeformableAlignment, DeformConv2d
    ctx = nn.Ctx("cuda:0", dname, 3)
    P = Params(3)
    B, C, H, W, dg = 1, 128, 12, 20, 16
    x = _rand((B, C, H, W), 6)
    off = _rand((B, 2 * dg * 9, H, W), 7, 2.0)
    msk = torch.sigmoid(_rand((B, dg * 9, H, W), 8))
    w, b = P.conv("n1.dcn", C, C, 3)
    ref = D.deform_conv2d(x, off, w, b, 1, 1, 1, msk)
    dcn = DeformConv2d(ctx, "n1.dcn", C, C, 3, 1, 1, 1, dg)
    got, _, _ = dcn(_rows(x).to(gpu), B, H, W, offset=_rows(off).to(gpu), mask=_rows(msk).to(gpu))
    rel = ((got.cpu() - _rows(ref)).abs().max() / ref.abs().max()).item()
    assert rel <= tol, rel
    # the whole alignment module (ProPainter: cond = [warped feature | current feature | flow], here 2C + 2 channels padded to 2C + 8)
    cond = _rand((B, 2 * C + 2, H, W), 9)
    flow = _rand((B, 2, H, W), 10, 1.5)
    ref_a = D.deformable_alignment(P, "n1.align", x, cond, flow, C, deform_groups=dg)
    align = DeformableAlignment(ctx, "n1.align", C, 2 * C + 2, deform_groups=dg)
    condp = torch.zeros(B * H * W, 2 * C + 8)
    condp[:, :2 * C + 2] = _rows(cond)
    got_a = align(_rows(x).to(gpu), condp.to(gpu), _rows(flow).to(gpu), B, H, W)
    rel_a = ((got_a.cpu() - _rows(ref_a)).abs().max() / ref_a.abs().max()).item()
    assert rel_a <= 3 * tol, rel_a


def test_argument_errors(gpu):
    from videovanish_amd import hip
    x = torch.zeros(64, 12, dtype=torch.float16, device=gpu)
    with pytest.raises(RuntimeError, match="multiple of 8"):
        hip.deform_im2col(hip.F16, x, B=1, H=8, W=8, deform_groups=1, offset=torch.zeros(64, 18, device=gpu))
    x = torch.zeros(64, 16, dtype=torch.float16, device=gpu)
    with pytest.raises(RuntimeError, match="raw / offset"):
        hip.deform_im2col(hip.F16, x, B=1, H=8, W=8, deform_groups=1)
