"""Pins of the deformable-convolution oracle (oracle/deform_ref.py) that need no reference operator: the cases where DCNv2 reduces to
something torch itself computes."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import deform_ref as D


def _rand(shape, seed, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


@pytest.mark.parametrize("stride,pad,dil,k", [(1, 1, 1, 3), (2, 1, 1, 3), (1, 2, 2, 3), (1, 0, 1, 1)])
def test_zero_offsets_unit_mask_is_a_plain_convolution(stride, pad, dil, k):
    B, C, H, W, Co, dg = 2, 8, 9, 11, 6, 4
    x, w, b = _rand((B, C, H, W), 1), _rand((Co, C, k, k), 2, 0.3), _rand((Co,), 3)
    Ho = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1
    Wo = (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
    off = torch.zeros(B, 2 * dg * k * k, Ho, Wo)
    got = D.deform_conv2d(x, off, w, b, stride, pad, dil, torch.ones(B, dg * k * k, Ho, Wo))
    ref = F.conv2d(x, w, b, stride=stride, padding=pad, dilation=dil)
    assert got.shape == ref.shape
    assert (got - ref).abs().max() < 1e-5
    assert (D.deform_conv2d(x, off, w, b, stride, pad, dil, None) - ref).abs().max() < 1e-5


def test_integer_offsets_shift_the_image_and_the_mask_scales_taps():
    B, C, H, W, Co, dg, k = 1, 4, 8, 10, 3, 1, 3
    x, w = _rand((B, C, H, W), 4), _rand((Co, C, k, k), 5, 0.3)
    off = torch.zeros(B, 2 * dg * 9, H, W)
    off[:, 0::2] = 2.0          # dy = +2 for every tap
    off[:, 1::2] = -1.0         # dx = -1
    shifted = torch.zeros_like(x)
    shifted[:, :, :H - 2, 1:] = x[:, :, 2:, :W - 1]                 # shifted[y, x] = x[y + 2, x - 1], zero outside
    ref = F.conv2d(shifted, w, None, padding=1)
    got = D.deform_conv2d(x, off, w, None, 1, 1, 1, None)
    # the zero padding of the shifted image and the sampler's "outside is zero" coincide except where a tap leaves the ORIGINAL image
    # but not the shifted frame: compare where every tap of both stays inside
    assert (got - ref)[:, :, 1:H - 3, 2:W - 1].abs().max() < 1e-5
    m = _rand((B, 9, H, W), 6).abs()
    got_m = D.deform_conv2d(x, torch.zeros_like(off), w, None, 1, 1, 1, m)
    ref_m = sum(F.conv2d(x * 1.0, (w * (torch.arange(9).view(1, 1, 3, 3) == t)), None, padding=1) * m[:, t:t + 1] for t in range(9))
    assert (got_m - ref_m).abs().max() < 1e-5


def test_sampler_matches_grid_sample_inside_and_is_zero_outside():
    B, C, H, W = 1, 3, 7, 9
    img = _rand((B, C, H, W), 7)
    g = torch.Generator().manual_seed(8)
    py = torch.rand(B, 1, 5, 6, generator=g) * (H + 3) - 2        # includes positions outside [-1, H]
    px = torch.rand(B, 1, 5, 6, generator=g) * (W + 3) - 2
    got = D.bilinear_zero(img, py, px)
    grid = torch.stack([2 * px[:, 0] / (W - 1) - 1, 2 * py[:, 0] / (H - 1) - 1], -1)
    ref = F.grid_sample(img, grid, mode="bilinear", padding_mode="zeros", align_corners=True)
    inside = ((py > -1) & (py < H) & (px > -1) & (px < W)).expand_as(got)
    assert (got - ref)[inside].abs().max() < 1e-5
    assert got[~inside].abs().max() == 0
    # exactly on the border rows the two rules agree as well (grid_sample blends with the zero padding)
    edge = D.bilinear_zero(img, torch.full((B, 1, 1, 1), -0.5), torch.full((B, 1, 1, 1), 2.0))
    assert torch.allclose(edge[0, :, 0, 0], 0.5 * img[0, :, 0, 2], atol=1e-6)


def test_deform_groups_use_their_own_offsets():
    B, C, H, W, Co, dg = 1, 8, 6, 6, 2, 4
    x, w = _rand((B, C, H, W), 9), _rand((Co, C, 3, 3), 10, 0.3)
    off = _rand((B, 2 * dg * 9, H, W), 11, 1.5)
    m = torch.sigmoid(_rand((B, dg * 9, H, W), 12))
    full = D.deform_conv2d(x, off, w, None, 1, 1, 1, m)
    parts = 0
    for g in range(dg):      # a group's channels through a dg = 1 operator with that group's offsets / mask
        sl = slice(2 * g, 2 * g + 2)
        parts = parts + D.deform_conv2d(x[:, sl], off[:, 18 * g: 18 * g + 18], w[:, sl], None, 1, 1, 1, m[:, 9 * g: 9 * g + 9])
    assert (full - parts).abs().max() < 1e-5


def test_alignment_module_shapes_and_flow_guidance():
    from oracle.model_ref import Params
    P = Params(3)
    B, C, H, W = 1, 16, 8, 12
    x = _rand((B, C, H, W), 13)
    cond = _rand((B, 2 * C + 2, H, W), 14)
    flow = _rand((B, 2, H, W), 15, 2.0)
    out = D.deformable_alignment(P, "align", x, cond, flow, C, deform_groups=4)
    assert out.shape == (B, C, H, W) and torch.isfinite(out).all()
    out0 = D.deformable_alignment(P, "align", x, cond, torch.zeros_like(flow), C, deform_groups=4)
    assert (out - out0).abs().max() > 1e-3          # the flow really moves the sampling positions
