"""Structural pins of the architecture the HIP host modules build (the third-party model code is absent from the
reference tree, so parity of the model stack is unpinned -- SURVEY 8c; these are the pins that ARE available offline):
the parameter counts of the modules constructed by videovanish_amd/{unet,vae}.py must equal the published sizes of the
checkpoints the reference names at diffuerase.py:41-43 -- SD-1.5 UNet2DConditionModel 859,520,964 parameters,
sd-vae-ft-mse AutoencoderKL 83,653,863 -- and the AnimateDiff-style motion modules must follow 21 x (22 C^2 + 21 C).

The constructors run on CPU here with a shape-only weight source (meta tensors, no arithmetic, no GPU call)."""
import math

import pytest
import torch

from videovanish_amd import nn as vnn
from videovanish_amd import packing
from videovanish_amd.config import UNetConfig, VAEConfig


class _ShapeWeights:
    def __init__(self):
        self.seen = {}

    def _t(self, name, shape):
        if name in self.seen:
            assert self.seen[name] == tuple(shape), name
        self.seen[name] = tuple(shape)
        return torch.empty(tuple(shape), device="meta")

    def conv(self, name, cin, cout, k, gain=1.0):
        return self._t(name + ".weight", (cout, cin, k, k)), self._t(name + ".bias", (cout,))

    def linear(self, name, cin, cout, gain=1.0, bias=True):
        return self._t(name + ".weight", (cout, cin)), (self._t(name + ".bias", (cout,)) if bias else None)

    def norm(self, name, c):
        return self._t(name + ".weight", (c,)), self._t(name + ".bias", (c,))

    def normal(self, name, shape, std=1.0, mean=0.0):
        return torch.empty(tuple(shape), device="meta")

    def count(self, prefix, exclude=()):
        return sum(math.prod(s) for n, s in self.seen.items() if n.startswith(prefix) and not any(e in n for e in exclude))


class _ShapeCtx:
    device, dt, h16 = torch.device("meta"), 0, torch.bfloat16

    def __init__(self):
        self.src = _ShapeWeights()

    def dev(self, t, dtype=None):
        return t


def _build(monkeypatch):
    monkeypatch.setattr(packing, "pack_matrix", lambda w, h16, geglu=False: w)
    monkeypatch.setattr(packing, "pack_conv", lambda w, h16, cin_pad=None: (w, w.shape[1] * w.shape[2] * w.shape[3]))
    monkeypatch.setattr(packing, "geglu_interleave", lambda w, b: (w, b))
    monkeypatch.setattr(packing, "pack_motion_stream", lambda w, h16, heads=8: (w["proj_in.w"], w["proj_in.b"]))
    monkeypatch.setattr(vnn.Linear, "__call__", lambda self, *a, **k: None)      # CrossAttention projects the text K/V at build time
    from videovanish_amd.unet import BrushNet, UNetMotion
    from videovanish_amd.vae import VAE
    ctx = _ShapeCtx()
    ucfg, vcfg = UNetConfig(), VAEConfig()
    text = torch.empty((ucfg.text_len, ucfg.cross_dim), device="meta")
    UNetMotion(ctx, ucfg, text)
    BrushNet(ctx, ucfg, text)
    VAE(ctx, vcfg)
    return ctx.src, ucfg


def test_parameter_counts_match_published_checkpoints(monkeypatch):
    src, ucfg = _build(monkeypatch)
    unet2d = src.count("unet.", exclude=("motion_modules",))
    assert unet2d == 859_520_964                     # stable-diffusion-v1-5 UNet2DConditionModel
    assert src.count("vae.") == 83_653_863           # stabilityai/sd-vae-ft-mse AutoencoderKL
    motion = src.count("unet.") - unet2d
    chans = [320, 320, 640, 640, 1280, 1280, 1280, 1280, 1280] + [1280] * 3 + [1280] * 3 + [640] * 3 + [320] * 3
    assert len(chans) == 21 and motion == sum(22 * c * c + 21 * c for c in chans)      # 8 down + 1 mid + 12 up temporal transformers
    assert 0.44e9 < motion < 0.46e9                  # the public AnimateDiff v2 motion adapter is ~0.45 B
    # BrushNet = the UNet's down/mid/up blocks without cross-attention removed + 9-channel conv_in + 25 zero convs; no conv_out
    brush = src.count("brushnet.")
    zero = sum(math.prod(s) for n, s in src.seen.items() if n.startswith("brushnet.brushnet_"))
    conv_out = src.count("unet.conv_out") + src.count("unet.conv_norm_out")
    conv_in_extra = 320 * 5 * 9                      # conv_in takes 9 channels instead of 4
    assert brush - zero == unet2d - conv_out + conv_in_extra
    assert len([n for n in src.seen if n.startswith("brushnet.brushnet_") and n.endswith(".weight")]) == 12 + 1 + 15


def test_key_names_follow_diffusers_layout(monkeypatch):
    """Every parameter name maps (checkpoint.map_name) to the diffusers state-dict key families the checkpoints use."""
    from videovanish_amd.checkpoint import map_name
    src, _ = _build(monkeypatch)
    keys = {"unet": set(), "brushnet": set(), "vae": set()}
    for n in src.seen:
        base, suffix = n.rsplit(".", 1)
        comp, key = map_name(base)
        keys[comp].add(key + "." + suffix)
    u = keys["unet"]
    for k in ("conv_in.weight", "time_embedding.linear_1.weight", "time_embedding.linear_2.bias", "conv_norm_out.weight", "conv_out.bias",
              "down_blocks.0.resnets.0.time_emb_proj.weight", "down_blocks.0.attentions.1.transformer_blocks.0.attn2.to_k.weight",
              "down_blocks.0.attentions.0.transformer_blocks.0.ff.net.0.proj.weight", "down_blocks.0.attentions.0.transformer_blocks.0.ff.net.2.bias",
              "down_blocks.2.downsamplers.0.conv.weight", "down_blocks.3.resnets.1.conv2.weight", "mid_block.attentions.0.proj_in.weight",
              "up_blocks.0.resnets.2.conv_shortcut.weight", "up_blocks.1.upsamplers.0.conv.bias", "up_blocks.3.attentions.2.proj_out.weight",
              "down_blocks.1.motion_modules.0.temporal_transformer.transformer_blocks.0.attn2.to_out.0.bias",
              "mid_block.motion_modules.0.temporal_transformer.proj_in.weight", "up_blocks.0.motion_modules.2.temporal_transformer.norm.weight"):
        assert k in u, k
    assert not any("attentions" in k for k in u if k.startswith("down_blocks.3.") or k.startswith("up_blocks.0."))      # DownBlock2D / UpBlock2D
    assert not any(k.endswith("to_q.bias") or k.endswith("to_k.bias") or k.endswith("to_v.bias") for k in u)
    b = keys["brushnet"]
    assert "conv_in_condition.weight" in b and "brushnet_mid_block.weight" in b and "brushnet_down_blocks.11.bias" in b and "brushnet_up_blocks.14.weight" in b
    v = keys["vae"]
    for k in ("encoder.conv_in.weight", "encoder.down_blocks.3.resnets.1.norm2.bias", "encoder.mid_block.attentions.0.to_q.bias",
              "decoder.mid_block.attentions.0.group_norm.weight", "decoder.up_blocks.0.upsamplers.0.conv.weight",
              "decoder.up_blocks.3.resnets.2.conv2.bias", "quant_conv.weight", "post_quant_conv.bias", "decoder.conv_out.weight"):
        assert k in v, k


@pytest.mark.parametrize("H,W,Hv,Wv", [(5, 6, 10, 12), (5, 6, 9, 12), (2, 3, 3, 6), (23, 8, 45, 16), (5, 6, 10, 11), (5, 6, 9, 11), (2, 2, 3, 3)])
def test_upconv2x_phase_weights_equal_the_upsampled_convolution(H, W, Hv, Wv):
    """packing.upconv2x_phase_weight (nn.UpConv2x): 2x2 convolutions over the source image, one per output parity, with pad = 1 - parity and
    the 3x3 taps that fall on one source pixel summed, ARE conv3x3(interpolate(x, nearest)) -- in fp64, even sizes and Hv = 2 H - 1 / Wv = 2 W - 1
    (the last row / column / corner from the "last" tap table: the tap beyond the edge is zero padding, not the duplicated source pixel).  Mirrors the
    launch plan of UpConv2x.__call__: bulk, last row, last column, corner."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(5)
    Fr, C, N = 2, 3, 4
    x = torch.randn(Fr, C, H, W, generator=g, dtype=torch.float64)
    w = torch.randn(N, C, 3, 3, generator=g, dtype=torch.float64)
    ref = F.conv2d(F.interpolate(x, size=(Hv, Wv), mode="nearest"), w, padding=1)
    out = torch.full_like(ref, float("nan"))
    oh, ow = 2 * H - Hv, 2 * W - Wv

    def conv(src, key, pt, pl, hout, wout):                                   # 2x2 taps, top / left padding (pt, pl), bottom / right by bounds
        wp = packing.upconv2x_phase_weight(w, *key).double()
        return F.conv2d(F.pad(src, (pl, 1, pt, 1)), wp)[:, :, :hout, :wout]
    for py in (0, 1):
        for px in (0, 1):
            out[:, :, py:py + 2 * (H - oh):2, px:px + 2 * (W - ow):2] = conv(x, (py, px), 1 - py, 1 - px, H - oh, W - ow)
    if oh:
        for px in (0, 1):
            out[:, :, Hv - 1:Hv, px:px + 2 * (W - ow):2] = conv(x[:, :, H - 2:], ("last", px), 0, 1 - px, 1, W - ow)
    if ow:
        for py in (0, 1):
            out[:, :, py:py + 2 * (H - oh):2, Wv - 1:Wv] = conv(x[:, :, :, W - 2:], (py, "last"), 1 - py, 0, H - oh, 1)
    if oh and ow:
        out[:, :, Hv - 1:Hv, Wv - 1:Wv] = conv(x[:, :, H - 2:, W - 2:], ("last", "last"), 0, 0, 1, 1)
    assert not torch.isnan(out).any()
    assert (out - ref).abs().max().item() <= 1e-5          # (the phase weights are summed in fp32)
