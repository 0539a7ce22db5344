"""ORACLE (test infrastructure, not product code): fp32 CPU PyTorch restatement of the DiffuEraser model stack.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.  The product path
(videovanish_amd/) never does: it fails loudly if the HIP extension is missing.

PARITY UNPINNED for everything in this file: the reference's arithmetic for this part of the path lives in
un-vendored third-party code that is absent from /root/reference:
  * github.com/calledit/DiffuEraser_np_array, default branch, no commit pin (install_videovanish.sh:78) --
    packages `diffueraser`, `propainter`, `libs`; call sites reference diffuerase.py:8-9,39-45,52-57,62-67
  * diffusers==0.29.2 (install_videovanish.sh:79)
The reference holds no tests, golden vectors or fixtures for it and passes no seed (diffuerase.py:62-67).  What
follows restates the *published architectures* those packages instantiate: SD-1.5
UNet2DConditionModel + AnimateDiff motion modules (diffusers 0.29 UNetMotionModel), BrushNet with cross
attention, AutoencoderKL (sd-vae-ft-mse), DDIM / TCD schedulers.  Layout here is plain NCHW torch; the HIP path
uses frames-major NHWC, so agreement between the two is a genuine check of the kernels' indexing.

Weights come from videovanish_amd.weights.SyntheticWeights (name-seeded, identical for oracle and product).
"""
import math

import torch
import torch.nn.functional as F

from videovanish_amd.config import UNetConfig, VAEConfig
from videovanish_amd.weights import SyntheticWeights


class Params:
    """Lazy fp32 parameter cache keyed by diffusers-style names."""

    def __init__(self, seed=0):
        self.src = SyntheticWeights(seed)
        self.cache = {}

    def conv(self, name, cin, cout, k, gain=1.0):
        if name not in self.cache:
            self.cache[name] = self.src.conv(name, cin, cout, k, gain)
        return self.cache[name]

    def linear(self, name, cin, cout, gain=1.0, bias=True):
        if name not in self.cache:
            self.cache[name] = self.src.linear(name, cin, cout, gain, bias)
        return self.cache[name]

    def norm(self, name, c):
        if name not in self.cache:
            self.cache[name] = self.src.norm(name, c)
        return self.cache[name]


# ----------------------------------------------------------------------------------------------------------------
# primitives
# ----------------------------------------------------------------------------------------------------------------
def conv2d(P, name, x, cout, k=3, stride=1, pad=1, gain=1.0):
    w, b = P.conv(name, x.shape[1], cout, k, gain)
    return F.conv2d(x, w, b, stride=stride, padding=pad)


def linear(P, name, x, cout, bias=True, gain=1.0):
    w, b = P.linear(name, x.shape[-1], cout, gain, bias)
    return F.linear(x, w, b)


def group_norm(P, name, x, groups, eps):
    g, b = P.norm(name, x.shape[1])
    return F.group_norm(x, groups, g, b, eps)


def layer_norm(P, name, x):
    g, b = P.norm(name, x.shape[-1])
    return F.layer_norm(x, (x.shape[-1],), g, b, 1e-5)


def attention(q, k, v, heads):
    """q [B,Nq,C], k/v [B,Nk,C] -> [B,Nq,C]; scale d^-1/2, softmax in fp32."""
    B, Nq, C = q.shape
    d = C // heads
    q = q.view(B, Nq, heads, d).transpose(1, 2)
    k = k.view(B, -1, heads, d).transpose(1, 2)
    v = v.view(B, -1, heads, d).transpose(1, 2)
    Nk = k.shape[2]
    if B * heads * Nq * Nk <= (1 << 28):
        s = (q @ k.transpose(-1, -2)) * (d ** -0.5)
        o = torch.softmax(s, dim=-1) @ v
        return o.transpose(1, 2).reshape(B, Nq, C)
    # the benchmarked geometries (N = 14400 at 720p, 32400 at 1080p: 13 / 34 GB of scores for 2 / 1 frames): torch's fused fp32 CPU kernel for the same
    # expression (softmax(q k^T d^-1/2) v, no score matrix in memory, ~5x faster than blocks of query rows through matmul + softmax; agrees with the direct
    # form to fp32 rounding: tests/test_oracle_cpu.py pins the two against each other just above the switch)
    o = F.scaled_dot_product_attention(q, k, v, scale=d ** -0.5)
    return o.transpose(1, 2).reshape(B, Nq, C)


def timestep_embedding(t, dim):
    """diffusers get_timestep_embedding(flip_sin_to_cos=True, downscale_freq_shift=0): [cos | sin]."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    a = float(t) * freqs
    return torch.cat([torch.cos(a), torch.sin(a)])[None]


def sinusoidal_pos_emb(n, dim):
    """diffusers SinusoidalPositionalEmbedding: pe[p,2i]=sin(p*w_i), pe[p,2i+1]=cos(p*w_i) (App. D.2)."""
    pos = torch.arange(n, dtype=torch.float32)[:, None]
    div = torch.exp(torch.arange(0, dim, 2, dtype=torch.float32) * (-math.log(10000.0) / dim))
    pe = torch.zeros(n, dim)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


# ----------------------------------------------------------------------------------------------------------------
# UNet / BrushNet blocks (App. D.1-D.3)
# ----------------------------------------------------------------------------------------------------------------
def resnet_block(P, name, x, temb, cout, groups, eps=1e-5, out_gain=1.0):
    h = conv2d(P, name + ".conv1", F.silu(group_norm(P, name + ".norm1", x, groups, eps)), cout)
    if temb is not None:
        h = h + linear(P, name + ".time_emb_proj", F.silu(temb), cout)[:, :, None, None]
    h = conv2d(P, name + ".conv2", F.silu(group_norm(P, name + ".norm2", h, groups, eps)), cout, gain=out_gain)
    if x.shape[1] != cout:
        x = conv2d(P, name + ".conv_shortcut", x, cout, k=1, pad=0)
    return x + h


def feed_forward(P, name, x):
    C = x.shape[-1]
    h = linear(P, name + ".net.0.proj", x, 8 * C)
    a, g = h.chunk(2, dim=-1)
    return linear(P, name + ".net.2", a * F.gelu(g), C)


def attn_layer(P, name, x, ctx, heads):
    C = x.shape[-1]
    q = linear(P, name + ".to_q", x, C, bias=False)
    k = linear(P, name + ".to_k", ctx, C, bias=False)
    v = linear(P, name + ".to_v", ctx, C, bias=False)
    return linear(P, name + ".to_out.0", attention(q, k, v, heads), C)


def spatial_transformer(P, name, x, text, cfg):
    """Transformer2DModel with one BasicTransformerBlock; proj_in/out are 1x1 convs (SD-1.5)."""
    B, C, H, W = x.shape
    res = x
    h = group_norm(P, name + ".norm", x, cfg.groups, 1e-6)
    h = conv2d(P, name + ".proj_in", h, C, k=1, pad=0)
    h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
    b = name + ".transformer_blocks.0"
    n = layer_norm(P, b + ".norm1", h)
    h = h + attn_layer(P, b + ".attn1", n, n, cfg.heads)
    n = layer_norm(P, b + ".norm2", h)
    h = h + attn_layer(P, b + ".attn2", n, text.expand(B, -1, -1), cfg.heads)
    h = h + feed_forward(P, b + ".ff", layer_norm(P, b + ".norm3", h))
    h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
    return conv2d(P, name + ".proj_out", h, C, k=1, pad=0) + res


def motion_module(P, name, x, cfg):
    """AnimateDiff temporal transformer over the frame axis (App. D.2).  x: [F,C,H,W] = one clip."""
    Fr, C, H, W = x.shape
    res = x
    h = x.permute(1, 0, 2, 3)[None]                       # [1,C,F,H,W]: GroupNorm statistics pool over the clip
    g, b = P.norm(name + ".norm", C)
    h = F.group_norm(h, cfg.groups, g, b, 1e-6)
    h = h[0].permute(2, 3, 1, 0).reshape(H * W, Fr, C)    # [HW, F, C]
    h = linear(P, name + ".proj_in", h, C)
    pe = sinusoidal_pos_emb(cfg.motion_max_seq, C)[:Fr]
    blk = name + ".transformer_blocks.0"
    n = layer_norm(P, blk + ".norm1", h) + pe
    h = h + attn_layer(P, blk + ".attn1", n, n, cfg.heads)
    n = layer_norm(P, blk + ".norm2", h) + pe
    h = h + attn_layer(P, blk + ".attn2", n, n, cfg.heads)
    h = h + feed_forward(P, blk + ".ff", layer_norm(P, blk + ".norm3", h))
    h = linear(P, name + ".proj_out", h, C)
    return h.reshape(H, W, Fr, C).permute(2, 3, 0, 1) + res


def time_embed(P, name, t, cfg):
    e = timestep_embedding(t, cfg.block_out[0])
    e = linear(P, name + ".linear_1", e, cfg.temb_dim)
    return linear(P, name + ".linear_2", F.silu(e), cfg.temb_dim)


def _backbone(P, pre, x, temb, text, cfg, motion, add_down=None, add_mid=None, add_up=None, collect_up=False):
    """Shared down/mid/up traversal.  Returns (x, down_skips, mid, up_outputs)."""
    L = len(cfg.block_out)
    skips = [x]
    # the ONE switch for the BrushNet down-residual site (config.UNetConfig.brushnet_add, [UNVERIFIED-3P]): "hidden" = into the running hidden state
    # (the skip taken after a layer then carries it; the conv_in skip is taken before), "skip" = onto the skip copies after the down path
    into_hidden = add_down is not None and cfg.brushnet_add == "hidden"
    assert cfg.brushnet_add in ("skip", "hidden")
    hid = list(add_down) if into_hidden else None
    if into_hidden:
        x = x + hid.pop(0)
    cin = cfg.block_out[0]
    for i, cout in enumerate(cfg.block_out):
        for j in range(cfg.layers_per_block):
            x = resnet_block(P, f"{pre}.down_blocks.{i}.resnets.{j}", x, temb, cout, cfg.groups)
            if cfg.attn_levels[i]:
                x = spatial_transformer(P, f"{pre}.down_blocks.{i}.attentions.{j}", x, text, cfg)
            if motion:
                x = motion_module(P, f"{pre}.down_blocks.{i}.motion_modules.{j}", x, cfg)
            if into_hidden:
                x = x + hid.pop(0)
            skips.append(x)
            cin = cout
        if i < L - 1:
            x = conv2d(P, f"{pre}.down_blocks.{i}.downsamplers.0.conv", x, cout, stride=2)
            if into_hidden:
                x = x + hid.pop(0)
            skips.append(x)
    down_skips = list(skips)
    if add_down is not None and not into_hidden:
        skips = [s + a for s, a in zip(skips, add_down)]
    C = cfg.block_out[-1]
    x = resnet_block(P, f"{pre}.mid_block.resnets.0", x, temb, C, cfg.groups)
    x = spatial_transformer(P, f"{pre}.mid_block.attentions.0", x, text, cfg)
    if motion:
        x = motion_module(P, f"{pre}.mid_block.motion_modules.0", x, cfg)
    x = resnet_block(P, f"{pre}.mid_block.resnets.1", x, temb, C, cfg.groups)
    mid = x
    if add_mid is not None:
        x = x + add_mid
    ups = []
    add_up = list(add_up) if add_up is not None else None
    rev = list(reversed(cfg.block_out))
    rev_attn = list(reversed(cfg.attn_levels))
    for i, cout in enumerate(rev):
        for j in range(cfg.layers_per_block + 1):
            x = torch.cat([x, skips.pop()], dim=1)
            x = resnet_block(P, f"{pre}.up_blocks.{i}.resnets.{j}", x, temb, cout, cfg.groups)
            if rev_attn[i]:
                x = spatial_transformer(P, f"{pre}.up_blocks.{i}.attentions.{j}", x, text, cfg)
            if motion:
                x = motion_module(P, f"{pre}.up_blocks.{i}.motion_modules.{j}", x, cfg)
            if collect_up:
                ups.append(x)
            if add_up is not None:
                x = x + add_up.pop(0)
        if i < L - 1:
            x = F.interpolate(x, size=skips[-1].shape[-2:], mode="nearest")   # to the skip's size, not blindly x2
            x = conv2d(P, f"{pre}.up_blocks.{i}.upsamplers.0.conv", x, cout)
            if collect_up:
                ups.append(x)
            if add_up is not None:
                x = x + add_up.pop(0)
    return x, down_skips, mid, ups


def brushnet_forward(P, x9, t, text, cfg: UNetConfig, scale=1.0):
    """BrushNet branch (App. D.3): x9 = cat[noisy latents(4), masked-image latents(4), mask(1)], per frame."""
    pre = "brushnet"
    temb = time_embed(P, pre + ".time_embedding", t, cfg)
    x = conv2d(P, pre + ".conv_in", x9, cfg.block_out[0])
    _, downs, mid, ups = _backbone(P, pre, x, temb, text, cfg, motion=False, collect_up=True)
    zg = cfg.zero_conv_gain
    d = [conv2d(P, f"{pre}.brushnet_down_blocks.{i}", s, s.shape[1], k=1, pad=0, gain=zg) * scale for i, s in enumerate(downs)]
    m = conv2d(P, f"{pre}.brushnet_mid_block", mid, mid.shape[1], k=1, pad=0, gain=zg) * scale
    u = [conv2d(P, f"{pre}.brushnet_up_blocks.{i}", s, s.shape[1], k=1, pad=0, gain=zg) * scale for i, s in enumerate(ups)]
    return d, m, u


def unet_forward(P, latents, t, text, cfg: UNetConfig, brush=None):
    """UNetMotionModel forward on one clip: latents [F,4,h,w] -> eps [F,4,h,w]."""
    pre = "unet"
    temb = time_embed(P, pre + ".time_embedding", t, cfg)
    x = conv2d(P, pre + ".conv_in", latents, cfg.block_out[0])
    d, m, u = brush if brush is not None else (None, None, None)
    x, _, _, _ = _backbone(P, pre, x, temb, text, cfg, motion=True, add_down=d, add_mid=m, add_up=u)
    x = F.silu(group_norm(P, pre + ".conv_norm_out", x, cfg.groups, 1e-5))
    return conv2d(P, pre + ".conv_out", x, cfg.out_ch)


def text_states(P, cfg: UNetConfig):
    """Prompt is "" => one constant [1,77,768] tensor; synthetic setup uses a seeded random tensor (App. D.5)."""
    return P.src.normal("text_states", (1, cfg.text_len, cfg.cross_dim))


# ----------------------------------------------------------------------------------------------------------------
# VAE (App. D.4)
# ----------------------------------------------------------------------------------------------------------------
def vae_attn(P, name, x, groups):
    B, C, H, W = x.shape
    h = group_norm(P, name + ".group_norm", x, groups, 1e-6).permute(0, 2, 3, 1).reshape(B, H * W, C)
    q = linear(P, name + ".to_q", h, C)
    k = linear(P, name + ".to_k", h, C)
    v = linear(P, name + ".to_v", h, C)
    o = linear(P, name + ".to_out.0", attention(q, k, v, 1), C)
    return o.reshape(B, H, W, C).permute(0, 3, 1, 2) + x


def vae_mid(P, pre, x, cfg):
    C = x.shape[1]
    x = resnet_block(P, pre + ".mid_block.resnets.0", x, None, C, cfg.groups, 1e-6)
    x = vae_attn(P, pre + ".mid_block.attentions.0", x, cfg.groups)
    return resnet_block(P, pre + ".mid_block.resnets.1", x, None, C, cfg.groups, 1e-6)


def vae_encode(P, img, cfg: VAEConfig):
    """img [B,3,H,W] in [-1,1] -> latent mean [B,4,H/8,W/8] * scaling (posterior mode; deterministic)."""
    pre = "vae.encoder"
    x = conv2d(P, pre + ".conv_in", img, cfg.block_out[0])
    L = len(cfg.block_out)
    for i, cout in enumerate(cfg.block_out):
        for j in range(cfg.layers_per_block):
            x = resnet_block(P, f"{pre}.down_blocks.{i}.resnets.{j}", x, None, cout, cfg.groups, 1e-6)
        if i < L - 1:
            x = F.pad(x, (0, 1, 0, 1))
            x = conv2d(P, f"{pre}.down_blocks.{i}.downsamplers.0.conv", x, cout, stride=2, pad=0)
    x = vae_mid(P, pre, x, cfg)
    x = F.silu(group_norm(P, pre + ".conv_norm_out", x, cfg.groups, 1e-6))
    x = conv2d(P, pre + ".conv_out", x, 2 * cfg.latent_ch)
    x = conv2d(P, "vae.quant_conv", x, 2 * cfg.latent_ch, k=1, pad=0)
    return x[:, :cfg.latent_ch] * cfg.scaling


def vae_decode(P, z, cfg: VAEConfig):
    """z [B,4,h,w] (scaled latents) -> image [B,3,8h,8w] in [-1,1]."""
    pre = "vae.decoder"
    x = conv2d(P, "vae.post_quant_conv", z / cfg.scaling, cfg.latent_ch, k=1, pad=0)
    rev = list(reversed(cfg.block_out))
    x = conv2d(P, pre + ".conv_in", x, rev[0])
    x = vae_mid(P, pre, x, cfg)
    for i, cout in enumerate(rev):
        for j in range(cfg.layers_per_block + 1):
            x = resnet_block(P, f"{pre}.up_blocks.{i}.resnets.{j}", x, None, cout, cfg.groups, 1e-6)
        if i < len(rev) - 1:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            x = conv2d(P, f"{pre}.up_blocks.{i}.upsamplers.0.conv", x, cout)
    x = F.silu(group_norm(P, pre + ".conv_norm_out", x, cfg.groups, 1e-6))
    return conv2d(P, pre + ".conv_out", x, 3)


# ----------------------------------------------------------------------------------------------------------------
# schedulers (App. D.5)
# ----------------------------------------------------------------------------------------------------------------
def alphas_cumprod():
    betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float64) ** 2
    return torch.cumprod(1.0 - betas, 0)


def ddim_timesteps(steps):
    ratio = 1000 // steps
    return [int(round(i * ratio)) + 1 for i in range(steps)][::-1]


def ddim_step(x, eps, t, steps, ac):
    prev = t - 1000 // steps
    a_t = float(ac[t])
    a_p = float(ac[prev]) if prev >= 0 else float(ac[0])
    x0 = (x - (1 - a_t) ** 0.5 * eps) / a_t ** 0.5
    return a_p ** 0.5 * x0 + (1 - a_p) ** 0.5 * eps


def tcd_timesteps(steps):
    """diffusers TCDScheduler.set_timesteps restated (original_inference_steps = 50, strength 1): the 50 origin timesteps
    20*k - 1, reversed, picked at floor(linspace(0, 50, steps, endpoint=False)): 2 -> [999, 499], 4 -> [999, 759, 499, 259]."""
    origin = [20 * k - 1 for k in range(1, 51)][::-1]
    return [origin[int(math.floor(i * 50.0 / steps))] for i in range(steps)]


def tcd_step(x, eps, t, t_prev, ac, z, gamma=0.3):
    """diffusers TCDScheduler.step restated: t_prev=None on the last step (prev timestep 0, no re-noising)."""
    a_t = float(ac[t])
    x0 = (x - (1 - a_t) ** 0.5 * eps) / a_t ** 0.5
    tp = 0 if t_prev is None else t_prev
    s = int(math.floor((1 - gamma) * tp))
    a_s, a_p = float(ac[s]), float(ac[tp])
    x_s = a_s ** 0.5 * x0 + (1 - a_s) ** 0.5 * eps
    if t_prev is None:
        return x_s
    return (a_p / a_s) ** 0.5 * x_s + (1 - a_p / a_s) ** 0.5 * z


def add_noise(x0, noise, t, ac):
    a = float(ac[t])
    return a ** 0.5 * x0 + (1 - a) ** 0.5 * noise
