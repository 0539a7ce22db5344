"""SD VAE (AutoencoderKL, sd-vae-ft-mse shapes) encoder / decoder on HIP kernels (reference call
site diffuerase.py:62-67 -> DiffuEraser.forward -> vae.encode / vae.decode)."""
import torch

from . import hip
from .nn import Conv, GroupNorm, Linear, ResBlock, SelfAttention, UpConv2x


class _MidAttn:
    def __init__(self, ctx, name, C, groups, precise=False):
        self.norm = GroupNorm(ctx, name + ".group_norm", C, groups, 1e-6, precise=precise)
        self.attn = SelfAttention(ctx, name, C, 1, qkv_bias=True, precise=precise)

    def __call__(self, x, F, H, W):
        return self.attn.spatial(self.norm(x, F, H * W), x, F, H * W)


class _Mid:
    def __init__(self, ctx, pre, C, groups, precise=False):
        self.r0 = ResBlock(ctx, pre + ".mid_block.resnets.0", C, C, groups, 1e-6, precise=precise)
        self.a = _MidAttn(ctx, pre + ".mid_block.attentions.0", C, groups, precise=precise)
        self.r1 = ResBlock(ctx, pre + ".mid_block.resnets.1", C, C, groups, 1e-6, precise=precise)

    def __call__(self, x, F, H, W):
        return self.r1(self.a(self.r0(x, F, H, W), F, H, W), F, H, W)


class VAE:
    def __init__(self, ctx, cfg, precise_decoder=False):
        """precise_decoder: every decoder GEMM runs in split precision (three products in one K-concatenated launch, nn.split3_weight): the decoder's operand
        rounding goes straight to the pixels and is the largest single term of the end-to-end error (profiles/r2_parity_*)."""
        self.ctx, self.cfg = ctx, cfg
        pd = self.precise = bool(precise_decoder)
        bo, lpb, g = cfg.block_out, cfg.layers_per_block, cfg.groups
        self.factor = 2 ** (len(bo) - 1)
        # ---- encoder
        pre = "vae.encoder"
        self.e_in = Conv(ctx, pre + ".conv_in", 3, bo[0], cin_pad=8)
        self.e_blocks, self.e_down = [], []
        cin = bo[0]
        for i, cout in enumerate(bo):
            rs = []
            for j in range(lpb):
                rs.append(ResBlock(ctx, f"{pre}.down_blocks.{i}.resnets.{j}", cin, cout, g, 1e-6))
                cin = cout
            self.e_blocks.append(rs)
            if i < len(bo) - 1:
                self.e_down.append(Conv(ctx, f"{pre}.down_blocks.{i}.downsamplers.0.conv", cout, cout))
        self.e_mid = _Mid(ctx, pre, bo[-1], g)
        self.e_norm = GroupNorm(ctx, pre + ".conv_norm_out", bo[-1], g, 1e-6)
        self.e_out = Conv(ctx, pre + ".conv_out", bo[-1], 2 * cfg.latent_ch)
        self.quant = Conv(ctx, "vae.quant_conv", 2 * cfg.latent_ch, 2 * cfg.latent_ch, k=1, rows=cfg.latent_ch)   # mean half only
        # ---- decoder
        pre = "vae.decoder"
        rev = list(reversed(bo))
        self.post_quant = Conv(ctx, "vae.post_quant_conv", cfg.latent_ch, cfg.latent_ch, k=1, cin_pad=8, precise=pd)
        self.d_in = Conv(ctx, pre + ".conv_in", cfg.latent_ch, rev[0], cin_pad=8, precise=pd)
        self.d_mid = _Mid(ctx, pre, rev[0], g, precise=pd)
        self.d_blocks, self.d_up = [], []
        cin = rev[0]
        for i, cout in enumerate(rev):
            rs = []
            for j in range(lpb + 1):
                rs.append(ResBlock(ctx, f"{pre}.up_blocks.{i}.resnets.{j}", cin, cout, g, 1e-6, precise=pd))
                cin = cout
            self.d_blocks.append(rs)
            if i < len(rev) - 1:
                self.d_up.append(UpConv2x(ctx, f"{pre}.up_blocks.{i}.upsamplers.0.conv", cout, cout, precise=pd))
        self.d_norm = GroupNorm(ctx, pre + ".conv_norm_out", rev[-1], g, 1e-6, precise=pd)
        self.d_out = Conv(ctx, pre + ".conv_out", rev[-1], 3, precise=pd)

    def encode(self, img8, F, H, W):
        """img8: h16 [F*H*W, 8] (3 real channels, [-1,1]) -> scaled latent means fp32 [F, H/f, W/f, 4]."""
        x, _, _ = self.e_in(img8, F, H, W)
        for i, rs in enumerate(self.e_blocks):
            last = i < len(self.e_blocks) - 1
            for j, r in enumerate(rs):
                # the block's last ResBlock feeds only the strided conv: hand over h16 (the conv rounds its input to h16 anyway,
                # so the values are identical) and the conv runs on the h16 LDS-DMA loader instead of the fp32 one
                x = r(x, F, H, W, out_dtype=self.ctx.h16 if (last and j == len(rs) - 1) else torch.float32)
            if i < len(self.e_blocks) - 1:
                # F.pad(x,(0,1,0,1)) + stride-2 conv, pad 0: bottom/right zeros come from the bounds check
                x, H, W = self.e_down[i](x, F, H, W, stride=2, pad=0, Hout=H // 2, Wout=W // 2)
        x = self.e_mid(x, F, H, W)
        h = self.e_norm(x, F, H * W, silu=True)
        m, _, _ = self.e_out(h, F, H, W)
        z, _, _ = self.quant(m, F, H, W, scale=self.cfg.scaling)
        return z.view(F, H, W, self.cfg.latent_ch)

    def decode(self, lat, F, h, w):
        """lat: fp32 [F,h,w,4] scaled latents -> fp32 [F, H, W, 3] in [-1,1] (un-clamped)."""
        cfg = self.cfg
        if self.precise:
            z8 = hip.pad_channels_f32(lat, 8, 1.0 / cfg.scaling).view(F * h * w, 8)      # fp32: the precise conv splits it into hi + lo
        else:
            z8 = hip.pad_channels(self.ctx.dt, lat, 8, 1.0 / cfg.scaling).view(F * h * w, 8)
        if self.precise:
            pq4, _, _ = self.post_quant(z8, F, h, w)                   # dense [M, 4] (the accumulation passes read it back as res0)
            pq = hip.pad_channels_f32(pq4, 8)
        else:
            pq = torch.zeros((F * h * w, 8), dtype=torch.float32, device=lat.device)
            self.post_quant(z8, F, h, w, out=pq)
        x, _, _ = self.d_in(pq, F, h, w)
        H, W = h, w
        x = self.d_mid(x, F, H, W)
        for i, rs in enumerate(self.d_blocks):
            last = i < len(self.d_blocks) - 1 and not self.precise
            for j, r in enumerate(rs):
                x = r(x, F, H, W, out_dtype=self.ctx.h16 if (last and j == len(rs) - 1) else torch.float32)   # feeds only the upsample conv
            if i < len(self.d_blocks) - 1:
                x, H, W = self.d_up[i](x, F, H, W, Hv=2 * H, Wv=2 * W)      # nearest x2 fused into the conv gather
        hh = self.d_norm(x, F, H * W, silu=True)
        out, _, _ = self.d_out(hh, F, H, W)
        return out.view(F, H, W, 3)
