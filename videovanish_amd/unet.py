"""Motion-UNet and BrushNet branch on HIP kernels (architecture: SURVEY App. D.1-D.3; the reference reaches them
through `DiffuEraser.forward`, call site reference diffuerase.py:62-67)."""
import math
import threading

import torch

from . import hip
from .nn import Conv, GroupNorm, Linear, MotionModule, ResBlock, SpatialTransformer, UpConv2x


def timestep_embedding(t, dim):
    """[cos | sin] sinusoidal embedding (diffusers flip_sin_to_cos=True, freq shift 0); tiny host-side table."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    a = float(t) * freqs
    return torch.cat([torch.cos(a), torch.sin(a)])[None]


def sinusoidal_pos_emb(n, dim):
    pos = torch.arange(n, dtype=torch.float32)[:, None]
    div = torch.exp(torch.arange(0, dim, 2, dtype=torch.float32) * (-math.log(10000.0) / dim))
    pe = torch.zeros(n, dim)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


class _Backbone:
    """Shared down/mid/up structure of the UNet and of BrushNet."""

    def __init__(self, ctx, pre, cfg, text_h16, motion, in_pad, precise_io=True):
        """precise_io (round 4, tools/parity_rank.py): the layers whose operand rounding buys the most error for the least work run in split
        precision (nn.split3_weight: hi / lo operands, 3 x the layer's FLOPs) -- the time-embedding linears and every ResBlock's time_emb_proj
        (one-row GEMMs, hoisted out of the denoise loop), and the UNet's conv_in / conv_out (4 <-> 320 channels: 0.1 % of a step's FLOPs, but
        their rounding sits directly on the network's input latents and on the predicted noise)."""
        self.ctx, self.cfg, self.pre, self.motion = ctx, cfg, pre, motion
        self.precise_io = bool(precise_io)
        bo, L, lpb = cfg.block_out, len(cfg.block_out), cfg.layers_per_block
        td = cfg.temb_dim
        self.t1 = Linear(ctx, pre + ".time_embedding.linear_1", bo[0], td, precise=self.precise_io)
        self.t2 = Linear(ctx, pre + ".time_embedding.linear_2", td, td, precise=self.precise_io)
        cin_real = cfg.in_ch if motion else cfg.brush_in_ch
        self.conv_in = Conv(ctx, pre + ".conv_in", cin_real, bo[0], cin_pad=in_pad, precise=self.precise_io and motion)
        pes = {}

        def pe(C):
            if C not in pes:
                pes[C] = ctx.dev(sinusoidal_pos_emb(cfg.motion_max_seq, C))
            return pes[C]

        def layer(name_res, name_attn, name_mot, cin, cout, attn):
            r = ResBlock(ctx, name_res, cin, cout, cfg.groups, 1e-5, td, precise_temb=self.precise_io, h16_mid=True)
            a = SpatialTransformer(ctx, name_attn, cout, cfg, text_h16) if attn else None
            m = MotionModule(ctx, name_mot, cout, cfg, pe(cout)) if motion else None
            return r, a, m

        self.down, self.downs = [], []
        skip_ch = [bo[0]]
        cin = bo[0]
        for i, cout in enumerate(bo):
            layers = []
            for j in range(lpb):
                layers.append(layer(f"{pre}.down_blocks.{i}.resnets.{j}", f"{pre}.down_blocks.{i}.attentions.{j}",
                                    f"{pre}.down_blocks.{i}.motion_modules.{j}", cin, cout, cfg.attn_levels[i]))
                cin = cout
                skip_ch.append(cout)
            self.down.append(layers)
            if i < L - 1:
                self.downs.append(Conv(ctx, f"{pre}.down_blocks.{i}.downsamplers.0.conv", cout, cout))
                skip_ch.append(cout)
        self.skip_ch = list(skip_ch)
        C = bo[-1]
        self.mid_r0 = ResBlock(ctx, f"{pre}.mid_block.resnets.0", C, C, cfg.groups, 1e-5, td, precise_temb=self.precise_io, h16_mid=True)
        self.mid_a = SpatialTransformer(ctx, f"{pre}.mid_block.attentions.0", C, cfg, text_h16)
        self.mid_m = MotionModule(ctx, f"{pre}.mid_block.motion_modules.0", C, cfg, pe(C)) if motion else None
        self.mid_r1 = ResBlock(ctx, f"{pre}.mid_block.resnets.1", C, C, cfg.groups, 1e-5, td, precise_temb=self.precise_io, h16_mid=True)
        self.up, self.ups, self.up_ch = [], [], []
        rev, rev_attn = list(reversed(bo)), list(reversed(cfg.attn_levels))
        x_ch = C
        for i, cout in enumerate(rev):
            layers = []
            for j in range(lpb + 1):
                sc = skip_ch.pop()
                layers.append(layer(f"{pre}.up_blocks.{i}.resnets.{j}", f"{pre}.up_blocks.{i}.attentions.{j}",
                                    f"{pre}.up_blocks.{i}.motion_modules.{j}", x_ch + sc, cout, rev_attn[i]))
                x_ch = cout
                self.up_ch.append(cout)
            self.up.append(layers)
            if i < L - 1:
                self.ups.append(UpConv2x(ctx, f"{pre}.up_blocks.{i}.upsamplers.0.conv", cout, cout))
                self.up_ch.append(cout)

    def temb(self, t, key=None):
        """SiLU(time embedding) of timestep t -- or, after prepare_temb(), the table {id(ResBlock): conv1 bias + time_emb_proj row} of t.
        key: the stream the tables were prepared on (default: the current one; the two-stream schedule passes its main stream)."""
        hit = self.__dict__.get("_temb_tables", {}).get(torch.cuda.current_stream().cuda_stream if key is None else key, {}).get(int(t))
        if hit is not None:
            return hit
        e = self.ctx.dev(timestep_embedding(t, self.cfg.block_out[0]))
        e = hip.silu(self.t1(e))
        return hip.silu(self.t2(e))            # every consumer applies SiLU first (ResnetBlock2D.time_emb_proj)

    def resblocks(self):
        out = [r for layers in self.down for (r, _, _) in layers] + [self.mid_r0, self.mid_r1]
        return out + [r for layers in self.up for (r, _, _) in layers]

    def prepare_temb(self, ts):
        """Hoist the time-embedding work out of the denoise loop: nothing in it depends on the latents, so the two embedding linears and the
        time_emb_proj of every ResBlock run ONCE for all S timesteps of the schedule as GEMMs with S rows (instead of ~27 one-row launches of
        ~50 us latency each per backbone and step: 0.7 % of a step).  Row s of an S-row GEMM is computed exactly like the one-row launch
        (same tile kernel, same k order), so the values are bit-identical.  Tables are kept per launch stream (chunks of one rank may run
        on several streams: a table is produced and consumed on the same stream, or on its side stream behind a wait).
        The chunk lanes are host threads that share this object: one dict PER STREAM KEY, built aside and published by a single
        item assignment, so no lane ever iterates or resizes a dict another lane is reading."""
        ts = [int(t) for t in ts]
        key = torch.cuda.current_stream().cuda_stream
        tables = self.__dict__.setdefault("_temb_tables", {})
        mine = tables.get(key)
        if mine is not None and all(t in mine for t in ts):
            return
        e = self.ctx.dev(torch.cat([timestep_embedding(t, self.cfg.block_out[0]) for t in ts], 0))
        st = hip.silu(self.t2(hip.silu(self.t1(e))))
        rows = {id(r): r.temb_bias(st) for r in self.resblocks()}
        tables[key] = {t: {rid: b[i] for rid, b in rows.items()} for i, t in enumerate(ts)}      # another schedule on this stream: the old rows go

    def run_down(self, x_in, F, h, w, st, into_hidden=None):
        """x_in: h16 [F*h*w, in_pad].  Returns (x, skips=[(tensor,H,W)...], (H,W)).
        into_hidden (UNetConfig.brushnet_add == "hidden"): one callable per skip position, x -> x + BrushNet residual of that position (the zero
        convolution with the UNet state as its fused residual: BrushNet.hidden_adders) -- applied to the running hidden state after conv_in (whose
        skip is taken BEFORE), after every layer and after every downsampler, so the skips taken there carry the residual."""
        x, _, _ = self.conv_in(x_in, F, h, w)
        H, W = h, w
        skips = [(x, H, W)]
        hid = list(into_hidden) if into_hidden is not None else None
        if hid is not None:
            x = hid.pop(0)(x)
        L = len(self.cfg.block_out)
        for i, layers in enumerate(self.down):
            for (r, a, m) in layers:
                x = r(x, F, H, W, silu_temb=st, want_gn=a is not None)      # (its output feeds the spatial transformer's GroupNorm)
                if a is not None:
                    x = a(x, F, H, W)
                if m is not None:
                    x = m(x, F, H, W)
                if hid is not None:
                    x = hid.pop(0)(x)
                skips.append((x, H, W))
            if i < L - 1:
                x, H, W = self.downs[i](x, F, H, W, stride=2)
                if hid is not None:
                    x = hid.pop(0)(x)
                skips.append((x, H, W))
        return x, skips, (H, W)

    def run_mid(self, x, F, H, W, st):
        x = self.mid_r0(x, F, H, W, silu_temb=st, want_gn=True)
        x = self.mid_a(x, F, H, W)
        if self.mid_m is not None:
            x = self.mid_m(x, F, H, W)
        return self.mid_r1(x, F, H, W, silu_temb=st)

    def run_up(self, x, F, H, W, skips, st, add_up=None, collect=False):
        """skips: list of (tensor,H,W) consumed from the end.  add_up: per-layer residual tensors fused into the last
        GEMM of each layer (res1).  collect: return the per-layer outputs (BrushNet)."""
        outs = []
        add_up = list(add_up) if add_up is not None else None
        L = len(self.cfg.block_out)
        f32 = torch.float32
        for i, layers in enumerate(self.up):
            for li, (r, a, m) in enumerate(layers):
                s, sh, sw = skips.pop()
                au = add_up.pop(0) if add_up is not None else None
                last = "m" if m is not None else ("a" if a is not None else "r")
                # the output of a block's last layer feeds only MFMA A operands (upsampler conv, BrushNet zero conv):
                # store it as h16 (same operand values, half the bytes, LDS-DMA fast path downstream)
                od = self.ctx.h16 if (li == len(layers) - 1 and i < L - 1 and not (last == "a" and au is not None)) else f32
                x = r(x, F, H, W, x1=s, silu_temb=st, res1=au if last == "r" else None, out_dtype=od if last == "r" else f32, want_gn=a is not None)
                if a is not None:
                    x = a(x, F, H, W, out_dtype=od if last == "a" else f32)
                    if last == "a" and au is not None:
                        hip.add_inplace(self.ctx.dt, x, au)
                if m is not None:
                    x = m(x, F, H, W, res1=au, out_dtype=od)
                if collect:
                    outs.append((x, H, W))
            if i < L - 1:
                _, Hn, Wn = skips[-1]
                au = add_up.pop(0) if add_up is not None else None
                x, H, W = self.ups[i](x, F, H, W, Hv=Hn, Wv=Wn, res1=au)
                if collect:
                    outs.append((x, H, W))
        return x, outs, (H, W)


class BrushNet(_Backbone):
    def __init__(self, ctx, cfg, text_h16, precise_io=True):
        super().__init__(ctx, "brushnet", cfg, text_h16, motion=False, in_pad=16, precise_io=precise_io)
        zg = cfg.zero_conv_gain
        self.zd = [Conv(ctx, f"brushnet.brushnet_down_blocks.{i}", c, c, k=1, gain=zg) for i, c in enumerate(self.skip_ch)]
        self.zm = Conv(ctx, "brushnet.brushnet_mid_block", cfg.block_out[-1], cfg.block_out[-1], k=1, gain=zg)
        self.zu = [Conv(ctx, f"brushnet.brushnet_up_blocks.{i}", c, c, k=1, gain=zg) for i, c in enumerate(self.up_ch)]

    def backbone(self, x16, t, F, h, w, temb_key=None):
        """the part that does not depend on the UNet: down, mid and up path of the branch -> (down outputs, mid, up outputs, (H, W) at the bottom)."""
        st = self.temb(t, temb_key)
        x, skips, (H, W) = self.run_down(x16, F, h, w, st)
        down_raw = list(skips)
        mid = self.run_mid(x, F, H, W, st)
        _, ups, _ = self.run_up(mid, F, H, W, skips, st, collect=True)
        return down_raw, mid, ups, (H, W)

    def hidden_adders(self, pre, F, scale=1.0):
        """UNetConfig.brushnet_add == "hidden": the down residuals as callables x -> zero_conv(BrushNet skip) + x, one per skip position, for
        _Backbone.run_down(into_hidden=...) -- the same fused kernel as the "skip" site (residual in the GEMM epilogue), launched inside the UNet's
        down path instead of after it."""
        down_raw = pre[0]
        return [(lambda x, z=z, s=s, sh=sh, sw=sw: z(s, F, sh, sw, res0=x, scale=scale)[0]) for z, (s, sh, sw) in zip(self.zd, down_raw)]

    def mid_up(self, pre, F, unet_mid, scale=1.0):
        """mid + up residuals only (the down residuals already went in through hidden_adders)."""
        _, mid, ups, (H, W) = pre
        new_mid, _, _ = self.zm(mid, F, H, W, res0=unet_mid, scale=scale)
        return new_mid, [z(s, F, sh, sw, scale=scale)[0] for z, (s, sh, sw) in zip(self.zu, ups)]

    def __call__(self, x16, t, F, h, w, unet_skips, unet_mid, scale=1.0, pre=None):
        """x16: h16 [F*h*w,16] BrushNet input.  unet_skips / unet_mid: the UNet's own down skips and mid output;
        the zero-conv GEMMs add them in their epilogue, so the returned tensors are already (skip + residual).
        pre: the result of backbone() when it was computed ahead (on another stream)."""
        down_raw, mid, ups, (H, W) = pre if pre is not None else self.backbone(x16, t, F, h, w)
        new_skips = []
        for z, (s, sh, sw), (us, _, _) in zip(self.zd, down_raw, unet_skips):
            o, _, _ = z(s, F, sh, sw, res0=us, scale=scale)
            new_skips.append((o, sh, sw))
        new_mid, _, _ = self.zm(mid, F, H, W, res0=unet_mid, scale=scale)
        add_up = [z(s, F, sh, sw, scale=scale)[0] for z, (s, sh, sw) in zip(self.zu, ups)]
        return new_skips, new_mid, add_up


class UNetMotion(_Backbone):
    def __init__(self, ctx, cfg, text_h16, precise_io=True):
        super().__init__(ctx, "unet", cfg, text_h16, motion=True, in_pad=8, precise_io=precise_io)
        self.norm_out = GroupNorm(ctx, "unet.conv_norm_out", cfg.block_out[0], cfg.groups, 1e-5, precise=self.precise_io)      # emits the split operand itself
        self.conv_out = Conv(ctx, "unet.conv_out", cfg.block_out[0], cfg.out_ch, precise=self.precise_io)

    def down_mid(self, lat8, t, F, h, w, into_hidden=None):
        st = self.temb(t)
        x, skips, (H, W) = self.run_down(lat8, F, h, w, st, into_hidden=into_hidden)
        mid = self.run_mid(x, F, H, W, st)
        return st, skips, mid, (H, W)

    def up_out(self, st, skips, mid, F, H, W, add_up):
        x, _, (H, W) = self.run_up(mid, F, H, W, skips, st, add_up=add_up)
        h = self.norm_out(x, F, H * W, silu=True)
        eps, _, _ = self.conv_out(h, F, H, W)
        return eps            # fp32 [F*h*w, 4]


class Denoiser:
    """eps = UNet(latents, t | BrushNet(cat[latents, cond, mask], t)) for one clip."""

    def __init__(self, ctx, cfg, text_states, precise_io=True):
        self.ctx, self.cfg = ctx, cfg
        text_h16 = ctx.dev(text_states[0], ctx.h16)
        self.unet = UNetMotion(ctx, cfg, text_h16, precise_io=precise_io)
        self.brush = BrushNet(ctx, cfg, text_h16, precise_io=precise_io)

    # Two-stream schedule (default since round 4; measured in profiles/r3_two_stream_ab.txt: +2.4 %, bit-identical): the BrushNet backbone does
    # not depend on the UNet's down / mid path (only its zero convolutions add the UNet skips), so the two share the GPU.  Kernels of two streams
    # overlap, so per-kernel HIP-event durations are only meaningful on ONE stream: while hip.PROFILE is set (bench.py's pricing pass) the
    # schedule falls back to one stream.
    OVERLAP = True
    # set by the chunk lanes of pipeline.forward_device: with two chunks in flight the GPU is already shared by two streams and a third / fourth
    # stream per chunk measured slower (profiles/r4_schedule_ab_*.txt: 2 lanes x 1 stream 20.2 s per chunk, 2 lanes x 2 streams 20.4 s, 1 lane x 2
    # streams 20.7 s, 1 lane x 1 stream 21.2 s) -- the second stream pays only when a rank runs ONE chunk at a time
    lane = threading.local()

    def prepare(self, ts):
        """once per schedule, before the denoise loop (pipeline.denoise_chunk): see _Backbone.prepare_temb"""
        self.unet.prepare_temb(ts)
        self.brush.prepare_temb(ts)

    def _side_stream(self, main):
        """one side stream per launch stream (chunks of one rank may run on several streams at once: pipeline.forward_device)"""
        sides = self.__dict__.setdefault("_sides", {})
        key = main.cuda_stream
        if key not in sides:
            sides[key] = torch.cuda.Stream(device=self.ctx.device)
        return sides[key]

    def __call__(self, lat, cond, mask2d, t, F, h, w, H, W):
        """lat, cond: fp32 [F,h,w,4] device; mask2d: u8 [F,H,W]."""
        ctx = self.ctx
        # (precise_io: conv_in takes the fp32 latents and splits them itself -- no rounding of the network input to h16)
        lat8 = (hip.pad_channels_f32(lat, 8) if self.unet.precise_io else hip.pad_channels(ctx.dt, lat, 8)).view(F * h * w, 8)
        x16 = hip.brushnet_input(ctx.dt, lat, cond, mask2d, H, W).view(F * h * w, 16)
        if self.cfg.brushnet_add == "hidden":      # the UNet's down path consumes a BrushNet residual after every layer: the branch runs first
            pre = self.brush.backbone(x16, t, F, h, w)
            st, skips, mid, (Hm, Wm) = self.unet.down_mid(lat8, t, F, h, w, into_hidden=self.brush.hidden_adders(pre, F))
            new_mid, add_up = self.brush.mid_up(pre, F, mid)
            return self.unet.up_out(st, skips, new_mid, F, Hm, Wm, add_up).view(F, h, w, 4)
        assert self.cfg.brushnet_add == "skip", self.cfg.brushnet_add
        pre = None
        if Denoiser.OVERLAP and hip.PROFILE is None and not getattr(Denoiser.lane, "concurrent", False):
            main = torch.cuda.current_stream()
            side = self._side_stream(main)
            side.wait_stream(main)                             # x16 / t are ready
            with torch.cuda.stream(side):
                pre = self.brush.backbone(x16, t, F, h, w, temb_key=main.cuda_stream)
            x16.record_stream(side)
        st, skips, mid, (Hm, Wm) = self.unet.down_mid(lat8, t, F, h, w)
        if pre is not None:
            main.wait_stream(side)
            for grp in (pre[0], pre[2]):                        # produced on the side stream, consumed on the main one
                for ten, _, _ in grp:
                    ten.record_stream(main)
            pre[1].record_stream(main)
        new_skips, new_mid, add_up = self.brush(x16, t, F, h, w, skips, mid, pre=pre)
        eps = self.unet.up_out(st, new_skips, new_mid, F, Hm, Wm, add_up)
        return eps.view(F, h, w, 4)
