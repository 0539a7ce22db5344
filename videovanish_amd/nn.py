"""HIP-backed layers of the DiffuEraser stack.  Every arithmetic op is a launch of a libvvhip kernel through the
C ABI (videovanish_amd.hip); torch only owns the device buffers.

Activation convention: a 2-D tensor [M, C] = frames-major NHWC with M = F*H*W.  The residual trunk is fp32; tensors
that feed an MFMA contraction are h16 (bf16 / fp16, chosen by Ctx.dtype).
"""
import torch

from . import hip, packing
from .weights import SyntheticWeights


def normalize_device(device):
    """str ("cuda", "cuda:1"), torch.device or int -> torch.device("cuda", index).  A bare "cuda" means the CURRENT device
    (one process per GPU sets it with torch.cuda.set_device(LOCAL_RANK)); anything that is not a HIP device is refused."""
    if device is None:
        device = "cuda"
    if isinstance(device, int):
        device = torch.device("cuda", device)
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError(f"videovanish_amd: device {device} is not a HIP device (there is no CPU fallback)")
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    return device


class Ctx:
    """Per-process model context: device, MFMA operand dtype, weight source."""

    def __init__(self, device="cuda:0", dtype="bf16", weight_seed=0, weights=None):
        hip.lib()  # fail loudly before anything else if the extension is missing
        if not torch.cuda.is_available():
            raise RuntimeError("videovanish_amd: no HIP device visible (there is no CPU fallback)")
        self.device = normalize_device(device)
        if torch.cuda.current_device() != self.device.index:
            torch.cuda.set_device(self.device)      # one process per GPU: every launch goes to the current device's stream
        self.dt = hip.dtype_id(dtype)
        self.h16 = hip.h16(self.dt)
        self.src = weights if weights is not None else SyntheticWeights(weight_seed)

    def dev(self, t, dtype=None):
        return t.to(dtype if dtype is not None else t.dtype).contiguous().to(self.device)


# ---- split precision ("precise" layers: the VAE decoder, RunConfig.precise_decoder) --------------------------------------
# A 16-bit operand keeps 11 (fp16) / 8 (bf16) significant bits; x = hi + lo and w = wh + wl keep ~2x that, and
# x*w ~= hi*wh + lo*wh + hi*wl (the dropped lo*wl term is ~2^-22 relative).  Since round 3 the three products are ONE launch of the MFMA kernel
# over a K-concatenated operand pair (3x the input channels): activations [hi | lo * 2^4 | hi * 2^-10] (hip.split3) against weights
# [wh | wh * 2^-4 | wl * 2^10], all three accumulated in the same fp32 accumulator and finished by one epilogue -- round 2 ran three launches
# that read-modify-wrote the fp32 output twice.  The power-of-two scales keep the small parts out of the fp16 subnormal range.
# Cost: 3x the layer's FLOPs -- used only where the emulation (tools/parity_emulate.py) shows the error budget is spent: the VAE decoder
# (its rounding errors go straight to the pixels).
LO_UP, LO_DOWN, WL_UP = 16.0, 1.0 / 16.0, 1024.0


def split3_weight(w, h16, cdim=1):
    """fp32 weight -> the K-concatenated split-precision weight: [wh | wh * 2^-4 | (w - wh) * 2^10] along the input-channel dimension."""
    wh = w.to(h16).float()
    return torch.cat([wh, wh * LO_DOWN, (w.float() - wh) * WL_UP], cdim)


class Conv:
    """conv2d / linear as implicit GEMM.  `in_pad`: activations carry zero-padded channels (conv_in layers)."""

    def __init__(self, ctx, name, cin, cout, k=3, gain=1.0, cin_pad=None, bias=True, weight=None, bias_t=None, rows=None, geglu=False,
                 precise=False):
        self.ctx, self.k, self.cout = ctx, k, cout
        self.precise = precise
        if weight is None:
            weight, b = ctx.src.conv(name, cin, cout, k, gain)
            bias_t = b if bias else None
        if rows is not None:                      # keep only the first `rows` output channels (VAE quant_conv mean half)
            weight, bias_t, self.cout = weight[:rows], (bias_t[:rows] if bias_t is not None else None), rows
        if geglu:
            w2, bias_t = packing.geglu_interleave(weight.reshape(weight.shape[0], -1), bias_t)
            wp, self.K = packing.pack_matrix(w2, ctx.h16, geglu=True), w2.shape[1]
        elif precise and weight.device.type == "meta":                      # shape-only construction (modelhub.manifest)
            self.cp = weight.shape[1] if cin_pad is None else cin_pad
            wp, self.K = weight, 3 * self.cp * k * k
        elif precise:
            cp = weight.shape[1] if cin_pad is None else cin_pad            # pad the input channels first: the three groups are cp channels each
            wpad = torch.zeros((weight.shape[0], cp) + tuple(weight.shape[2:]), dtype=torch.float32)
            wpad[:, :weight.shape[1]] = weight.float()
            wp, self.K = packing.pack_conv(split3_weight(wpad, ctx.h16), ctx.h16)
            self.cp = cp
        else:
            wp, self.K = packing.pack_conv(weight, ctx.h16, cin_pad)
        if precise and geglu:
            raise RuntimeError("precise GEGLU layers are not supported")
        self.geglu = geglu
        self.w = ctx.dev(wp)
        self.b = ctx.dev(bias_t.float()) if bias_t is not None else None

    def __call__(self, x0, F, H, W, x1=None, stride=1, Hv=None, Wv=None, Hout=None, Wout=None, pad=None, bias=True, rowvec=None,
                 res0=None, res1=None, out_dtype=torch.float32, out=None, scale=1.0, bias_override=None, gn_partials=False):
        k = self.k
        pad = (k // 2) if pad is None else pad
        Hv = H if Hv is None else Hv
        Wv = W if Wv is None else Wv
        if Hout is None:
            Hout = (Hv + 2 * pad - k) // stride + 1
            Wout = (Wv + 2 * pad - k) // stride + 1
        b = bias_override if bias_override is not None else (self.b if bias else None)
        if self.precise:
            if rowvec is not None or out_dtype != torch.float32 or x1 is not None:
                raise RuntimeError("precise conv: one source, fp32 output, no rowvec")
            # [M, 3 C]: hi | lo | hi' -- the three products become ONE launch over 3 C channels (a precise GroupNorm emits that form itself)
            x3 = x0 if (x0.dtype == self.ctx.h16 and x0.shape[1] == 3 * self.cp) else hip.split3(self.ctx.dt, x0)
            return hip.conv_gemm(self.ctx.dt, x3, self.w, self.cout, self.K, F=F, Hin=H, Win=W, Hv=Hv, Wv=Wv, Hout=Hout, Wout=Wout, ksize=k,
                                 stride=stride, pad_t=pad, pad_l=pad, bias=b, res0=res0, res1=res1, out=out, out_dtype=torch.float32,
                                 out_scale=scale), Hout, Wout
        return hip.conv_gemm(self.ctx.dt, x0, self.w, self.cout, self.K, x1=x1, F=F, Hin=H, Win=W, Hv=Hv, Wv=Wv, Hout=Hout, Wout=Wout,
                             ksize=k, stride=stride, pad_t=pad, pad_l=pad, bias=b, rowvec=rowvec, res0=res0, res1=res1, out=out,
                             out_dtype=out_dtype, epilogue=hip.EPI_GEGLU if self.geglu else hip.EPI_NONE, out_scale=scale, gn_partials=gn_partials), Hout, Wout


class UpConv2x:
    """The 3x3 convolution (padding 1) of an Upsample2D layer, taken over the nearest-neighbour upsampled image WITHOUT building it and without the
    taps the upsampling duplicates: the 3 x 3 taps of output pixel (2y + py, 2x + px) fall on only 2 x 2 source pixels, so each of the four output
    parities is a 2x2 convolution over the SOURCE image whose weights are the 3x3 taps summed per source pixel (in fp32, before the one rounding to
    h16: packing.upconv2x_phase_weight) -- 4/9 of the multiply-adds of the fused-gather form Conv(..., Hv=, Wv=) runs.  Four launches of
    vv_conv_gemm (ksize 2, pad = 1 - parity) scatter into the [F, Hv, Wv] output (vv_conv_params.sc_*); bias / residual ride in every launch.
      rows:  py = 0: source rows (y - 1: w[0], y: w[1] + w[2]);   py = 1: (y: w[0] + w[1], y + 1: w[2]);   columns alike.
    Sizes: Hv in (2 H, 2 H - 1) and Wv in (2 W, 2 W - 1) -- every size a stride-2 downsampling followed by "resize to the skip's size" produces
    (torch's nearest map floor(r H / Hv) is floor(r / 2) for both; 45 rows from 23 at 720p).  In the odd case the last output row (column) sees zero
    padding where the even case sees the duplicated source row, so it is computed by small launches of its own over the last two source rows
    (columns) with the tap beyond the edge dropped ("last" weights); both odd: one more launch for the corner pixel.  Anything else falls back to
    the fused-gather 3x3 form."""

    def __init__(self, ctx, name, cin, cout, gain=1.0, precise=False):
        self.ctx, self.name, self.cin, self.cout, self.gain, self.precise = ctx, name, cin, cout, gain, precise
        weight, b = ctx.src.conv(name, cin, cout, 3, gain)
        self.b = ctx.dev(b.float()) if (b is not None and b.device.type != "meta") else None
        self.meta = weight.device.type == "meta"
        self.cp = cin
        self.w, self._fallback = {}, None
        if self.meta:                                    # shape-only construction (modelhub.manifest): the tensor was requested, nothing is packed
            self.K = 4 * cin * (3 if precise else 1)
            return
        w32 = weight.float()
        for vy in (0, 1, "last"):
            for vx in (0, 1, "last"):
                wp = packing.upconv2x_phase_weight(w32, vy, vx)                                                    # [cout, cin, 2, 2]
                packed, self.K = packing.pack_conv(split3_weight(wp, ctx.h16) if precise else wp, ctx.h16)
                self.w[(vy, vx)] = ctx.dev(packed)

    def fallback(self):
        if self._fallback is None:
            self._fallback = Conv(self.ctx, self.name, self.cin, self.cout, gain=self.gain, precise=self.precise)
        return self._fallback

    def __call__(self, x, F, H, W, Hv=None, Wv=None, res1=None):
        Hv = 2 * H if Hv is None else Hv
        Wv = 2 * W if Wv is None else Wv
        oh, ow = 2 * H - Hv, 2 * W - Wv                      # 1: the odd case of that dimension
        if oh not in (0, 1) or ow not in (0, 1) or (oh and H < 2) or (ow and W < 2) or self.cin % 8:
            return self.fallback()(x, F, H, W, Hv=Hv, Wv=Wv, res1=res1)
        ctx = self.ctx
        x = x.reshape(-1, x.shape[-1])
        if self.precise:      # [M, 3 C]: hi | lo | hi' (see Conv)
            x = x if (x.dtype == ctx.h16 and x.shape[1] == 3 * self.cp) else hip.split3(ctx.dt, x)
        Cx = x.shape[-1]
        x4 = x.view(F, H, W, Cx)
        out = torch.empty((F * Hv * Wv, self.cout), dtype=torch.float32, device=x.device)

        def launch(src, hin, win, hout, wout, wkey, pt, pl, oy, ox):
            hip.conv_gemm(ctx.dt, src, self.w[wkey], self.cout, self.K, F=F, Hin=hin, Win=win, Hout=hout, Wout=wout, ksize=2, pad_t=pt, pad_l=pl,
                          bias=self.b, res1=res1, out=out, scatter=(Hv, Wv, 2, 2, oy, ox))
        for py in (0, 1):
            for px in (0, 1):                                # the bulk: every row / column whose taps do not reach beyond an odd edge
                launch(x, H, W, H - oh, W - ow, (py, px), 1 - py, 1 - px, py, px)
        if oh:
            xr = x4[:, H - 2:].contiguous()                  # the last two source rows of every frame -> output row Hv - 1 (parity 0)
            for px in (0, 1):
                launch(xr, 2, W, 1, W - ow, ("last", px), 0, 1 - px, Hv - 1, px)
        if ow:
            xc = x4[:, :, W - 2:].contiguous()               # the last two source columns -> output column Wv - 1
            for py in (0, 1):
                launch(xc, H, 2, H - oh, 1, (py, "last"), 1 - py, 0, py, Wv - 1)
        if oh and ow:
            launch(x4[:, H - 2:, W - 2:].contiguous(), 2, 2, 1, 1, ("last", "last"), 0, 0, Hv - 1, Wv - 1)
        return out, Hv, Wv


class Linear:
    """y = x W^T + b on a [M, K] matrix (fp32 or h16 input)."""

    def __init__(self, ctx, name=None, cin=None, cout=None, bias=True, gain=1.0, weight=None, bias_t=None, geglu=False, precise=False):
        self.ctx = ctx
        if weight is None:
            weight, bias_t = ctx.src.linear(name, cin, cout, gain, bias)
        self.cout = weight.shape[0]
        self.geglu = geglu
        self.precise = precise
        if geglu:
            weight, bias_t = packing.geglu_interleave(weight, bias_t)
        if precise:
            if geglu:
                raise RuntimeError("precise GEGLU layers are not supported")
            self.kin = weight.shape[1]
            if weight.device.type != "meta":
                weight = split3_weight(weight, ctx.h16)
        self.K = weight.shape[1]
        self.w = ctx.dev(packing.pack_matrix(weight, ctx.h16, geglu=geglu))
        self.b = ctx.dev(bias_t.float()) if bias_t is not None else None

    def __call__(self, x, res0=None, res1=None, out_dtype=torch.float32, rows_per_frame=None, out=None, split=None):
        """split = (heads, head_dim, tokens_per_batch): head-major store of a fused QKV projection (hip.conv_gemm split_heads)."""
        if x.device.type == "meta":          # shape-only construction (modelhub.manifest): CrossAttention projects the text K/V at build time
            return None
        M = x.shape[0]
        kw = dict(split_heads=split[0], split_dim=split[1], split_tokens=split[2]) if split else {}
        if self.precise:
            if split or out_dtype != torch.float32:
                raise RuntimeError("precise linear: fp32 row-major output only")
            if not (x.dtype == self.ctx.h16 and x.shape[1] == 3 * self.kin):
                x = hip.split3(self.ctx.dt, x)
        return hip.conv_gemm(self.ctx.dt, x, self.w, self.cout, self.K, F=1, Hin=M, Win=1, bias=self.b, res0=res0, res1=res1,
                             out_dtype=out_dtype, out=out, epilogue=hip.EPI_GEGLU if self.geglu else hip.EPI_NONE, **kw)


class GroupNorm:
    def __init__(self, ctx, name, C, groups, eps, precise=False):
        self.ctx, self.groups, self.eps = ctx, groups, eps
        self.out_dtype = "split3" if precise else None           # precise consumers take the K-concatenated split operand [M, 3C] (hip.split3)
        g, b = ctx.src.norm(name, C)
        self.g, self.b = ctx.dev(g), ctx.dev(b)

    def __call__(self, x0, F, HW, x1=None, silu=False, pool_frames=False, partials=None):
        return hip.groupnorm(self.ctx.dt, x0, self.g, self.b, self.groups, self.eps, x1=x1, F=F, HW=HW, silu=silu, pool_frames=pool_frames,
                             out_dtype=self.out_dtype, partials=partials)


class LayerNorm:
    def __init__(self, ctx, name, C):
        self.ctx = ctx
        g, b = ctx.src.norm(name, C)
        self.g, self.b = ctx.dev(g), ctx.dev(b)

    def __call__(self, x, pe=None, rows_per_frame=1):
        return hip.layernorm(self.ctx.dt, x, self.g, self.b, pe=pe, rows_per_frame=rows_per_frame)


class ResBlock:
    """ResnetBlock2D: GN+SiLU -> conv3 (+temb) -> GN+SiLU -> conv3 -> + shortcut(x)."""

    # conv1's output feeds norm2 and nothing else, so it could be stored in h16 (conv1 writes 2 bytes per element, norm2's two passes read 2 instead of 4).
    # Built and measured in round 5, NOT adopted (H16_MID = False): +0.47 % on the bench line (GroupNorm 1.22 -> 1.14 s per chunk), but the extra rounding costs
    # parity -- predicted on the CPU (tools/parity_h16_conv1.py: rms +1.5 % at full width), measured on the GPU: rms +1.8 % at c1, +6.7 % at 50 steps full width, and the smoke clip's
    # per-pixel maximum 8.3e-4 -> 1.015e-3, over the 1e-3 bound (profiles/r5_h16_mid_ab.txt).  The switch stays for the A/B (tools/bench_with.py ResBlock.H16_MID=1).
    H16_MID = False
    # conv2 leaves per-channel partial sums of the block's output for the GroupNorm of the spatial transformer that follows (vv_conv_params.gn_partials, round 6):
    # that GroupNorm then needs no statistics pass over HBM.  Only where conv2 runs on the 128 x 160 halo-tile kernel anyway (C = 320 / 640 at 720p-class sizes).
    GN_FROM_EPILOGUE = True

    def __init__(self, ctx, name, cin, cout, groups, eps, temb_dim=None, precise=False, precise_temb=False, h16_mid=False):
        self.ctx, self.cin, self.cout = ctx, cin, cout
        self.h16_mid = bool(h16_mid) and not precise
        self.norm1 = GroupNorm(ctx, name + ".norm1", cin, groups, eps, precise=precise)
        self.conv1 = Conv(ctx, name + ".conv1", cin, cout, precise=precise)
        # precise_temb: the time-embedding projection in split precision -- a one-row GEMM per ResBlock (hoisted out of the denoise loop: free)
        self.temb = Linear(ctx, name + ".time_emb_proj", temb_dim, cout, precise=precise_temb) if temb_dim else None
        self.norm2 = GroupNorm(ctx, name + ".norm2", cout, groups, eps, precise=precise)
        self.conv2 = Conv(ctx, name + ".conv2", cout, cout, precise=precise)
        self.short = Conv(ctx, name + ".conv_shortcut", cin, cout, k=1, precise=precise) if cin != cout else None

    def temb_bias(self, silu_temb):
        """[S, temb_dim] -> [S, cout]: conv1 bias + time_emb_proj(silu(temb)), one row per timestep (rows are independent GEMM rows: the
        row of a batch of timesteps equals the single-timestep launch bit for bit)."""
        S = silu_temb.shape[0]
        res = self.conv1.b.view(1, -1) if S == 1 else self.conv1.b.view(1, -1).repeat(S, 1)
        return self.temb(silu_temb, res0=res)

    def wants_gn_partials(self, H, W, res1, out_dtype):
        """conv2 can emit the GroupNorm partials of the block's output: the shapes the dispatcher sends to the 128 x 160 halo-tile kernel"""
        cover = ((H + 7) // 8) * 8 * ((W + 15) // 16) * 16
        return (ResBlock.GN_FROM_EPILOGUE and self.cout in (320, 640) and not self.conv2.precise and res1 is None and out_dtype == torch.float32
                and cover * 100 <= H * W * 115 and H * W >= 2048)

    def __call__(self, x0, F, H, W, x1=None, silu_temb=None, res1=None, out_dtype=torch.float32, want_gn=False):
        HW = H * W
        h = self.norm1(x0, F, HW, x1=x1, silu=True)
        b1 = None
        if self.temb is not None:
            # conv1 bias + time_emb_proj(silu(temb)) is the same vector for every frame -> fold into the bias
            if isinstance(silu_temb, dict):      # hoisted out of the denoise loop (unet._Backbone.prepare_temb): this block's row for this timestep
                b1 = silu_temb[id(self)]
            else:
                b1 = self.temb_bias(silu_temb).view(-1)
        mid16 = self.h16_mid and ResBlock.H16_MID
        h, _, _ = self.conv1(h, F, H, W, bias_override=b1, out_dtype=self.ctx.h16 if mid16 else torch.float32,
                             gn_partials=not mid16 and not self.conv1.precise and self.wants_gn_partials(H, W, None, torch.float32))      # conv1's output feeds norm2 only
        h = self.norm2(h, F, HW, silu=True, partials=getattr(h, "vv_gn", None))
        if self.short is not None:
            xs, _, _ = self.short(x0, F, H, W, x1=x1)
        else:
            xs = x0
        out, _, _ = self.conv2(h, F, H, W, res0=xs, res1=res1, out_dtype=out_dtype,
                               gn_partials=bool(want_gn) and self.wants_gn_partials(H, W, res1, out_dtype))
        return out


class SelfAttention:
    """attn with fused QKV projection (no bias) + output projection (bias) + residual."""

    def __init__(self, ctx, name, C, heads, qkv_bias=False, precise=False, prescale_q=False):
        self.ctx, self.C, self.heads, self.precise = ctx, C, heads, precise
        # prescale_q (spatial use only): softmax scale * log2(e) folded into the fp32 query weights BEFORE their one rounding to h16,
        # so the attention kernel gets c*q at the same precision as q (vv_attention, q_prescaled; the d = 40 kernel wants it)
        self.prescale_q = bool(prescale_q) and not precise
        ws, bs = [], []
        for n in ("to_q", "to_k", "to_v"):
            w, b = ctx.src.linear(f"{name}.{n}", C, C, 1.0, qkv_bias)
            if n == "to_q" and self.prescale_q:
                w = w * hip.attention_q_scale(C // heads)
                b = b * hip.attention_q_scale(C // heads) if b is not None else None
            ws.append(w); bs.append(b)
        self.qkv = Linear(ctx, weight=torch.cat(ws, 0), bias_t=torch.cat(bs, 0) if qkv_bias else None, precise=precise)
        self.out = Linear(ctx, name + ".to_out.0", C, C, precise=precise)

    def spatial(self, n, res, B, N):
        C, dt, D = self.C, self.ctx.dt, self.C // self.heads
        if self.precise:
            # split-precision projections around the (h16-operand) attention core: fp32 [M, 3C] QKV, rounded once to h16
            qkv32 = self.qkv(n)
            qkv, _ = hip.split_f32(dt, qkv32, 1.0)
            del qkv32
            o = torch.empty((B * N, C), dtype=self.ctx.h16, device=n.device)
            hip.attention(dt, qkv, qkv, qkv, o, B=B, heads=self.heads, Nq=N, Nkv=N, D=D, q_bs=N * 3 * C, k_bs=N * 3 * C, v_bs=N * 3 * C,
                          o_bs=N * C, q_rs=3 * C, k_rs=3 * C, v_rs=3 * C, o_rs=C, k_off=C, v_off=2 * C)
            return self.out(o, res0=res)
        return self.out(self.core(n, B, N), res0=res)

    def core(self, n, B, N):
        """the attention core alone: fused QKV projection + softmax(QK^T)V -> h16 [B*N, C] (the output projection is the caller's)."""
        C, dt, D = self.C, self.ctx.dt, self.C // self.heads
        # head-major QKV ([frame][q|k|v][head][token][D]): a head's K/V rows are contiguous 2*D-byte records, so the K/V tile
        # DMA of the attention kernel reads whole cache lines (the [token][3C] layout over-fetched 2.6x at D = 40)
        return self.core_qkv(self.qkv(n, out_dtype=self.ctx.h16, split=(self.heads, D, N)), B, N)

    def core_qkv(self, qkv, B, N, head_major_out=False):
        """softmax(QK^T)V on a head-major QKV buffer ([frame][q|k|v][head][token][D]) -> h16 [B*N, C], or (head_major_out) [B, heads, N, D]: every wave
        then stores whole contiguous 2 D-byte records instead of an 80-byte slice of a 640-byte row (vv_attn_params.o_hs; the fused chain tail reads it)."""
        C, dt, D = self.C, self.ctx.dt, self.C // self.heads
        o = torch.empty((B, self.heads, N, D) if head_major_out else (B * N, C), dtype=self.ctx.h16, device=qkv.device)
        hip.attention(dt, qkv, qkv, qkv, o, B=B, heads=self.heads, Nq=N, Nkv=N, D=D, q_bs=N * 3 * C, k_bs=N * 3 * C,
                      v_bs=N * 3 * C, o_bs=N * C, q_rs=D, k_rs=D, v_rs=D, o_rs=D if head_major_out else C, k_off=N * C, v_off=2 * N * C,
                      q_hs=N * D, k_hs=N * D, v_hs=N * D, q_prescaled=self.prescale_q, o_hs=N * D if head_major_out else 0)
        return o

    def temporal(self, n, res, Fr, HW, res1=None):
        """sequence = frames; rows of the [F*HW, 3C] QKV matrix gathered with stride HW*3C inside the kernel."""
        assert not self.prescale_q
        C, dt = self.C, self.ctx.dt
        # (a per-pixel head-major QKV -- split_tokens < 0 -- makes this core 15 % faster but scatters the QKV GEMM's stores over
        # records 3*C*F elements apart: +9 % on that GEMM, a net loss; measured in profiles/r1_gemm_ab.txt)
        qkv = self.qkv(n, out_dtype=self.ctx.h16)
        o = torch.empty((Fr * HW, C), dtype=self.ctx.h16, device=n.device)
        hip.attention(dt, qkv, qkv, qkv, o, B=HW, heads=self.heads, Nq=Fr, Nkv=Fr, D=C // self.heads, q_bs=3 * C, k_bs=3 * C, v_bs=3 * C,
                      o_bs=C, q_rs=HW * 3 * C, k_rs=HW * 3 * C, v_rs=HW * 3 * C, o_rs=HW * C, k_off=C, v_off=2 * C)
        return self.out(o, res0=res, res1=res1)


class CrossAttention:
    """attn2: queries from the image tokens, K/V from the constant text states (projected ONCE at build time)."""

    def __init__(self, ctx, name, C, heads, text_h16):
        self.ctx, self.C, self.heads = ctx, C, heads
        cross = text_h16.shape[-1]
        self.q = Linear(ctx, name + ".to_q", C, C, bias=False)
        wk, _ = ctx.src.linear(name + ".to_k", cross, C, 1.0, False)
        wv, _ = ctx.src.linear(name + ".to_v", cross, C, 1.0, False)
        kvp = Linear(ctx, weight=torch.cat([wk, wv], 0), bias_t=None)
        self.kv = kvp(text_h16, out_dtype=ctx.h16)            # [77, 2C], hoisted out of the denoise loop
        self.nkv = text_h16.shape[0]
        self.out = Linear(ctx, name + ".to_out.0", C, C)

    def __call__(self, n, res, B, N):
        C = self.C
        q = self.q(n, out_dtype=self.ctx.h16)
        o = torch.empty((B * N, C), dtype=self.ctx.h16, device=n.device)
        hip.attention(self.ctx.dt, q, self.kv, self.kv, o, B=B, heads=self.heads, Nq=N, Nkv=self.nkv, D=C // self.heads, q_bs=N * C, k_bs=0,
                      v_bs=0, o_bs=N * C, q_rs=C, k_rs=2 * C, v_rs=2 * C, o_rs=C, v_off=C)
        return self.out(o, res0=res)


class FeedForward:
    def __init__(self, ctx, name, C):
        self.ctx = ctx
        self.proj = Linear(ctx, name + ".net.0.proj", C, 8 * C, geglu=True)
        self.out = Linear(ctx, name + ".net.2", 4 * C, C)

    def __call__(self, n, res):
        # the FF result is consumed only as the A operand of proj_out (which rounds it to h16 anyway): store it as h16
        return self.out(self.proj(n, out_dtype=self.ctx.h16), res0=res, out_dtype=self.ctx.h16)


class SpatialTransformer:
    """Transformer2D block.  At C = 320, 8 heads and 77 text tokens (level 0: 14 blocks per denoise step) everything after the self-attention
    core runs as ONE kernel (csrc/vv_chain.hip: output projection, cross-attention, GEGLU feed-forward, proj_out, all residuals; the token's row
    stays in registers, weights and the per-head text K / V stream through an LDS ring); other widths run layer by layer."""
    FUSED = True          # class-level switch (tests / A-B runs compare both paths on the same weights)
    HEAD_MAJOR_O = True   # the self-attention core's output in head-major layout between the two fused kernels (round 6: whole 80-byte records per store)

    def __init__(self, ctx, name, C, cfg, text_h16):
        self.ctx = ctx
        self.norm = GroupNorm(ctx, name + ".norm", C, cfg.groups, 1e-6)
        self.proj_in = Conv(ctx, name + ".proj_in", C, C, k=1)
        b = name + ".transformer_blocks.0"
        self.n1, self.n2, self.n3 = (LayerNorm(ctx, f"{b}.norm{i}", C) for i in (1, 2, 3))
        self.attn1 = SelfAttention(ctx, b + ".attn1", C, cfg.heads, prescale_q=(C // cfg.heads == 40))
        self.attn2 = CrossAttention(ctx, b + ".attn2", C, cfg.heads, text_h16)
        self.ff = FeedForward(ctx, b + ".ff", C)
        self.proj_out = Conv(ctx, name + ".proj_out", C, C, k=1)
        self.fused = None
        if C == 320 and cfg.heads == 8 and text_h16.shape[0] == 77 and torch.device(ctx.device).type != "meta":
            src, w = ctx.src, {}
            w["o1.w"], w["o1.b"] = src.linear(b + ".attn1.to_out.0", C, C)
            w["ln2.g"], w["ln2.b"] = src.norm(b + ".norm2", C)
            w["q2.w"], _ = src.linear(b + ".attn2.to_q", C, C, 1.0, False)
            kv = self.attn2.kv.float().cpu()                                  # [77, 2C]: the text tokens projected once at build time (h16 values)
            w["k2"], w["v2"] = kv[:, :C].contiguous(), kv[:, C:].contiguous()
            w["o2.w"], w["o2.b"] = src.linear(b + ".attn2.to_out.0", C, C)
            w["ln3.g"], w["ln3.b"] = src.norm(b + ".norm3", C)
            w["ff1.w"], w["ff1.b"] = src.linear(b + ".ff.net.0.proj", C, 8 * C)
            w["ff2.w"], w["ff2.b"] = src.linear(b + ".ff.net.2", 4 * C, C)
            wo, w["out.b"] = src.conv(name + ".proj_out", C, C, 1)
            w["out.w"] = wo.reshape(C, C)
            stream, params = packing.pack_chain_stream(w, ctx.h16, cfg.heads)
            self.fused = (ctx.dev(stream), ctx.dev(params))
            # the front of the block (GroupNorm apply, proj_in, LN1, fused q|k|v projection) as one kernel too
            f = {}
            wi, f["in.b"] = src.conv(name + ".proj_in", C, C, 1)
            f["in.w"] = wi.reshape(C, C)
            f["ln1.g"], f["ln1.b"] = src.norm(b + ".norm1", C)
            wq, _ = src.linear(b + ".attn1.to_q", C, C, 1.0, False)
            wk, _ = src.linear(b + ".attn1.to_k", C, C, 1.0, False)
            wv, _ = src.linear(b + ".attn1.to_v", C, C, 1.0, False)
            f["qkv.w"] = torch.cat([wq * hip.attention_q_scale(C // cfg.heads), wk, wv], 0)      # as SelfAttention(prescale_q=True)
            fs, fp = packing.pack_chain_front_stream(f, ctx.h16)
            self.front = (ctx.dev(fs), ctx.dev(fp))

    def __call__(self, x, F, H, W, out_dtype=torch.float32):
        HW = H * W
        gn = getattr(x, "vv_gn", None)      # partial sums of x left by the convolution that produced it (ResBlock.GN_FROM_EPILOGUE)
        if self.fused is not None and SpatialTransformer.FUSED and x.dtype == torch.float32:
            t, qkv = hip.spatial_chain_front_c320(self.ctx.dt, x, self.norm.g, self.norm.b, self.norm.groups, self.norm.eps, self.front[0], self.front[1],
                                                  F=F, HW=HW, partials=gn)
            hm = SpatialTransformer.HEAD_MAJOR_O
            o = self.attn1.core_qkv(qkv, F, HW, head_major_out=hm)
            return hip.spatial_chain_c320(self.ctx.dt, o, t, x, self.fused[0], self.fused[1], out_dtype=out_dtype, o_hw=HW if hm else 0)
        h = self.norm(x, F, HW, partials=gn)
        t, _, _ = self.proj_in(h, F, H, W)
        t = self.attn1.spatial(self.n1(t), t, F, HW)
        t = self.attn2(self.n2(t), t, F, HW)
        t = self.ff(self.n3(t), t)
        out, _, _ = self.proj_out(t, F, H, W, res0=x, out_dtype=out_dtype)
        return out


class MotionModule:
    """AnimateDiff temporal transformer.  At C = 320, 8 heads and 32-frame clips the whole module runs as ONE kernel
    (csrc/vv_motion.hip: one wave per pixel, trunk and activations in registers, weights streamed through an LDS ring); every other
    shape runs the layer-by-layer path below."""
    FUSED = True          # class-level switch (tests / A-B runs compare both paths on the same weights)
    HEAD_MAJOR_O = True   # the self-attention core's output in head-major layout between the two fused kernels (round 6: whole 80-byte records per store)

    def __init__(self, ctx, name, C, cfg, pe_table):
        self.ctx = ctx
        self.C, self.groups = C, cfg.groups
        self.fused = None
        if C == 320 and cfg.heads == 8:
            src = ctx.src
            b = name + ".transformer_blocks.0"
            w = {}
            w["proj_in.w"], w["proj_in.b"] = src.linear(name + ".proj_in", C, C)
            w["proj_out.w"], w["proj_out.b"] = src.linear(name + ".proj_out", C, C)
            for a in ("attn1", "attn2"):
                for nm, key in (("q", "to_q"), ("k", "to_k"), ("v", "to_v")):
                    w[f"{a}.{nm}"], _ = src.linear(f"{b}.{a}.{key}", C, C, 1.0, False)
                w[f"{a}.o"], w[f"{a}.ob"] = src.linear(f"{b}.{a}.to_out.0", C, C)
            for i in (1, 2, 3):
                w[f"ln{i}.g"], w[f"ln{i}.b"] = src.norm(f"{b}.norm{i}", C)
            w["ff1.w"], w["ff1.b"] = src.linear(b + ".ff.net.0.proj", C, 8 * C)
            w["ff2.w"], w["ff2.b"] = src.linear(b + ".ff.net.2", 4 * C, C)
            w["pe"] = pe_table.detach().float().cpu()
            stream, params = packing.pack_motion_stream(w, ctx.h16, cfg.heads)
            self.fused = (ctx.dev(stream), ctx.dev(params))
        self.norm = GroupNorm(ctx, name + ".norm", C, cfg.groups, 1e-6)
        self.proj_in = Linear(ctx, name + ".proj_in", C, C)
        b = name + ".transformer_blocks.0"
        self.n1, self.n2, self.n3 = (LayerNorm(ctx, f"{b}.norm{i}", C) for i in (1, 2, 3))
        self.attn1 = SelfAttention(ctx, b + ".attn1", C, cfg.heads)
        self.attn2 = SelfAttention(ctx, b + ".attn2", C, cfg.heads)
        self.ff = FeedForward(ctx, b + ".ff", C)
        self.proj_out = Linear(ctx, name + ".proj_out", C, C)
        self.pe = pe_table          # [max_seq, C] fp32 on device

    def __call__(self, x, F, H, W, res1=None, out_dtype=torch.float32):
        if hip.PROFILE is None:          # (the tag is process-global state: only touched in bench.py's one-thread pricing pass)
            return self._run(x, F, H, W, res1, out_dtype)
        prev, hip.PROFILE_TAG = hip.PROFILE_TAG, "motion:"
        try:
            return self._run(x, F, H, W, res1, out_dtype)
        finally:
            hip.PROFILE_TAG = prev

    def _run(self, x, F, H, W, res1, out_dtype):
        HW = H * W
        # (clips of 16..32 frames: a wave always carries 32 token rows, the rows past F are masked -- e.g. the reference's own 22-frame windows)
        if (self.fused is not None and MotionModule.FUSED and 16 <= F <= 32 and HW % 4 == 0 and x.dtype == torch.float32
                and (res1 is None or res1.dtype == torch.float32)):
            return hip.motion_module_c320(self.ctx.dt, x, self.fused[0], self.fused[1], self.norm.g, self.norm.b, self.groups, self.norm.eps,
                                          F=F, HW=HW, res1=res1, out_dtype=out_dtype)
        if F > self.pe.shape[0]:
            raise RuntimeError(f"motion module: clip of {F} frames exceeds the positional table ({self.pe.shape[0]})")
        h = self.norm(x, F, HW, pool_frames=True)
        t = self.proj_in(h)
        t = self.attn1.temporal(self.n1(t, pe=self.pe, rows_per_frame=HW), t, F, HW)
        t = self.attn2.temporal(self.n2(t, pe=self.pe, rows_per_frame=HW), t, F, HW)
        t = self.ff(self.n3(t), t)
        return self.proj_out(t, res0=x, res1=res1, out_dtype=out_dtype)
