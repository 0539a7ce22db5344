"""SAM 2.1 video-predictor arithmetic on the gfx950 HIP kernels (reference sam2_masker.py:88-150 drives the third-party
`sam2` package this replaces).  Implements the five-method model interface of videovanish_amd/sam2_predictor.py:
encode_image, track_step, encode_memory_from_low_res, fill_holes, masks_to_video_res.

Every arithmetic op is a launch through the C ABI (videovanish_amd.hip); torch owns device buffers and does data movement only (cat /
view).  No CPU fallback.  Layout: activations are row-major [tokens, C] (= NHWC); trunks fp32, MFMA operands h16.
  * Hiera trunk: windowed blocks gather their windows with vv_gather_rows (index tensors built once per geometry), fused QKV projection stored
    head-major, head dim 72 zero-padded to 80 (the attention kernel's softmax scale stays 72^-1/2), q pooling = vv_maxpool2x2 on the q block.
  * memory attention: one head of d_model (256): rotary encoding applied in place on the h16 q / k blocks (vv_rope_apply), the memory bank
    (<= 7 x 4096 + 64 tokens of 64 channels) projected per layer.
  * SAM decoder: two-way transformer on ~10 tokens x 4096 image tokens, transposed convolutions as GEMM + vv_pixel_shuffle2, hypernetwork
    product / mask selection / object-pointer mixing on the device (no host round trip inside a frame).
Restates the published modules (see oracle/sam2_ref.py for the citations) [UNVERIFIED-3P]; parity is tested against that oracle.
"""
import math

import numpy as np
import torch

from . import hip, packing
from .nn import Ctx
from .sam2_config import Sam2Config, hiera_blocks, select_memories
from .sam2_weights import Sam2Weights

NO_OBJ_SCORE = -1024.0
IMG_MEAN = (0.485, 0.456, 0.406)
IMG_STD = (0.229, 0.224, 0.225)
_HEAD_DIMS = (32, 40, 64, 80, 128, 160, 256, 512)          # head dims vv_attention is built for


def _up64(n):
    return (n + 63) // 64 * 64


def _pad_dim(d):
    for k in _HEAD_DIMS:
        if k >= d:
            return k
    raise RuntimeError(f"head dim {d} > 512")


class _Lin:
    """y = act(x W^T + b) (+ res0); W fp32 [N, K] packed once to the MFMA layout."""

    def __init__(self, ctx, w, b=None):
        self.ctx, self.N, self.K = ctx, w.shape[0], w.shape[1]
        self.w = ctx.dev(packing.pack_matrix(w.float(), ctx.h16))
        self.b = ctx.dev(b.float()) if b is not None else None

    def __call__(self, x, res0=None, out_dtype=torch.float32, act=hip.ACT_NONE, split=None, out=None):
        kw = dict(split_heads=split[0], split_dim=split[1], split_tokens=split[2]) if split else {}
        return hip.conv_gemm(self.ctx.dt, x, self.w, self.N, self.K, F=1, Hin=x.shape[0], Win=1, bias=self.b, res0=res0, out_dtype=out_dtype,
                             act=act, out=out, **kw)


class _Conv:
    """conv2d (any k / stride / pad) on NHWC h16 input with zero-padded input channels."""

    def __init__(self, ctx, w, b, cin_pad=None):
        self.ctx, self.k, self.N = ctx, w.shape[-1], w.shape[0]
        wp, self.K = packing.pack_conv(w.float(), ctx.h16, cin_pad)
        self.w, self.b = ctx.dev(wp), ctx.dev(b.float())

    def __call__(self, x, H, W, stride, pad, res0=None, out_dtype=torch.float32):
        Ho, Wo = (H + 2 * pad - self.k) // stride + 1, (W + 2 * pad - self.k) // stride + 1
        return hip.conv_gemm(self.ctx.dt, x, self.w, self.N, self.K, F=1, Hin=H, Win=W, Hout=Ho, Wout=Wo, ksize=self.k, stride=stride, pad_t=pad,
                             pad_l=pad, bias=self.b, res0=res0, out_dtype=out_dtype), Ho, Wo


def _pad_heads_rows(w, b, heads, d, dp):
    """[heads * d, K] -> [heads * dp, K] (zero rows): projection whose OUTPUT feeds the attention kernel at head dim dp."""
    K = w.shape[1]
    wp = torch.zeros(heads, dp, K)
    wp[:, :d] = w.reshape(heads, d, K)
    bp = None
    if b is not None:
        bp = torch.zeros(heads, dp)
        bp[:, :d] = b.reshape(heads, d)
        bp = bp.reshape(-1)
    return wp.reshape(heads * dp, K), bp


def _pad_heads_cols(w, heads, d, dp):
    """[N, heads * d] -> [N, heads * dp] (zero columns): projection whose INPUT is the attention output at head dim dp."""
    N = w.shape[0]
    wp = torch.zeros(N, heads, dp)
    wp[:, :, :d] = w.reshape(N, heads, d)
    return wp.reshape(N, heads * dp)


class _Attention:
    """sam/transformer.py::Attention: q / k / v projections, multi-head softmax attention, output projection.  Inputs fp32 [N, C]."""

    def __init__(self, m, name, heads, rope=False):
        W = m.W.get
        self.m, self.heads = m, heads
        wq, bq = W(name + ".q_proj.weight"), W(name + ".q_proj.bias")
        wk, bk = W(name + ".k_proj.weight"), W(name + ".k_proj.bias")
        wv, bv = W(name + ".v_proj.weight"), W(name + ".v_proj.bias")
        wo, bo = W(name + ".out_proj.weight"), W(name + ".out_proj.bias")
        internal = wq.shape[0]
        self.d = internal // heads
        self.dp = _pad_dim(self.d)
        self.q = _Lin(m.ctx, *_pad_heads_rows(wq, bq, heads, self.d, self.dp))
        self.k = _Lin(m.ctx, *_pad_heads_rows(wk, bk, heads, self.d, self.dp))
        self.v = _Lin(m.ctx, *_pad_heads_rows(wv, bv, heads, self.d, self.dp))
        self.o = _Lin(m.ctx, _pad_heads_cols(wo, heads, self.d, self.dp), bo)
        self.rope = rope

    def __call__(self, q_in, k_in, v_in, res0=None, rows_rope_k=None):
        m, h16 = self.m, self.m.ctx.h16
        q = self.q(q_in, out_dtype=h16)
        k = self.k(k_in, out_dtype=h16)
        v = self.v(v_in, out_dtype=h16)
        if self.rope:
            hip.rope_apply(m.ctx.dt, q, q.shape[0], m.rope_cs, self.dp)
            hip.rope_apply(m.ctx.dt, k, k.shape[0] if rows_rope_k is None else rows_rope_k, m.rope_cs, self.dp)
        Nq, Nk, I = q.shape[0], k.shape[0], self.heads * self.dp
        o = torch.empty((Nq, I), dtype=h16, device=q.device)
        if self.heads == 1 and Nq >= 1024 and Nk >= 2048 and Nk % 4 == 0:
            # one long head is only Nq / 128 blocks: split the keys 4 ways over the batch index and merge by log-sum-exp (4 x the blocks)
            hip.attention_split_kv(m.ctx.dt, q, k, v, o, heads=1, Nq=Nq, Nkv=Nk, D=self.dp, S=4, q_rs=I, k_rs=I, v_rs=I, o_rs=I, q_hs=self.dp,
                                   k_hs=self.dp, v_hs=self.dp, scale=float(self.d) ** -0.5)
        else:
            hip.attention(m.ctx.dt, q, k, v, o, B=1, heads=self.heads, Nq=Nq, Nkv=Nk, D=self.dp, q_bs=0, k_bs=0, v_bs=0, o_bs=0, q_rs=I, k_rs=I,
                          v_rs=I, o_rs=I, q_hs=self.dp, k_hs=self.dp, v_hs=self.dp, scale=float(self.d) ** -0.5)
        return self.o(o, res0=res0)


class HipSam2:
    def __init__(self, cfg: Sam2Config = Sam2Config(), weights=None, seed=0, device=None, dtype="fp16"):
        self.cfg = cfg
        self.W = weights if weights is not None else Sam2Weights(cfg, seed)
        self.ctx = Ctx(device if device is not None else "cuda", dtype)
        self.device = self.ctx.device
        ctx, W, dev = self.ctx, self.W.get, self.ctx.dev
        self.blocks, self.stage_ends = hiera_blocks(cfg)
        S, fs, D, Mm = cfg.image_size, cfg.feat_size, cfg.d_model, cfg.mem_dim
        self._idx = {}
        # ---- Hiera trunk
        T = "image_encoder.trunk."
        # patch embedding: Conv2d(3, E, 7, stride 4, padding 3).  vv_conv_gemm has strides 1 / 2: the image is stored space-to-depth by 4
        # (48 channels per 4x4 block, vv_u8_normalize) and the 7x7 window -- it starts 3 pixels before block i, i.e. covers blocks i-1 and i --
        # becomes a 2x2 stride-1 convolution over blocks with one block of zero padding on the top / left
        w7 = W(T + "patch_embed.proj.weight")
        w2 = torch.zeros(w7.shape[0], 4, 4, 3, 2, 2)                                    # [co][dy][dx][c][by][bx]
        for ky in range(7):
            by, dy = (0, ky + 1) if ky < 3 else (1, ky - 3)
            for kx in range(7):
                bx, dx = (0, kx + 1) if kx < 3 else (1, kx - 3)
                w2[:, dy, dx, :, by, bx] = w7[:, :, ky, kx]
        self.patch = _Conv(ctx, w2.reshape(w7.shape[0], 48, 2, 2), W(T + "patch_embed.proj.bias"))
        h0 = S // 4
        pe = torch.nn.functional.interpolate(W(T + "pos_embed"), size=(h0, h0), mode="bicubic")       # parameter folding at build time
        win = W(T + "pos_embed_window")
        pe = pe + win.tile([x // y for x, y in zip(pe.shape, win.shape)])
        self.pos_embed = dev(pe.permute(0, 2, 3, 1).reshape(h0 * h0, -1))
        self.hb = []
        for i, b in enumerate(self.blocks):
            n = f"{T}blocks.{i}"
            heads, d = b["heads"], b["dim_out"] // b["heads"]
            dp = _pad_dim(d)
            wq, bq = W(n + ".attn.qkv.weight"), W(n + ".attn.qkv.bias")
            wq, bq = _pad_heads_rows(wq, bq, 3 * heads, d, dp)
            # reduction lengths that are not multiples of 64 (144, 288; the attention output of stage 1: 2 x 80) would take vv_conv_gemm's generic
            # register-staged loader, 5-10x slower than the LDS-DMA path on the 65536-token stage: the h16 operands carry zero-padded channels instead
            k1, k2, ko = _up64(b["dim"]), _up64(b["dim_out"]), _up64(heads * dp)
            padk = lambda w, k: torch.nn.functional.pad(w, (0, k - w.shape[1]))
            L = dict(b, d=d, dp=dp, k1=k1, k2=k2, ko=ko,
                     n1=(dev(W(n + ".norm1.weight")), dev(W(n + ".norm1.bias"))), n2=(dev(W(n + ".norm2.weight")), dev(W(n + ".norm2.bias"))),
                     qkv=_Lin(ctx, padk(wq, k1), bq),
                     proj=_Lin(ctx, padk(_pad_heads_cols(W(n + ".attn.proj.weight"), heads, d, dp), ko), W(n + ".attn.proj.bias")),
                     mlp0=_Lin(ctx, padk(W(n + ".mlp.layers.0.weight"), k2), W(n + ".mlp.layers.0.bias")),
                     mlp1=_Lin(ctx, W(n + ".mlp.layers.1.weight"), W(n + ".mlp.layers.1.bias")))
            if b["dim"] != b["dim_out"]:
                L["sc"] = _Lin(ctx, padk(W(n + ".proj.weight"), k1), W(n + ".proj.bias"))
            self.hb.append(L)
        # ---- neck + the decoder's high-resolution 1x1 convolutions (SAM2Base.forward_image)
        dims = list(reversed(cfg.stage_dims))
        self.neck = [_Lin(ctx, W(f"image_encoder.neck.convs.{j}.conv.weight").flatten(1), W(f"image_encoder.neck.convs.{j}.conv.bias"))
                     for j in range(len(dims))]
        Q = "sam_mask_decoder."
        self.conv_s0 = _Lin(ctx, W(Q + "conv_s0.weight").flatten(1), W(Q + "conv_s0.bias"))
        self.conv_s1 = _Lin(ctx, W(Q + "conv_s1.weight").flatten(1), W(Q + "conv_s1.bias"))
        self.pos_top = dev(_sine_pos_2d(D, fs, fs))
        # ---- memory attention
        cis = _axial_cis(D, fs, fs, cfg.rope_theta)
        self.rope_cs = dev(torch.stack([cis.real, cis.imag], dim=-1).float())           # [fs*fs, D/2, 2]
        self.ma = []
        for i in range(cfg.mem_attn_layers):
            n = f"memory_attention.layers.{i}"
            self.ma.append(dict(sa=_Attention(self, n + ".self_attn", 1, rope=True), ca=_Attention(self, n + ".cross_attn_image", 1, rope=True),
                                l1=_Lin(ctx, W(n + ".linear1.weight"), W(n + ".linear1.bias")), l2=_Lin(ctx, W(n + ".linear2.weight"), W(n + ".linear2.bias")),
                                n=[(dev(W(f"{n}.norm{k}.weight")), dev(W(f"{n}.norm{k}.bias"))) for k in (1, 2, 3)]))
        self.ma_norm = (dev(W("memory_attention.norm.weight")), dev(W("memory_attention.norm.bias")))
        # ---- memory encoder
        E = "memory_encoder."
        self.md, c = [], 1
        for j in range(4):
            n = f"{E}mask_downsampler.encoder"
            self.md.append((_Conv(ctx, W(f"{n}.{3 * j}.weight"), W(f"{n}.{3 * j}.bias"), cin_pad=max(8, c)),
                            dev(W(f"{n}.{3 * j + 1}.weight")), dev(W(f"{n}.{3 * j + 1}.bias")), c * 4))
            c *= 4
        self.md_out = _Lin(ctx, W(E + "mask_downsampler.encoder.12.weight").flatten(1), W(E + "mask_downsampler.encoder.12.bias"))
        n = f"{E}mask_downsampler.encoder"
        self.md_direct = [tuple(dev(W(f"{n}.{k}.{t}")) for k, t in ((3 * j, "weight"), (3 * j, "bias"), (3 * j + 1, "weight"), (3 * j + 1, "bias"))) for j in (0, 1)]
        self.pix_proj = _Lin(ctx, W(E + "pix_feat_proj.weight").flatten(1), W(E + "pix_feat_proj.bias"))
        self.fuser = []
        for i in range(cfg.fuser_layers):
            n = f"{E}fuser.layers.{i}"
            g = W(n + ".gamma")
            self.fuser.append(dict(dw=dev(W(n + ".dwconv.weight").reshape(D, 7, 7)), dwb=dev(W(n + ".dwconv.bias")),
                                   n=(dev(W(n + ".norm.weight")), dev(W(n + ".norm.bias"))),
                                   p1=_Lin(ctx, W(n + ".pwconv1.weight"), W(n + ".pwconv1.bias")),
                                   p2=_Lin(ctx, W(n + ".pwconv2.weight") * g[:, None], W(n + ".pwconv2.bias") * g)))     # layer scale folded in
        self.me_out = _Lin(ctx, W(E + "out_proj.weight").flatten(1), W(E + "out_proj.bias"))
        pos_mem = _sine_pos_2d(Mm, fs, fs)
        tpos = W("maskmem_tpos_enc")                                                     # [num_maskmem, 1, 1, Mm]
        self.mem_pos = [dev(pos_mem + tpos[k].reshape(1, Mm)) for k in range(cfg.num_maskmem)]     # sine encoding + temporal encoding k
        self.mem_pos_plain = dev(pos_mem)
        self.no_obj_embed_spatial = dev(W("no_obj_embed_spatial").reshape(Mm))
        self.no_mem_embed = dev(W("no_mem_embed").reshape(1, D).expand(fs * fs, D))
        self.no_obj_ptr = dev(W("no_obj_ptr").reshape(1, D))
        self.mask_downsample_w, self.mask_downsample_b = W("mask_downsample.weight").float().cpu(), W("mask_downsample.bias").float().cpu()      # host: prompt preparation
        self.obj_ptr_proj = [_Lin(ctx, W(f"obj_ptr_proj.layers.{j}.weight"), W(f"obj_ptr_proj.layers.{j}.bias")) for j in range(3)]
        self.tpos_proj = _Lin(ctx, W("obj_ptr_tpos_proj.weight"), W("obj_ptr_tpos_proj.bias"))
        # ---- prompt encoder
        P = "sam_prompt_encoder."
        self.gauss = dev(W(P + "pe_layer.positional_encoding_gaussian_matrix"))
        self.point_table = dev(torch.cat([W(P + "not_a_point_embed.weight")] + [W(f"{P}point_embeddings.{i}.weight") for i in range(4)], dim=0))
        self.no_mask_dense = dev(W(P + "no_mask_embed.weight").reshape(1, D).expand(fs * fs, D))
        mic = cfg.mask_in_chans
        self.mdn0 = _Conv(ctx, W(P + "mask_downscaling.0.weight"), W(P + "mask_downscaling.0.bias"), cin_pad=8)
        self.mdn1 = (dev(W(P + "mask_downscaling.1.weight")), dev(W(P + "mask_downscaling.1.bias")))
        self.mdn3 = _Conv(ctx, W(P + "mask_downscaling.3.weight"), W(P + "mask_downscaling.3.bias"), cin_pad=max(8, mic // 4))
        self.mdn4 = (dev(W(P + "mask_downscaling.4.weight")), dev(W(P + "mask_downscaling.4.bias")))
        self.mdn6 = _Lin(ctx, W(P + "mask_downscaling.6.weight").flatten(1), W(P + "mask_downscaling.6.bias"))
        g = (torch.arange(fs, dtype=torch.float32) + 0.5) / fs
        xy = torch.stack([g.view(1, fs).expand(fs, fs), g.view(fs, 1).expand(fs, fs)], dim=-1).reshape(-1, 2)
        c2 = (2 * xy - 1) @ W(P + "pe_layer.positional_encoding_gaussian_matrix") * (2 * math.pi)
        self.dense_pe = dev(torch.cat([torch.sin(c2), torch.cos(c2)], dim=-1))          # get_dense_pe(): a constant of the parameters
        # ---- mask decoder
        H = cfg.dec_heads
        self.dec = []
        for i in range(cfg.dec_depth):
            n = f"{Q}transformer.layers.{i}"
            self.dec.append(dict(sa=_Attention(self, n + ".self_attn", H), t2i=_Attention(self, n + ".cross_attn_token_to_image", H),
                                 i2t=_Attention(self, n + ".cross_attn_image_to_token", H),
                                 m0=_Lin(ctx, W(n + ".mlp.layers.0.weight"), W(n + ".mlp.layers.0.bias")),
                                 m1=_Lin(ctx, W(n + ".mlp.layers.1.weight"), W(n + ".mlp.layers.1.bias")),
                                 n=[(dev(W(f"{n}.norm{k}.weight")), dev(W(f"{n}.norm{k}.bias"))) for k in (1, 2, 3, 4)]))
        self.dec_final = _Attention(self, Q + "transformer.final_attn_token_to_image", H)
        self.dec_final_n = (dev(W(Q + "transformer.norm_final_attn.weight")), dev(W(Q + "transformer.norm_final_attn.bias")))
        self.out_tokens = dev(torch.cat([W(Q + "obj_score_token.weight"), W(Q + "iou_token.weight"), W(Q + "mask_tokens.weight")], dim=0))
        w1, w3 = W(Q + "output_upscaling.0.weight"), W(Q + "output_upscaling.3.weight")     # ConvTranspose2d [cin, cout, 2, 2] -> rows (dy, dx, cout)
        self.up1 = _Lin(ctx, w1.permute(2, 3, 1, 0).reshape(-1, w1.shape[0]))
        self.up1_b = dev(W(Q + "output_upscaling.0.bias"))
        self.up_ln = (dev(W(Q + "output_upscaling.1.weight")), dev(W(Q + "output_upscaling.1.bias")))
        self.up2 = _Lin(ctx, w3.permute(2, 3, 1, 0).reshape(-1, w3.shape[0]))
        self.up2_b = dev(W(Q + "output_upscaling.3.bias"))
        nm = cfg.num_multimask + 1
        self.hyper = [[_Lin(ctx, W(f"{Q}output_hypernetworks_mlps.{i}.layers.{j}.weight"), W(f"{Q}output_hypernetworks_mlps.{i}.layers.{j}.bias"))
                       for j in range(3)] for i in range(nm)]
        self.iou_head = [_Lin(ctx, W(f"{Q}iou_prediction_head.layers.{j}.weight"), W(f"{Q}iou_prediction_head.layers.{j}.bias")) for j in range(3)]
        self.obj_head = [_Lin(ctx, W(f"{Q}pred_obj_score_head.layers.{j}.weight"), W(f"{Q}pred_obj_score_head.layers.{j}.bias")) for j in range(3)]

    # ---- helpers -----------------------------------------------------------------------------------------------------------------
    def _ln(self, x, gb, eps, act=hip.ACT_NONE, out_dtype=None, cpad=None):
        return hip.layernorm_ex(self.ctx.dt, x, gb[0], gb[1], eps, act=act, out_dtype=out_dtype, cpad=cpad)

    def _mlp(self, x, layers, sigmoid=False):
        for j, l in enumerate(layers):
            last = j == len(layers) - 1
            x = l(x, act=hip.ACT_NONE if last else hip.ACT_RELU)
        return hip.act_inplace(x, hip.ACT_SIGMOID) if sigmoid else x

    def _index(self, key, builder):
        if key not in self._idx:
            self._idx[key] = torch.from_numpy(builder().astype(np.int32)).to(self.device)
        return self._idx[key]

    def _index_free(self, key, builder):
        if key not in self._idx:
            self._idx[key] = builder()
        return self._idx[key]

    def _part_idx(self, Bf, H, W, ws):
        """row of window-ordered token (frame, window, wy, wx) in the [Bf*H*W] map."""
        def b():
            y, x = np.arange(H).reshape(H // ws, ws), np.arange(W).reshape(W // ws, ws)
            one = (y[:, None, :, None] * W + x[None, :, None, :]).reshape(-1)
            return (np.arange(Bf)[:, None] * (H * W) + one[None, :]).reshape(-1)
        return self._index(("part", Bf, H, W, ws), b)

    def _unpart_idx(self, Bf, H, W, ws):
        """row of map token (frame, y, x) in the window-ordered matrix."""
        def b():
            y, x = np.arange(H)[:, None], np.arange(W)[None, :]
            one = (((y // ws) * (W // ws) + x // ws) * ws * ws + (y % ws) * ws + x % ws).reshape(-1)
            return (np.arange(Bf)[:, None] * (H * W) + one[None, :]).reshape(-1)
        return self._index(("unpart", Bf, H, W, ws), b)

    # ---- image encoder -------------------------------------------------------------------------------------------------------------
    def _block(self, x, Bf, H, W, L):
        ctx, dt = self.ctx, self.ctx.dt
        heads, dp, ws = L["heads"], L["dp"], L["window"]
        xn = self._ln(x, L["n1"], 1e-6, cpad=L["k1"])
        sc = x
        if "sc" in L:
            sc = hip.maxpool2x2(L["sc"](xn), Bf, H, W)
        if ws > 0:
            if H % ws or W % ws:
                raise RuntimeError(f"Hiera window {ws} does not divide the {H}x{W} map (padded windows are not built)")
            xn = hip.gather_rows(xn, self._part_idx(Bf, H, W, ws))
            B, N, wh = Bf * (H // ws) * (W // ws), ws * ws, ws
        else:
            B, N, wh = Bf, H * W, H
        ww = N // wh
        qkv = L["qkv"](xn, out_dtype=ctx.h16, split=(heads, dp, N))                      # [B][q|k|v][head][token][dp]
        blk = heads * N * dp
        Nq, Ho, Wo = N, H, W
        if L["q_stride"]:
            # q of window b = `heads` stacked [wh, ww, dp] images: one pooling launch over B images of height heads * wh
            q = hip.maxpool2x2(qkv, B, heads * wh, ww, Cc=dp, in_bs=3 * blk)
            Nq, Ho, Wo = N // 4, H // 2, W // 2
            q_bs = heads * Nq * dp
        else:
            q, q_bs = qkv, 3 * blk
        ko = L["ko"]
        if ko == heads * dp:
            o = torch.empty((B * Nq, ko), dtype=ctx.h16, device=x.device)
        else:           # padded row: the attention kernel writes the first heads * dp columns, the rest stays zero (one buffer per shape, reused)
            o = self._index_free(("opad", B * Nq, ko), lambda: torch.zeros((B * Nq, ko), dtype=ctx.h16, device=x.device))
        hip.attention(dt, q, qkv, qkv, o, B=B, heads=heads, Nq=Nq, Nkv=N, D=dp, q_bs=q_bs, k_bs=3 * blk, v_bs=3 * blk, o_bs=Nq * ko,
                      q_rs=dp, k_rs=dp, v_rs=dp, o_rs=ko, k_off=blk, v_off=2 * blk, q_hs=Nq * dp, k_hs=N * dp, v_hs=N * dp,
                      scale=float(L["d"]) ** -0.5)
        if ws > 0:
            y = hip.gather_rows(L["proj"](o), self._unpart_idx(Bf, Ho, Wo, ws // 2 if L["q_stride"] else ws))
            x = hip.axpby(sc, y, 1.0, 1.0)
        else:
            x = L["proj"](o, res0=sc)
        h = L["mlp0"](self._ln(x, L["n2"], 1e-6, cpad=L["k2"]), out_dtype=ctx.h16)
        hip.act_inplace(h, hip.ACT_GELU)
        return L["mlp1"](h, res0=x), Ho, Wo

    def encode_images(self, frames_u8):
        """a batch of frames through the image encoder in one pass (every GEMM sees Bf x the rows: the stage-3 blocks of a single 1024^2 frame are
        only 4096 tokens); results are per frame and do not depend on the batch they were computed in."""
        cfg, ctx = self.cfg, self.ctx
        S, Bf = cfg.image_size, len(frames_u8)
        img = torch.from_numpy(np.ascontiguousarray(np.stack(frames_u8))).to(self.device)
        if tuple(img.shape[1:3]) != (S, S):
            img = hip.resize_u8(img.contiguous(), S, S, mode="bilinear")                    # cv2.INTER_LINEAR semantics (bit exact, vv_image.hip)
        x48 = hip.u8_normalize(ctx.dt, img.reshape(Bf * S, S, 3).contiguous(), IMG_MEAN, IMG_STD, 48, s2d=4)     # frames stacked vertically (S % 4 == 0)
        H = W = S // 4
        pos = self._index_free(("pos", Bf), lambda: self.pos_embed.repeat(Bf, 1))
        x = hip.conv_gemm(ctx.dt, x48, self.patch.w, self.patch.N, self.patch.K, F=Bf, Hin=H, Win=W, Hout=H, Wout=W, ksize=2, stride=1, pad_t=1, pad_l=1,
                          bias=self.patch.b, res0=pos, out_dtype=torch.float32)
        outs = []
        for i, L in enumerate(self.hb):
            x, H, W = self._block(x, Bf, H, W, L)
            if i in self.stage_ends:
                outs.append((x, H, W))
        n = len(outs) - 1
        feats, prev = [None] * len(outs), None
        for i in range(n, -1, -1):
            xi, Hi, Wi = outs[i]
            lat = self.neck[n - i](xi)
            if i in cfg.fpn_top_down_levels and prev is not None:
                one = lambda: ((np.arange(Hi)[:, None] // 2) * (Wi // 2) + np.arange(Wi)[None, :] // 2).reshape(-1)
                up = hip.gather_rows(prev, self._index(("up2", Bf, Hi, Wi), lambda: (np.arange(Bf)[:, None] * ((Hi // 2) * (Wi // 2)) + one()[None, :]).reshape(-1)))
                prev = hip.axpby(lat, up, 1.0, 1.0)
            else:
                prev = lat
            feats[i] = prev
        if cfg.scalp > 0:
            feats = feats[:-cfg.scalp]
        f0, f1, top = feats[-3:]
        s0, s1 = self.conv_s0(f0), self.conv_s1(f1)
        n0, n1, nt = s0.shape[0] // Bf, s1.shape[0] // Bf, top.shape[0] // Bf
        if getattr(self.W, "sd", None) is not None and not getattr(self, "_range_checked", False):
            # a real checkpoint: the first encoded batch is range-checked once (fp16 operands saturate to inf above 65504; every parity result of
            # this build was measured on synthetic weights) -- fail loudly instead of painting garbage masks
            self._range_checked = True
            if not bool(torch.isfinite(top).all()):
                raise RuntimeError("SAM 2 image encoder produced non-finite features with this checkpoint: configure(dtype=\"bf16\") "
                                   "(the reference's own autocast type) instead of fp16 operands")
        return [{"s0": s0[f * n0:(f + 1) * n0], "s1": s1[f * n1:(f + 1) * n1], "top": top[f * nt:(f + 1) * nt]} for f in range(Bf)]

    def encode_image(self, frame_u8):
        return self.encode_images([frame_u8])[0]

    # ---- memory attention ----------------------------------------------------------------------------------------------------------
    def _memory_attention(self, cur, memory, memory_pos, n_ptr_tokens):
        ctx = self.ctx
        out = hip.axpby(cur, self.pos_top, 1.0, 0.1)
        kin = hip.axpby(memory, memory_pos, 1.0, 1.0)
        for L in self.ma:
            t2 = self._ln(out, L["n"][0], 1e-5)
            out = L["sa"](t2, t2, t2, res0=out)
            t2 = self._ln(out, L["n"][1], 1e-5)
            out = L["ca"](t2, kin, memory, res0=out, rows_rope_k=memory.shape[0] - n_ptr_tokens)
            t2 = self._ln(out, L["n"][2], 1e-5)
            out = L["l2"](L["l1"](t2, out_dtype=ctx.h16, act=hip.ACT_RELU), res0=out)
        return self._ln(out, self.ma_norm, 1e-5, out_dtype=torch.float32)

    def _memory_conditioned(self, frame_idx, is_init_cond_frame, feats, output_dict, num_frames, track_in_reverse):
        cfg = self.cfg
        D, Mm = cfg.d_model, cfg.mem_dim
        if is_init_cond_frame:
            return hip.axpby(feats["top"], self.no_mem_embed, 1.0, 1.0)
        mems, ptrs, max_ptrs = select_memories(cfg, frame_idx, output_dict, num_frames, track_in_reverse)
        mem = [prev["maskmem_features"] for _, prev in mems]
        pos = [self.mem_pos[cfg.num_maskmem - t_pos - 1] for t_pos, _ in mems]
        n_ptr_tokens = 0
        if ptrs:
            split = D // Mm
            plist = torch.tensor([p for p, _ in ptrs], dtype=torch.float32) / float(max_ptrs - 1)
            obj_pos = self.tpos_proj(hip.sine_pe_1d(plist.to(self.device), D))                      # [n, Mm]
            rep = self._index(("rep", len(ptrs), split), lambda: np.repeat(np.arange(len(ptrs)), split))
            mem.append(torch.cat([o["obj_ptr"] for _, o in ptrs], dim=0).reshape(len(ptrs) * split, Mm))
            pos.append(hip.gather_rows(obj_pos, rep))
            n_ptr_tokens = len(ptrs) * split
        return self._memory_attention(feats["top"], torch.cat(mem, dim=0), torch.cat(pos, dim=0), n_ptr_tokens)

    # ---- memory encoder ------------------------------------------------------------------------------------------------------------
    FUSED_MASKDOWN = True     # class-level switch: tests compare the fused front of the mask path with the layer-by-layer one

    def _memory_encoder(self, pix_feat, mask_in, S, first=0):
        """mask_in: h16 [S*S, 8] (layer-by-layer form, first = 0) or the output of vv_sam2_maskdown ([(S/4)^2, 16], first = 2 layers done)."""
        ctx, fs = self.ctx, self.cfg.feat_size
        m, H, W = mask_in, S >> first, S >> first
        for conv, g, b, cout in self.md[first:]:
            y, H, W = conv(m, H, W, 2, 1)
            m = hip.layernorm_ex(ctx.dt, y, g, b, 1e-6, act=hip.ACT_GELU, cpad=max(8, cout))
        x = self.pix_proj(pix_feat, res0=self.md_out(m))
        for L in self.fuser:
            h = hip.dwconv(x, fs, fs, L["dw"], L["dwb"])
            h = L["p1"](self._ln(h, L["n"], 1e-6), out_dtype=ctx.h16)
            hip.act_inplace(h, hip.ACT_GELU)
            x = L["p2"](h, res0=x)
        return self.me_out(x)

    def encode_memory_from_low_res(self, feats, pred_masks, object_score_logits, is_mask_from_pts):
        cfg = self.cfg
        S, lo = cfg.image_size, 4 * cfg.feat_size
        binarize = cfg.binarize_mask_from_pts_for_mem_enc and is_mask_from_pts
        if HipSam2.FUSED_MASKDOWN:
            m = hip.sam2_maskdown(self.ctx.dt, pred_masks.reshape(-1), lo, S, binarize, cfg.sigmoid_scale_for_mem_enc, cfg.sigmoid_bias_for_mem_enc,
                                  self.md_direct[0], self.md_direct[1])
            f = self._memory_encoder(feats["top"], m, S, first=2)
        else:
            high = hip.resize_bilinear_f32(pred_masks.reshape(lo * lo, 1), lo, lo, S, S)
            m = hip.mask_mem_input(self.ctx.dt, high, binarize, cfg.sigmoid_scale_for_mem_enc, cfg.sigmoid_bias_for_mem_enc)
            f = self._memory_encoder(feats["top"], m, S)
        hip.add_rowvec_unless(f, self.no_obj_embed_spatial, object_score_logits)
        return f, self.mem_pos_plain

    # ---- SAM heads -----------------------------------------------------------------------------------------------------------------
    def _two_way(self, src, tokens):
        keys, queries = src, tokens
        qpe, kpe = tokens, self.dense_pe
        add = lambda a, b: hip.axpby(a, b, 1.0, 1.0)
        for i, L in enumerate(self.dec):
            if i == 0:
                queries = L["sa"](queries, queries, queries)
            else:
                q = add(queries, qpe)
                queries = L["sa"](q, q, queries, res0=queries)
            queries = self._ln(queries, L["n"][0], 1e-5, out_dtype=torch.float32)
            q, k = add(queries, qpe), add(keys, kpe)
            queries = self._ln(L["t2i"](q, k, keys, res0=queries), L["n"][1], 1e-5, out_dtype=torch.float32)
            queries = self._ln(L["m1"](L["m0"](queries, out_dtype=self.ctx.h16, act=hip.ACT_RELU), res0=queries), L["n"][2], 1e-5, out_dtype=torch.float32)
            q = add(queries, qpe)
            keys = self._ln(L["i2t"](k, q, queries, res0=keys), L["n"][3], 1e-5, out_dtype=torch.float32)
        q, k = add(queries, qpe), add(keys, kpe)
        queries = self._ln(self.dec_final(q, k, keys, res0=queries), self.dec_final_n, 1e-5, out_dtype=torch.float32)
        return queries, keys

    def _sam_heads(self, pix_feat, feats, point_inputs, mask_inputs, multimask_output):
        cfg, ctx = self.cfg, self.ctx
        fs, D = cfg.feat_size, cfg.d_model
        nm = cfg.num_multimask + 1
        if point_inputs is not None:
            coords = torch.cat([point_inputs["point_coords"].float().reshape(-1, 2), torch.zeros(1, 2)], dim=0)
            labels = torch.cat([point_inputs["point_labels"].int().reshape(-1), torch.tensor([-1], dtype=torch.int32)], dim=0)
        else:
            coords, labels = torch.zeros(2, 2), -torch.ones(2, dtype=torch.int32)
        sparse = hip.prompt_points(coords.contiguous().to(self.device), labels.contiguous().to(self.device), 1.0 / cfg.image_size, self.gauss, self.point_table)
        if mask_inputs is not None:
            lo = 4 * fs
            m8 = hip.pad_channels(ctx.dt, mask_inputs.reshape(lo * lo, 1), 8)
            y, H, W = self.mdn0(m8, lo, lo, 2, 0)
            mic = cfg.mask_in_chans
            y = hip.layernorm_ex(ctx.dt, y, self.mdn1[0], self.mdn1[1], 1e-6, act=hip.ACT_GELU, cpad=max(8, mic // 4))
            y, H, W = self.mdn3(y, H, W, 2, 0)
            y = hip.layernorm_ex(ctx.dt, y, self.mdn4[0], self.mdn4[1], 1e-6, act=hip.ACT_GELU)
            src = self.mdn6(y, res0=pix_feat)
        else:
            src = hip.axpby(pix_feat, self.no_mask_dense, 1.0, 1.0)
        tokens = torch.cat([self.out_tokens, sparse], dim=0)
        hs, keys = self._two_way(src, tokens)
        # upscaling with the high-resolution features
        y = self.up1(keys)
        u = hip.pixel_shuffle2(ctx.dt, y, self.up1_b, fs, fs, add=feats["s1"])
        u = self._ln(u, self.up_ln, 1e-6, act=hip.ACT_GELU)
        y = self.up2(u)
        up = hip.pixel_shuffle2(ctx.dt, y, self.up2_b, 2 * fs, 2 * fs, add=feats["s0"], act=hip.ACT_GELU)        # fp32 [(4 fs)^2, D/8]
        hyper = torch.cat([self._mlp(hs[2 + i:3 + i].contiguous(), self.hyper[i]) for i in range(nm)], dim=0)
        masks = hip.hyper_masks(hyper, up)                                                # [nm, (4 fs)^2]
        iou = self._mlp(hs[1:2].contiguous(), self.iou_head, sigmoid=True).reshape(-1)
        obj = self._mlp(hs[0:1].contiguous(), self.obj_head).reshape(-1)
        sel = hip.sam_select(masks, iou, obj, multimask_output, cfg.stability_delta, cfg.stability_thresh)
        low = hip.sam_pick(masks, sel, NO_OBJ_SCORE)
        token = hip.gather_rows(hs[2:2 + nm].contiguous(), sel[2:3].contiguous())
        ptr = hip.select_f32(self._mlp(token, self.obj_ptr_proj), self.no_obj_ptr, sel[1:2].contiguous())
        return low, ptr, obj

    def use_multimask(self, is_init_cond_frame, point_inputs):
        n = 0 if point_inputs is None else point_inputs["point_labels"].shape[1]
        return self.cfg.multimask_min_pt_num <= n <= self.cfg.multimask_max_pt_num

    def use_mask_as_output(self, feats, mask_inputs):
        """SAM2Base._use_mask_as_output (oracle/sam2_ref.py::use_mask_as_output): a caller-supplied binary mask [1, 1, S, S] (host tensor: it is a PROMPT) is
        the frame's output.  Prompt preparation stays on the host like the click coordinates do -- logits -10 / +10, their antialiased low-resolution copy,
        the learned 4x4 / stride-4 `mask_downsample` of the mask (17 parameters) --; the SAM heads then run on the device for the object pointer, on the frame's
        RAW top-level features, with that downsampled mask as their mask prompt; "does the object appear" is read off the mask."""
        cfg = self.cfg
        lo = 4 * cfg.feat_size
        m = mask_inputs.detach().float().cpu()
        low = torch.nn.functional.interpolate(m * 20.0 - 10.0, size=(lo, lo), mode="bilinear", align_corners=False, antialias=True)
        md = torch.nn.functional.conv2d(m, self.mask_downsample_w, self.mask_downsample_b, stride=4)
        appears = bool((m > 0).any())
        _, ptr, _ = self._sam_heads(feats["top"], feats, None, md.reshape(-1).contiguous().to(self.device), False)
        if not appears:
            ptr = self.no_obj_ptr.clone()
        obj = torch.tensor([10.0 if appears else -10.0], dtype=torch.float32, device=self.device)
        return low.reshape(-1).contiguous().to(self.device), ptr, obj

    def track_step(self, frame_idx, is_init_cond_frame, feats, point_inputs, output_dict, num_frames, track_in_reverse=False,
                   run_mem_encoder=True, prev_sam_mask_logits=None, mask_inputs=None):
        if mask_inputs is not None:
            masks, ptr, obj = self.use_mask_as_output(feats, mask_inputs)
            out = {"pred_masks": masks, "obj_ptr": ptr, "object_score_logits": obj, "maskmem_features": None, "maskmem_pos_enc": None}
            if run_mem_encoder:
                # upstream SAM2Base._encode_memory_in_output: is_mask_from_pts = (point_inputs is not None) -- False for a mask prompt: the
                # memory encoder sees sigmoid(logits), not the binarised mask
                out["maskmem_features"], out["maskmem_pos_enc"] = self.encode_memory_from_low_res(feats, masks, obj, point_inputs is not None)
            return out
        pix = self._memory_conditioned(frame_idx, is_init_cond_frame, feats, output_dict, num_frames, track_in_reverse)
        masks, ptr, obj = self._sam_heads(pix, feats, point_inputs, prev_sam_mask_logits, self.use_multimask(is_init_cond_frame, point_inputs))
        out = {"pred_masks": masks, "obj_ptr": ptr, "object_score_logits": obj, "maskmem_features": None, "maskmem_pos_enc": None}
        if run_mem_encoder:
            out["maskmem_features"], out["maskmem_pos_enc"] = self.encode_memory_from_low_res(feats, masks, obj, point_inputs is not None)
        return out

    # ---- predictor-facing utilities ------------------------------------------------------------------------------------------------
    def fill_holes(self, pred_masks):
        a, lo = self.cfg.fill_hole_area, 4 * self.cfg.feat_size
        if a <= 0:
            return pred_masks
        return hip.fill_holes(pred_masks.clone(), lo, lo, a)

    def masks_to_video_res(self, pred_masks, H, W):
        lo = 4 * self.cfg.feat_size
        if (H, W) == (lo, lo):
            return pred_masks.reshape(1, 1, H, W)
        return hip.resize_bilinear_f32(pred_masks.reshape(lo * lo, 1), lo, lo, H, W).reshape(1, 1, H, W)

    def clamp_prev_logits(self, pred_masks):
        return hip.clamp_f32(pred_masks, -32.0, 32.0)

    def to_numpy(self, t):
        return t.detach().cpu().numpy()


# ---- constants of the configuration (no parameters involved), computed once on the host ---------------------------------------------------
def _sine_pos_2d(num_pos_feats, h, w, temperature=10000.0):
    """PositionEmbeddingSine(normalize=True, scale=2 pi) as rows [h*w, num_pos_feats]."""
    npf = num_pos_feats // 2
    y = torch.arange(1, h + 1, dtype=torch.float32).view(h, 1).expand(h, w)
    x = torch.arange(1, w + 1, dtype=torch.float32).view(1, w).expand(h, w)
    y, x = y / (h + 1e-6) * 2 * math.pi, x / (w + 1e-6) * 2 * math.pi
    dim_t = temperature ** (2 * (torch.arange(npf, dtype=torch.float32) // 2) / npf)
    px, py = x[:, :, None] / dim_t, y[:, :, None] / dim_t
    px = torch.stack((px[:, :, 0::2].sin(), px[:, :, 1::2].cos()), dim=3).flatten(2)
    py = torch.stack((py[:, :, 0::2].sin(), py[:, :, 1::2].cos()), dim=3).flatten(2)
    return torch.cat((py, px), dim=2).reshape(h * w, num_pos_feats)


def _axial_cis(dim, end_x, end_y, theta):
    fr = 1.0 / (theta ** (torch.arange(0, dim, 4)[: dim // 4].float() / dim))
    t = torch.arange(end_x * end_y, dtype=torch.float32)
    tx, ty = (t % end_x).float(), torch.div(t, end_x, rounding_mode="floor").float()
    fx, fy = torch.outer(tx, fr), torch.outer(ty, fr)
    return torch.cat([torch.polar(torch.ones_like(fx), fx), torch.polar(torch.ones_like(fy), fy)], dim=-1)
