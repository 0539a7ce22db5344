"""ProPainter's inpainting generator on the HIP kernels (oracle: oracle/inpaintgen_ref.py).

Reference: third-party `propainter` model/propainter.py (InpaintGenerator) + model/modules/sparse_transformer.py, reached from reference
diffuerase.py:52-57 through `Propainter.forward` (ref_stride / neighbor_length of diffuerase.py:53-54 pick this network's frames).
Layout: NHWC rows; tokens [t * f_h * f_w, 512].  How the pieces map onto the kernels:
  grouped encoder convs       -> ONE dense two-source `vv_conv_gemm` per layer with block-sparse weights (the group-wise skip concatenation
                                 becomes the kernel's (x0 | x1) channel concat; 2-8x the FLOPs of a grouped conv on 1/4-resolution maps: ~1 ms)
  flow_warp, soft split, unfold-> `vv_deform_im2col` (1x1 tap with offset = flow; 7x7 stride 3 / 4x4 stride 4 taps with zero offsets)
  DeformableAlignment          -> deform.DeformableAlignment (fused tanh / sigmoid / flow front end)
  sparse window attention      -> fused QKV GEMM, `vv_gather_rows` with host-built index tables (window partition, rolled neighbour keys,
                                 pooled global keys, inverse scatter), `vv_attention` (d = 128) once for the windows that touch a hole
                                 (queries of all frames x keys of every t_dilation-th frame) and once for the clean windows (per frame)
  fusion feed-forward          -> GEMM, `vv_fold_patches` (overlap-add + count normalisation + GELU, which commutes with the unfold gather), GEMM
  soft composition             -> GEMM, `vv_fold_patches`, 3x3 conv with the residual in the epilogue
Which windows touch a hole is decided on the host from the caller's uint8 masks (no device synchronisation).  No CPU fallback."""
import math

import numpy as np
import torch

from . import hip, packing
from .deform import DeformableAlignment
from .flowcomplete import _Conv

K7, S3, P3 = 7, 3, 3
GROUPS = [1, 2, 4, 8, 1]


def token_grid(h, w):
    return (h + 2 * P3 - K7) // S3 + 1, (w + 2 * P3 - K7) // S3 + 1


class _Linear:
    def __init__(self, ctx, weight, bias):
        self.ctx, self.N, self.K = ctx, weight.shape[0], weight.shape[1]
        self.w = ctx.dev(packing.pack_matrix(weight.contiguous(), ctx.h16))
        self.b = ctx.dev(bias.float()) if bias is not None else None

    def __call__(self, x, res0=None, out_dtype=None):
        return hip.conv_gemm(self.ctx.dt, x, self.w, self.N, self.K, F=1, Hin=1, Win=x.shape[0], bias=self.b, res0=res0,
                             out_dtype=out_dtype if out_dtype is not None else self.ctx.h16)


class _Encoder:
    SPEC = [(64, 2, 1), (64, 1, 1), (128, 2, 1), (256, 1, 1), (384, 1, 1), (512, 1, 2), (384, 1, 4), (256, 1, 8), (128, 1, 1)]

    def __init__(self, ctx, name):
        self.ctx, self.layers = ctx, []
        cin = 5
        for li, (co, s, g) in enumerate(self.SPEC):
            i = 2 * li
            if i > 8:
                # reference: groups g over cat_g([x0_g | prev_g]).  Dense two-source form: columns [x0 (256) | prev (cprev)], zero outside the group
                cprev = cin
                w, b = ctx.src.conv(f"{name}.layers.{i}", (256 + cprev) // g, co, 3, 1.0)
                wd = torch.zeros(co, 256 + cprev, 3, 3)
                a, p, og = 256 // g, cprev // g, co // g
                for j in range(g):
                    wd[j * og:(j + 1) * og, j * a:(j + 1) * a] = w[j * og:(j + 1) * og, :a]
                    wd[j * og:(j + 1) * og, 256 + j * p:256 + (j + 1) * p] = w[j * og:(j + 1) * og, a:]
                wp, K = packing.pack_conv(wd, ctx.h16, None)
                self.layers.append((ctx.dev(wp), K, ctx.dev(b.float()), co, s, True))
            else:
                cpad = (cin + 7) // 8 * 8
                w, b = ctx.src.conv(f"{name}.layers.{i}", cin, co, 3, 1.0)
                wp, K = packing.pack_conv(w, ctx.h16, cpad if cpad != cin else None)
                self.layers.append((ctx.dev(wp), K, ctx.dev(b.float()), co, s, False))
            cin = co

    def __call__(self, x, F, H, W):
        """x fp32 [F*H*W, 8] -> fp32 [F*(H/4)*(W/4), 128]."""
        out, x0 = x, None
        for li, (w, K, b, co, s, skip) in enumerate(self.layers):
            if 2 * li == 8:
                x0 = out
            Ho, Wo = (H + 2 - 3) // s + 1, (W + 2 - 3) // s + 1
            last = li == len(self.layers) - 1
            out = hip.conv_gemm(self.ctx.dt, x0 if skip else out, w, co, K, x1=out if skip else None, F=F, Hin=H, Win=W, Hout=Ho, Wout=Wo, ksize=3,
                                stride=s, pad_t=1, pad_l=1, bias=b, out_dtype=torch.float32 if last else self.ctx.h16, act=hip.ACT_LRELU, act_slope=0.2)
            H, W = Ho, Wo
        return out, H, W


class _FeaturePropagation:
    def __init__(self, ctx, name, C, deform_groups):
        self.ctx, self.C = ctx, C
        self.align, self.bb0, self.bb2 = {}, {}, {}
        for mod in ("backward_1", "forward_1"):
            self.align[mod] = DeformableAlignment(ctx, f"{name}.deform_align.{mod}", C, 2 * C + 5, deform_groups, 3.0)
            self.bb0[mod] = _Conv(ctx, f"{name}.backbone.{mod}.0", 2 * C + 2, C, 3)
            self.bb2[mod] = _Conv(ctx, f"{name}.backbone.{mod}.2", C, C, 3)
        self.fuse0 = _Conv(ctx, f"{name}.fuse.0", 2 * C + 2, C, 3)
        self.fuse2 = _Conv(ctx, f"{name}.fuse.2", C, C, 3)

    def __call__(self, x, flows_f, flows_b, masks2, T, H, W):
        """x fp32 [T*H*W, C]; flows fp32 [T-1, H, W, 2]; masks2 fp32 [T*H*W, 2] -> fp32 [T*H*W, C]."""
        C, HW, dev = self.C, H * W, x.device
        pad6 = torch.zeros((HW, 6), dtype=torch.float32, device=dev)
        pad3 = torch.zeros((HW, 3), dtype=torch.float32, device=dev)
        feats = {"input": [x[t * HW:(t + 1) * HW] for t in range(T)]}
        cache = ["input", "backward_1", "forward_1"]
        for p_i, mod in enumerate(("backward_1", "forward_1")):
            feats[mod] = []
            if mod == "backward_1":
                frame_idx = list(range(T))[::-1]
                flow_idx = frame_idx
                f_prop, f_check = flows_f, flows_b
            else:
                frame_idx = list(range(T))
                flow_idx = list(range(-1, T - 1))
                f_prop, f_check = flows_b, flows_f
            prop = None
            for i, idx in enumerate(frame_idx):
                cur = feats[cache[p_i]][idx]
                mcur = masks2[idx * HW:(idx + 1) * HW]
                if i == 0:
                    prop = cur
                else:
                    fp, fc = f_prop[flow_idx[i]], f_check[flow_idx[i]]
                    valid = hip.fb_valid(fp, fc).reshape(HW, 1).float()
                    fp2 = fp.reshape(HW, 2)
                    warped, _, _ = hip.deform_im2col(self.ctx.dt, prop, B=1, H=H, W=W, kh=1, kw=1, stride=1, pad=0, dil=1, deform_groups=1,
                                                     offset=fp2.flip(1).contiguous())
                    cond = torch.cat([cur, warped.float(), fp2, valid, mcur, pad3], 1)
                    prop = self.align[mod](prop, cond, fp2, 1, H, W)
                h, _, _ = self.bb0[mod](torch.cat([cur, prop, mcur, pad6], 1), 1, H, W, act=0.2)
                prop, _, _ = self.bb2[mod](h, 1, H, W, res0=prop, out_dtype=torch.float32)
                feats[mod].append(prop)
            if mod == "backward_1":
                feats[mod] = feats[mod][::-1]
        ob, of = torch.cat(feats["backward_1"], 0), torch.cat(feats["forward_1"], 0)
        pad6t = torch.zeros((T * HW, 6), dtype=torch.float32, device=dev)
        h, _, _ = self.fuse0(torch.cat([ob, of, masks2, pad6t], 1), T, H, W, act=0.2)
        out, _, _ = self.fuse2(h, T, H, W, res0=x, out_dtype=torch.float32)
        return out


def _window_tables(t, fh, fw, ws, pool, T_ind):
    """Index tables of the sparse window attention over a [t, fh, fw] token grid padded to whole windows.  Rows of the gathered table:
    first the t * nh * nw padded tokens, then the t * ph * pw pooled tokens."""
    wh, ww = ws
    n_wh, n_ww = math.ceil(fh / wh), math.ceil(fw / ww)
    nh, nw = n_wh * wh, n_ww * ww
    ph, pw = nh // pool[0], nw // pool[1]
    e = ((wh + 1) // 2, (ww + 1) // 2)
    nwin, L = n_wh * n_ww, wh * ww
    pad_idx = np.full((t, nh, nw), -1, np.int32)
    pad_idx[:, :fh, :fw] = np.arange(t * fh * fw, dtype=np.int32).reshape(t, fh, fw)
    # window (wy, wx), position (iy, ix) -> padded-grid coordinates
    wy, wx = np.divmod(np.arange(nwin), n_ww)
    iy, ix = np.divmod(np.arange(L), ww)
    oy = (wy[:, None] * wh + iy[None, :])                                    # [nwin, L]
    ox = (wx[:, None] * ww + ix[None, :])
    frames = np.arange(t)
    q_idx = ((frames[None, :, None] * nh + oy[:, None, :]) * nw + ox[:, None, :]).astype(np.int32)        # [nwin, t, L]
    # rolled neighbour-window keys: torch.roll(k, (sy, sx)) then window partition = token ((y - sy) % nh, (x - sx) % nw); only the
    # positions that really come from OUTSIDE the window are kept (the reference's four corner masks)
    m_tl = np.ones(ws, bool); m_tl[:-e[0], :-e[1]] = False
    m_tr = np.ones(ws, bool); m_tr[:-e[0], e[1]:] = False
    m_bl = np.ones(ws, bool); m_bl[e[0]:, :-e[1]] = False
    m_br = np.ones(ws, bool); m_br[e[0]:, e[1]:] = False
    ry, rx = [], []
    for (sy, sx), m in (((-e[0], -e[1]), m_tl), ((-e[0], e[1]), m_tr), ((e[0], -e[1]), m_bl), ((e[0], e[1]), m_br)):
        sel = np.nonzero(m.reshape(-1))[0]
        ry.append((oy[:, sel] - sy) % nh)
        rx.append((ox[:, sel] - sx) % nw)
    ky = np.concatenate([oy] + ry, 1)                                        # [nwin, L + roll_N]: own window, then the rolled keys
    kx = np.concatenate([ox] + rx, 1)
    tf = np.asarray(list(T_ind))
    tok_k = (tf[None, :, None] * nh + ky[:, None, :]) * nw + kx[:, None, :]                                # [nwin, |T_ind|, L + roll_N]
    pooled = t * nh * nw + tf[:, None] * (ph * pw) + np.arange(ph * pw)[None, :]                          # [|T_ind|, ph * pw]
    k_idx = np.concatenate([tok_k, np.broadcast_to(pooled[None], (nwin, len(tf), ph * pw))], 2).reshape(nwin, -1).astype(np.int32)
    # inverse: token (f, y, x) of the UNPADDED grid -> its slot in the [window][frame][position] output order
    y, x = np.arange(fh), np.arange(fw)
    win = (y[:, None] // wh) * n_ww + x[None, :] // ww
    pos = (y[:, None] % wh) * ww + x[None, :] % ww
    inv = (win[None] * (t * L) + frames[:, None, None] * L + pos[None]).astype(np.int32)
    return dict(nh=nh, nw=nw, ph=ph, pw=pw, nwin=nwin, n_wh=n_wh, n_ww=n_ww, pad_idx=pad_idx.reshape(-1), q_idx=q_idx, k_idx=k_idx, inv=inv.reshape(-1))


class _TransformerBlock:
    def __init__(self, ctx, name, dim, n_head, ws, pool, hidden=1960):
        self.ctx, self.dim, self.n_head, self.ws, self.pool, self.hidden = ctx, dim, n_head, ws, pool, hidden
        src = ctx.src
        g1, b1 = src.norm(f"{name}.norm1", dim)
        g2, b2 = src.norm(f"{name}.norm2", dim)
        self.n1, self.n2 = (ctx.dev(g1.float()), ctx.dev(b1.float())), (ctx.dev(g2.float()), ctx.dev(b2.float()))
        ws_, bs_ = [], []
        for n in ("query", "key", "value"):
            w, b = src.linear(f"{name}.attention.{n}", dim, dim)
            ws_.append(w); bs_.append(b)
        self.qkv = _Linear(ctx, torch.cat(ws_, 0), torch.cat(bs_, 0))
        pw_, pb_ = src.conv(f"{name}.attention.pool_layer", 1, dim, pool[0], 1.0)          # depth-wise [dim, 1, 4, 4]
        kk = pool[0] * pool[1]
        wd = torch.zeros(dim, kk, dim)                                                       # dense form over tap-major columns k * dim + c
        ar = torch.arange(dim)
        wd[ar, :, ar] = pw_.reshape(dim, kk)
        self.poolw = _Linear(ctx, wd.reshape(dim, kk * dim), pb_)
        w, b = src.linear(f"{name}.attention.proj", dim, dim)
        self.proj = _Linear(ctx, w, b)
        ch = hidden // 49
        self.ch = ch
        w1, b1_ = src.linear(f"{name}.mlp.fc1.0", dim, hidden)                              # rows c * 49 + k  ->  tap-major k * ch + c
        perm = torch.arange(hidden).view(ch, 49).t().reshape(-1)
        self.fc1 = _Linear(ctx, w1[perm], b1_[perm])
        w2, b2_ = src.linear(f"{name}.mlp.fc2.1", hidden, dim)
        self.fc2 = _Linear(ctx, w2[:, perm], b2_)

    def __call__(self, x, t, fh, fw, h, w, tabs, masked_win, zero7, zero4):
        """x fp32 [t*fh*fw, dim] (residual stream) -> fp32, same shape."""
        ctx, dt, dim = self.ctx, self.ctx.dt, self.dim
        wh, ww = self.ws
        L = wh * ww
        y = hip.layernorm(dt, x, self.n1[0], self.n1[1])
        xp = hip.gather_rows(y, tabs["pad_idx"])                                   # [t*nh*nw, dim], zero rows in the padding (as F.pad after norm1)
        qkv = self.qkv(xp)
        nh, nw, ph, pw = tabs["nh"], tabs["nw"], tabs["ph"], tabs["pw"]
        pcol, _, _ = hip.deform_im2col(dt, xp, B=t, H=nh, W=nw, kh=self.pool[0], kw=self.pool[1], stride=self.pool[0], pad=0, dil=1, deform_groups=1,
                                       offset=zero4)
        table = torch.cat([qkv, self.qkv(self.poolw(pcol))], 0)                      # token rows, then pooled-token rows (their q part is unused)
        outs = []
        heads, D = self.n_head, dim // self.n_head
        if masked_win["m_q"] is not None:
            nm, Lk = masked_win["nm"], masked_win["Lk"]
            qg, kg = hip.gather_rows(table, masked_win["m_q"]), hip.gather_rows(table, masked_win["m_k"])
            o = torch.empty((nm * t * L, dim), dtype=ctx.h16, device=x.device)
            hip.attention(dt, qg, kg, kg, o, B=nm, heads=heads, Nq=t * L, Nkv=Lk, D=D, q_bs=t * L * 3 * dim, k_bs=Lk * 3 * dim, v_bs=Lk * 3 * dim,
                          o_bs=t * L * dim, q_rs=3 * dim, k_rs=3 * dim, v_rs=3 * dim, o_rs=dim, k_off=dim, v_off=2 * dim)
            outs.append(o)
        if masked_win["u_q"] is not None:
            nu = masked_win["nu"]
            qg = hip.gather_rows(table, masked_win["u_q"])
            o = torch.empty((nu * t * L, dim), dtype=ctx.h16, device=x.device)
            hip.attention(dt, qg, qg, qg, o, B=nu * t, heads=heads, Nq=L, Nkv=L, D=D, q_bs=L * 3 * dim, k_bs=L * 3 * dim, v_bs=L * 3 * dim,
                          o_bs=L * dim, q_rs=3 * dim, k_rs=3 * dim, v_rs=3 * dim, o_rs=dim, k_off=dim, v_off=2 * dim)
            outs.append(o)
        att = hip.gather_rows(torch.cat(outs, 0) if len(outs) > 1 else outs[0], masked_win["inv"])
        x = self.proj(att, res0=x, out_dtype=torch.float32)
        y = hip.layernorm(dt, x, self.n2[0], self.n2[1])
        hmid = self.fc1(y)                                                            # [n, 49 * ch] tap-major
        grid = hip.fold_patches(dt, hmid, t, fh, fw, self.ch, h, w, K7, S3, P3, normalise=True, gelu=True, out_dtype=ctx.h16)
        hcol, _, _ = hip.deform_im2col(dt, grid, B=t, H=h, W=w, kh=K7, kw=K7, stride=S3, pad=P3, dil=1, deform_groups=1, offset=zero7)
        return self.fc2(hcol, res0=x, out_dtype=torch.float32)


class InpaintGenerator:
    def __init__(self, ctx, depths=8, t_dilation=2, name="gen", C=128, hidden=512, n_head=4, ws=(5, 9), pool=(4, 4), deform_groups=16):
        assert depths % t_dilation == 0
        self.ctx, self.C, self.hidden, self.depths, self.t_dilation, self.ws, self.pool = ctx, C, hidden, depths, t_dilation, ws, pool
        self.encoder = _Encoder(ctx, f"{name}.encoder")
        self.prop = _FeaturePropagation(ctx, f"{name}.feat_prop_module", C, deform_groups)
        w, b = ctx.src.linear(f"{name}.ss.embedding", 49 * C, hidden)                      # columns c * 49 + k -> tap-major k * C + c
        perm = torch.arange(49 * C).view(C, 49).t().reshape(-1)
        self.ss = _Linear(ctx, w[:, perm], b)
        self.blocks = [_TransformerBlock(ctx, f"{name}.transformers.transformer.{d}", hidden, n_head, ws, pool) for d in range(depths)]
        w, b = ctx.src.linear(f"{name}.sc.embedding", hidden, 49 * C)
        self.sc = _Linear(ctx, w[perm], b[perm])
        self.sc_conv = _Conv(ctx, f"{name}.sc.bias_conv", C, C, 3)
        self.d0, self.d2 = _Conv(ctx, f"{name}.decoder.0.conv", C, 128, 3), _Conv(ctx, f"{name}.decoder.2", 128, 64, 3)
        self.d4, self.d6 = _Conv(ctx, f"{name}.decoder.4.conv", 64, 64, 3), _Conv(ctx, f"{name}.decoder.6", 64, 3, 3)
        self._tabs = {}

    def _tables(self, t, fh, fw, parity, dev):
        key = (t, fh, fw, parity)
        if key not in self._tabs:
            T_ind = list(range(parity, t, self.t_dilation))
            tb = _window_tables(t, fh, fw, self.ws, self.pool, T_ind)
            tb["pad_idx_host"] = tb["pad_idx"]
            tb["pad_idx"] = torch.from_numpy(tb["pad_idx"]).to(dev)
            self._tabs[key] = tb
        return self._tabs[key]

    def forward(self, frames_u8, flows_f, flows_b, masks_in_u8, masks_up_u8, l_t):
        """frames u8 [t,H,W,3] (local frames first, then reference frames; already holding the propagated content), completed flows fp32
        [l_t-1,H,W,2] at full resolution, masks u8 [t,H,W] (dilated input masks / masks still open after the image propagation).
        Returns the raw prediction of the local frames, fp32 [l_t*H*W, 3] (before tanh)."""
        ctx, dt, C = self.ctx, self.ctx.dt, self.C
        t, H, W, _ = frames_u8.shape
        if H % 4 or W % 4:
            raise RuntimeError(f"InpaintGenerator: H={H}, W={W} must be multiples of 4")
        dev = frames_u8.device
        enc, h, w = self.encoder(hip.gen_input(frames_u8.contiguous(), masks_in_u8.contiguous(), masks_up_u8.contiguous()), t, H, W)
        hw = h * w
        # 1/4-resolution masks (nearest = every 4th pixel) and the window classification: integer work on the caller's masks, on the host
        m_in = (masks_in_u8.cpu().numpy() > 0)[:, ::4, ::4]
        m_up = (masks_up_u8[:l_t].cpu().numpy() > 0)[:, ::4, ::4]
        masks2 = torch.from_numpy(np.stack([m_in[:l_t], m_up], -1).astype(np.float32).reshape(l_t * hw, 2)).to(dev)
        if l_t > 1:
            local = self.prop(enc[:l_t * hw], hip.flow_down4(flows_f.contiguous()), hip.flow_down4(flows_b.contiguous()), masks2, l_t, h, w)
            enc = torch.cat([local, enc[l_t * hw:]], 0)
        fh, fw = token_grid(h, w)
        zero7 = torch.zeros((t * fh * fw, 2 * 49), dtype=torch.float32, device=dev)
        col, _, _ = hip.deform_im2col(dt, enc, B=t, H=h, W=w, kh=K7, kw=K7, stride=S3, pad=P3, dil=1, deform_groups=1, offset=zero7)
        tok = self.ss(col, out_dtype=torch.float32)
        # hole mask of the local frames on the token grid (max pool 7/3/3), then per window
        mpad = np.pad(m_in[:l_t], ((0, 0), (P3, P3), (P3, P3)))
        mp = np.lib.stride_tricks.sliding_window_view(mpad, (K7, K7), axis=(1, 2))[:, ::S3, ::S3][:, :fh, :fw].any((3, 4))
        wh, ww = self.ws
        per_parity = {}
        for parity in range(self.t_dilation):
            tb = self._tables(t, fh, fw, parity, dev)
            mgrid = np.zeros((l_t, tb["nh"], tb["nw"]), bool)
            mgrid[:, :fh, :fw] = mp
            wmask = mgrid.reshape(l_t, tb["n_wh"], wh, tb["n_ww"], ww).any((0, 2, 4)).reshape(-1)
            mi, ui = np.nonzero(wmask)[0], np.nonzero(~wmask)[0]
            L = wh * ww
            slot = np.empty(tb["nwin"], np.int64)            # position of every window's output block in cat([masked outputs, clean outputs])
            slot[mi] = np.arange(len(mi))
            slot[ui] = len(mi) + np.arange(len(ui))
            inv = tb["inv"].astype(np.int64)
            win_of = inv // (t * L)
            inv_cat = (slot[win_of] * (t * L) + inv % (t * L)).astype(np.int32)
            to_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a.reshape(-1))).to(dev)
            per_parity[parity] = dict(
                nm=len(mi), nu=len(ui), Lk=tb["k_idx"].shape[1],
                m_q=to_dev(tb["q_idx"][mi]) if len(mi) else None, m_k=to_dev(tb["k_idx"][mi]) if len(mi) else None,
                u_q=to_dev(tb["q_idx"][ui]) if len(ui) else None, inv=to_dev(inv_cat), tabs=tb)
        tb0 = per_parity[0]["tabs"]
        zero4 = torch.zeros((t * tb0["ph"] * tb0["pw"], 2 * self.pool[0] * self.pool[1]), dtype=torch.float32, device=dev)
        for d, blk in enumerate(self.blocks):
            pp = per_parity[d % self.t_dilation]
            tok = blk(tok, t, fh, fw, h, w, pp["tabs"], pp, zero7, zero4)
        comp = hip.fold_patches(dt, self.sc(tok), t, fh, fw, C, h, w, K7, S3, P3, out_dtype=ctx.h16)
        enc, _, _ = self.sc_conv(comp, t, h, w, res0=enc, out_dtype=torch.float32)
        x = enc[:l_t * hw]
        x, _, _ = self.d0(hip.upsample2x_bilinear(dt, x, l_t, h, w), l_t, 2 * h, 2 * w, act=0.2)
        x, _, _ = self.d2(x, l_t, 2 * h, 2 * w, act=0.2)
        x, _, _ = self.d4(hip.upsample2x_bilinear(dt, x, l_t, 2 * h, 2 * w), l_t, H, W, act=0.2)
        x, _, _ = self.d6(x, l_t, H, W, out_dtype=torch.float32)
        return x


def get_ref_index(neighbor_ids, length, ref_stride=10, mid=0, ref_num=-1):
    """ProPainter inference script: reference frames of one neighbour window."""
    if ref_num == -1:
        return [i for i in range(0, length, ref_stride) if i not in neighbor_ids]
    out = []
    for i in range(max(0, mid - ref_stride * (ref_num // 2)), min(length, mid + ref_stride * (ref_num // 2)), ref_stride):
        if i not in neighbor_ids:
            if len(out) > ref_num:
                break
            out.append(i)
    return out


def window_schedule(T, neighbor_length=10, ref_stride=10, subvideo_length=80):
    ns = neighbor_length // 2
    ref_num = subvideo_length // ref_stride if T > subvideo_length else -1
    out = []
    for f in range(0, T, ns):
        nb = list(range(max(0, f - ns), min(T, f + ns + 1)))
        out.append((nb, get_ref_index(nb, T, ref_stride, f, ref_num)))
    return out


def inpaint_clip(gen, updated_u8, ori_u8, flows_f, flows_b, masks_u8, updated_masks_u8, neighbor_length=10, ref_stride=10, subvideo_length=80):
    """Sliding-window inference of the generator over a clip (all tensors on the device): updated_u8 [T,H,W,3] frames after the image
    propagation, ori_u8 the originals, completed flows fp32 [T-1,H,W,2], masks u8 [T,H,W].  Returns u8 [T,H,W,3]."""
    T, H, W, _ = updated_u8.shape
    dev = updated_u8.device
    acc = torch.zeros((T, H, W, 3), dtype=torch.float32, device=dev)
    seen = [False] * T
    for nb, ref in window_schedule(T, neighbor_length, ref_stride, subvideo_length):
        ids = torch.tensor(nb + ref, dtype=torch.long, device=dev)
        fl = torch.tensor(nb[:-1], dtype=torch.long, device=dev)
        raw = gen.forward(updated_u8[ids], flows_f[fl], flows_b[fl], masks_u8[ids], updated_masks_u8[ids], len(nb)).reshape(len(nb), H * W, 3)
        for i, idx in enumerate(nb):
            hip.gen_compose(raw[i], ori_u8[idx], masks_u8[idx], acc[idx], not seen[idx])
            seen[idx] = True
    return acc.to(torch.uint8)
