"""ORACLE (test infrastructure): ProPainter's inpainting generator (encoder, flow-guided deformable feature propagation, soft split,
temporal sparse-window transformer, soft composition, decoder), fp32 torch on the CPU.

SURVEY.md row n1, the last of the three learned networks of the full ProPainter prior (third-party `Propainter.forward`, call site
reference diffuerase.py:52-57; the knobs the reference passes -- ref_stride = 10, neighbor_length = 10, diffuerase.py:53-54 -- select this
network's reference and neighbour frames).  PARITY UNPINNED: `propainter` is un-vendored, un-pinned third-party code absent from
/root/reference (install_videovanish.sh:78), its weights (`ruffy369/propainter`) are unreachable; this file restates the published
architecture (ProPainter model/propainter.py, model/modules/sparse_transformer.py) with seeded synthetic weights:

  encoder      9 convs (5 -> 64 s2, 64, 128 s2, 256, 384, then grouped convs 640 -> 512 (g2), 768 -> 384 (g4), 640 -> 256 (g8), 512 -> 128) with
               the E2FGVI group-wise skip concatenation of the 256-channel feature, LeakyReLU(0.2)                         1/4 resolution
  feat_prop    bidirectional propagation of the local frames' features: flow_warp + forward/backward consistency, DeformableAlignment
               (DCNv2, 16 groups, offset = 3 tanh(.) + flow, cond = [current | warped | flow | valid | masks]), 2-conv backbone, 2-conv fuse
  soft split   unfold 7x7 stride 3 padding 3 + Linear(49*128 -> 512)                                                       tokens f_h x f_w
  transformer  8 x [LayerNorm, sparse window attention (windows 5x9, 4 heads; masked windows also see the rolled neighbour-window keys and 4x4
               pooled global keys of every t_dilation-th frame; clean windows attend inside the window, frame by frame), LayerNorm,
               fusion feed-forward (Linear 512 -> 1960, fold / normalise / unfold over the 7x7 patches, GELU, Linear 1960 -> 512)]
  soft comp    Linear(512 -> 49*128) + fold + Conv2d(128,128,3)
  decoder      [bilinear x2 + Conv(128,128)] LReLU, Conv(128,64) LReLU, [x2 + Conv(64,64)] LReLU, Conv(64,3), tanh         full resolution
"""
import math

import torch
import torch.nn.functional as F

from .deform_ref import bilinear_zero, deform_conv2d

LR = 0.2
GROUPS = [1, 2, 4, 8, 1]


def _conv(P, name, x, cout, k=3, stride=1, pad=1, groups=1, gain=1.0):
    cin = x.shape[1]
    w, b = P.conv(name, cin // groups, cout, k, gain)         # [cout, cin/groups, k, k]
    return F.conv2d(x, w, b, stride=stride, padding=pad, groups=groups)


def _linear(P, name, x, cout):
    w, b = P.linear(name, x.shape[-1], cout)
    return F.linear(x, w, b)


def _ln(P, name, x):
    g, b = P.norm(name, x.shape[-1])
    return F.layer_norm(x, (x.shape[-1],), g, b, 1e-5)


def _deconv(P, name, x, cout):
    x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
    return _conv(P, name + ".conv", x, cout)


# ------------------------------------------------------------------------------------------------------------------------ encoder
def encoder(P, name, x):
    """x [BT, 5, H, W] (masked frame | mask_in | mask_updated) -> [BT, 128, H/4, W/4]."""
    spec = [(64, 2, 1), (64, 1, 1), (128, 2, 1), (256, 1, 1), (384, 1, 1), (512, 1, 2), (384, 1, 4), (256, 1, 8), (128, 1, 1)]
    bt = x.shape[0]
    out, x0 = x, None
    for li, (co, s, g) in enumerate(spec):
        i = 2 * li                                   # index in the reference's ModuleList (conv, act, conv, act, ...)
        if i == 8:
            x0 = out
        if i > 8:
            gg = GROUPS[(i - 8) // 2]
            h, w = x0.shape[-2:]
            out = torch.cat([x0.view(bt, gg, -1, h, w), out.view(bt, gg, -1, h, w)], 2).view(bt, -1, h, w)
        out = F.leaky_relu(_conv(P, f"{name}.layers.{i}", out, co, 3, s, 1, g), LR)
    return out


# ------------------------------------------------------------------------------------------------------------ feature propagation
def flow_warp(x, flow):
    """x [B, C, H, W], flow [B, 2, H, W] (dx, dy): grid_sample(bilinear, zeros, align_corners=True) at p + flow."""
    B, C, H, W = x.shape
    ys, xs = torch.meshgrid(torch.arange(H, dtype=x.dtype), torch.arange(W, dtype=x.dtype), indexing="ij")
    return bilinear_zero(x, ys[None, None] + flow[:, 1:2], xs[None, None] + flow[:, 0:1])


def fb_consistency(flow_fw, flow_bw, a1=0.01, a2=0.5):
    wb = flow_warp(flow_bw, flow_fw)
    d = flow_fw + wb
    mag = (flow_fw ** 2).sum(1, keepdim=True) + (wb ** 2).sum(1, keepdim=True)
    return ((d ** 2).sum(1, keepdim=True) < a1 * mag + a2).to(flow_fw.dtype)


def deformable_alignment(P, name, x, cond, flow, C, deform_groups=16, max_residue=3.0):
    h = cond
    for i, co in enumerate([C, C, C, 27 * deform_groups]):
        w, b = P.conv(f"{name}.conv_offset.{2 * i}", h.shape[1], co, 3, 0.1 if i == 3 else 1.0)
        h = F.conv2d(h, w, b, padding=1)
        if i < 3:
            h = F.leaky_relu(h, 0.1)
    o1, o2, m = torch.chunk(h, 3, dim=1)
    offset = max_residue * torch.tanh(torch.cat([o1, o2], 1))
    offset = offset + flow.flip(1).repeat(1, offset.shape[1] // 2, 1, 1)
    w, b = P.conv(name, C, C, 3)
    return deform_conv2d(x, offset, w, b, 1, 1, 1, torch.sigmoid(m))


def feature_propagation(P, name, x, flows_f, flows_b, mask, C=128, deform_groups=16):
    """x [B, T, C, H, W]; flows_f / flows_b [B, T-1, 2, H, W] (t -> t+1 / t+1 -> t); mask [B, T, 2, H, W] -> fused features [B, T, C, H, W]."""
    B, T, _, H, W = x.shape
    feats = {"input": [x[:, i] for i in range(T)]}
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
            mcur = mask[:, idx]
            if i == 0:
                prop = cur
            else:
                fp, fc = f_prop[:, flow_idx[i]], f_check[:, flow_idx[i]]
                valid = fb_consistency(fp, fc)
                warped = flow_warp(prop, fp)
                cond = torch.cat([cur, warped, fp, valid, mcur], 1)
                prop = deformable_alignment(P, f"{name}.deform_align.{mod}", prop, cond, fp, C, deform_groups)
            feat = torch.cat([cur, prop, mcur], 1)
            h = F.leaky_relu(_conv(P, f"{name}.backbone.{mod}.0", feat, C), LR)
            prop = prop + _conv(P, f"{name}.backbone.{mod}.2", h, C)
            feats[mod].append(prop)
        if mod == "backward_1":
            feats[mod] = feats[mod][::-1]
    ob = torch.stack(feats["backward_1"], 1).reshape(-1, C, H, W)
    of = torch.stack(feats["forward_1"], 1).reshape(-1, C, H, W)
    h = F.leaky_relu(_conv(P, f"{name}.fuse.0", torch.cat([ob, of, mask.reshape(-1, 2, H, W)], 1), C), LR)
    out = _conv(P, f"{name}.fuse.2", h, C) + x.reshape(-1, C, H, W)
    return out.view(B, T, C, H, W)


# ------------------------------------------------------------------------------------------------------- soft split / composition
K7, S3, P3 = (7, 7), (3, 3), (3, 3)


def token_grid(h, w):
    return (h + 2 * P3[0] - (K7[0] - 1) - 1) // S3[0] + 1, (w + 2 * P3[1] - (K7[1] - 1) - 1) // S3[1] + 1


def soft_split(P, name, x, b, hidden=512):
    """x [BT, C, h, w] -> tokens [b, t, f_h, f_w, hidden]."""
    fh, fw = token_grid(*x.shape[-2:])
    feat = F.unfold(x, K7, stride=S3, padding=P3).permute(0, 2, 1)        # [BT, n, C*49] (channel-major patches)
    feat = _linear(P, f"{name}.embedding", feat, hidden)
    return feat.view(b, -1, fh, fw, hidden)


def soft_comp(P, name, x, t, out_size, C=128):
    b = x.shape[0]
    feat = _linear(P, f"{name}.embedding", x.reshape(b, -1, x.shape[-1]), 49 * C)
    feat = feat.view(b * t, -1, 49 * C).permute(0, 2, 1)
    feat = F.fold(feat, out_size, K7, stride=S3, padding=P3)
    return _conv(P, f"{name}.bias_conv", feat, C)


# ------------------------------------------------------------------------------------------------------------------- transformer
def window_partition(x, ws, n_head):
    B, T, H, W, C = x.shape
    x = x.view(B, T, H // ws[0], ws[0], W // ws[1], ws[1], n_head, C // n_head)
    return x.permute(0, 2, 4, 6, 1, 3, 5, 7).contiguous()                 # [B, n_wh, n_ww, head, T, w_h, w_w, c_head]


def rolled_valid_index(ws):
    e = tuple((i + 1) // 2 for i in ws)
    m_tl = torch.ones(ws); m_tl[:-e[0], :-e[1]] = 0
    m_tr = torch.ones(ws); m_tr[:-e[0], e[1]:] = 0
    m_bl = torch.ones(ws); m_bl[e[0]:, :-e[1]] = 0
    m_br = torch.ones(ws); m_br[e[0]:, e[1]:] = 0
    return torch.stack((m_tl, m_tr, m_bl, m_br), 0).flatten().nonzero(as_tuple=False).view(-1), e


def sparse_window_attention(P, name, x, mask, T_ind, n_head=4, ws=(5, 9), pool=(4, 4)):
    """x [b, t, h, w, c] tokens; mask [b, l_t, h, w, 1] (pooled hole mask of the local frames); T_ind: frames whose keys the masked
    windows see."""
    b, t, h, w, c = x.shape
    wh, ww = ws
    ch = c // n_head
    n_wh, n_ww = math.ceil(h / wh), math.ceil(w / ww)
    nh, nw = n_wh * wh, n_ww * ww
    if nh > h or nw > w:
        x = F.pad(x, (0, 0, 0, nw - w, 0, nh - h, 0, 0))
        mask = F.pad(mask, (0, 0, 0, nw - w, 0, nh - h, 0, 0))
    q, k, v = _linear(P, f"{name}.query", x, c), _linear(P, f"{name}.key", x, c), _linear(P, f"{name}.value", x, c)
    nwin = n_wh * n_ww
    part = lambda a: window_partition(a.contiguous(), ws, n_head).view(b, nwin, n_head, t, wh * ww, ch)
    win_q, win_k, win_v = part(q), part(k), part(v)
    valid, e = rolled_valid_index(ws)
    rk, rv = [], []
    for sh in ((-e[0], -e[1]), (-e[0], e[1]), (e[0], -e[1]), (e[0], e[1])):
        rk.append(part(torch.roll(k, shifts=sh, dims=(2, 3))))
        rv.append(part(torch.roll(v, shifts=sh, dims=(2, 3))))
    win_k = torch.cat([win_k, torch.cat(rk, 4)[:, :, :, :, valid]], 4)
    win_v = torch.cat([win_v, torch.cat(rv, 4)[:, :, :, :, valid]], 4)
    # pooled (global) tokens: depth-wise 4x4 mean with a learnable kernel (initialised to the mean)
    pw, pb = P.conv(f"{name}.pool_layer", 1, c, pool[0], 1.0)               # depth-wise [c, 1, 4, 4]
    px = F.conv2d(x.reshape(b * t, nh, nw, c).permute(0, 3, 1, 2), pw, pb, stride=pool, groups=c)
    ph, pww = px.shape[-2:]
    px = px.permute(0, 2, 3, 1).view(b, t, ph, pww, c)

    def pooled(lin):
        pk = _linear(P, f"{name}.{lin}", px, c).unsqueeze(1).repeat(1, nwin, 1, 1, 1, 1)
        pk = pk.view(b, nwin, t, ph, pww, n_head, ch).permute(0, 1, 5, 2, 3, 4, 6)
        return pk.contiguous().view(b, nwin, n_head, t, ph * pww, ch)

    win_k = torch.cat([win_k, pooled("key")], 4)
    win_v = torch.cat([win_v, pooled("value")], 4)
    out = torch.zeros_like(win_q)
    l_t = mask.shape[1]
    mwin = F.max_pool2d(mask.reshape(b * l_t, 1, nh, nw), ws, ws).view(b, l_t, nwin).sum(1)      # > 0: the window touches a hole
    scale = 1.0 / math.sqrt(ch)
    for i in range(b):
        mi = mwin[i].nonzero(as_tuple=False).view(-1)
        if len(mi) > 0:
            qt = win_q[i, mi].reshape(len(mi), n_head, t * wh * ww, ch)
            kt = win_k[i, mi][:, :, T_ind].reshape(len(mi), n_head, -1, ch)
            vt = win_v[i, mi][:, :, T_ind].reshape(len(mi), n_head, -1, ch)
            att = F.softmax(qt @ kt.transpose(-2, -1) * scale, -1)
            out[i, mi] = (att @ vt).view(-1, n_head, t, wh * ww, ch)
        ui = (mwin[i] == 0).nonzero(as_tuple=False).view(-1)
        if len(ui) > 0:
            qs, ks, vs = win_q[i, ui], win_k[i, ui, :, :, :wh * ww], win_v[i, ui, :, :, :wh * ww]
            att = F.softmax(qs @ ks.transpose(-2, -1) * scale, -1)
            out[i, ui] = att @ vs
    out = out.view(b, n_wh, n_ww, n_head, t, wh, ww, ch).permute(0, 4, 1, 5, 2, 6, 3, 7).contiguous().view(b, t, nh, nw, c)
    out = out[:, :, :h, :w]
    return _linear(P, f"{name}.proj", out, c)


def fusion_feed_forward(P, name, x, out_size, hidden=1960):
    """x [b, n, c]: Linear -> fold the 49-patch hidden vectors onto the feature grid, normalise by the overlap count, unfold -> GELU -> Linear."""
    nvec = token_grid(*out_size)[0] * token_grid(*out_size)[1]
    h = _linear(P, f"{name}.fc1.0", x, hidden)
    b, n, c = h.shape
    ones = h.new_ones(b, n, 49).view(-1, nvec, 49).permute(0, 2, 1)
    norm = F.fold(ones, out_size, K7, stride=S3, padding=P3)
    h = F.fold(h.view(-1, nvec, c).permute(0, 2, 1), out_size, K7, stride=S3, padding=P3)
    h = F.unfold(h / norm, K7, stride=S3, padding=P3).permute(0, 2, 1).contiguous().view(b, n, c)
    return _linear(P, f"{name}.fc2.1", F.gelu(h), x.shape[-1])


def transformer(P, name, x, fold_size, l_mask, depths=8, t_dilation=2, n_head=4):
    """x [b, t, f_h, f_w, c]."""
    T = x.shape[1]
    t_inds = [torch.arange(i, T, t_dilation) for i in range(t_dilation)] * (depths // t_dilation)
    for d in range(depths):
        blk = f"{name}.transformer.{d}"
        att = sparse_window_attention(P, f"{blk}.attention", _ln(P, f"{blk}.norm1", x), l_mask, t_inds[d], n_head)
        x = x + att
        b, t, h, w, c = x.shape
        y = fusion_feed_forward(P, f"{blk}.mlp", _ln(P, f"{blk}.norm2", x).view(b, t * h * w, c), fold_size)
        x = x + y.view(b, t, h, w, c)
    return x


# --------------------------------------------------------------------------------------------------------------------- generator
def generator(P, masked_frames, flows_f, flows_b, masks_in, masks_updated, l_t, depths=8, t_dilation=2, name="gen"):
    """masked_frames [b, t, 3, H, W] in [-1, 1] (local frames first, then reference frames); flows [b, l_t-1, 2, H, W] (completed, full
    resolution); masks [b, t, 1, H, W] in {0, 1}.  Returns the local frames' prediction [b, l_t, 3, H, W] (tanh range)."""
    b, t, _, H, W = masked_frames.shape
    C = 128
    enc = encoder(P, f"{name}.encoder", torch.cat([masked_frames.reshape(b * t, 3, H, W), masks_in.reshape(b * t, 1, H, W),
                                                  masks_updated.reshape(b * t, 1, H, W)], 1))
    h, w = enc.shape[-2:]
    enc = enc.view(b, t, C, h, w)
    local, ref = enc[:, :l_t], enc[:, l_t:]
    ds = lambda f: F.interpolate(f.reshape(-1, 2, H, W), scale_factor=0.25, mode="bilinear", align_corners=False).view(b, l_t - 1, 2, h, w) / 4.0
    dm = lambda m, n: F.interpolate(m.reshape(-1, 1, H, W), scale_factor=0.25, mode="nearest").view(b, n, 1, h, w)
    m_in = dm(masks_in, t)
    m_in_l, m_up_l = m_in[:, :l_t], dm(masks_updated[:, :l_t], l_t)
    mask_pool = F.max_pool2d(m_in_l.reshape(-1, 1, h, w), K7, S3, P3)
    mask_pool = mask_pool.view(b, l_t, 1, *mask_pool.shape[-2:]).permute(0, 1, 3, 4, 2).contiguous()
    local = feature_propagation(P, f"{name}.feat_prop_module", local, ds(flows_f), ds(flows_b), torch.cat([m_in_l, m_up_l], 2), C)
    enc = torch.cat([local, ref], 1)
    tok = soft_split(P, f"{name}.ss", enc.reshape(-1, C, h, w), b)
    tok = transformer(P, f"{name}.transformers", tok, (h, w), mask_pool, depths, t_dilation)
    comp = soft_comp(P, f"{name}.sc", tok, t, (h, w), C).view(b, t, C, h, w)
    enc = enc + comp
    x = enc[:, :l_t].reshape(-1, C, h, w)
    x = F.leaky_relu(_deconv(P, f"{name}.decoder.0", x, 128), LR)
    x = F.leaky_relu(_conv(P, f"{name}.decoder.2", x, 64), LR)
    x = F.leaky_relu(_deconv(P, f"{name}.decoder.4", x, 64), LR)
    x = _conv(P, f"{name}.decoder.6", x, 3)
    return torch.tanh(x).view(b, l_t, 3, H, W)


def get_ref_index(neighbor_ids, length, ref_stride=10, mid=0, ref_num=-1):
    """reference frames of one neighbour window (ProPainter inference script): every ref_stride-th frame outside the window; for clips longer
    than subvideo_length only ref_num of them around the window centre."""
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
    """[(neighbor_ids, ref_ids)] of ProPainter's sliding-window inference loop."""
    ns = neighbor_length // 2
    ref_num = subvideo_length // ref_stride if T > subvideo_length else -1
    out = []
    for f in range(0, T, ns):
        nb = list(range(max(0, f - ns), min(T, f + ns + 1)))
        out.append((nb, get_ref_index(nb, T, ref_stride, f, ref_num)))
    return out


def inpaint_clip(P, updated_u8, ori_u8, flows_f, flows_b, masks_u8, updated_masks_u8, neighbor_length=10, ref_stride=10, subvideo_length=80,
                 depths=8, t_dilation=2):
    """The generator over a whole clip: updated_u8 [T,H,W,3] (frames after the image propagation), ori_u8 the original frames, flows
    [T-1,2,H,W] completed, masks [T,H,W] u8.  Returns u8 [T,H,W,3]: prediction inside the (dilated) masks, original outside; a frame visited
    by two windows is the truncated mean of both visits (the reference's uint8 arithmetic)."""
    T, H, W, _ = updated_u8.shape
    fr = torch.from_numpy(updated_u8).float().permute(0, 3, 1, 2) / 127.5 - 1.0
    mi = torch.from_numpy(masks_u8 > 0).float()[:, None]
    mu = torch.from_numpy(updated_masks_u8 > 0).float()[:, None]
    comp = [None] * T
    for nb, ref in window_schedule(T, neighbor_length, ref_stride, subvideo_length):
        ids = nb + ref
        with torch.no_grad():
            pred = generator(P, fr[ids][None], flows_f[nb[:-1]][None], flows_b[nb[:-1]][None], mi[ids][None], mu[ids][None], len(nb), depths, t_dilation)[0]
        pred = ((pred + 1) / 2).permute(0, 2, 3, 1).numpy() * 255
        for i, idx in enumerate(nb):
            bm = (masks_u8[idx] > 0)[..., None].astype(pred.dtype)
            img = pred[i].astype("uint8") * bm.astype("uint8") + ori_u8[idx] * (1 - bm.astype("uint8"))
            comp[idx] = img if comp[idx] is None else (comp[idx].astype("float32") * 0.5 + img.astype("float32") * 0.5).astype("uint8")
    import numpy as np
    return np.stack(comp)
