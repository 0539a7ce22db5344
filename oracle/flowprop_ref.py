"""ORACLE (test infrastructure): RAFT optical flow + flow-guided frame propagation prior, fp32 torch on the CPU.

This restates the part of the third-party `Propainter.forward` (call site reference diffuerase.py:52-57) that the
north star names: the RAFT all-pairs correlation volume + pyramid lookup + recurrent update, the bilinear warp and the
forward/backward-consistent propagation (App. D.7-D.8).  PARITY UNPINNED: `propainter` is
un-vendored third-party code absent from /root/reference (install_videovanish.sh:78); RAFT follows the published
princeton-vl architecture ("raft-things", 20 iterations).  The flow-completion network, deformable feature propagation
and the sparse transformer of the full ProPainter are row n1 (not built): holes the propagation cannot reach are
filled with the frame's mean unmasked colour.

Weights: videovanish_amd.weights.SyntheticWeights (name-seeded).  BatchNorm of the context encoder is in eval mode:
a per-channel affine (seeded running statistics), which the product folds into the conv weights.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from .model_ref import Params, conv2d

RADIUS, LEVELS, ITERS = 4, 4, 20


def _conv(P, name, x, cout, kh, kw, stride=1, gain=1.0):
    cin = x.shape[1]
    key = name
    if key not in P.cache:
        w = P.src.normal(name + ".weight", (cout, cin, kh, kw), std=gain / float(cin * kh * kw) ** 0.5)
        b = P.src.normal(name + ".bias", (cout,), std=0.02)
        P.cache[key] = (w, b)
    w, b = P.cache[key]
    return F.conv2d(x, w, b, stride=stride, padding=(kh // 2, kw // 2))


def bn_affine(P, name, c):
    """eval-mode BatchNorm as y = x*a + b with seeded gamma/beta/running stats."""
    if name not in P.cache:
        g = P.src.normal(name + ".weight", (c,), 0.1, 1.0)
        be = P.src.normal(name + ".bias", (c,), 0.1)
        mu = P.src.normal(name + ".running_mean", (c,), 0.1)
        var = P.src.normal(name + ".running_var", (c,), 0.1, 1.0).abs() + 0.5
        a = g / torch.sqrt(var + 1e-5)
        P.cache[name] = (a, be - mu * a)
    return P.cache[name]


def _norm(P, name, x, kind):
    if kind == "instance":
        return F.instance_norm(x, eps=1e-5)
    a, b = bn_affine(P, name, x.shape[1])
    return x * a[None, :, None, None] + b[None, :, None, None]


def _resblock(P, name, x, planes, kind, stride):
    y = F.relu(_norm(P, name + ".norm1", _conv(P, name + ".conv1", x, planes, 3, 3, stride), kind))
    y = F.relu(_norm(P, name + ".norm2", _conv(P, name + ".conv2", y, planes, 3, 3), kind))
    if stride != 1:
        x = _norm(P, name + ".norm3", _conv(P, name + ".downsample.0", x, planes, 1, 1, stride), kind)
    return F.relu(x + y)


def encoder(P, name, img, out_dim, kind):
    """BasicEncoder: img [B,3,H,W] in [-1,1] -> [B,out_dim,H/8,W/8]."""
    x = F.relu(_norm(P, name + ".norm1", _conv(P, name + ".conv1", img, 64, 7, 7, 2), kind))
    for i, (planes, stride) in enumerate([(64, 1), (96, 2), (128, 2)]):
        x = _resblock(P, f"{name}.layer{i + 1}.0", x, planes, kind, stride)
        x = _resblock(P, f"{name}.layer{i + 1}.1", x, planes, kind, 1)
    return _conv(P, name + ".conv2", x, out_dim, 1, 1)


def corr_pyramid(f1, f2):
    """f1,f2 [C,h,w] -> list of LEVELS tensors [h*w, h/2^l, w/2^l] (all-pairs correlation / sqrt(C), avg-pooled)."""
    C, h, w = f1.shape
    corr = (f1.reshape(C, -1).t() @ f2.reshape(C, -1)) / math.sqrt(C)
    corr = corr.reshape(h * w, 1, h, w)
    pyr = [corr[:, 0]]
    for _ in range(LEVELS - 1):
        corr = F.avg_pool2d(corr, 2, stride=2)
        pyr.append(corr[:, 0])
    return pyr


def bilinear_zero(plane, x, y):
    """plane [..., Hp, Wp]; sample at float pixel coords (x, y) (align_corners=True), zeros outside.  Fixed op order
    (the HIP kernel replicates it): v = (1-wx)(1-wy) v00 + wx(1-wy) v01 + (1-wx)wy v10 + wx wy v11."""
    Hp, Wp = plane.shape[-2:]
    x0, y0 = torch.floor(x), torch.floor(y)
    wx, wy = x - x0, y - y0
    x0, y0 = x0.long(), y0.long()

    def tap(yy, xx):
        ok = (xx >= 0) & (xx < Wp) & (yy >= 0) & (yy < Hp)
        v = plane[..., yy.clamp(0, Hp - 1), xx.clamp(0, Wp - 1)]
        return torch.where(ok, v, torch.zeros_like(v))

    return ((1 - wx) * (1 - wy)) * tap(y0, x0) + (wx * (1 - wy)) * tap(y0, x0 + 1) + ((1 - wx) * wy) * tap(y0 + 1, x0) + (wx * wy) * tap(y0 + 1, x0 + 1)


def corr_lookup(pyr, coords):
    """coords [h*w, 2] (x,y) in level-0 pixels -> [h*w, LEVELS*81]; channel (l, i, j) samples (x/2^l + i-4, y/2^l + j-4)."""
    N = coords.shape[0]
    d = torch.arange(-RADIUS, RADIUS + 1, dtype=torch.float32)
    out = []
    idx = torch.arange(N)
    for l, c in enumerate(pyr):
        cx = coords[:, 0] / 2 ** l
        cy = coords[:, 1] / 2 ** l
        xs = (cx[:, None, None] + d[None, :, None]).expand(N, 9, 9)
        ys = (cy[:, None, None] + d[None, None, :]).expand(N, 9, 9)
        Hp, Wp = c.shape[-2:]
        x0, y0 = torch.floor(xs), torch.floor(ys)
        wx, wy = xs - x0, ys - y0
        x0, y0 = x0.long(), y0.long()

        def tap(yy, xx):
            ok = (xx >= 0) & (xx < Wp) & (yy >= 0) & (yy < Hp)
            v = c[idx[:, None, None], yy.clamp(0, Hp - 1), xx.clamp(0, Wp - 1)]
            return torch.where(ok, v, torch.zeros_like(v))

        v = ((1 - wx) * (1 - wy)) * tap(y0, x0) + (wx * (1 - wy)) * tap(y0, x0 + 1) + ((1 - wx) * wy) * tap(y0 + 1, x0) + (wx * wy) * tap(y0 + 1, x0 + 1)
        out.append(v.reshape(N, 81))
    return torch.cat(out, 1)


def update_block(P, net, inp, corr, flow):
    """BasicUpdateBlock (hidden 128): all tensors [1,C,h,w]."""
    pre = "raft.update"
    cor = F.relu(_conv(P, pre + ".encoder.convc1", corr, 256, 1, 1))
    cor = F.relu(_conv(P, pre + ".encoder.convc2", cor, 192, 3, 3))
    flo = F.relu(_conv(P, pre + ".encoder.convf1", flow, 128, 7, 7))
    flo = F.relu(_conv(P, pre + ".encoder.convf2", flo, 64, 3, 3))
    mot = F.relu(_conv(P, pre + ".encoder.conv", torch.cat([cor, flo], 1), 126, 3, 3))
    x = torch.cat([inp, mot, flow], 1)                   # 128 + 126 + 2
    for tag, kh, kw in (("1", 1, 5), ("2", 5, 1)):       # SepConvGRU: horizontal then vertical
        hx = torch.cat([net, x], 1)
        z = torch.sigmoid(_conv(P, f"{pre}.gru.convz{tag}", hx, 128, kh, kw))
        r = torch.sigmoid(_conv(P, f"{pre}.gru.convr{tag}", hx, 128, kh, kw))
        q = torch.tanh(_conv(P, f"{pre}.gru.convq{tag}", torch.cat([r * net, x], 1), 128, kh, kw))
        net = (1 - z) * net + z * q
    dflow = _conv(P, pre + ".flow_head.conv2", F.relu(_conv(P, pre + ".flow_head.conv1", net, 256, 3, 3)), 2, 3, 3)
    mask = 0.25 * _conv(P, pre + ".mask.2", F.relu(_conv(P, pre + ".mask.0", net, 256, 3, 3)), 576, 1, 1)
    return net, mask, dflow


def convex_upsample(flow, mask):
    """flow [1,2,h,w], mask [1,576,h,w] -> [1,2,8h,8w]."""
    _, _, h, w = flow.shape
    m = torch.softmax(mask.view(1, 1, 9, 8, 8, h, w), dim=2)
    up = F.unfold(8 * flow, [3, 3], padding=1).view(1, 2, 9, 1, 1, h, w)
    up = torch.sum(m * up, dim=2).permute(0, 1, 4, 2, 5, 3)
    return up.reshape(1, 2, 8 * h, 8 * w)


def raft_flow(P, img1_u8, img2_u8, iters=ITERS, trace=None):
    """img [H,W,3] uint8 (H,W multiples of 8) -> flow img1->img2 [2,H,W] fp32 (x,y displacement in pixels)."""
    def prep(a):
        return (2.0 * (torch.from_numpy(a).float() / 255.0) - 1.0).permute(2, 0, 1)[None]
    i1, i2 = prep(img1_u8), prep(img2_u8)
    f1 = encoder(P, "raft.fnet", i1, 256, "instance")[0]
    f2 = encoder(P, "raft.fnet", i2, 256, "instance")[0]
    cn = encoder(P, "raft.cnet", i1, 256, "batch")
    net, inp = torch.tanh(cn[:, :128]), F.relu(cn[:, 128:])
    C, h, w = f1.shape
    pyr = corr_pyramid(f1, f2)
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
    coords0 = torch.stack([xs, ys], 0)[None]
    coords1 = coords0.clone()
    if trace is not None:
        trace.update(f1=f1, f2=f2, net0=net, inp=inp, corr0=pyr[0], corr3=pyr[3])
    mask = None
    for it in range(iters):
        corr = corr_lookup(pyr, coords1[0].reshape(2, -1).t()).t().reshape(1, LEVELS * 81, h, w)
        flow = coords1 - coords0
        net, mask, dflow = update_block(P, net, inp, corr, flow)
        coords1 = coords1 + dflow
        if trace is not None and it == 0:
            trace.update(lookup0=corr, dflow0=dflow, net1=net)
    flow_lo = coords1 - coords0
    if trace is not None:
        trace.update(flow_lo=flow_lo)
    return convex_upsample(flow_lo, mask)[0]


# ----------------------------------------------------------------------------------------------------------------
# flow-guided propagation (App. D.8)
# ----------------------------------------------------------------------------------------------------------------
def warp_frame(img, flow):
    """img [C,H,W] fp32, flow [2,H,W]: out(x,y) = img(x + fx, y + fy), bilinear, zeros outside (align_corners=True)."""
    C, H, W = img.shape
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    return bilinear_zero(img, xs + flow[0], ys + flow[1])


def fb_consistency(f_ab, f_ba):
    """valid(p) = |f_ab + warp(f_ba, f_ab)|^2 < 0.01 (|f_ab|^2 + |warp f_ba|^2) + 0.5."""
    wb = warp_frame(f_ba, f_ab)
    d = f_ab + wb
    lhs = d[0] * d[0] + d[1] * d[1]
    rhs = 0.01 * ((f_ab[0] * f_ab[0] + f_ab[1] * f_ab[1]) + (wb[0] * wb[0] + wb[1] * wb[1])) + 0.5
    return lhs < rhs


def propagate(frames_u8, masks_u8, flows_fw, flows_bw):
    """frames [T,H,W,3] u8, masks [T,H,W] u8 (non-zero = hole); flows_fw[t] = flow t->t+1, flows_bw[t] = flow t+1->t,
    each [2,H,W].  Two sweeps; a hole pixel of frame t is filled from the neighbour's current content warped by the flow
    towards the neighbour when the flow is forward/backward consistent and the (nearest) source pixel is known."""
    T, H, W, _ = frames_u8.shape
    img = torch.from_numpy(frames_u8).float().permute(0, 3, 1, 2)
    hole = torch.from_numpy(masks_u8 > 0)

    def sweep(order, flow_to_nb, flow_from_nb):
        cur = img.clone()
        known = ~hole.clone()
        filled = torch.zeros_like(hole)
        for t, nb, k in order:
            f = flow_to_nb[k]                       # flow from frame t to its neighbour nb
            valid = fb_consistency(f, flow_from_nb[k])
            w = warp_frame(cur[nb], f)
            ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
            sx = torch.floor(xs + f[0] + 0.5).long()
            sy = torch.floor(ys + f[1] + 0.5).long()
            inb = (sx >= 0) & (sx < W) & (sy >= 0) & (sy < H)
            src_known = known[nb][sy.clamp(0, H - 1), sx.clamp(0, W - 1)] & inb
            take = (~known[t]) & valid & src_known
            cur[t] = torch.where(take[None], w, cur[t])
            known[t] = known[t] | take
            filled[t] = take
        return cur, filled

    fwd_order = [(t, t - 1, t - 1) for t in range(1, T)]            # frame t from t-1: flow t->t-1 = flows_bw[t-1]
    bwd_order = [(t, t + 1, t) for t in range(T - 2, -1, -1)]       # frame t from t+1: flow t->t+1 = flows_fw[t]
    a, fa = sweep(fwd_order, flows_bw, flows_fw)
    b, fb = sweep(bwd_order, flows_fw, flows_bw)
    out = img.clone()
    both = fa & fb
    out = torch.where(both[:, None], (a + b) * 0.5, out)
    out = torch.where((fa & ~fb)[:, None], a, out)
    out = torch.where((fb & ~fa)[:, None], b, out)
    rest = hole & ~(fa | fb)
    res = []
    for t in range(T):
        o = out[t]
        if rest[t].any():
            keep = (~hole[t]).numpy()
            fr = frames_u8[t].astype(np.int64)
            cnt = int(keep.sum())
            sums = fr[keep].sum(0) if cnt else fr.reshape(-1, 3).sum(0)
            mean = torch.tensor((sums / float(cnt if cnt else H * W)).astype(np.float32))      # exact integer sums / count
            o = torch.where(rest[t][None], mean[:, None, None].expand_as(o), o)
        res.append(torch.clamp(torch.floor(o + 0.5), 0, 255).to(torch.uint8).permute(1, 2, 0).numpy())
    return res, (fa | fb).numpy()


def subvideo_ranges(T, subvideo_length, pad_len=10):
    """Sub-video schedule of ProPainter's image propagation ([UNVERIFIED-3P], public inference script): sub-videos of
    min(100, subvideo_length) frames, each propagated together with pad_len frames of context on both sides; only the inner
    frames are kept.  Returns [(s_f, e_f, keep_lo, keep_hi)] in frame indices."""
    L = min(100, int(subvideo_length))
    if L <= 0 or T <= L:
        return [(0, T, 0, T)]
    return [(max(0, f - pad_len), min(T, f + L + pad_len), f, min(T, f + L)) for f in range(0, T, L)]


def flow_propagation_prior(frames, masks2d, weight_seed=0, iters=ITERS, P=None, subvideo_length=0):
    """Restatement of the prior the reference obtains at diffuerase.py:52-57 (flow part only).  frames: list of (H,W,3)
    u8 with H,W multiples of 8; masks2d: list of (H,W) u8.  subvideo_length > 0: propagate per sub-video (reference passes 50)."""
    P = P or Params(weight_seed)
    T = len(frames)
    with torch.no_grad():
        fw = [raft_flow(P, frames[t], frames[t + 1], iters) for t in range(T - 1)]
        bw = [raft_flow(P, frames[t + 1], frames[t], iters) for t in range(T - 1)]
        out = [None] * T
        for (s, e, lo, hi) in subvideo_ranges(T, subvideo_length):
            sub, _ = propagate(np.stack(frames[s:e]), np.stack(masks2d[s:e]), fw[s:e - 1], bw[s:e - 1])
            for t in range(lo, hi):
                out[t] = sub[t - s]
    return out
