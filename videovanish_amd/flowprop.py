"""Flow-guided propagation prior on HIP kernels: RAFT flows in both temporal
directions, forward/backward consistency, bilinear warp, two propagation sweeps, mean-colour fill of what is left.
This is the flow part of the third-party `Propainter.forward` the reference calls at diffuerase.py:52-57; the
flow-completion network / deformable propagation / sparse transformer of the full ProPainter are row n1."""
import numpy as np
import torch

from . import hip
from .nn import Ctx, normalize_device
from .raft import ITERS, RAFT

_cache = {}


def _model(device, dtype, weight_seed):
    key = (str(device) if device is not None else "cuda:%d" % torch.cuda.current_device(), dtype, weight_seed)
    if key not in _cache:
        ctx = Ctx(normalize_device(device), dtype, weight_seed)
        _cache[key] = (ctx, RAFT(ctx))
    return _cache[key]


def _flow_net(ctx):
    key = ("fc", id(ctx))
    if key not in _cache:
        from .flowcomplete import FlowCompleteNet
        _cache[key] = FlowCompleteNet(ctx)
    return _cache[key]


PAIRS = 8      # pairs per stacked group


def flows_for_clip(raft, frames_u8, iters=ITERS):
    """frames u8 [T,H,W,3] (device, H,W % 8 == 0) -> (fw, bw): lists of fp32 [H,W,2]; fw[t] = flow t->t+1, bw[t] = flow t+1->t."""
    T = frames_u8.shape[0]
    f, c, h, w = raft.features(frames_u8)
    # forward pairs (t -> t+1) and backward pairs (t+1 -> t) go through the update block in stacked groups: one launch per
    # kernel and iteration for the whole group (the all-pairs pyramid is 1.1 GB of fp32 per pair at 720p -> groups of <= PAIRS)
    fw, bw = [], []
    for t0 in range(0, T - 1, PAIRS):
        t1 = min(t0 + PAIRS, T - 1)
        fw.extend(raft.flow_batch(f[t0:t1], f[t0 + 1:t1 + 1], c[t0:t1], h, w, iters))
        bw.extend(raft.flow_batch(f[t0 + 1:t1 + 1], f[t0:t1], c[t0 + 1:t1 + 1], h, w, iters))
    return fw, bw


def propagate(frames_u8, masks_u8, fw, bw):
    """frames u8 [T,H,W,3], masks u8 [T,H,W] (device) + flows -> (prior u8 [T,H,W,3], filled u8 [T,H,W])."""
    T, H, W, _ = frames_u8.shape
    dev = frames_u8.device
    img = hip.u8_to_f32(frames_u8)
    hole = masks_u8

    def sweep(order, flow_to_nb, flow_from_nb):
        cur = img.clone()
        known = hip.u8_is_zero(hole)
        filled = torch.zeros((T, H, W), dtype=torch.uint8, device=dev)
        for t, nb, k in order:
            valid = hip.fb_valid(flow_to_nb[k], flow_from_nb[k])
            hip.prop_fill(cur[t], cur[nb], known[t], known[nb], valid, flow_to_nb[k], filled[t])
        return cur, filled

    a, fa = sweep([(t, t - 1, t - 1) for t in range(1, T)], bw, fw)
    b, fb = sweep([(t, t + 1, t) for t in range(T - 2, -1, -1)], fw, bw)
    out = torch.empty((T, H, W, 3), dtype=torch.uint8, device=dev)
    filled = torch.empty((T, H, W), dtype=torch.uint8, device=dev)
    sums = torch.stack([hip.masked_sum_u8(frames_u8[t], hole[t]) for t in range(T)]).cpu().numpy()     # [T,4] exact integer sums
    for t in range(T):
        cnt = int(sums[t, 3])
        if cnt:
            mean = (sums[t, :3] / float(cnt)).astype(np.float32)
        else:
            mean = (frames_u8[t].cpu().numpy().astype(np.int64).reshape(-1, 3).sum(0) / float(H * W)).astype(np.float32)
        o, f = hip.prop_combine(img[t], a[t], b[t], fa[t], fb[t], hole[t], torch.from_numpy(mean).to(dev))
        out[t], filled[t] = o, f
    return out, filled


def subvideo_ranges(T, subvideo_length, pad_len=5):
    """Sub-video schedule of ProPainter's image propagation (third-party, restated from the public inference script): sub-videos
    of min(100, subvideo_length) frames propagated with pad_len frames of context on both sides, inner frames kept."""
    L = min(100, int(subvideo_length))
    if L <= 0 or T <= L:
        return [(0, T, 0, T)]
    return [(max(0, f - pad_len), min(T, f + L + pad_len), f, min(T, f + L)) for f in range(0, T, L)]


def flow_propagation_prior(frames, masks, device=None, progress=None, dtype="fp16", weight_seed=0, iters=ITERS, subvideo_length=0,
                           flow_completion=False):
    """list of (H0,W0,3) u8 + list of (H0,W0) u8 masks -> list of (H0,W0,3) u8 prior frames.
    subvideo_length > 0: the propagation runs per sub-video as the reference's ProPainter call asks (diffuerase.py:55).
    flow_completion: complete the RAFT flows inside the holes with the recurrent flow-completion network first (flowcomplete.py)."""
    ctx, raft = _model(device, dtype, weight_seed)
    dev = ctx.device
    H0, W0 = frames[0].shape[:2]
    H, W = max(64, H0 // 8 * 8), max(64, W0 // 8 * 8)
    fr = torch.from_numpy(np.stack(frames)).to(dev)
    mk = torch.from_numpy(np.stack([m if m.ndim == 2 else np.any(m > 0, axis=2).astype(np.uint8) * 255 for m in masks])).to(dev)
    if (H, W) != (H0, W0):
        fr = hip.resize_u8(fr.contiguous(), H, W, mode="bilinear")
        mk = hip.resize_u8(mk.contiguous(), H, W, mode="nearest")
    if progress is not None:
        progress(25, "running flow prior (RAFT)")
    fw, bw = flows_for_clip(raft, fr.contiguous(), iters)
    T = fr.shape[0]
    if flow_completion and T > 1:
        if progress is not None:
            progress(35, "running flow prior (flow completion)")
        cf, cb = _flow_net(ctx).complete_flows(torch.stack(fw), torch.stack(bw), mk.contiguous())
        fw, bw = list(cf), list(cb)
    if progress is not None:
        progress(40, "running flow prior (propagation)")
    ranges = subvideo_ranges(T, subvideo_length)
    if len(ranges) == 1:
        out, _ = propagate(fr.contiguous(), mk.contiguous(), fw, bw)
    else:
        out = torch.empty_like(fr)
        for (s, e, lo, hi) in ranges:
            sub, _ = propagate(fr[s:e].contiguous(), mk[s:e].contiguous(), fw[s:e - 1], bw[s:e - 1])
            out[lo:hi] = sub[lo - s: hi - s]
    if (H, W) != (H0, W0):
        out = hip.resize_u8(out.contiguous(), H0, W0, mode="bilinear")
    return list(out.cpu().numpy())
