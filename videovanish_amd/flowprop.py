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


def _model(device, dtype, weight_seed, weights=None):
    key = (str(device) if device is not None else "cuda:%d" % torch.cuda.current_device(), dtype, weight_seed, id(weights) if weights is not None else None)
    if key not in _cache:
        ctx = Ctx(normalize_device(device), dtype, weight_seed, weights=weights)
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


def subvideo_ranges(T, subvideo_length, pad_len=10):
    """Sub-video schedule of ProPainter's image propagation (third-party, restated from the public inference script): sub-videos
    of min(100, subvideo_length) frames propagated with pad_len frames of context on both sides, inner frames kept.
    pad_len = 10 for the image propagation [UNVERIFIED-3P: recalled from the public script; 5 is what its flow-completion sub-videos use]."""
    L = min(100, int(subvideo_length))
    if L <= 0 or T <= L:
        return [(0, T, 0, T)]
    return [(max(0, f - pad_len), min(T, f + L + pad_len), f, min(T, f + L)) for f in range(0, T, L)]


def _generator(ctx):
    key = ("gen", id(ctx))
    if key not in _cache:
        from .inpaintgen import InpaintGenerator
        _cache[key] = InpaintGenerator(ctx)
    return _cache[key]


def flow_propagation_prior(frames, masks, device=None, progress=None, dtype="fp16", weight_seed=0, iters=ITERS, subvideo_length=0,
                           flow_completion=False, generator=False, ref_stride=10, neighbor_length=10, weights=None):
    """list of (H0,W0,3) u8 + list of (H0,W0) u8 masks -> list of (H0,W0,3) u8 prior frames.
    subvideo_length > 0: the propagation runs per sub-video as the reference's ProPainter call asks (diffuerase.py:55).
    flow_completion: complete the RAFT flows inside the holes with the recurrent flow-completion network first (flowcomplete.py).
    generator: run ProPainter's inpainting generator (inpaintgen.py) over the propagated frames in sliding windows of `neighbor_length`
    frames with every `ref_stride`-th frame as reference (the last stage of the real ProPainter)."""
    ctx, raft = _model(device, dtype, weight_seed, weights)
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
        out, filled = propagate(fr.contiguous(), mk.contiguous(), fw, bw)
    else:
        out, filled = torch.empty_like(fr), torch.empty_like(mk)
        for (s, e, lo, hi) in ranges:
            sub, fsub = propagate(fr[s:e].contiguous(), mk[s:e].contiguous(), fw[s:e - 1], bw[s:e - 1])
            out[lo:hi], filled[lo:hi] = sub[lo - s: hi - s], fsub[lo - s: hi - s]
    if generator and T > 1:
        if progress is not None:
            progress(45, "running flow prior (inpainting generator)")
        from .inpaintgen import inpaint_clip
        # frames after the image propagation: propagated content inside the holes, mid-grey (0 in the network's [-1, 1] range) where nothing
        # arrived; the "updated" masks mark what is still open
        open_mask = ((mk > 0) & (filled == 0)).to(torch.uint8) * 255
        updated = torch.where(open_mask[..., None] > 0, torch.full_like(out, 127), out)
        out = inpaint_clip(_generator(ctx), updated.contiguous(), fr.contiguous(), torch.stack(fw), torch.stack(bw), mk.contiguous(), open_mask.contiguous(),
                           neighbor_length=neighbor_length, ref_stride=ref_stride, subvideo_length=max(int(subvideo_length), 1))
    if (H, W) != (H0, W0):
        out = hip.resize_u8(out.contiguous(), H0, W0, mode="bilinear")
    return list(out.cpu().numpy())
