"""ORACLE (test infrastructure): the DiffuEraser.forward pipeline + the in-tree orchestration, fp32 on CPU.

Follows reference diffuerase.py:20-114 (in-tree: dilation, model call, resize-back + feather composite) and
SURVEY.md section 3.3 / App. D for the third-party `DiffuEraser.forward` it calls at diffuerase.py:62-67
(PARITY UNPINNED for the third-party part -- see oracle/model_ref.py header).

Chunking is build-defined (the reference lists chunk+overlap+blend as a TODO, README.md:76): independent
`chunk`-frame clips starting at s_i = min((chunk-overlap)*i, T-chunk), blended by a sequential linear cross-fade
in fp32 pixel space, in chunk order.
"""
import numpy as np
import torch

from . import imageops_ref as I
from . import model_ref as M


def model_size(H0, W0, max_img_size):
    """Inference size: max side <= max_img_size, both sides floored to multiples of 8."""
    Hs, Ws = H0, W0
    if max(H0, W0) > max_img_size:
        s = max_img_size / float(max(H0, W0))
        Hs, Ws = int(round(H0 * s)), int(round(W0 * s))
    return max(8, Hs - Hs % 8), max(8, Ws - Ws % 8)


def chunk_plan(T, chunk, overlap):
    if T <= chunk:
        return [(0, T)]
    starts, i = [], 0
    while True:
        s = min((chunk - overlap) * i, T - chunk)
        starts.append(s)
        if s + chunk >= T:
            break
        i += 1
    return [(s, s + chunk) for s in starts]


def blend_weights(plan):
    """Per chunk: fp32 weight of the NEW chunk for each of its frames under the sequential cross-fade
    (frames beyond the running coverage get 1).  w_new(k) = (k+1)/(O+1) over the O overlapping frames."""
    out, covered = [], 0
    for (s, e) in plan:
        w = np.ones(e - s, np.float32)
        O = max(0, covered - s)
        for k in range(O):
            w[k] = np.float32(k + 1) / np.float32(O + 1)
        out.append(w)
        covered = max(covered, e)
    return out


def chunk_noise(seed, index, shape):
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed) * 1000003 + int(index))
    return torch.randn(shape, generator=g, dtype=torch.float32)


def to_model_tensor(frames_u8):
    x = torch.from_numpy(np.stack(frames_u8)).permute(0, 3, 1, 2).float()
    return x / 127.5 - 1.0


def denoise_chunk(P, img, m, prior, noise, steps, ucfg, vcfg, scheduler="ddim", tcd_noise=None, trace=None):
    """One clip: img/prior [F,3,H,W] in [-1,1], m [F,1,H,W] in {0,1} -> decoded [F,3,H,W] in [0,1]."""
    ac = M.alphas_cumprod()
    text = M.text_states(P, ucfg)
    prior_lat = M.vae_encode(P, prior, vcfg)
    cond_lat = M.vae_encode(P, img * (1.0 - m), vcfg)
    h, w = prior_lat.shape[-2:]
    m_lat = torch.nn.functional.interpolate(m, size=(h, w), mode="nearest")
    ts = M.ddim_timesteps(steps) if scheduler == "ddim" else M.tcd_timesteps(steps)
    lat = M.add_noise(prior_lat, noise, ts[0], ac)
    if trace is not None:
        trace.update(prior_lat=prior_lat, cond_lat=cond_lat, m_lat=m_lat, lat0=lat.clone())
    for i, t in enumerate(ts):
        brush = M.brushnet_forward(P, torch.cat([lat, cond_lat, m_lat], 1), t, text, ucfg)
        eps = M.unet_forward(P, lat, t, text, ucfg, brush)
        if trace is not None and i == 0:
            trace.update(eps0=eps.clone())
        if scheduler == "ddim":
            lat = M.ddim_step(lat, eps, t, steps, ac)
        else:
            lat = M.tcd_step(lat, eps, t, ts[i + 1] if i + 1 < len(ts) else None, ac,
                             tcd_noise[i] if (tcd_noise is not None and i < len(tcd_noise)) else torch.zeros_like(lat))
        if trace is not None:
            trace.setdefault("lat_steps", []).append(lat.clone())
    if trace is not None:
        trace.update(lat_final=lat.clone())
    dec = M.vae_decode(P, lat, vcfg)
    if trace is not None:
        trace.update(decoded_raw=dec.clone())      # before the clamp to [0,1]: the tests also bound the error of the unclamped decode
    return (dec / 2 + 0.5).clamp(0, 1)


def diffueraser_forward(frames, masks2d, priori, max_img_size=960, steps=50, chunk=32, overlap=8, seed=42,
                        weight_seed=0, ucfg=None, vcfg=None, scheduler="ddim", P=None, return_float=False):
    """Restatement of `DiffuEraser.forward(frames, masks, priori, max_img_size=, ...)` (call site reference
    diffuerase.py:62-67): list of uint8 RGB frames at the inference size."""
    from videovanish_amd.config import UNetConfig, VAEConfig
    ucfg = ucfg or UNetConfig()
    vcfg = vcfg or VAEConfig()
    P = P or M.Params(weight_seed)
    T = len(frames)
    H0, W0 = frames[0].shape[:2]
    H, W = model_size(H0, W0, max_img_size)
    fr = [I.resize_bilinear_u8(f, W, H) for f in frames]
    pr = [I.resize_bilinear_u8(f, W, H) for f in priori]
    mk = [I.resize_nearest_u8(np.where(m > 0, 255, 0).astype(np.uint8), W, H) for m in masks2d]
    plan = chunk_plan(T, chunk, overlap)
    wts = blend_weights(plan)
    acc = torch.zeros(T, 3, H, W)
    with torch.no_grad():
        for ci, (s, e) in enumerate(plan):
            img = to_model_tensor(fr[s:e])
            prior = to_model_tensor(pr[s:e])
            m = torch.from_numpy(np.stack(mk[s:e]) > 0).float()[:, None]
            f = 2 ** (len(vcfg.block_out) - 1)                # VAE down-factor (8 for the SD VAE)
            noise = chunk_noise(seed, ci, (e - s, 4, H // f, W // f))
            tcd_noise = None
            if scheduler == "tcd":
                tcd_noise = [chunk_noise(seed + 104729 * (i + 1), ci, (e - s, 4, H // f, W // f)) for i in range(steps - 1)]
            dec = denoise_chunk(P, img, m, prior, noise, steps, ucfg, vcfg, scheduler, tcd_noise=tcd_noise)
            w = torch.from_numpy(wts[ci])[:, None, None, None]
            acc[s:e] = (acc[s:e] * (1.0 - w)) + (dec * w)        # sequential cross-fade, fp32, chunk order
    out = []
    pix = acc.permute(0, 2, 3, 1).contiguous().numpy()
    if return_float:
        return pix
    for t in range(T):
        out.append(I.blur_compose(pix[t], fr[t], mk[t]))
    return out


def run_infill_on_frames(frames_rgb, mask_frames, mask_dilation_iter=8, propainer_frames=None, max_img_size=960,
                         keep_unmasked_original=True, feather_px=3, compat_reference_early_return=False, **kw):
    """Restatement of reference diffuerase.py:20-114 with the model call replaced by diffueraser_forward above.
    `propainer_frames` must be supplied (the flow-propagation prior has its own oracle in flowprop_ref.py)."""
    H0, W0 = frames_rgb[0].shape[:2]
    dil = I.collapse_and_dilate(mask_frames, mask_dilation_iter)               # :27-31
    out = diffueraser_forward(frames_rgb, dil, propainer_frames, max_img_size=max_img_size, **kw)   # :62-67
    for i, f in enumerate(out):                                                 # :70-112
        if f.shape[0] != H0 or f.shape[1] != W0:
            out[i] = I.resize_bilinear_u8(f, W0, H0)
        if keep_unmasked_original:
            out[i] = I.composite(out[i], frames_rgb[i], I.feather_alpha(dil[i], feather_px))
        if compat_reference_early_return:                                       # :114 (return inside the loop)
            break
    return out


# ---- reference temporal windowing ([UNVERIFIED-3P]: restated from the public DiffuEraser pipeline) -------
def reference_contexts(n, nframes=22, overlap=4):
    """get_frames_context_swap restated: (contexts, contexts_swap) as [a, b) ranges; even steps use the first list, odd the second."""
    npc = min(nframes, n)
    if n <= npc:
        return [(0, n)], [(0, n)]
    ctx, swap = [], []
    k = 0
    for k in range(0, n - npc, npc - overlap):
        ctx.append((k, k + npc))
    if k + npc < n:
        ctx.append((n - npc, n))
    swap.append((0, npc))
    for k in range(npc // 2, n - npc, npc - overlap):
        swap.append((k, k + npc))
    if k + npc < n:
        swap.append((n - npc, n))
    return ctx, swap


def _denoise_windows(P, lat, cond, m_lat, steps, ucfg, nframes, overlap, scheduler="ddim", tcd_noise=None):
    ac = M.alphas_cumprod()
    text = M.text_states(P, ucfg)
    n = lat.shape[0]
    ctxs, swap = reference_contexts(n, nframes, overlap)
    ts = M.ddim_timesteps(steps) if scheduler == "ddim" else M.tcd_timesteps(steps)
    for i, t in enumerate(ts):
        value, count = torch.zeros_like(lat), torch.zeros(n, 1, 1, 1)
        for (a, b) in (ctxs if i % 2 == 0 else swap):
            brush = M.brushnet_forward(P, torch.cat([lat[a:b], cond[a:b], m_lat[a:b]], 1), t, text, ucfg)
            value[a:b] += M.unet_forward(P, lat[a:b], t, text, ucfg, brush)
            count[a:b] += 1
        if scheduler == "ddim":
            lat = M.ddim_step(lat, value / count, t, steps, ac)
        else:
            lat = M.tcd_step(lat, value / count, t, ts[i + 1] if i + 1 < len(ts) else None, ac,
                             tcd_noise[i] if (tcd_noise is not None and i < len(tcd_noise)) else torch.zeros_like(lat))
    return lat


def diffueraser_forward_reference_windows(frames, masks2d, priori, max_img_size=960, steps=50, seed=42, weight_seed=0, ucfg=None, vcfg=None,
                                          nframes=22, overlap=4, P=None, return_float=False, scheduler="ddim", trace=None):
    """DiffuEraser.forward with the third-party pipeline's own temporal scheme: windows of `nframes` shifted by half a window on
    odd steps, noise prediction averaged over the covering windows, key-frame pre-inference when T > 2 * nframes."""
    from videovanish_amd.config import UNetConfig, VAEConfig
    ucfg = ucfg or UNetConfig()
    vcfg = vcfg or VAEConfig()
    P = P or M.Params(weight_seed)
    T = len(frames)
    H0, W0 = frames[0].shape[:2]
    H, W = model_size(H0, W0, max_img_size)
    fr = [I.resize_bilinear_u8(f, W, H) for f in frames]
    pr = [I.resize_bilinear_u8(f, W, H) for f in priori]
    mk = [I.resize_nearest_u8(np.where(m > 0, 255, 0).astype(np.uint8), W, H) for m in masks2d]
    fr_orig, mk_orig = list(fr), list(mk)
    f = 2 ** (len(vcfg.block_out) - 1)
    h, w = H // f, W // f
    ac = M.alphas_cumprod()
    t0 = (M.ddim_timesteps(steps) if scheduler == "ddim" else M.tcd_timesteps(steps))[0]
    reps = (T + nframes - 1) // nframes
    z_pre = z_all = None
    if scheduler == "tcd":      # re-noising tensors: seeded per step, tiled over the windows like the initial noise
        z_pre = [chunk_noise(seed + 104729 * (i + 1), 0, (nframes, 4, h, w)) for i in range(steps - 1)]
        z_all = [z.repeat(reps, 1, 1, 1)[:T] for z in z_pre]
    with torch.no_grad():
        mt = lambda lst: torch.from_numpy(np.stack(lst) > 0).float()[:, None]
        prior_lat = M.vae_encode(P, to_model_tensor(pr), vcfg)
        cond_lat = M.vae_encode(P, to_model_tensor(fr) * (1.0 - mt(mk)), vcfg)
        noise_pre = chunk_noise(seed, 0, (nframes, 4, h, w))
        if T > 2 * nframes:
            step = T / nframes
            idx = [int(i * step) for i in range(nframes)][:nframes]
            m_lat = torch.nn.functional.interpolate(mt([mk[i] for i in idx]), size=(h, w), mode="nearest")
            lat_pre = M.add_noise(prior_lat[idx], noise_pre, t0, ac)
            out_pre = _denoise_windows(P, lat_pre, cond_lat[idx], m_lat, steps, ucfg, nframes, overlap, scheduler, z_pre)
            pix = (M.vae_decode(P, out_pre, vcfg) / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1).numpy()
            keys = []
            for j, i in enumerate(idx):
                key = I.blur_compose(pix[j], fr[i], np.full((H, W), 255, np.uint8))
                keys.append(key)
                fr[i], mk[i] = key, np.zeros((H, W), np.uint8)
                prior_lat[i] = out_pre[j]
                cond_lat[i] = M.vae_encode(P, to_model_tensor([key]), vcfg)[0]
            if trace is not None:      # the quantised key frames (the path's one discontinuity): tests hand them to the HIP path to separate float error from flips
                trace.update(key_u8=np.stack(keys), key_idx=list(idx))
        noise = noise_pre.repeat(reps, 1, 1, 1)[:T]
        m_lat = torch.nn.functional.interpolate(mt(mk), size=(h, w), mode="nearest")
        lat = _denoise_windows(P, M.add_noise(prior_lat, noise, t0, ac), cond_lat, m_lat, steps, ucfg, nframes, overlap, scheduler, z_all)
        pix = (M.vae_decode(P, lat, vcfg) / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1).contiguous().numpy()
    if return_float:
        return pix
    return [I.blur_compose(pix[t], fr_orig[t], mk_orig[t]) for t in range(T)]
