"""GPU parity at the BASELINE configurations and at the benchmarked step count.

* 50 DDIM steps (the count bench.py times) on the tiny and small configs, bf16 and fp16, error logged per step
* c1 (8 frames 256x256, 10 steps) at FULL width against the fp32 oracle
* c3 / c4 / c5 geometries at full width through the assembled pipeline (properties; the oracle cannot run these sizes)

The default precision plan (fp16 operands + split-precision VAE decoder) is asserted at the north-star bound itself: per-pixel
max-abs <= 1e-3 in [0,1], and the same bound on the UNCLAMPED decoder output per unit of its range; the other dtypes at <= 1.5 x
their measured error (profiles/r3_parity_gpu.txt)."""
import os
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from videovanish_amd.config import SMALL_UNET, SMALL_VAE, TINY_UNET, TINY_VAE, RunConfig, UNetConfig, VAEConfig

REPORT = os.environ.get("VV_PARITY_REPORT")


def _log(msg):
    print(msg)
    if REPORT:
        with open(REPORT, "a") as f:
            f.write(msg + "\n")


def _clip(T, H, W, seed=1234):
    rng = np.random.default_rng(seed)
    frames = [rng.integers(0, 256, (H, W, 3), dtype=np.uint8) for _ in range(T)]
    m2d = []
    for t in range(T):
        m = np.zeros((H, W), np.uint8)
        m[H // 4: H // 2, W // 4 + 2 * t: W // 2 + 2 * t] = 255
        m2d.append(m)
    prior = []
    for f, m in zip(frames, m2d):
        p = f.copy()
        p[m > 0] = f.reshape(-1, 3).mean(0).astype(np.uint8)
        prior.append(p)
    return frames, m2d, prior


def _one_chunk_both(ucfg, vcfg, dname, T, H, W, steps, seed=7, **run_kw):
    """One clip through oracle and HIP path with per-step latent traces.  Returns (pixel err array, per-step RELATIVE latent max-abs)."""
    from oracle import model_ref as M
    from oracle import pipeline_ref as R
    from videovanish_amd.pipeline import DiffuEraserHIP, chunk_noise
    frames, m2d, prior = _clip(T, H, W)
    f = 2 ** (len(vcfg.block_out) - 1)
    noise = chunk_noise(seed, 0, (T, 4, H // f, W // f))
    P = M.Params(0)
    tr_ref, tr = {}, {}
    with torch.no_grad():
        img = R.to_model_tensor(frames)
        pr = R.to_model_tensor(prior)
        m = torch.from_numpy(np.stack(m2d) > 0).float()[:, None]
        ref = R.denoise_chunk(P, img, m, pr, noise, steps, ucfg, vcfg, trace=tr_ref).permute(0, 2, 3, 1).numpy()
    model = DiffuEraserHIP(RunConfig(steps=steps, chunk=T, overlap=0, seed=seed, dtype=dname, unet=ucfg, vae=vcfg, **run_kw))
    dev = model.ctx.device
    dec = model.denoise_chunk(torch.from_numpy(np.stack(frames)).to(dev), torch.from_numpy(np.stack(prior)).to(dev),
                              torch.from_numpy(np.stack(m2d)).to(dev), noise.permute(0, 2, 3, 1).contiguous().to(dev), steps=steps, trace=tr)
    raw = dec[..., :3].float().cpu().numpy() / 2 + 0.5
    got = raw.clip(0, 1)
    # relative latent error per step (with random-init weights the DDIM iterate grows in magnitude, so absolute numbers mislead)
    lat_err = [float((a.cpu().permute(0, 3, 1, 2) - b).abs().max() / b.abs().max()) for a, b in zip(tr["lat_steps"], tr_ref["lat_steps"])]
    # UNCLAMPED decoder output (the clamp to [0,1] hides errors of the saturated pixels): error relative to the reference's own range
    ref_raw = tr_ref["decoded_raw"].permute(0, 2, 3, 1).numpy() / 2 + 0.5
    UNCLAMPED.clear()
    UNCLAMPED.update(abs_max=float(np.abs(raw - ref_raw).max()), ref_range=float(ref_raw.max() - ref_raw.min()),
                     saturated=float(((ref_raw <= 0) | (ref_raw >= 1)).mean()))
    return np.abs(got - ref), lat_err


def _stats(err):
    return float(err.mean()), float(np.sqrt((err.astype(np.float64) ** 2).mean()))


UNCLAMPED = {}      # filled by _one_chunk_both: max-abs error of the unclamped decode, the reference's range, the saturated fraction


# measured (profiles/r2_parity_table.txt): see the tolerances below; the error does NOT grow with the step count
@pytest.mark.parametrize("dname,precise,tol", [("bf16", False, 1.6e-2), ("fp16", False, 1.8e-3), ("fp16", True, 1.0e-3)])
@pytest.mark.parametrize("cname,ucfg,vcfg,T,H,W", [("tiny", TINY_UNET, TINY_VAE, 4, 32, 40), ("small", SMALL_UNET, SMALL_VAE, 3, 48, 64)])
def test_parity_50_steps(gpu, dname, precise, tol, cname, ucfg, vcfg, T, H, W):
    """50 DDIM steps (the count bench.py times).  fp16 + split-precision VAE decoder meets the north-star 1e-3 per-pixel bound."""
    err, lat_err = _one_chunk_both(ucfg, vcfg, dname, T, H, W, steps=50, precise_decoder=precise)
    _log(f"parity50[{cname},{dname}{',precise-decoder' if precise else ''}] pixel max_abs={err.max():.3e} mean_abs={err.mean():.3e} | latent rel. max-abs at steps 1/5/10/25/50: "
         + " ".join(f"{lat_err[i - 1]:.2e}" for i in (1, 5, 10, 25, 50)) + _unclamped_str())
    mean, rms = _stats(err)
    _log(f"parity50[{cname},{dname}{',precise-decoder' if precise else ''}] pixel rms={rms:.3e}")
    assert err.max() <= tol
    assert lat_err[-1] <= 12 * lat_err[0]                      # relative latent error grows ~5x over the 50 steps, no blow-up
    if precise:
        assert UNCLAMPED["abs_max"] <= 1.0e-3                                          # the same bound on the UNCLAMPED decode
        # the maximum is an extreme-value statistic (+-10 % with any rounding change); the quantity an optimisation must not degrade is the mean / rms:
        # <= 1.5 x the round-4 measurement (profiles/r4_parity_gpu.txt: mean 6.96e-5 tiny, 7.08e-5 small)
        assert mean <= 1.06e-4 and rms <= 1.6e-4


def _unclamped_str():
    return (f" | unclamped decode: max_abs={UNCLAMPED['abs_max']:.3e} over a range of {UNCLAMPED['ref_range']:.2f} "
            f"({100 * UNCLAMPED['saturated']:.1f} % of the reference pixels saturate)")


def test_parity_50_steps_full_width(gpu):
    """50 DDIM steps at FULL SD-1.5 / SD-VAE width (4 frames 64x64: ~6 TFLOP of oracle work), the default precision plan
    (fp16 operands + split-precision VAE decoder): the north-star bound 1e-3 per pixel, asserted as such."""
    t0 = time.time()
    err, lat_err = _one_chunk_both(UNetConfig(), VAEConfig(), "fp16", 4, 64, 64, steps=50, seed=11, precise_decoder=True)
    _log(f"parity50[full,fp16,precise-decoder] pixel max_abs={err.max():.3e} mean_abs={err.mean():.3e} | latent rel. max-abs at steps 1/5/10/25/50: "
         + " ".join(f"{lat_err[i - 1]:.2e}" for i in (1, 5, 10, 25, 50)) + _unclamped_str() + f" ({time.time() - t0:.0f} s)")
    mean, rms = _stats(err)
    _log(f"parity50[full,fp16,precise-decoder] pixel rms={rms:.3e}")
    assert err.max() <= 1.0e-3
    assert UNCLAMPED["abs_max"] <= 1.0e-3
    assert lat_err[-1] <= 12 * lat_err[0]
    assert mean <= 1.18e-4 and rms <= 1.8e-4                        # <= 1.5 x the round-4 mean (7.89e-5); rms logged since round 5


# (round 6: the -m gpu suite has a 1200 s step limit at the driver; seeds 7 and 1234 -- ~70 s each, measured every round since round 5 in
#  profiles/r*_parity_gpu.txt -- are opt-in: VV_ALL_SEEDS=1)
_MORE_SEEDS = pytest.mark.skipif(not os.environ.get("VV_ALL_SEEDS"), reason="extra noise seeds of c1: opt-in with VV_ALL_SEEDS=1 (suite time budget)")


@pytest.mark.parametrize("seed", [42, pytest.param(7, marks=_MORE_SEEDS), pytest.param(1234, marks=_MORE_SEEDS)])
def test_config_c1_full_width_vs_oracle(gpu, seed):
    """BASELINE config 1 as stated: 8 frames 256x256, 10 DDIM steps, FULL SD-1.5 / SD-VAE width, one 8-frame clip, against the
    fp32 oracle on the host cores (~45 TFLOP of CPU work: minutes).  Three noise seeds (round 5): the per-pixel maximum of 1.5 M pixels is an
    extreme-value statistic, the mean / rms asserts beside it guard the quantity a numerics change actually moves."""
    t0 = time.time()
    err, lat_err = _one_chunk_both(UNetConfig(), VAEConfig(), "fp16", 8, 256, 256, steps=10, seed=seed, precise_decoder=True)
    mean, rms = _stats(err)
    tag = "" if seed == 42 else f",seed {seed}"
    _log(f"c1_full_width[fp16,precise-decoder,10 steps{tag}] pixel max_abs={err.max():.3e} mean_abs={err.mean():.3e} rms={rms:.3e} latent rel. max-abs per step: "
         + " ".join(f"{e:.2e}" for e in lat_err) + _unclamped_str() + f" ({time.time() - t0:.0f} s)")
    assert err.max() <= 1.0e-3                                      # the north-star bound itself (measured 8.0e-4 .. 8.9e-4)
    assert UNCLAMPED["abs_max"] <= 1.0e-3
    assert mean <= 1.45e-4 and rms <= 2.0e-4                        # <= 1.5 x the round-4 measurement (mean 9.65e-5, rms ~1.3e-4)


def test_raft_20_iterations_mid_size_vs_oracle(gpu):
    """RAFT at the iteration count the product runs (20 GRU updates) against oracle/flowprop_ref.py at 184 x 320 (23 x 40 feature grid): the recurrent
    error growth is measured, not inferred from the 3-iteration 720p check.  Translating block texture, random-init weights."""
    from oracle import flowprop_ref as FP
    from oracle.model_ref import Params
    from videovanish_amd import flowprop
    H, W = 184, 320
    rng = np.random.default_rng(77)
    base = rng.integers(0, 256, (H // 8 + 1, W // 8 + 2, 3), dtype=np.uint8)
    base = np.repeat(np.repeat(base, 8, 0), 8, 1)
    f0, f1 = np.ascontiguousarray(base[:H, :W]), np.ascontiguousarray(base[:H, 3: 3 + W])
    ctx, raft = flowprop._model(None, "fp16", 0)
    errs = {}
    for iters in (3, 20):
        with torch.no_grad():
            ref = FP.raft_flow(Params(0), f0, f1, iters=iters)
        f, c, h, w = raft.features(torch.from_numpy(np.stack([f0, f1])).to(ctx.device))
        flow = raft.flow(f[0], f[1], c[0], h, w, iters=iters).cpu().permute(2, 0, 1)
        d = (flow - ref).abs()
        errs[iters] = (float(d.max() / ref.abs().max()), float(d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()))
        _log(f"raft {W}x{H} fp16 vs oracle ({iters} it.): flow rel max {errs[iters][0]:.3e} rel rms {errs[iters][1]:.3e} (|flow| max {ref.abs().max().item():.2f} px)")
    assert errs[3][0] <= 5e-2
    assert errs[20][0] <= 5e-2 and errs[20][1] <= 2e-2             # 20 recurrent updates: the error may grow, it must not take off


def _rect_masks(T, H, W):
    m2d = []
    for t in range(T):
        m = np.zeros((H, W), np.uint8)
        x0 = (W // 8 + 2 * t) % (W - W // 4)
        m[H // 3: H // 3 + H // 4, x0: x0 + W // 4] = 255
        m2d.append(m)
    return m2d


def _properties(model, T, H, W, steps=1):
    rng = np.random.default_rng(7)
    frames = [rng.integers(0, 256, (H, W, 3), dtype=np.uint8) for _ in range(T)]
    m2d = _rect_masks(T, H, W)
    a = np.stack(model.forward(frames, m2d, frames, max_img_size=max(H, W), steps=steps))
    assert a.shape == (T, H, W, 3) and a.dtype == np.uint8
    F0 = np.stack(frames)
    far = np.ones((T, H, W), bool)
    for t in range(T):
        ys, xs = np.nonzero(m2d[t])
        far[t, max(0, ys.min() - 12): ys.max() + 13, max(0, xs.min() - 12): xs.max() + 13] = False
    assert (a[far] == F0[far]).all()                     # compose keeps pixels far from the mask
    inside = np.stack(m2d) > 0
    assert (a[inside] != F0[inside]).mean() > 0.5        # the hole was repainted
    return frames, m2d, a


def test_config_c3_720p_assembled_pipeline(gpu):
    """c3 geometry: 1280x720, 32/8 chunks -- 40 frames = 2 chunks (0..32, 8..40) blended, 1 DDIM step, full width, bf16.
    Properties + bit-reproducibility + independence of the chunk that does not cover a frame."""
    from videovanish_amd.pipeline import DiffuEraserHIP, chunk_plan
    T, H, W = 40, 720, 1280
    assert chunk_plan(T, 32, 8) == [(0, 32), (8, 40)]
    model = DiffuEraserHIP(RunConfig(steps=1, chunk=32, overlap=8, seed=1, dtype="bf16"))
    frames, m2d, a = _properties(model, T, H, W)
    b = np.stack(model.forward(frames[:32], m2d[:32], frames[:32], max_img_size=1280, steps=1))
    assert np.array_equal(a[:8], b[:8])                  # frames 0..7 are covered by chunk 0 only: identical to a 32-frame run


def test_config_c4_1080p_chunk(gpu):
    """c4 geometry: one 32-frame 1920x1080 chunk (latent 135x240 -> 68x120 -> 34x60 -> 17x30: odd sizes), 1 step, full width."""
    from videovanish_amd.pipeline import DiffuEraserHIP
    model = DiffuEraserHIP(RunConfig(steps=1, chunk=32, overlap=8, seed=1, dtype="bf16"))
    _properties(model, 32, 1080, 1920)


def test_config_c5_720p_flow_prior_dilation_fp16(gpu):
    """c5: 720p + RAFT flow-propagation prior + mask dilation 8 + feather composite, fp16, through the drop-in entry point.
    The clip is a translating block texture (+2 px/frame); the RAFT flow of the first pair is checked against the oracle at the
    FULL 720p size (random-init RAFT weights: parity of the kernels, not a tracking result)."""
    import diffuerase
    from oracle import flowprop_ref as FP
    from videovanish_amd import flowprop
    T, H, W = 8, 720, 1280
    rng = np.random.default_rng(1234)
    base = rng.integers(0, 256, (H, W + 2 * T, 3), dtype=np.uint8)
    base = np.repeat(np.repeat(base[::8, ::8], 8, 0), 8, 1)[:H, :W + 2 * T]          # 8x8 blocks: trackable texture
    frames = [np.ascontiguousarray(base[:, 2 * t: 2 * t + W]) for t in range(T)]
    masks = []
    for t in range(T):
        m = np.zeros((H, W, 3), np.uint8)
        m[H // 4: H // 2, W // 4 + 2 * t: W // 2 + 2 * t] = 255
        masks.append(m)
    diffuerase.configure(RunConfig(steps=2, chunk=32, overlap=8, seed=3, dtype="fp16"))
    diffuerase.propainter = None
    progs = []
    out = diffuerase.run_infill_on_frames(frames, masks, mask_dilation_iter=8, max_img_size=1280, prog=lambda p, s, *a: progs.append((p, s)),
                                          num_inference_steps=2, scheduler="ddim")
    assert len(out) == T and all(o.shape == (H, W, 3) and o.dtype == np.uint8 for o in out)
    assert (20, "running propainter prior") in progs
    O, F0 = np.stack(out), np.stack(frames)
    assert (O[:, : H // 8] == F0[:, : H // 8]).all()                                # far from the dilated + feathered hole
    inside = np.stack([m[..., 0] for m in masks]) > 0
    assert (O[inside] != F0[inside]).mean() > 0.5
    # RAFT at the full 720p size vs the fp32 oracle: one pair, 3 update iterations (all-pairs 14400 x 14400 correlation volume)
    from oracle.model_ref import Params
    ctx, raft = flowprop._model(None, "fp16", 0)
    with torch.no_grad():
        ref = FP.raft_flow(Params(0), frames[0], frames[1], iters=3)
    f, c, h, w = raft.features(torch.from_numpy(np.stack(frames[:2])).to(ctx.device))
    flow = raft.flow(f[0], f[1], c[0], h, w, iters=3)
    e = ((flow.cpu().permute(2, 0, 1) - ref).abs().max() / ref.abs().max()).item()
    _log(f"c5 raft 1280x720 fp16 vs oracle (3 it.): flow rel max {e:.3e} (|flow| max {ref.abs().max().item():.2f} px)")
    assert e <= 5e-2
    diffuerase.configure(None)


def test_reference_windowing_vs_oracle(gpu):
    """the third-party pipeline's own temporal scheme (22-frame windows, half-window shift on odd steps, value/count
    averaging, key-frame pre-inference because T = 46 > 44) against its fp32 restatement; tiny width, 3 DDIM steps."""
    from oracle import pipeline_ref as R
    from videovanish_amd.pipeline import DiffuEraserHIP, key_frame_indices, reference_contexts
    T, H, W = 46, 16, 24
    frames, m2d, prior = _clip(T, H, W, seed=21)
    assert reference_contexts(T) == ([(0, 22), (18, 40), (24, 46)], [(0, 22), (11, 33), (24, 46)]) and len(key_frame_indices(T)) == 22
    ref = R.diffueraser_forward_reference_windows(frames, m2d, prior, steps=3, seed=7, ucfg=TINY_UNET, vcfg=TINY_VAE, return_float=True)
    model = DiffuEraserHIP(RunConfig(steps=3, seed=7, dtype="fp16", unet=TINY_UNET, vae=TINY_VAE, windowing="reference"))
    tm = {}
    got, (lo, hi) = model.forward(frames, m2d, prior, steps=3, return_float=True, scheduler=None, timings=tm)      # same contract as the chunked path
    assert (lo, hi) == (0, T) and tm["compute_s"] > 0
    err = np.abs(got - ref)
    _log(f"reference_windowing[tiny,fp16,precise,T=46] pixel max_abs={err.max():.3e} mean_abs={err.mean():.3e}")
    assert err.max() <= 6e-3          # the key frames pass through a uint8 quantisation: a 1-level flip there is 3.9e-3 on its own
    out = model.forward(frames, m2d, prior, steps=3)
    refu = R.diffueraser_forward_reference_windows(frames, m2d, prior, steps=3, seed=7, ucfg=TINY_UNET, vcfg=TINY_VAE)
    du = np.abs(np.stack(out).astype(int) - np.stack(refu).astype(int))
    assert len(out) == T and du.max() <= 2
    # the reference's own regime (round 5): the same windows under the 2-step TCD schedule of its "2-Step" checkpoint
    tr_ref, tr_got = {}, {}
    ref2 = R.diffueraser_forward_reference_windows(frames, m2d, prior, steps=2, seed=7, ucfg=TINY_UNET, vcfg=TINY_VAE, return_float=True, scheduler="tcd", trace=tr_ref)
    got2 = model.forward_reference_windows(frames, m2d, prior, steps=2, return_float=True, scheduler="tcd", trace=tr_got)
    e2 = np.abs(got2 - ref2)
    # the ONE discontinuity of this regime is the uint8 hand-off of the 22 key frames: count the levels that flipped, then repeat the run with
    # the oracle's key frames handed in -- what remains is the float error of the path itself, and THAT is held to the north-star bound
    kd = np.abs(tr_got["key_u8"].astype(int) - tr_ref["key_u8"].astype(int))
    assert tr_got["key_idx"] == list(tr_ref["key_idx"]) and kd.max() <= 1
    got3 = model.forward_reference_windows(frames, m2d, prior, steps=2, return_float=True, scheduler="tcd", key_override=tr_ref["key_u8"])
    e3 = np.abs(got3 - ref2)
    _log(f"reference_windowing[tiny,fp16,precise,T=46,2-step TCD] pixel max_abs={e2.max():.3e} mean_abs={e2.mean():.3e} | key-frame uint8 levels flipped: "
         f"{int(kd.sum())} of {kd.size} | with the oracle's key frames handed in (float error only): max_abs={e3.max():.3e} mean_abs={e3.mean():.3e}")
    assert e3.max() <= 1.0e-3                                   # float error of the reference-default regime: the north-star bound
    assert e2.max() <= (1.0e-3 if kd.sum() == 0 else 6e-3)      # a flipped key level is 3.9e-3 on its own pixel (and is what the GUI user's path contains)
