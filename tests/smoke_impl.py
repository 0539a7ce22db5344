"""__graft_entry__.smoke(): one small invocation of the hot path on cuda:0, checked against the oracle."""
import numpy as np


def run():
    import torch
    from oracle import pipeline_ref as R
    from videovanish_amd.config import TINY_UNET, TINY_VAE, RunConfig
    from videovanish_amd.pipeline import DiffuEraserHIP
    rng = np.random.default_rng(1234)
    T, H, W = 4, 32, 40
    frames = [rng.integers(0, 256, (H, W, 3), dtype=np.uint8) for _ in range(T)]
    m2d = []
    for t in range(T):
        m = np.zeros((H, W), np.uint8)
        m[8:16, 10 + 2 * t:20 + 2 * t] = 255
        m2d.append(m)
    prior = [f.copy() for f in frames]
    run_cfg = RunConfig(steps=2, chunk=4, overlap=2, seed=5, dtype="fp16", precise_decoder=True, unet=TINY_UNET, vae=TINY_VAE)
    got, _ = DiffuEraserHIP(run_cfg).forward(frames, m2d, prior, return_float=True)
    ref = R.diffueraser_forward(frames, m2d, prior, steps=2, chunk=4, overlap=2, seed=5, ucfg=TINY_UNET, vcfg=TINY_VAE, return_float=True)
    err = float(np.abs(got - ref).max())
    assert np.isfinite(got).all() and err <= 1.0e-3, f"smoke parity failed: max-abs {err}"
    print(f"smoke ok: 4-frame 32x40 clip, 2 DDIM steps on {torch.cuda.get_device_name(0)}; max-abs vs oracle {err:.2e} (fp16 operands, split-precision VAE decoder)")
