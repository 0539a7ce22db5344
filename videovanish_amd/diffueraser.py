"""API mirror of the third-party `diffueraser.diffueraser.DiffuEraser` class the reference constructs and calls
(reference diffuerase.py:8,39-45,62-67) -- same constructor / forward signature, MI355X-native underneath."""
import numpy as np

from .config import RunConfig
from .nn import normalize_device
from .pipeline import DiffuEraserHIP


class DiffuEraser:
    def __init__(self, device, base_model_path="stable-diffusion-v1-5/stable-diffusion-v1-5", vae_path="stabilityai/sd-vae-ft-mse",
                 diffueraser_path="lixiaowen/diffuEraser", ckpt="2-Step", run: RunConfig = None, dist=None, weights=None, gather="all"):
        # model ids are accepted for signature compatibility; weights are seeded random-init of the same
        # architecture (no network on the build/bench machines; real-weight loading is row n2 of SURVEY 8f)
        self.ids = (base_model_path, vae_path, diffueraser_path)
        self.ckpt = ckpt
        self.run = run or RunConfig()
        self.dist = dist
        self.gather = gather      # multi-GPU: "all" = every rank returns all frames, "rank0" = only rank 0 does (pipeline.gather_frames)
        self.model = DiffuEraserHIP(self.run, normalize_device(device), weights=weights)

    def forward(self, frames, masks, priori, max_img_size=960, mask_dilation_iter=0, guidance_scale=None, progress=None,
                num_inference_steps=None, scheduler=None):
        if guidance_scale not in (None, 0, 0.0):
            raise NotImplementedError("classifier-free guidance is not used on this path (reference passes None)")
        masks2d = [np.any(m > 0, axis=2).astype(np.uint8) * 255 if m.ndim == 3 else m for m in masks]
        if mask_dilation_iter:
            import torch
            from . import hip
            t = hip.mask_collapse_dilate(torch.from_numpy(np.stack(masks2d)).to(self.model.ctx.device), mask_dilation_iter)
            masks2d = list(t.cpu().numpy())
        if num_inference_steps is None:
            num_inference_steps = 2 if self.ckpt == "2-Step" else self.run.steps
        if scheduler is None:
            scheduler = "tcd" if self.ckpt == "2-Step" else "ddim"
        cb = None
        if progress is not None:
            cb = lambda i, n: progress(50 + int(40 * i / max(n, 1)), "running DiffuEraser")
        return self.model.forward(frames, masks2d, priori, max_img_size=max_img_size, steps=num_inference_steps, scheduler=scheduler,
                                  progress=cb, dist=self.dist, gather=self.gather)
