"""API mirror of the third-party `propainter.inference` module (reference diffuerase.py:9,36,49,52-57)."""
import torch


def get_device():
    """reference diffuerase.py:36 -- the product path is GPU only: fail loudly without a HIP device."""
    if not torch.cuda.is_available():
        raise RuntimeError("videovanish_amd: no HIP device visible (there is no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


class Propainter:
    """Flow-guided propagation prior (RAFT correlation + bilinear warp)."""

    def __init__(self, model_dir="ruffy369/propainter", device=None):
        self.model_dir, self.device = model_dir, device

    def forward(self, frames, masks, ref_stride=10, neighbor_length=10, subvideo_length=50, mask_dilation=0, progress=None):
        from .flowprop import flow_propagation_prior
        return flow_propagation_prior(frames, masks, device=self.device, progress=progress)
