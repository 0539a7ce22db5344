"""API mirror of the third-party `propainter.inference` module (reference diffuerase.py:9,36,49,52-57)."""
import torch


def get_device():
    """reference diffuerase.py:36 -- the product path is GPU only: fail loudly without a HIP device."""
    if not torch.cuda.is_available():
        raise RuntimeError("videovanish_amd: no HIP device visible (there is no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


class Propainter:
    """Flow-guided propagation prior (RAFT correlation + bilinear warp)."""

    def __init__(self, model_dir="ruffy369/propainter", device=None, flow_completion=False, generator=False, weights=None):
        # flow_completion: run the recurrent flow-completion network (flowcomplete.py) between RAFT and the propagation, as the real
        # ProPainter does.  Off by default: the network's trained weights (ruffy369/propainter, reference diffuerase.py:49) are not
        # reachable from the build image and seeded random weights would replace the measured flow inside the holes by noise.
        # generator: run the inpainting generator (inpaintgen.py) over the propagated frames -- the last stage of the real ProPainter; off by
        # default for the same reason.  With both switches on, `forward` runs the complete ProPainter pipeline: RAFT -> flow completion ->
        # image propagation -> generator.
        # weights: a checkpoint.CheckpointWeights with the components "raft" (+ "fc", "gen") -- diffuerase.configure(weights=...) resolves the
        # reference's model id to local files (modelhub.py); None = seeded random init
        self.model_dir, self.device, self.flow_completion, self.generator, self.weights = model_dir, device, flow_completion, generator, weights

    def forward(self, frames, masks, ref_stride=10, neighbor_length=10, subvideo_length=50, mask_dilation=0, progress=None):
        """Same signature as the third-party call (reference diffuerase.py:52-57).  Built: RAFT flows (both directions) + flow-guided
        image propagation, run per sub-video of min(100, subvideo_length) frames with 5 frames of context (`subvideo_length`), after an
        optional mask dilation (`mask_dilation`, L1 ball like the reference's own dilation); with `flow_completion=True` (constructor) the
        recurrent flow-completion network fills the flows inside the holes first, and with `generator=True` the inpainting generator
        (deformable feature propagation + sparse-window transformer) runs over the propagated frames in sliding windows of `neighbor_length`
        frames with every `ref_stride`-th frame as reference.  Without the generator those two knobs have nothing to act on and hole pixels
        no consistent flow reaches keep the frame's mean colour."""
        import numpy as np
        from .flowprop import flow_propagation_prior
        if ref_stride <= 0 or neighbor_length <= 0 or subvideo_length <= 0:
            raise ValueError("Propainter.forward: ref_stride, neighbor_length and subvideo_length must be positive")
        if mask_dilation:
            from . import hip
            m = torch.from_numpy(np.stack([mm if mm.ndim == 3 else mm[..., None] for mm in masks])).to(self.device if self.device is not None else get_device())
            masks = list(hip.mask_collapse_dilate(m.contiguous(), int(mask_dilation)).cpu().numpy())
        return flow_propagation_prior(frames, masks, device=self.device, progress=progress, subvideo_length=subvideo_length,
                                      flow_completion=self.flow_completion, generator=self.generator, ref_stride=ref_stride,
                                      neighbor_length=neighbor_length, weights=self.weights)
