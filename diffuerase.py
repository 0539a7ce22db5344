"""Drop-in replacement for the reference module `diffuerase` (reference diffuerase.py:1-158): same module name, same
`run_infill_on_frames` signature and CLI, so `videovanish.py` (reference :46,1518,1593) keeps working unchanged --
with the hot path running on hand-written gfx950 HIP kernels (videovanish_amd/) instead of torch/cuDNN.

Extra keyword-only knobs (old callers are unaffected): num_inference_steps, scheduler, chunk, overlap, dtype, seed,
compat_reference_early_return.  There is no CPU fallback: without the HIP extension / a GPU this raises.
"""
import argparse
import os

import numpy as np
import torch

from videovanish_amd import hip
from videovanish_amd.config import RunConfig
from videovanish_amd.diffueraser import DiffuEraser
from videovanish_amd.propainter import Propainter, get_device

# module-level singletons, as in the reference (diffuerase.py:15-18): not re-entrant, one worker thread at a time
device = None
last_ckpt = None
video_inpainting_sd = None
propainter = None
_run_config = None      # set through configure(); None = full SD-1.5 / sd-vae-ft-mse shapes
_dist = None
_gather = "all"
_prior_stages = {}      # configure(prior=...): optional learned stages of the ProPainter prior (flow_completion, generator)
_weights = None         # configure(weights=...) / $VV_WEIGHTS_DIR: a local model store (videovanish_amd/modelhub.py) or a CheckpointWeights
_loaded = None          # (CheckpointWeights, prior stages) resolved from _weights, cached until configure() is called again


def configure(run: RunConfig = None, dist=None, gather="all", prior=None, weights=None, reference_defaults=False):
    """Select architecture / chunking / dtype for subsequently constructed models (tests use small configs).
    dist = (rank, world) with torch.distributed initialised, one process per GPU (torchrun); gather = "all": every rank returns
    every frame; "rank0": only rank 0 does (the other ranks get None for frames they do not own and should not write a file).
    prior = {"flow_completion": bool, "generator": bool}: run the learned stages of the full ProPainter prior (videovanish_amd/propainter.py).
    weights = a directory holding the four checkpoints the reference names (reference :41-43,49; layout: videovanish_amd/modelhub.py), or a
    checkpoint.CheckpointWeights.  Without it (and without $VV_WEIGHTS_DIR) the models are seeded random-init of the same architecture.
    With it every tensor is checked against the architecture first, the empty prompt is CLIP-encoded once, the PCM "2-Step" LoRA is merged,
    and the learned ProPainter stages switch ON when their files are present (unless `prior` says otherwise).
    reference_defaults=True: ONE switch for the computation the reference app runs by default (reference diffuerase.py:20-21,37,47-57 + the third-party
    forward): the pipeline's own temporal scheme (22-frame windows shifted on odd steps, value / count averaging, key-frame pre-inference:
    RunConfig.windowing="reference", one GPU), the 2-step TCD schedule of the "2-Step" checkpoint (already the default of this module), and the
    COMPLETE ProPainter prior (recurrent flow completion + inpainting generator) when no prior is handed over -- instead of this build's defaults
    (independent 32 / 8 chunks that shard over GPUs; RAFT + propagation only).  `run` / `prior` given explicitly still win field by field."""
    global _run_config, _dist, _gather, last_ckpt, _prior_stages, propainter, _weights, _loaded
    if reference_defaults:
        import dataclasses
        run = dataclasses.replace(run or RunConfig(), windowing="reference")
        prior = dict({"flow_completion": True, "generator": True}, **(prior or {}))
    _run_config, _dist, _gather, last_ckpt = run, dist, gather, None
    _prior_stages, propainter = dict(prior or {}), None
    _weights, _loaded = weights, None


def _resolve_weights(ckpt):
    """(weight source | None, prior stages) for the configured / environment-named model store."""
    global _loaded
    src = _weights if _weights is not None else os.environ.get("VV_WEIGHTS_DIR")
    if src is None:
        return None, dict(_prior_stages)
    if _loaded is None:
        if isinstance(src, (str, os.PathLike)):
            from videovanish_amd import modelhub
            run = _run_config or RunConfig()
            _loaded = modelhub.load(os.fspath(src), ckpt=ckpt, ucfg=run.unet, vcfg=run.vae)
        else:
            _loaded = (src, {"flow_completion": "fc" in src.components, "generator": "gen" in src.components and "fc" in src.components})
    w, stages = _loaded
    stages = dict(stages)
    stages.update(_prior_stages)            # an explicit configure(prior=...) wins
    return w, stages


def run_infill_on_frames(frames_rgb, mask_frames, mask_dilation_iter=8, ckpt="2-Step",
                         propainer_frames=None, max_img_size=960, keep_unmasked_original=True, feather_px=3, prog=None,
                         *, num_inference_steps=None, scheduler=None, compat_reference_early_return=False):
    global device, last_ckpt, video_inpainting_sd, propainter

    H0, W0 = frames_rgb[0].shape[:2]

    if prog is not None: prog(5, "dilating frames")
    dev = get_device()
    m = torch.from_numpy(np.stack([mm if mm.ndim == 3 else mm[..., None] for mm in mask_frames])).to(dev)
    dil_t = hip.mask_collapse_dilate(m.contiguous(), mask_dilation_iter)       # reference :27-31
    dilated_mask_frames = list(dil_t.cpu().numpy())

    if prog is not None: prog(10, "loading weights")
    if last_ckpt != ckpt or video_inpainting_sd is None:                        # reference :35-45 (ckpt forced to "2-Step")
        device = dev
        ckpt = "2-Step"
        last_ckpt = ckpt
        video_inpainting_sd = DiffuEraser(device, "stable-diffusion-v1-5/stable-diffusion-v1-5", "stabilityai/sd-vae-ft-mse",
                                          "lixiaowen/diffuEraser", ckpt=ckpt, run=_run_config, dist=_dist, gather=_gather,
                                          weights=_resolve_weights(ckpt)[0])

    if propainer_frames is None:                                                # reference :47-57
        if propainter is None:
            w, stages = _resolve_weights("2-Step")
            propainter = Propainter("ruffy369/propainter", device=device, weights=w if (w is not None and "raft" in w.components) else None, **stages)
        if prog is not None: prog(20, "running propainter prior")
        prev_tag, hip.PROFILE_TAG = hip.PROFILE_TAG, "prior:"                   # bench.py --prior raft prices the prior's kernels under this prefix
        try:
            propainer_frames = propainter.forward(frames_rgb, dilated_mask_frames, ref_stride=10, neighbor_length=10,
                                                  subvideo_length=50, mask_dilation=0, progress=prog)
        finally:
            hip.PROFILE_TAG = prev_tag

    if prog is not None: prog(50, "running DiffuEraser")
    guidance_scale = None
    inpainted_frames = video_inpainting_sd.forward(frames_rgb, dilated_mask_frames, propainer_frames, max_img_size=max_img_size,
                                                   mask_dilation_iter=0, guidance_scale=guidance_scale, progress=prog,
                                                   num_inference_steps=num_inference_steps, scheduler=scheduler)

    if prog is not None: prog(90, "resizing and merging finished frames")
    # reference :69-112.  The reference returns from inside its loop (:114) so only frame 0 is post-processed; the
    # evident intent (all frames) is the default here, compat_reference_early_return=True reproduces the quirk.
    n_post = 1 if compat_reference_early_return else len(inpainted_frames)
    idx = [i for i in range(n_post) if inpainted_frames[i] is not None]        # multi-GPU "rank0" gather: other ranks hold only their own frames
    if not idx:
        return inpainted_frames
    Hm, Wm = inpainted_frames[idx[0]].shape[:2]
    out = torch.from_numpy(np.stack([inpainted_frames[i] for i in idx])).to(dev)
    if (Hm, Wm) != (H0, W0):
        out = hip.resize_u8(out.contiguous(), H0, W0, mode="bilinear")          # cv2.resize(f,(W0,H0)), :73
    if keep_unmasked_original:
        orig = torch.from_numpy(np.stack([frames_rgb[i] for i in idx])).to(dev)
        out = hip.feather_composite(out.contiguous(), orig.contiguous(), dil_t[idx].contiguous(), float(feather_px))   # :77-112
    out = out.cpu().numpy()
    for j, i in enumerate(idx):
        inpainted_frames[i] = out[j]
    return inpainted_frames


def _frame_io():
    """The reference's own frame I/O helper `tools` (cv2; reference tools.py:4-45) when the drop-in sits next to the GUI, else the
    cv2-free FFV1 / Matroska module with the same two functions.  The test is for the API, not for the import: a
    directory called tools/ next to this file imports as an (empty) namespace package."""
    try:
        import tools
        if callable(getattr(tools, "load_video_frames_from_path", None)) and callable(getattr(tools, "write_video_frames_to_path", None)):
            return tools
    except ImportError:
        pass
    from videovanish_amd import frameio
    return frameio


# =============================
# CLI entry point (reference diffuerase.py:121-155)
# =============================
def main():
    tools = _frame_io()
    ap = argparse.ArgumentParser(description="Remove masked objects from a video (DiffuEraser hot path on MI355X).")
    ap.add_argument("--color_video", required=True, type=str, help="Input color video path.")
    ap.add_argument("--mask_video", required=True, type=str, help="Input mask video path.")
    ap.add_argument("--prior_video", required=False, type=str, help="Input prior video path.")
    ap.add_argument("--start_frame", type=int, default=0, help="Index of first frame to process (default: 0).")
    ap.add_argument("--max_frames", type=int, default=-1, help="Max number of frames to process after start_frame.")
    ap.add_argument("--out", type=str, default=None, help="Output video path (default: <input>_vanished.mkv)")
    args = ap.parse_args()

    assert os.path.isfile(args.color_video), "input video missing"
    out_video = args.out or (args.color_video + "_vanished.mkv")
    frames, fps = tools.load_video_frames_from_path(args.color_video, args.start_frame, args.max_frames)
    H0, W0 = frames[0].shape[:2]
    mask_frames, mask_fps = tools.load_video_frames_from_path(args.mask_video, args.start_frame, args.max_frames)
    Hm, Wm = mask_frames[0].shape[:2]
    prior_frames = None
    if args.prior_video is not None:      # the reference's test is inverted (:142); a supplied prior is used here
        prior_frames, prior_fps = tools.load_video_frames_from_path(args.prior_video, args.start_frame, args.max_frames)
        Hp, Wp = prior_frames[0].shape[:2]
        assert (H0 == Hp and W0 == Wp), "prior and color video are diffrent sizes"
    assert (H0 == Hm and W0 == Wm), "mask and color video are diffrent sizes"
    out_frames = run_infill_on_frames(frames, mask_frames, propainer_frames=prior_frames)
    tools.write_video_frames_to_path(out_video, out_frames, fps, H0, W0)


if __name__ == '__main__':
    main()
