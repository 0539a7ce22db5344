"""Drop-in replacement for the reference module `sam2_masker` (reference sam2_masker.py:1-210): same module name, same
`run_sam2_on_frames(frames_rgb, annotations, device=None, prog=None)` and CLI, so `videovanish.py` (reference :46,1461,1555) keeps working
unchanged -- with SAM 2.1 running on the gfx950 HIP kernels (videovanish_amd/sam2_model.py) behind the `SAM2VideoPredictor` API mirror
(videovanish_amd/sam2_predictor.py).  No cv2: `color_for_obj` restates OpenCV's 8-bit HSV -> BGR conversion.  No CPU fallback.

`configure(predictor=...)` / `configure(checkpoint=..., cfg=...)` select the model (tests use small synthetic-weight configurations; the
default resolves the reference's hard-coded checkpoint path, reference :19, or $VV_SAM2_CHECKPOINT, and fails loudly if it is missing).
"""
import argparse
import json
import os

import numpy as np

# reference :19-20
SAM2_CHECKPOINT = "sam2_numpy_frames/checkpoints/sam2.1_hiera_large.pt"
SAM2_MODEL_CFG = "configs/sam2.1/sam2.1_hiera_l.yaml"

predictor = None


def configure(predictor_=None, checkpoint=None, cfg=None, seed=None, dtype=None, device=None, **kw):
    """predictor_ (or the keyword `predictor=`): a ready Sam2VideoPredictor.  Otherwise one is built on the HIP path: from `checkpoint` (the
    published .pt), or -- only when `seed` is given explicitly -- from seeded synthetic weights of configuration `cfg` (tests / benchmarks
    without network access).  dtype: MFMA operand type; default "bf16" for a real checkpoint -- the reference runs SAM 2 under bfloat16
    autocast (reference :76) and published-weight activations are not range-checked for fp16 here -- and "fp16" (11 significant bits, what the
    parity tests measure) for synthetic weights."""
    global predictor
    if "predictor" in kw:
        predictor_ = kw.pop("predictor")
    if kw:
        raise TypeError(f"configure() got unexpected keyword arguments {sorted(kw)}")
    if dtype is None:
        dtype = "bf16" if checkpoint is not None else "fp16"
    if predictor_ is not None or (checkpoint is None and seed is None):
        predictor = predictor_
        return
    from videovanish_amd.sam2_config import Sam2Config
    from videovanish_amd.sam2_model import HipSam2
    from videovanish_amd.sam2_predictor import Sam2VideoPredictor
    from videovanish_amd.sam2_weights import Sam2Weights
    cfg = cfg or Sam2Config()
    w = Sam2Weights.from_checkpoint(checkpoint, cfg) if checkpoint is not None else Sam2Weights(cfg, seed)
    predictor = Sam2VideoPredictor(HipSam2(cfg, w, device=device, dtype=dtype))


# =============================
# Color mapping per object id (reference :27-37)
# =============================
_SECTOR = ((1, 3, 0), (1, 0, 2), (3, 0, 1), (0, 2, 1), (0, 1, 3), (2, 1, 0))


def color_for_obj(obj_id):
    """Deterministic, bright BGR color for a given obj_id using HSV cycling: cv2.cvtColor(uint8 [[[h, 200, 255]]], COLOR_HSV2BGR) with
    h = (obj_id * 37) % 180, restated (OpenCV's 8-bit path: float arithmetic on h * 6/180, s/255, v/255; result * 255 rounded to nearest)."""
    h = np.float32(int((obj_id * 37) % 180)) * np.float32(6.0 / 180.0)
    s, v = np.float32(200.0 / 255.0), np.float32(1.0)
    sector = int(np.floor(h))
    f = h - np.float32(sector)
    sector %= 6
    tab = (v, v * (np.float32(1) - s), v * (np.float32(1) - s * f), v * (np.float32(1) - s * (np.float32(1) - f)))
    b, g, r = (tab[i] for i in _SECTOR[sector])
    return tuple(int(min(255, max(0, np.rint(x * np.float32(255.0))))) for x in (b, g, r))  # (B, G, R)


# =============================
# Library API (reference :43-177)
# =============================
def run_sam2_on_frames(frames_rgb, annotations, device=None, prog=None):
    """
    Run SAM2 segmentation on a list of frames, then return COLORED mask frames
    (black background; each obj_id rendered in its own solid color).

    Args / returns: as the reference (list of (H,W,3) uint8 frames; annotations = {"keyframes": [{frame_idx, pos_clicks: [{x,y,obj}],
    neg_clicks: [{x,y,obj}], rects: [{x,y,w,h,obj}]}]}, coordinates normalised [0..1] or absolute pixels) -> list of (H,W,3) uint8.
    """
    global predictor
    assert isinstance(frames_rgb, (list, tuple)) and len(frames_rgb) > 0, "frames must be a non-empty list"
    H0, W0 = frames_rgb[0].shape[:2]

    if prog is not None: prog(1, "Setting up sam2")
    if predictor is None:                                                       # reference :87-88
        from videovanish_amd.sam2_predictor import build_sam2_video_predictor
        predictor = build_sam2_video_predictor(SAM2_MODEL_CFG, os.environ.get("VV_SAM2_CHECKPOINT", SAM2_CHECKPOINT), device=device)

    if prog is not None: prog(25, "Loading frames in to sam2")
    # one forward pass after the prompts: outputs no later frame can select are freed as tracking advances.  The flag is restored afterwards --
    # a predictor the caller handed in through configure(predictor=...) must behave like upstream when it is reused
    had_trim = hasattr(predictor, "trim_memory")
    old_trim = getattr(predictor, "trim_memory", None)
    if had_trim:
        predictor.trim_memory = True
    try:
        return _run(predictor, frames_rgb, annotations, W0, H0, prog)
    finally:
        if had_trim:
            predictor.trim_memory = old_trim


def _run(predictor, frames_rgb, annotations, W0, H0, prog):
    inference_state = predictor.init_state(video_path=frames_rgb)

    # prompts in the order the reference issues them (reference :96-141): per keyframe (sorted by frame), the clicks of each object in one
    # call, then every rectangle in its own call (which replaces that object's clicks on the frame: clear_old_points defaults to True)
    for frame_idx, obj_id, kw in _prompts(annotations, W0, H0):
        predictor.add_new_points_or_box(inference_state=inference_state, frame_idx=frame_idx, obj_id=obj_id, **kw)

    if prog is not None: prog(45, "Infering masks with sam2")
    segments = {}
    for t, obj_ids, logits in predictor.propagate_in_video(inference_state):
        segments[t] = {int(o): (logits[i] > 0.0).detach().cpu().numpy() for i, o in enumerate(obj_ids)}

    if prog is not None: prog(80, "Creating color mask from sam2 data")
    return [_paint(segments.get(t, {}), H0, W0) for t in range(len(frames_rgb))]


def _px(v, size):
    """a coordinate in [0, 1] is a fraction of the frame, anything else is already in pixels (reference :96-97)."""
    return float(v) * size if 0.0 <= v <= 1.0 else float(v)


def _prompts(annotations, W0, H0):
    """annotations -> [(frame_idx, obj_id, kwargs of add_new_points_or_box)] (reference :98-141)."""
    calls = []
    for kf in sorted(annotations.get("keyframes", []), key=lambda k: int(k["frame_idx"])):
        t = int(kf["frame_idx"])
        per_obj = {}
        for label, key in ((1, "pos_clicks"), (0, "neg_clicks")):
            for c in kf.get(key, []):
                pts, labs = per_obj.setdefault(int(c.get("obj", 1)), ([], []))
                pts.append((_px(c["x"], W0), _px(c["y"], H0)))
                labs.append(label)
        for obj, (pts, labs) in per_obj.items():
            calls.append((t, obj, dict(points=np.asarray(pts, dtype=np.float32).reshape(-1, 2), labels=np.asarray(labs, dtype=np.int32))))
        for r in kf.get("rects", []):
            x1, y1 = _px(r["x"], W0), _px(r["y"], H0)
            x2 = _px(r["x"] + r["w"], W0) if 0.0 <= r["w"] <= 1.0 else x1 + float(r["w"])
            y2 = _px(r["y"] + r["h"], H0) if 0.0 <= r["h"] <= 1.0 else y1 + float(r["h"])
            calls.append((t, int(r.get("obj", 1)), dict(box=np.array([min(x1, x2), min(y1, y2), max(x1, x2), max(y1, y2)], dtype=np.float32))))
    return calls


def _paint(masks, H0, W0):
    """black canvas, one solid colour per object, higher object ids painted last (reference :155-175)."""
    canvas = np.zeros((H0, W0, 3), dtype=np.uint8)
    for obj in sorted(masks):
        m = masks[obj]
        if m is None or m.size == 0:
            continue
        m = np.squeeze(np.asarray(m)) if np.asarray(m).ndim > 2 else np.asarray(m)
        if m.shape != (H0, W0):                  # cv2.resize(..., INTER_NEAREST): source index = floor(destination index * scale)
            ys = np.minimum((np.arange(H0) * (m.shape[0] / H0)).astype(np.int64), m.shape[0] - 1)
            xs = np.minimum((np.arange(W0) * (m.shape[1] / W0)).astype(np.int64), m.shape[1] - 1)
            m = m[ys][:, xs]
        canvas[m.astype(bool)] = color_for_obj(obj)
    return canvas


def _frame_io():
    """as diffuerase._frame_io: the reference's `tools` (cv2) when it is importable with the right API, else the cv2-free FFV1 / Matroska module."""
    try:
        import tools
        if callable(getattr(tools, "load_video_frames_from_path", None)) and callable(getattr(tools, "write_video_frames_to_path", None)):
            return tools
    except ImportError:
        pass
    from videovanish_amd import frameio
    return frameio


# =============================
# CLI entry point (reference :183-205)
# =============================
def main():
    tools = _frame_io()
    ap = argparse.ArgumentParser(description="Create colored mask video with SAM2 (one color per object, black background).")
    ap.add_argument("--color_video", required=True, type=str, help="Input color video path.")
    ap.add_argument("--annotations", required=True, type=str, help="JSON annotation file.")
    ap.add_argument("--start_frame", type=int, default=0, help="Index of first frame to process (default: 0).")
    ap.add_argument("--max_frames", type=int, default=-1, help="Max number of frames to process after start_frame.")
    ap.add_argument("--out", type=str, default=None, help="Output video path (default: <input>_sam2_mask.mkv)")
    args = ap.parse_args()

    assert os.path.isfile(args.color_video), "input video missing"
    out_video = args.out or (args.color_video + "_sam2_mask.mkv")

    frames, fps = tools.load_video_frames_from_path(args.color_video, args.start_frame, args.max_frames)
    H0, W0 = frames[0].shape[:2]

    with open(args.annotations, "r") as f:
        ann = json.load(f)

    mask_frames = run_sam2_on_frames(frames, ann)
    tools.write_video_frames_to_path(out_video, mask_frames, fps, H0, W0)


if __name__ == "__main__":
    main()
