#!/usr/bin/env python3
"""Generate golden fixtures from the REAL reference `diffuerase.py` (run in the build container only).

The reference module (/root/reference/diffuerase.py) cannot be imported as-is: `cv2`, `diffueraser`,
`propainter` are absent.  We inject stub modules for those three names into `sys.modules`, import the
reference, and record what its *in-tree* code does (diffuerase.py:20-114):

  (i)   dilated masks for seeded random masks at k in {0,1,3,8}  (real scipy, diffuerase.py:27-31)
  (ii)  the exact positional/keyword arguments crossing into the third-party boundary
        (diffuerase.py:39-45, 49, 52-57, 62-67)
  (iii) the `prog` call sequence (diffuerase.py:26,33,51,59,69)
  (iv)  the early-return quirk (diffuerase.py:114): only frame 0 is resized/composited
  (v)   the composite arithmetic of diffuerase.py:99-112 given KNOWN d_in/d_out planes (the stub
        distanceTransform returns planes we choose, so the alpha/rint/clip arithmetic is the reference's)

Nothing of the reference's source text is stored: only inputs and outputs (`reference_intree.npz`,
`reference_calls.json`).  /root/reference does not exist on the GPU box; tests read only the fixtures.
"""
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _install_stubs(rec):
    cv2 = types.ModuleType("cv2")
    cv2.INTER_NEAREST, cv2.INTER_LINEAR, cv2.THRESH_BINARY, cv2.DIST_L2 = 0, 1, 0, 2

    def resize(img, dsize, interpolation=1):
        W, H = dsize
        rec["resize"].append((tuple(img.shape), (W, H), int(interpolation)))
        ys = (np.arange(H) * img.shape[0] // H)
        xs = (np.arange(W) * img.shape[1] // W)
        return np.ascontiguousarray(img[ys][:, xs])

    def threshold(src, thresh, maxval, typ):
        return thresh, np.where(src > thresh, maxval, 0).astype(src.dtype)

    def bitwise_not(a):
        return np.bitwise_not(a)

    def distanceTransform(src, dist_type, mask_size):
        rec["dt"].append((int(dist_type), int(mask_size)))
        # stub: a deterministic plane chosen by the fixture (NOT a distance transform)
        return rec["dt_planes"].pop(0).astype(np.float32)

    cv2.resize, cv2.threshold, cv2.bitwise_not, cv2.distanceTransform = resize, threshold, bitwise_not, distanceTransform
    sys.modules["cv2"] = cv2

    de_pkg = types.ModuleType("diffueraser")
    de_mod = types.ModuleType("diffueraser.diffueraser")

    class DiffuEraser:
        def __init__(self, *a, **k):
            rec["de_ctor"] = (list(a), dict(k))

        def forward(self, frames, masks, priori, **k):
            prog = k.pop("progress")
            rec["de_fwd"] = dict(kw=k, n_frames=len(frames), mask_shape=list(masks[0].shape),
                                 mask_dtype=str(masks[0].dtype), mask_values=sorted(set(np.unique(masks[0]).tolist())),
                                 priori_is_given=priori is rec.get("prior_obj"), has_progress=prog is not None)
            rec["masks_seen"] = [m.copy() for m in masks]
            return [f.copy() for f in rec["model_out"]]

    de_mod.DiffuEraser = DiffuEraser
    sys.modules["diffueraser"] = de_pkg
    sys.modules["diffueraser.diffueraser"] = de_mod

    pp_pkg = types.ModuleType("propainter")
    pp_mod = types.ModuleType("propainter.inference")

    class Propainter:
        def __init__(self, *a, **k):
            rec["pp_ctor"] = (list(a), {kk: str(v) for kk, v in k.items()})

        def forward(self, frames, masks, **k):
            k.pop("progress")
            rec["pp_fwd"] = dict(kw=k, mask_shape=list(masks[0].shape))
            return [f.copy() for f in frames]

    pp_mod.Propainter = Propainter
    pp_mod.get_device = lambda: "cpu"
    sys.modules["propainter"] = pp_pkg
    sys.modules["propainter.inference"] = pp_mod


def main():
    rec = dict(resize=[], dt=[], dt_planes=[])
    _install_stubs(rec)
    sys.path.insert(0, REF)
    import diffuerase as ref  # the real reference module

    rng = np.random.default_rng(20251205)
    T, H0, W0 = 3, 40, 56
    Hm, Wm = 32, 48  # model ("inference") size != input size -> exercises the resize at :72-73
    frames = [rng.integers(0, 256, (H0, W0, 3), dtype=np.uint8) for _ in range(T)]
    out = {}
    calls = {}

    # ---- (i) dilation fixtures, k in {0,1,3,8} (+ an all-empty mask for k=0)
    masks_sparse = []
    for t in range(T):
        m = np.zeros((H0, W0, 3), np.uint8)
        pts = rng.integers(0, [H0, W0], (4, 2))
        for (y, x) in pts:
            m[y, x, rng.integers(0, 3)] = rng.integers(1, 256)
        m[10 + t:14 + t, 20:27, :] = 255
        masks_sparse.append(m)
    out["frames"] = np.stack(frames)
    out["masks"] = np.stack(masks_sparse)
    for k in (0, 1, 3, 8):
        progs = []
        rec["model_out"] = [rng.integers(0, 256, (Hm, Wm, 3), dtype=np.uint8) for _ in range(T)]
        out[f"model_out_k{k}"] = np.stack(rec["model_out"])
        # d_in / d_out planes handed to the composite arithmetic (stub DT), chosen to hit the ramp + ties
        d_in = rng.choice(np.array([0, 0.5, 1, 1.4, 2, 2.1969, 2.8, 3, 4.2, 9], np.float32), (H0, W0))
        d_out = rng.choice(np.array([0, 0.5, 1, 1.4, 2, 2.1969, 2.8, 3, 4.2, 9], np.float32), (H0, W0))
        rec["dt_planes"] = [d_in, d_out]
        out[f"d_in_k{k}"], out[f"d_out_k{k}"] = d_in, d_out
        ref.last_ckpt = None  # force ctor path each time (module-global cache, diffuerase.py:15-18)
        ref.propainter = None
        res = ref.run_infill_on_frames(frames, masks_sparse, mask_dilation_iter=k, prog=lambda p, s: progs.append([int(p), s]))
        out[f"dilated_k{k}"] = np.stack(rec["masks_seen"])
        out[f"result0_k{k}"] = res[0]  # frame 0: resized + composited (only frame processed: early return :114)
        calls[f"k{k}"] = dict(prog=progs, n_out=len(res), out_shapes=[list(r.shape) for r in res],
                              de_ctor=rec["de_ctor"], de_fwd=rec["de_fwd"], pp_ctor=rec["pp_ctor"], pp_fwd=rec["pp_fwd"],
                              resize=rec["resize"][-2:], dt=rec["dt"][-2:])
    # empty mask + k=0 and k=2
    empty = [np.zeros((H0, W0, 3), np.uint8) for _ in range(T)]
    for k in (0, 2):
        rec["model_out"] = [f.copy() for f in frames]
        rec["dt_planes"] = [np.zeros((H0, W0), np.float32), np.full((H0, W0), 50, np.float32)]
        ref.run_infill_on_frames(frames, empty, mask_dilation_iter=k)
        out[f"dilated_empty_k{k}"] = np.stack(rec["masks_seen"])

    # ---- prior supplied => Propainter must NOT be called; feather_px=0 hard composite; keep_unmasked False
    rec.pop("pp_fwd", None)
    prior = [f.copy() for f in frames]
    rec["prior_obj"] = prior
    rec["model_out"] = [rng.integers(0, 256, (H0, W0, 3), dtype=np.uint8) for _ in range(T)]
    out["model_out_hard"] = np.stack(rec["model_out"])
    progs = []
    res = ref.run_infill_on_frames(frames, masks_sparse, mask_dilation_iter=3, propainer_frames=prior, feather_px=0,
                                   max_img_size=512, prog=lambda p, s: progs.append([int(p), s]))
    out["result0_hard"] = res[0]
    calls["prior_given"] = dict(prog=progs, pp_called="pp_fwd" in rec, de_fwd=rec["de_fwd"])
    res = ref.run_infill_on_frames(frames, masks_sparse, mask_dilation_iter=3, propainer_frames=prior, keep_unmasked_original=False)
    out["result0_nokeep"] = res[0]

    np.savez_compressed(os.path.join(OUT, "reference_intree.npz"), **out)
    with open(os.path.join(OUT, "reference_calls.json"), "w") as f:
        json.dump(calls, f, default=str)
    print("wrote", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
