#!/usr/bin/env python3
"""Generate golden fixtures from the REAL reference `sam2_masker.py` (run in the build container only).

/root/reference/sam2_masker.py cannot be imported as-is (`cv2`, `sam2` absent).  Stub modules are injected for those names, the reference's
own `run_sam2_on_frames` (sam2_masker.py:43-177) is run on seeded inputs, and what its IN-TREE code does is recorded:

  (i)   the exact sequence of predictor calls for a set of annotations -- init_state, add_new_points_or_box (frame, object id, pixel
        coordinates after the reference's normalised-or-absolute rule, labels, boxes as x1 y1 x2 y2) -- sam2_masker.py:93-141
  (ii)  the `prog` sequence (sam2_masker.py:66,90,143,153)
  (iii) the painted output frames for masks the stub predictor yields (per-object colour, higher ids on top, frames the predictor never
        yields stay black) -- sam2_masker.py:155-175.  The stub cv2.cvtColor returns a colour table chosen by this script (NOT a colour
        conversion), so the painting logic is the reference's and the HSV arithmetic is not pinned (no cv2 here).

Only inputs and outputs are stored (`sam2_masker_calls.json`, `sam2_masker_frames.npz`), none of the reference's source text.
"""
import json
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
H0, W0, T = 36, 48, 6

ANNOTATIONS = {"keyframes": [
    {"frame_idx": 3, "pos_clicks": [{"x": 0.25, "y": 0.75, "obj": 2}], "neg_clicks": [], "rects": []},
    {"frame_idx": 1,
     "pos_clicks": [{"x": 0.5, "y": 0.5, "obj": 1}, {"x": 30, "y": 20, "obj": 2}, {"x": 1.0, "y": 0.0}],
     "neg_clicks": [{"x": 0.1, "y": 0.2, "obj": 1}, {"x": 47.5, "y": 2, "obj": 5}],
     "rects": [{"x": 0.2, "y": 0.25, "w": 0.5, "h": 0.5, "obj": 3}, {"x": 10, "y": 8, "w": 20, "h": 12, "obj": 1}, {"x": 0.9, "y": 0.9, "w": -0.5, "h": 1.5}]},
]}


def stub_colour(h):
    """the table the stub cvtColor answers with: any injective map works (the fixture stores the colours it produced)."""
    return (int(h) % 256, (3 * int(h) + 17) % 256, 255 - int(h) % 200)


def main():
    rec = {"calls": [], "prog": [], "hsv": []}
    cv2 = types.ModuleType("cv2")
    cv2.COLOR_HSV2BGR, cv2.INTER_NEAREST = 54, 0

    def cvtColor(a, code):
        rec["hsv"].append([int(v) for v in a[0, 0]] + [int(code)])
        return np.uint8([[stub_colour(a[0, 0, 0])]])

    def resize(img, dsize, interpolation=0):
        W, H = dsize
        return np.ascontiguousarray(img[(np.arange(H) * img.shape[0] // H)][:, (np.arange(W) * img.shape[1] // W)])

    cv2.cvtColor, cv2.resize = cvtColor, resize
    cv2.VideoCapture = cv2.VideoWriter = cv2.VideoWriter_fourcc = None
    sys.modules["cv2"] = cv2

    class StubPredictor:
        def init_state(self, video_path=None):
            rec["calls"].append({"op": "init_state", "n_frames": len(video_path), "shape": list(video_path[0].shape)})
            self.first, self.objs = None, []
            return {"state": True}

        def add_new_points_or_box(self, inference_state, frame_idx, obj_id, points=None, labels=None, box=None):
            c = {"op": "add", "frame_idx": int(frame_idx), "obj_id": int(obj_id), "frame_type": type(frame_idx).__name__}
            if points is not None:
                c.update(points=np.asarray(points).tolist(), points_dtype=str(np.asarray(points).dtype), labels=np.asarray(labels).tolist(),
                         labels_dtype=str(np.asarray(labels).dtype))
            if box is not None:
                c.update(box=np.asarray(box).tolist(), box_dtype=str(np.asarray(box).dtype))
            rec["calls"].append(c)
            if obj_id not in self.objs:
                self.objs.append(obj_id)
            self.first = frame_idx if self.first is None else min(self.first, frame_idx)

        def propagate_in_video(self, inference_state):
            rec["calls"].append({"op": "propagate"})
            g = torch.Generator().manual_seed(123)
            for t in range(self.first, T - 1):                      # the last frame is never yielded: it must stay black
                yield t, list(self.objs), torch.randn(len(self.objs), 1, H0, W0, generator=g) - 0.4

    sam2 = types.ModuleType("sam2")
    build = types.ModuleType("sam2.build_sam")
    build.build_sam2_video_predictor = lambda cfg, ckpt, device=None: (rec.__setitem__("build", [cfg, ckpt, str(device)]), StubPredictor())[1]
    sys.modules["sam2"], sys.modules["sam2.build_sam"] = sam2, build
    sys.path.insert(0, REF)
    import sam2_masker as ref                                       # the reference module

    frames = [np.random.default_rng(5).integers(0, 256, (H0, W0, 3), dtype=np.uint8) for _ in range(T)]
    out = ref.run_sam2_on_frames(frames, ANNOTATIONS, prog=lambda p, s: rec["prog"].append([p, s]))
    colours = {str(i): list(ref.color_for_obj(i)) for i in (1, 2, 3, 5)}
    json.dump({"annotations": ANNOTATIONS, "H0": H0, "W0": W0, "T": T, "calls": rec["calls"], "prog": rec["prog"], "build": rec["build"],
               "hsv_requests": rec["hsv"], "stub_colours": colours}, open(os.path.join(OUT, "sam2_masker_calls.json"), "w"), indent=1)
    np.savez_compressed(os.path.join(OUT, "sam2_masker_frames.npz"), out=np.stack(out))
    print("wrote", len(rec["calls"]), "calls,", len(out), "frames; painted fraction", float(np.stack(out).any(axis=3).mean()))


if __name__ == "__main__":
    main()
