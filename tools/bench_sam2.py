#!/usr/bin/env python3
"""SAM 2.1 Hiera-L video predictor on the HIP path at the published size (1024 x 1024 model input, 224.4 M parameters, seeded synthetic
weights: no checkpoint is reachable offline): frames/s of `propagate_in_video` for N objects on a 720p clip + per-stage times.
    python tools/bench_sam2.py [frames=12] [objects=1] [fp16|bf16]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from videovanish_amd import hip
from videovanish_amd.sam2_config import Sam2Config
from videovanish_amd.sam2_model import HipSam2
from videovanish_amd.sam2_predictor import Sam2VideoPredictor
from videovanish_amd.sam2_weights import Sam2Weights

T = int(sys.argv[1]) if len(sys.argv) > 1 else 12
NOBJ = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dname = sys.argv[3] if len(sys.argv) > 3 else "fp16"
H, W = 720, 1280
cfg = Sam2Config()
t0 = time.time()
model = HipSam2(cfg, Sam2Weights(cfg, 0), device="cuda:0", dtype=dname)
torch.cuda.synchronize()
print(f"model built in {time.time() - t0:.1f} s ({torch.cuda.memory_allocated() / 2**30:.2f} GiB of weights + constants on the device)", flush=True)
rng = np.random.default_rng(0)
base = rng.integers(0, 256, (H + 4 * T, W + 4 * T, 3), dtype=np.uint8)
frames = [np.ascontiguousarray(base[2 * t:2 * t + H, 3 * t:3 * t + W]) for t in range(T)]
pred = Sam2VideoPredictor(model)


def sync_time(fn, n=3):
    fn(); torch.cuda.synchronize()
    t = time.time()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize()
    return (time.time() - t) / n, r


dt_enc, feats = sync_time(lambda: model.encode_image(frames[0]))
print(f"image encoder (Hiera-L + FPN, 1024^2): {dt_enc * 1e3:.1f} ms / frame", flush=True)
st = pred.init_state(video_path=frames)
for o in range(NOBJ):
    pred.add_new_points_or_box(st, 0, o + 1, points=np.array([[200.0 + 150 * o, 300.0]], dtype=np.float32), labels=np.array([1], dtype=np.int32))
torch.cuda.synchronize()
t = time.time()
n = 0
per = []
for fi, ids, logits in pred.propagate_in_video(st):
    torch.cuda.synchronize()
    per.append(time.time() - t)
    t = time.time()
    n += 1
steady = per[min(8, len(per) - 1):]        # after the memory bank has filled (7 frames)
print(f"propagate_in_video: {n} frames x {NOBJ} object(s) at 1280x720: {sum(per):.2f} s total; steady state {np.mean(steady) * 1e3:.1f} ms / frame "
      f"= {1.0 / np.mean(steady):.2f} frames/s (first tracked frames: {[round(p * 1e3) for p in per[:4]]} ms)", flush=True)
od = st["output_dict_per_obj"][0]
last = max(od["non_cond_frame_outputs"])
f = model.encode_image(frames[last])
dt_track, _ = sync_time(lambda: model.track_step(last, False, f, None, od, T, run_mem_encoder=False))
dt_mem, _ = sync_time(lambda: model.encode_memory_from_low_res(f, od["non_cond_frame_outputs"][last]["pred_masks"], od["non_cond_frame_outputs"][last]["object_score_logits"], False))
print(f"per object and frame with a full memory bank: memory attention + SAM heads {dt_track * 1e3:.1f} ms, memory encoder {dt_mem * 1e3:.1f} ms", flush=True)
import json
print(json.dumps({"metric": "SAM 2.1 Hiera-L tracked frames/sec at 720p (propagate_in_video, steady state)", "value": round(1.0 / float(np.mean(steady)), 2), "unit": "frames/s",
                  "n_gpus": 1, "objects": NOBJ, "frames": T, "dtype": dname, "data": "synthetic", "higher_is_better": True,
                  "config": {"workload": f"{T}-frame 1280x720 clip -> 1024x1024 model input, {NOBJ} object(s), one click on frame 0, SAM 2.1 Hiera-L (224.4 M parameters), "
                                         "seeded synthetic weights"},
                  "stages_ms": {"image_encoder_single_frame": round(dt_enc * 1e3, 2), "memory_attention_and_sam_heads_per_object": round(dt_track * 1e3, 2),
                                "memory_encoder_per_object": round(dt_mem * 1e3, 2), "frame_steady_state": round(float(np.mean(steady)) * 1e3, 2)}}), flush=True)
