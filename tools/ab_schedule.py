#!/usr/bin/env python3
"""A/B of the chunk schedules on ONE device, ONE model build: python tools/ab_schedule.py [chunks per config] [configs ...]
config = "<lanes><o|s>": lanes = RunConfig.concurrent_chunks, o = BrushNet on a second stream (Denoiser.OVERLAP), s = one stream per chunk.
Prints seconds per 32-frame 720p / 50-step chunk for every config, in the order given (repeat a config to see the box's drift)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from dataclasses import replace
from videovanish_amd import unet
from videovanish_amd.config import RunConfig
from videovanish_amd.pipeline import DiffuEraserHIP, chunk_plan

K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
configs = sys.argv[2:] or ["1s", "1o", "2o", "3o", "2s", "1s"]
steps = int(os.environ.get("VV_AB_DENOISE_STEPS", "50"))
H, W, chunk, overlap = 720, 1280, 32, 8
run = RunConfig(steps=steps, chunk=chunk, overlap=overlap, seed=42, dtype="fp16")
model = DiffuEraserHIP(run, "cuda:0")
dev = model.ctx.device
T = (chunk - overlap) * K + overlap
assert len(chunk_plan(T, chunk, overlap)) == K
fr, mk, pr = bench.synth_clip(T, H, W)
fr, mk, pr = torch.from_numpy(fr).to(dev), torch.from_numpy(mk).to(dev), torch.from_numpy(pr).to(dev)
model.forward_device(fr[:chunk], pr[:chunk], mk[:chunk], chunk, 0, steps=min(steps, 4), scheduler="ddim")      # warm-up
torch.cuda.synchronize()
ref = None


def masked_streams(kind):
    """Lab: two lane streams restricted to disjoint halves of the chip (hipExtStreamCreateWithCUMask).  The KFD deals mask bit i to XCD i % 8:
    'x' = XCDs 0-3 / 4-7 (own L2s per lane), 'c' = half of the CUs of EVERY XCD per lane, 'q' = 3/4 + 3/4 overlapping in the middle."""
    import ctypes
    path = next(l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l)
    rt = ctypes.CDLL(path)
    out = []
    for lane in range(2):
        if kind == "x":
            word = 0x0F0F0F0F if lane == 0 else 0xF0F0F0F0
            words = [word] * 8
        elif kind == "c":
            word = 0x00FF00FF if lane == 0 else 0xFF00FF00
            words = [word] * 8
        else:
            words = [0xFFFFFFFF] * 6 + [0, 0] if lane == 0 else [0, 0] + [0xFFFFFFFF] * 6
        arr = (ctypes.c_uint32 * 8)(*words)
        h = ctypes.c_void_p()
        rc = rt.hipExtStreamCreateWithCUMask(ctypes.byref(h), 8, arr)
        assert rc == 0, rc
        out.append(torch.cuda.ExternalStream(h.value, device=dev))
    return out


for c in configs:
    c, _, mask = c.partition("%")               # "2s%x": lanes on CU-masked streams (lab)
    model.__dict__["_lane_streams"] = masked_streams(mask) if mask else []
    c, _, vb = c.partition("/")                 # "2s/8": VAE encode / decode in batches of 8 frames (DiffuEraserHIP.vae_batch; default 4)
    model.vae_batch = int(vb) if vb else 4
    c0, _, stag = c.partition("@")              # "2s@0.4": lane k starts 0.4 k seconds late (RunConfig.lane_stagger_s)
    lanes, mode = int(c0[:-1]), c0[-1]
    unet.Denoiser.OVERLAP = mode == "o"      # (with lanes > 1 the product keeps one stream per chunk whatever this says: profiles/r4_schedule_ab_*.txt
                                             #  were measured before that rule went in)
    model.run = replace(run, concurrent_chunks=lanes, lane_stagger_s=float(stag or 0.0))
    torch.cuda.synchronize(); t0 = time.time()
    out, _ = model.forward_device(fr, pr, mk, T, 0, steps=steps, scheduler="ddim")
    torch.cuda.synchronize(); dt = time.time() - t0
    same = "" if ref is None else f" identical={bool(torch.equal(out, ref))}"
    ref = out if ref is None else ref
    print(f"{c}{'/' + vb if vb else ''}{'%' + mask if mask else ''}: {dt / K:8.3f} s/chunk  {(chunk - overlap) * K / dt:.4f} frames/s  ({K} chunks){same}", flush=True)
