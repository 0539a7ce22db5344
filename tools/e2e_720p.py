import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
import diffuerase
T, H, W = 56, 720, 1280
rng = np.random.default_rng(1234)
base = rng.integers(0, 256, (H + 2 * T, W + 2 * T, 3), dtype=np.uint8)
frames = [np.ascontiguousarray(base[t: t + H, 2 * t: 2 * t + W]) for t in range(T)]
masks = []
for t in range(T):
    m = np.zeros((H, W, 3), np.uint8); m[H // 4: H // 2, W // 4 + 2 * t: W // 2 + 2 * t] = 255; masks.append(m)
log = []; tl = []
t0 = time.time()
out = diffuerase.run_infill_on_frames(frames, masks, mask_dilation_iter=8, max_img_size=1280, prog=lambda p, s, *a: (log.append((p, s)), tl.append((round(time.time() - t0, 1), p))), num_inference_steps=4)
dt = time.time() - t0
assert len(out) == T and all(o.shape == (H, W, 3) and o.dtype == np.uint8 for o in out)
unm = np.stack([m[..., 0] for m in masks]) == 0
# far from the (dilated + feathered) hole the original pixels are kept exactly
far = np.zeros_like(unm); far[:, : H // 8, :] = True
assert (np.stack(out)[far] == np.stack(frames)[far]).all()
print(tl[:12], tl[-3:]); print(f"e2e ok: {T} frames 720p, prior + 4 steps + composite in {dt:.1f} s (incl. model build); prog: {log[:3]} ... {log[-1]}")
