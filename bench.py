#!/usr/bin/env python3
"""bench.py -- inpainted frames/sec of the DiffuEraser hot path at 720p / 50 denoise steps on N MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one pass of the hot path over one batch = ONE 32-frame 1280x720 chunk: VAE encode (prior + masked
image) -> 50 DDIM steps of BrushNet + motion-UNet -> VAE decode -> overlap blend -> soft-mask compose, inputs
(uint8 frames / prior / masks, noise) already resident in HBM.  Every rank runs K chunks of one long synthetic video
(weak scaling; contiguous chunk blocks per rank; the 8-frame overlaps at rank boundaries are exchanged over RCCL at
blend time).  Credited output = DISTINCT frames (24 new frames per chunk; overlap recompute is not credited).

The timed region runs the product schedule (RunConfig.concurrent_chunks chunks in flight on their own HIP streams, BrushNet beside the
UNet's down / mid path on a second stream: kernels overlap, so per-kernel durations mean nothing there).  `roofline` / `temporal_block` /
`kernel_times_s` therefore come from a SECOND pass over ONE chunk of the same inputs in the same process, on ONE stream with every MFMA
kernel launch bracketed by HIP events on its launch stream ("kernel_pricing": "single-stream pass"; rank 0 only).

Prints ONE JSON line (rank 0) with the driver's contract fields + `roofline` (dominant kernel) + `cpu_baseline` (the fp32 oracle on the
host cores, bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = 2500.0     # dense bf16/fp16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md (chip-level parameters)
HBM_PEAK_GBS = 8000.0


def synth_frame(t, H, W, seed=1234):
    """One frame of the SURVEY 8d synthetic clip, seeded by its index (so a rank can build just the frames it needs)."""
    rng = np.random.default_rng(seed * 1000003 + t)
    frame = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    mask = np.zeros((H, W), np.uint8)
    rh, rw = H // 4, W // 4
    x0 = (W // 8 + 2 * t) % (W - rw)
    mask[H // 3: H // 3 + rh, x0: x0 + rw] = 255
    prior = frame.copy()
    prior[mask > 0] = frame.reshape(-1, 3).mean(0).astype(np.uint8)
    return frame, mask, prior


def strong_scaling(args, model, dist, rank, world, H, W, ucfg, vcfg):
    """--frames T: ONE fixed T-frame clip (c3: 256 x 720p, c4: 1024 x 1080p) sharded over the ranks, timed host memory ->
    host memory: upload of this rank's frames, VAE + denoise of its chunks, overlap exchange + blend over RCCL,
    compose, collection of all uint8 frames in rank 0's host memory."""
    from videovanish_amd import flops
    from videovanish_amd.pipeline import chunk_plan, shard_chunks
    T = args.frames
    plan = chunk_plan(T, args.chunk, args.overlap)
    mine = shard_chunks(len(plan), world)[rank]
    need = set([0])
    if mine:
        need |= set(range(plan[mine[0]][0], plan[mine[-1]][1]))
    frames, masks, prior = [None] * T, [None] * T, [None] * T
    for t in sorted(need):
        frames[t], masks[t], prior[t] = synth_frame(t, H, W)
    dev = model.ctx.device

    def barrier():
        model._sync()
        if world > 1:
            import torch.distributed as td
            td.barrier()
            model._sync()

    if args.warmup > 0:      # one chunk, untimed: kernels loaded, allocator warm (and one RCCL round trip)
        fr = torch.from_numpy(np.stack([frames[0]] * args.chunk)).to(dev)
        mk = torch.from_numpy(np.stack([masks[0]] * args.chunk)).to(dev)
        model.forward_device(fr, fr, mk, args.chunk, 0, steps=min(2, args.denoise_steps), scheduler="ddim")
        del fr, mk
        if world > 1:
            import torch.distributed as td
            x = torch.zeros(1, device=dev)
            td.all_reduce(x)
    reps = max(1, args.steps)
    tms = []
    barrier()
    t0 = time.time()
    for _ in range(reps):
        tm = {}
        out = model.forward(frames, masks, prior, max_img_size=max(H, W), steps=args.denoise_steps, scheduler="ddim", dist=dist,
                            gather="rank0", timings=tm)
        tms.append(tm)
    barrier()
    dt = time.time() - t0
    per_rank = [tms[-1]]
    backend, ranks_seen = "none", 1
    if world > 1:
        import torch.distributed as td
        ranks_seen = td.get_world_size()
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        td.all_reduce(tt, op=td.ReduceOp.MAX)
        dt = float(tt.item())
        per_rank = [None] * world
        td.all_gather_object(per_rank, tms[-1])      # a few floats of control-plane data (the frames travelled as tensors)
        backend = td.get_backend()
    if rank != 0:
        return None
    assert all(o is not None and o.shape == (H, W, 3) and o.dtype == np.uint8 for o in out)
    digest = None
    if args.dry_run:
        import hashlib
        digest = hashlib.sha256(np.stack(out).tobytes()).hexdigest()
    fl = flops.per_output_frame(H, W, args.chunk, args.denoise_steps, ucfg, vcfg) * args.chunk * len(plan) * reps
    ideal = len(plan) / max(len(s) for s in shard_chunks(len(plan), world))
    return {
        "metric": f"inpainted frames/sec at {H}p, {args.denoise_steps} denoise steps", "value": round(T * reps / dt, 5), "unit": "frames/s", "n_gpus": world,
        "steps": reps, "warmup": args.warmup, "ms_per_step": round(dt / reps * 1e3, 2), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"{ {720: 'c3', 1080: 'c4', 480: 'c2'}.get(H, 'custom') }: fixed {T}-frame {W}x{H} clip, {args.denoise_steps} DDIM steps, "
                               f"{args.chunk}/{args.overlap} chunk/overlap = {len(plan)} chunks sharded {[len(s) for s in shard_chunks(len(plan), world)]}, "
                               f"timed host memory -> host memory (rank 0 collects every frame), {args.arch} width, random-init weights",
                   "frames": T, "chunks": len(plan), "parallelism": f"chunk-dp{world}", "ideal_speedup_from_chunk_quantisation": round(ideal, 3),
                   "precise_decoder": bool(args.precise_decoder)},
        "job_tflops": round(fl / dt / 1e12, 1), "collective_backend": backend, "ranks_seen": ranks_seen,
        "per_rank_seconds": per_rank, **({"dry_run": True, "output_sha256": digest} if args.dry_run else {}),
    }


def kernel_table(prof):
    """hip.PROFILE records -> {key: [launches, seconds, flops, algorithmic bytes]}"""
    kernels = {}
    for key, fl, nbytes, e0, e1 in prof:
        k = kernels.setdefault(key, [0, 0.0, 0.0, 0.0])
        k[0] += 1; k[1] += e0.elapsed_time(e1) * 1e-3; k[2] += fl; k[3] += nbytes
    return kernels


def full_pipeline_c5(args, run, rank, world, dist, H, W, ucfg, vcfg):
    """BASELINE config 5 (--prior raft --dilate K): the WHOLE drop-in call diffuerase.run_infill_on_frames -- mask collapse + dilation, RAFT
    (20 GRU iterations) + flow-guided propagation prior, VAE / denoise / decode per chunk, overlap blend, compose, resize-back, feathered
    composite -- on one synthetic clip of steps x 24 + 8 frames, timed host memory -> host memory.  Credited frames = 24 per chunk like the
    default line.  The priced pass (one stream, one 32-frame clip, HIP events per launch) gives the prior's kernels -- all-pairs correlation
    GEMM, pyramid pooling, correlation lookup, GRU element-wise kernels, convex upsampling, the bilinear warps of the propagation -- with
    their achieved rate against the HBM (or MFMA) peak; algorithmic bytes per launch as SURVEY 8(d) defines them (operands read once +
    result written once; the lookup: 4 levels x 10 x 10 window samples x 4 B + 324 x 2 B per pixel and iteration)."""
    import diffuerase
    from videovanish_amd import flops, hip
    stride = args.chunk - args.overlap
    K = max(1, args.steps)
    T = stride * K + args.overlap
    ref = bool(args.reference_defaults)
    diffuerase.configure(run=run, dist=dist, gather="rank0", reference_defaults=ref)
    steps_timed = None if ref else args.denoise_steps

    def clip(n, t0=0):
        fr, mk, _ = synth_clip(n, H, W, seed=1234, t0=t0)
        return list(fr), [np.repeat(m[..., None], 3, axis=2) for m in mk]      # the GUI hands over 3-channel mask frames (reference :29)

    def call(frames, masks, steps):
        if ref:      # every default of the reference call (diffuerase.py:20-21): 2-step TCD, max_img_size 960, dilation 8, feather 3
            return diffuerase.run_infill_on_frames(frames, masks, mask_dilation_iter=args.dilate)
        return diffuerase.run_infill_on_frames(frames, masks, mask_dilation_iter=args.dilate, propainer_frames=None, max_img_size=max(H, W),
                                               num_inference_steps=steps, scheduler="ddim")

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as td
            td.barrier()
            torch.cuda.synchronize()

    if args.warmup > 0:      # one chunk, 2 denoise steps: every kernel of the path loaded, models built, allocator warm
        fw, mw = clip(args.chunk)
        call(fw, mw, min(2, args.denoise_steps))
    frames, masks = clip(T)
    power = PowerTrace(int(os.environ.get("LOCAL_RANK", "0"))) if (rank == 0 and not args.no_power_trace) else None
    barrier()
    t0 = time.time()
    if power is not None:
        power.__enter__()
    out = call(frames, masks, steps_timed)
    barrier()
    dt = time.time() - t0
    if power is not None:
        power.__exit__()
    if world > 1:
        import torch.distributed as td
        tt = torch.tensor([dt], dtype=torch.float64, device=torch.device("cuda", torch.cuda.current_device()))
        td.all_reduce(tt, op=td.ReduceOp.MAX)
        dt = float(tt.item())
    if rank != 0:
        return None
    assert len(out) == T and all(o is not None and o.shape == (H, W, 3) and o.dtype == np.uint8 for o in out)
    # ---- priced pass: one 32-frame clip through the same call on one stream
    prior_tab, prior_s, priced_s, dom = None, None, None, None
    if not args.no_kernel_events:
        fw, mw = clip(args.chunk)
        torch.cuda.synchronize()
        hip.PROFILE = []
        t1 = time.time()
        call(fw, mw, steps_timed)
        torch.cuda.synchronize()
        priced_s = time.time() - t1
        prof, hip.PROFILE = hip.PROFILE, None
        kernels = kernel_table(prof)
        prior_s = sum(v[1] for k, v in kernels.items() if k.startswith("prior:"))
        prior_tab = {}
        for k, v in sorted(kernels.items(), key=lambda kv: -kv[1][1]):
            if not (k.startswith("prior:") or k in ("mask_collapse_dilate", "feather_composite", "resize_u8")):
                continue
            n, tsec, fl, by = v
            if fl > 0:
                prior_tab[k] = {"launches": n, "seconds": round(tsec, 4), "bound": "mfma", "achieved_tflops": round(fl / tsec / 1e12, 1), "frac": round(fl / tsec / 1e12 / MFMA_PEAK_TFLOPS, 4)}
            else:
                prior_tab[k] = {"launches": n, "seconds": round(tsec, 4), "bound": "hbm", "achieved_gbs": round(by / tsec / 1e9, 1), "frac": round(by / tsec / 1e9 / HBM_PEAK_GBS, 4),
                                "algorithmic_bytes_per_launch": by / n}
        dk = max(kernels, key=lambda k: kernels[k][1])
        n, tsec, fl, by = kernels[dk]
        dom = {"kernel": dk, "bound": "mfma" if fl > 0 else "hbm", "achieved": round((fl / tsec / 1e12) if fl > 0 else (by / tsec / 1e9), 2),
               "peak": MFMA_PEAK_TFLOPS if fl > 0 else HBM_PEAK_GBS, "unit": "TFLOP/s" if fl > 0 else "GB/s",
               "frac": round(((fl / tsec / 1e12) / MFMA_PEAK_TFLOPS) if fl > 0 else ((by / tsec / 1e9) / HBM_PEAK_GBS), 4), "traffic": None,
               "launches": n, "avg_launch_ms": round(tsec / n * 1e3, 4), "share_of_step_time": round(tsec / priced_s, 3)}
    credited = stride * K * world
    return {
        "metric": f"inpainted frames/sec at {H}p, {'2 (TCD, reference defaults)' if ref else args.denoise_steps} denoise steps", "value": round((T if ref else credited) / dt, 5), "unit": "frames/s", "n_gpus": world,
        "steps": K, "warmup": args.warmup, "ms_per_step": round(dt / K * 1e3, 2), "higher_is_better": True, "scaling": "weak" if world == 1 else "strong",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": (f"reference defaults (diffuerase.configure(reference_defaults=True)): {T}-frame {W}x{H} clip, every default of the reference call -- dilation {args.dilate}, "
                                f"complete ProPainter prior (RAFT + flow completion + propagation + generator), max_img_size 960, 2-step TCD over the third-party pipeline's 22-frame "
                                f"windows, feathered composite; all {T} frames credited; timed host memory -> host memory, {args.arch} width, random-init weights") if ref else
                               (f"c5: {T}-frame {W}x{H} clip through diffuerase.run_infill_on_frames with NO prior handed over: mask dilation {args.dilate} + RAFT (20 it.) "
                                f"flow-propagation prior + {args.denoise_steps} DDIM steps per {args.chunk}/{args.overlap} chunk + blend + compose + feathered composite, "
                                f"timed host memory -> host memory, {args.arch} width, random-init weights"),
                   "frames": T, "credited_frames": credited, "parallelism": f"chunk-dp{world}", "precise_decoder": bool(args.precise_decoder)},
        "roofline": dom, "prior": None if prior_tab is None else {"seconds_per_32_frames": round(prior_s, 3), "share_of_priced_clip": round(prior_s / priced_s, 4),
                                                                   "kernels": prior_tab},
        "kernel_pricing": None if priced_s is None else {"how": "single-stream pass", "frames": args.chunk, "seconds": round(priced_s, 3)},
        "cpu_baseline": None, "power": power.summary() if power is not None else None,
        "job_tflops": round(flops.per_output_frame(H, W, args.chunk, args.denoise_steps, ucfg, vcfg) * args.chunk * K * world / dt / 1e12, 1),
    }


class PowerTrace:
    """Socket power / shader clock of one GPU while the timed region runs, sampled by `rocm-smi` from a side thread (one short-lived child process
    every ~0.5 s; the timed region only launches kernels, nothing of it waits on this).  Context for `roofline`: the job runs at the board's power cap
    (DESIGN.md 5.00), so the nominal MFMA peak is not reachable whatever the kernels do.  Every failure (no rocm-smi, no permission, odd output) ends
    in `None`: the bench line never depends on it."""

    def __init__(self, device_index):
        import threading
        self.dev, self.samples, self._stop = int(device_index), [], threading.Event()
        self._thread = threading.Thread(target=self._run, name="vv-power-trace", daemon=True)

    def _run(self):
        import re, subprocess
        while not self._stop.is_set() and len(self.samples) < 4000:
            try:
                out = subprocess.run(["rocm-smi", "-d", str(self.dev), "--showpower", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
                pw = re.search(r"Package Power \(W\):\s*([0-9.]+)", out)
                ck = re.search(r"sclk clock level:[^(]*\((\d+)Mhz\)", out)
                if pw:
                    self.samples.append((float(pw.group(1)), int(ck.group(1)) if ck else 0))
            except Exception:
                return
            self._stop.wait(0.3)

    def __enter__(self):
        self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        self._thread.join(timeout=15)

    def summary(self):
        if len(self.samples) < 4:
            return None
        pw = [p for p, _ in self.samples]
        cap = None
        try:
            import re, subprocess
            m = re.search(r"Package Power \(W\):\s*([0-9.]+)", subprocess.run(["rocm-smi", "-d", str(self.dev), "--showmaxpower"], capture_output=True, text=True, timeout=10).stdout)
            cap = float(m.group(1)) if m else None
        except Exception:
            cap = None
        return {"source": "rocm-smi --showpower --showclocks, one sample every ~0.5 s during the timed region (rank 0's GPU)", "samples": len(pw),
                "mean_w": round(sum(pw) / len(pw), 1), "max_w": max(pw), "cap_w": cap, "share_of_samples_at_or_above_1200_w": round(sum(p >= 1200 for p in pw) / len(pw), 3),
                "mean_sclk_mhz": round(sum(c for _, c in self.samples) / len(self.samples))}


def parity_summary():
    """The measured parity lines of the default precision plan, read from the newest committed GPU parity log
    (profiles/r*_parity_gpu.txt, written by `VV_PARITY_REPORT=... pytest -m gpu tests/test_configs_gpu.py`): nothing is hard-coded here."""
    import glob
    import re
    logs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_parity_gpu.txt")), key=lambda p: int(re.search(r"r(\d+)_", os.path.basename(p)).group(1)))
    if not logs:
        return None
    keep = {}
    for line in open(logs[-1]):
        m = re.match(r"(parity50\[[^\]]*precise-decoder\]|c1_full_width\[[^\]]*\]) pixel max_abs=([0-9.e+-]+)", line)
        if m:
            keep[m.group(1)] = float(m.group(2))      # the last occurrence of a key wins (the log is appended per run)
    return {"source": os.path.relpath(logs[-1], ROOT), "per_pixel_max_abs_vs_fp32_oracle": keep, "bound": 1.0e-3}


def synth_clip(T, H, W, seed=1234, t0=0):
    """SURVEY 8d synthetic inputs: uniform random frames, one 25% x 25% rectangle drifting +2 px/frame, prior = frame with
    the masked pixels replaced by the per-frame mean colour (bypasses ProPainter exactly as diffuerase.py:47 allows)."""
    rng = np.random.default_rng(seed)
    frames = rng.integers(0, 256, (T, H, W, 3), dtype=np.uint8)
    masks = np.zeros((T, H, W), np.uint8)
    rh, rw = H // 4, W // 4
    for t in range(T):
        x0 = (W // 8 + 2 * (t0 + t)) % (W - rw)
        masks[t, H // 3: H // 3 + rh, x0: x0 + rw] = 255
    prior = frames.copy()
    for t in range(T):
        prior[t][masks[t] > 0] = frames[t].reshape(-1, 3).mean(0).astype(np.uint8)
    return frames, masks, prior


def cpu_baseline(H, W, steps, chunk, overlap, sample_hw=(256, 384), vae_hw=(192, 256), sample_frames=4):
    """The oracle (kind "port": fp32 torch restatement) timed on the host cores on a BOUNDED sample of the same workload (~10-20 s of
    CPU work): one BrushNet+UNet denoise step on a 4-frame clip at 384x256 (the temporal modules attend over the 4 frames) and one VAE
    encode+decode at 256x192 (full-width architecture, same weights), converted to CPU TFLOP/s with the algorithmic FLOP model and then
    to frames/s of the 720p/50-step job."""
    from oracle import model_ref as M
    from videovanish_amd import flops
    from videovanish_amd.config import UNetConfig, VAEConfig
    ucfg, vcfg = UNetConfig(), VAEConfig()
    # threads: SURVEY 8(d) says os.cpu_count(); on the GPU box (128+ logical CPUs) the small-tensor oracle is SLOWER with one thread per CPU than with a
    # modest pool, so 32 are used and both numbers are stated (`cores` = threads used, `host_cpus` = os.cpu_count()); `extrapolated`: the sample is one
    # denoise step + one VAE pass at reduced size, scaled to the job by the algorithmic FLOP model -- a baseline, not a measurement of the full job
    cores = min(32, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    P = M.Params(0)
    sh, sw = sample_hw
    h, w = sh // 8, sw // 8
    g = torch.Generator().manual_seed(0)
    lat = torch.randn(sample_frames, 4, h, w, generator=g)
    x9 = torch.cat([lat, lat, torch.ones(sample_frames, 1, h, w)], 1)
    text = M.text_states(P, ucfg)
    img = torch.rand(1, 3, vae_hw[0], vae_hw[1], generator=g) * 2 - 1
    with torch.no_grad():
        # materialise the weights outside the timed sample (a tiny input touches every layer)
        small = torch.randn(1, 4, 8, 8, generator=g)
        M.unet_forward(P, small, 500, text, ucfg, M.brushnet_forward(P, torch.cat([small, small, torch.ones(1, 1, 8, 8)], 1), 500, text, ucfg))
        z0 = M.vae_encode(P, torch.zeros(1, 3, 16, 16), vcfg); M.vae_decode(P, z0, vcfg)
        t0 = time.time()
        M.unet_forward(P, lat, 500, text, ucfg, M.brushnet_forward(P, x9, 500, text, ucfg))
        t_step = time.time() - t0
        t0 = time.time()
        z = M.vae_encode(P, img, vcfg)
        M.vae_decode(P, z, vcfg)
        t_vae = time.time() - t0
    enc, dec = flops.vae_per_frame(vae_hw[0], vae_hw[1], vcfg)
    fl = flops.denoise_step_per_frame(h, w, sample_frames, ucfg) * sample_frames + enc + dec
    tfs = fl / (t_step + t_vae) / 1e12
    per_frame = flops.per_output_frame(H, W, chunk, steps, ucfg, vcfg) * chunk / float(chunk - overlap)
    return {"value": tfs * 1e12 / per_frame, "unit": "frames/s", "cores": cores, "host_cpus": os.cpu_count(), "kind": "port", "extrapolated": True, "cpu_tflops": round(tfs, 4),
            "sample": f"{sample_frames}-frame clip: one denoise step at {sw}x{sh} ({t_step:.1f}s) + one VAE encode+decode at {vae_hw[1]}x{vae_hw[0]} ({t_vae:.1f}s), "
                      f"{fl / 1e12:.2f} TFLOP; scaled by the algorithmic FLOP model to {steps} steps + 2 enc + 1 dec per {W}x{H} frame, "
                      f"x{chunk / float(chunk - overlap):.2f} chunk overlap"}


def self_launch(n, argv, dry_run=False):
    """`python bench.py --gpus N ...` started WITHOUT a launcher (no WORLD_SIZE in the environment): start N fresh child processes of this script,
    one rank per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set, the contract torch.distributed.run gives its
    children), pass rank 0's stdout (the JSON line) through, forward the other ranks' stdout to stderr, and return the children's worst exit code.
    The parent makes NO GPU call before or after (importing torch and counting devices do not initialise HIP); nothing is ever exec'd over a
    process that has touched the GPU.  If a rank dies the others are ended by their exact PIDs (they would wait in a collective forever)."""
    import socket
    import subprocess
    import threading
    if not dry_run:
        have = torch.cuda.device_count()      # (counting devices does not initialise the GPU on this image)
        if have < n:
            print(f"bench.py --gpus {n}: only {have} GPU(s) visible on this node", file=sys.stderr)
            return 2
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        if dry_run:
            env.setdefault("OMP_NUM_THREADS", "1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env, stdout=subprocess.PIPE, text=True))

    def pump(p, sink):
        for line in p.stdout:
            sink.write(line)
            sink.flush()

    pumps = [threading.Thread(target=pump, args=(p, sys.stdout if r == 0 else sys.stderr), daemon=True) for r, p in enumerate(procs)]
    for t in pumps:
        t.start()
    codes = [None] * n
    while any(c is None for c in codes):
        for r, p in enumerate(procs):
            if codes[r] is None:
                codes[r] = p.poll()
        if any(c not in (None, 0) for c in codes):
            for r, p in enumerate(procs):      # a rank failed: end the rest (exact PIDs), they would hang in the next collective
                if codes[r] is None:
                    p.terminate()
            for r, p in enumerate(procs):
                if codes[r] is None:
                    try:
                        codes[r] = p.wait(timeout=30)
                    except subprocess.TimeoutExpired:
                        p.kill()
                        codes[r] = p.wait()
            break
        time.sleep(0.2)
    for t in pumps:
        t.join(timeout=10)
    bad = [c for c in codes if c != 0]
    if bad:
        print(f"bench.py --gpus {n}: rank exit codes {codes}", file=sys.stderr)
        return max(abs(c) for c in bad) or 1
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--chunk", type=int, default=32)
    ap.add_argument("--overlap", type=int, default=8)
    ap.add_argument("--denoise-steps", type=int, default=50)
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "bf16"], help="MFMA operand type (fp16 = the precision plan whose 50-step "
                    "parity is <= 1e-3; bf16 + --no-precise-decoder is ~5 %% faster at 1e-2)")
    ap.add_argument("--arch", default="full", choices=["full", "small", "tiny"])
    ap.add_argument("--frames", type=int, default=0, help="strong-scaling mode: one fixed clip of this many frames sharded over the ranks, "
                    "timed host memory -> host memory (c3: --frames 256; c4: --frames 1024 --height 1080 --width 1920)")
    ap.add_argument("--no-precise-decoder", dest="precise_decoder", action="store_false",
                    help="VAE decoder in one pass of h16 operands instead of split precision (3 MFMA passes per GEMM)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--no-power-trace", action="store_true", help="do not sample rocm-smi beside the timed region (the `power` object of the line)")
    ap.add_argument("--lanes", type=int, default=None, help="chunks of one rank in flight at once, each on its own HIP stream "
                    "(RunConfig.concurrent_chunks; default: the product default)")
    ap.add_argument("--one-stream", action="store_true", help="A/B: the round-3 schedule (one chunk at a time, BrushNet and UNet on one stream)")
    ap.add_argument("--dry-run", action="store_true", help="host logic only: gloo instead of RCCL, CPU tensors, pipeline.DryRunEraser instead of the model "
                    "(no kernels, nothing is inpainted, every number of the line is meaningless) -- the sharding / exchange / gather / timing "
                    "plumbing of the N-GPU lines exercised without a GPU (tests/test_dist_cpu.py); adds `output_sha256`")
    ap.add_argument("--prior", default="none", choices=["none", "raft"], help="raft = BASELINE config 5: no prior is handed over; the RAFT (20 iterations) + "
                    "flow-guided propagation prior, the mask dilation and the feathered composite run INSIDE the timed region through "
                    "diffuerase.run_infill_on_frames, host memory -> host memory (reference diffuerase.py:27-31,47-57,69-112)")
    ap.add_argument("--reference-defaults", action="store_true", help="with --prior raft: the regime the reference app runs by default (one switch: "
                    "diffuerase.configure(reference_defaults=True)): its pipeline's own 22-frame windows, the 2-step TCD schedule, max_img_size 960, the "
                    "complete ProPainter prior (flow completion + generator, random-init weights) -- one host-to-host line for the GUI user's path")
    ap.add_argument("--dilate", type=int, default=8, help="--prior raft: mask_dilation_iter of the drop-in call (reference default 8)")
    ap.add_argument("--dump-kernels", default=None, help="write the raw per-kernel table (launches, seconds, flops, bytes) to this JSON file")
    ap.add_argument("--profile-shapes", action="store_true", help="per-kernel keys carry the GEMM / attention shapes (M, N, K): tools/shape_table.py")
    args = ap.parse_args()
    if args.one_stream:
        from videovanish_amd import unet as _unet
        _unet.Denoiser.OVERLAP = False
        args.lanes = 1

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: this process becomes the launcher (it has made no GPU call and makes none)
        raise SystemExit(self_launch(args.gpus, sys.argv[1:], dry_run=args.dry_run))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} was started with WORLD_SIZE={world}: one rank per GPU (launch with torch.distributed.run "
                         f"--nproc-per-node {args.gpus}, or start it without WORLD_SIZE in the environment and it spawns its own ranks)")
    if not args.dry_run:
        torch.cuda.set_device(local_rank)
    dist = None
    # VV_BENCH_FORCE_DIST=1: take the multi-rank code path (process group, barriers, all_reduce / all_gather_object of the records) even at world 1 --
    # the RCCL first-contact test that one GPU allows (tests/test_fullsize_gpu.py::test_bench_line_with_rccl_initialised)
    multi = world > 1 or os.environ.get("VV_BENCH_FORCE_DIST") == "1"
    if multi:
        import torch.distributed as td
        if args.dry_run:
            td.init_process_group("gloo")
        else:
            import datetime
            # (the default collective watchdog is 10 minutes; a rank's block of 50-step chunks between two collectives can be longer at 1080p: one hour)
            td.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(minutes=60))
        dist = (rank, world)
        if args.dry_run and os.environ.get("VV_DRYRUN_FAIL_RANK") == str(rank):      # tests/test_dist_cpu.py: one rank dies after the rendezvous
            raise RuntimeError(f"dry run: rank {rank} told to fail (VV_DRYRUN_FAIL_RANK)")

    from videovanish_amd import hip
    hip.PROFILE_SHAPES = bool(args.profile_shapes)
    from videovanish_amd.config import SMALL_UNET, SMALL_VAE, TINY_UNET, TINY_VAE, RunConfig, UNetConfig, VAEConfig
    from videovanish_amd.pipeline import DiffuEraserHIP, chunk_plan, shard_chunks
    ucfg, vcfg = {"full": (UNetConfig(), VAEConfig()), "small": (SMALL_UNET, SMALL_VAE), "tiny": (TINY_UNET, TINY_VAE)}[args.arch]
    run = RunConfig(steps=args.denoise_steps, chunk=args.chunk, overlap=args.overlap, seed=42, weight_seed=0, dtype=args.dtype, unet=ucfg, vae=vcfg,
                    precise_decoder=args.precise_decoder, **({"concurrent_chunks": args.lanes} if args.lanes else {}))
    if args.prior == "raft":      # BASELINE config 5: the whole drop-in call, prior included (builds its own models through the drop-in's cache)
        res = full_pipeline_c5(args, run, rank, world, dist, args.height, args.width, ucfg, vcfg)
        if rank == 0:
            print(json.dumps(res))
        if multi:
            import torch.distributed as td
            td.destroy_process_group()
        return
    t_build = time.time()
    if args.dry_run:
        from videovanish_amd.pipeline import DryRunEraser
        model = DryRunEraser(run, "cpu")
        args.no_kernel_events = args.no_cpu_baseline = args.no_power_trace = True      # no kernels run: nothing to price, nothing to compare
    else:
        model = DiffuEraserHIP(run, f"cuda:{local_rank}")
        torch.cuda.synchronize()
    t_build = time.time() - t_build
    H, W = args.height, args.width
    stride = args.chunk - args.overlap
    dev = model.ctx.device
    if args.frames > 0:
        res = strong_scaling(args, model, dist, rank, world, H, W, ucfg, vcfg)
        if rank == 0:
            res["config"]["model_build_s"] = round(t_build, 1)
            print(json.dumps(res))
        if multi:
            import torch.distributed as td
            td.destroy_process_group()
        return

    host_inputs = []

    def resident_inputs(n_chunks_per_rank):
        """This rank's slice of a (world * n) -chunk synthetic video, uploaded BEFORE the timed region."""
        T = stride * n_chunks_per_rank * world + args.overlap
        plan = chunk_plan(T, args.chunk, args.overlap)
        assert len(plan) == n_chunks_per_rank * world
        mine = shard_chunks(len(plan), world)[rank]
        base, end = plan[mine[0]][0], plan[mine[-1]][1]
        # seeded per FRAME INDEX (synth_frame), so the video -- and with it every output frame -- does not depend on how many ranks share it
        trip = [synth_frame(t, H, W) for t in range(base, end)]
        fr, mk, pr = (np.stack([x[i] for x in trip]) for i in range(3))
        del trip
        host_inputs[:] = [fr, pr, mk]
        return T, base, torch.from_numpy(fr).to(dev), torch.from_numpy(pr).to(dev), torch.from_numpy(mk).to(dev)

    def barrier():
        model._sync()
        if multi:
            import torch.distributed as td
            td.barrier()
            model._sync()

    if args.warmup > 0:
        T, base, fr, pr, mk = resident_inputs(args.warmup)
        model.forward_device(fr, pr, mk, T, base, steps=args.denoise_steps, scheduler="ddim", dist=dist)
        del fr, pr, mk
    T, base, fr, pr, mk = resident_inputs(args.steps)
    power = PowerTrace(local_rank) if (rank == 0 and not args.no_power_trace) else None
    barrier()
    t0 = time.time()
    if power is not None:
        power.__enter__()
    tm = {} if multi else None      # (N > 1 only: one device sync either side of the blend-time exchange, to report its seconds per rank)
    out, (lo, hi) = model.forward_device(fr, pr, mk, T, base, steps=args.denoise_steps, scheduler="ddim", dist=dist, timings=tm)
    model._sync()
    dt_mine = time.time() - t0           # this rank's own seconds (before it waits for the slowest rank in the barrier)
    barrier()
    dt = time.time() - t0
    if power is not None:
        power.__exit__()
    # what the collective layer itself reports: a launcher that silently started fewer ranks, or a backend that is not RCCL, shows up in the line
    backend, ranks_seen, per_rank = "none", 1, [{"rank": 0, "seconds": round(dt_mine, 3), "chunks": args.steps, "owned_frames": hi - lo}]
    if multi:
        import torch.distributed as td
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        td.all_reduce(tt, op=td.ReduceOp.MAX)
        dt = float(tt.item())
        backend, ranks_seen = td.get_backend(), td.get_world_size()
        per_rank = [None] * world
        sent = model.last_blend.sent_bytes if model.last_blend is not None else 0
        td.all_gather_object(per_rank, {"rank": rank, "seconds": round(dt_mine, 3), "chunks": args.steps, "owned_frames": hi - lo,
                                        "exchange_blend_s": round(tm.get("exchange_blend_s", 0.0), 4), "overlap_bytes_sent": int(sent)})
    frame_sha = None
    if args.dry_run:      # per-frame fingerprints of what every rank owns (dry run only): the N-rank weak line computes the frames of the one-rank line
        import hashlib
        mine_sha = {lo + j: hashlib.sha256(out[j].numpy().tobytes()).hexdigest()[:16] for j in range(hi - lo)}
        alls = [mine_sha]
        if multi:
            alls = [None] * world
            td.all_gather_object(alls, mine_sha)
        frame_sha = [v for _, v in sorted((k, v) for d in alls for k, v in d.items())]
        assert len(frame_sha) == T, "dry run: the ranks' owned frames do not tile the video"
    distinct = stride * args.steps * world          # credited frames: every chunk contributes (chunk - overlap) new frames
    assert out is not None and out.dtype == torch.uint8 and out.shape[1:] == (H, W, 3)
    # SURVEY 8(d) defines the metric host memory -> host memory; `value` keeps the inputs resident (the driver contract), the copies this rank would add
    # are measured right here on the same tensors and reported beside it (`host_to_host`)
    model._sync()
    t_up = time.time()
    ups = [torch.from_numpy(a).to(dev) for a in host_inputs]
    model._sync()
    t_up = time.time() - t_up
    del ups
    t_down = time.time()
    out_host = out.cpu()
    t_down = time.time() - t_down
    del out_host

    if rank != 0:
        if multi:
            import torch.distributed as td
            td.destroy_process_group()
        return
    # ---- pricing pass (rank 0, after the timed region): ONE chunk of the same inputs on ONE stream, every launch bracketed by HIP events
    prof, dt_priced = None, None
    if not args.no_kernel_events:
        one = chunk_plan(T, args.chunk, args.overlap)[shard_chunks(len(chunk_plan(T, args.chunk, args.overlap)), world)[rank][0]]
        a, b = one[0] - base, one[1] - base
        torch.cuda.synchronize()
        hip.PROFILE = []
        t1 = time.time()
        model.forward_device(fr[a:b], pr[a:b], mk[a:b], args.chunk, 0, steps=args.denoise_steps, scheduler="ddim")
        torch.cuda.synchronize()
        dt_priced = time.time() - t1
        prof, hip.PROFILE = hip.PROFILE, None
    # ---- roofline of the dominant kernel (HIP events recorded on the launch stream in the pricing pass)
    roof = None
    kernels = {}
    if prof:
        for key, flops, nbytes, e0, e1 in prof:
            k = kernels.setdefault(key, [0, 0.0, 0.0, 0.0])
            k[0] += 1; k[1] += e0.elapsed_time(e1) * 1e-3; k[2] += flops; k[3] += nbytes
        dom = max(kernels, key=lambda k: kernels[k][1])
        n, tsec, fl, by = kernels[dom]
        mfma = fl > 0
        ach = (fl / tsec / 1e12) if mfma else (by / tsec / 1e9)
        peak = MFMA_PEAK_TFLOPS if mfma else HBM_PEAK_GBS
        traffic = None
        try:   # HBM bytes per launch from the committed PMC passes (profiles/traffic_table.json; rocprofv3 --pmc)
            traffic = json.load(open(os.path.join(ROOT, "profiles", "traffic_table.json"))).get(dom, {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
        roof = {"kernel": dom, "bound": "mfma" if mfma else "hbm", "achieved": round(ach, 2), "peak": peak,
                "unit": "TFLOP/s" if mfma else "GB/s", "frac": round(ach / peak, 4), "traffic": traffic,
                "traffic_source": "profiles/traffic_table.json (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command; gfx950 x2 FETCH correction)" if traffic else None,
                "launches": n,
                "avg_launch_ms": round(tsec / n * 1e3, 4), "algorithmic_per_launch": (fl if mfma else by) / n,
                "share_of_step_time": round(tsec / dt_priced, 3)}
    # ---- the north star's own target: the fused temporal block (21 motion modules) against the MFMA roofline
    temporal = None
    t_motion = sum(v[1] for k, v in kernels.items() if k.startswith("motion:"))
    if t_motion > 0:
        from videovanish_amd import flops as _fl
        f8 = 2 ** (len(vcfg.block_out) - 1)
        fl_motion = _fl.temporal_block_per_frame(H // f8, W // f8, args.chunk, ucfg) * args.chunk * args.denoise_steps      # the priced pass is ONE chunk
        temporal = {"tflops": round(fl_motion / t_motion / 1e12, 1), "frac_of_mfma_peak": round(fl_motion / t_motion / 1e12 / MFMA_PEAK_TFLOPS, 4),
                    "seconds_per_step": round(t_motion, 3), "share_of_step_time": round(t_motion / dt_priced, 3),
                    "note": "all kernels launched by the motion modules (GroupNorm, LayerNorm+pos-emb, projections, temporal attention core, GEGLU FF)"}
    cpu = None
    if not args.no_cpu_baseline and world == 1:      # reported on rank 0 at N=1 only
        cpu = cpu_baseline(H, W, args.denoise_steps, args.chunk, args.overlap)
    res = {
        "metric": f"inpainted frames/sec at {H}p, {args.denoise_steps} denoise steps", "value": round(distinct / dt, 5), "unit": "frames/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "ranks_seen": ranks_seen, "collective_backend": backend, "per_rank": per_rank,
        **({"dry_run": True, "frame_sha256_16": frame_sha} if args.dry_run else {}),
        "config": {"workload": f"{ {480: 'c2', 720: 'c3', 1080: 'c4'}.get(H, 'custom') } chunk: {args.chunk}-frame {W}x{H} chunk, {args.denoise_steps} DDIM steps, {args.chunk}/{args.overlap} chunk/overlap, "
                               f"{args.arch} SD-1.5 UNet+BrushNet+motion / SD-VAE, random-init weights",
                   "frames_per_step": args.chunk, "credited_frames_per_step": stride, "chunks_per_rank": args.steps, "chunks_total": args.steps * world,
                   "parallelism": f"chunk-dp{world}", "model_build_s": round(t_build, 1), "precise_decoder": bool(args.precise_decoder),
                   "concurrent_chunks": min(model.run.concurrent_chunks, args.steps),
                   "streams_per_chunk": 2 if (min(model.run.concurrent_chunks, args.steps) == 1 and __import__("videovanish_amd.unet", fromlist=["x"]).Denoiser.OVERLAP) else 1,
                   "parity": parity_summary()},
        "kernel_pricing": None if not prof else {"how": "single-stream pass", "chunks": 1, "seconds": round(dt_priced, 3),
                                                 "kernel_seconds": round(sum(v[1] for v in kernels.values()), 3),
                                                 "note": "one chunk of the same resident inputs re-run after the timed region on ONE stream, every launch bracketed "
                                                         "by HIP events on its launch stream; the timed region itself overlaps kernels of several streams"},
        "host_to_host": {"value": round(distinct / (dt + t_up + t_down), 5), "unit": "frames/s", "upload_s": round(t_up, 4), "download_s": round(t_down, 4),
                         "note": "SURVEY 8(d) form of the metric: the same timed region plus this rank's synchronous upload of its uint8 frames / prior / masks and "
                                 "download of its uint8 output (pageable host memory, measured on the same tensors right after the timed region)"},
        "roofline": roof, "temporal_block": temporal, "cpu_baseline": cpu, "power": power.summary() if power is not None else None,
        "job_tflops": round(__import__("videovanish_amd.flops", fromlist=["x"]).per_output_frame(H, W, args.chunk, args.denoise_steps, ucfg, vcfg)
                            * args.chunk * args.steps * world / dt / 1e12, 1),
        "kernel_times_s": {k: [v[0], round(v[1], 3), round(v[2] / v[1] / 1e12, 1) if v[2] else round(v[3] / v[1] / 1e9, 1)] for k, v in
                           sorted(kernels.items(), key=lambda kv: -kv[1][1])},
    }
    if args.dump_kernels:      # raw per-key table (launches, seconds, flops, algorithmic bytes); with --profile-shapes the GEMM keys carry M,N,K
        json.dump(kernels, open(args.dump_kernels, "w"))
    print(json.dumps(res))
    if multi:
        import torch.distributed as td
        td.destroy_process_group()


if __name__ == "__main__":
    main()
