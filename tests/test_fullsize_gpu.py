"""Parity at BASELINE.json's FULL sizes (720p latent 90x160, 32-frame chunk), where the whole oracle is too slow:
each kernel is run on the full-size problem and checked against the fp32 CPU oracle on a slice the oracle finishes in
seconds (one frame / one head / a few hundred rows), plus size-independent properties (run-to-run determinism,
per-frame independence of the per-frame branch)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

FR, H, W = 32, 90, 160
N = H * W


def _r(t, td):
    return t.to(td).float()


def test_spatial_attention_full_size(gpu):
    """N = 14400 tokens, 8 heads x d=40 (UNet level 0 at 720p): one (frame, head) against the full 14400^2 softmax."""
    from videovanish_amd import hip
    td, dt = torch.bfloat16, hip.BF16
    g = torch.Generator().manual_seed(61)
    B, heads, D = 2, 8, 40
    C = heads * D
    qkv = _r(torch.randn(B, N, 3 * C, generator=g), td)
    buf = qkv.to(td).to(gpu)
    out = torch.empty(B, N, C, dtype=td, device=gpu)
    args = dict(B=B, heads=heads, Nq=N, Nkv=N, D=D, q_bs=N * 3 * C, k_bs=N * 3 * C, v_bs=N * 3 * C, o_bs=N * C, q_rs=3 * C, k_rs=3 * C,
                v_rs=3 * C, o_rs=C, k_off=C, v_off=2 * C)
    hip.attention(dt, buf, buf, buf, out, **args)
    out2 = torch.empty_like(out)
    hip.attention(dt, buf, buf, buf, out2, **args)
    assert torch.equal(out, out2)                                   # deterministic
    b, h = 1, 5
    q, k, v = (qkv[b, :, i * C + h * D: i * C + (h + 1) * D] for i in range(3))
    ref = torch.softmax((q @ k.t()) * D ** -0.5, -1) @ v
    got = out[b, :, h * D:(h + 1) * D].float().cpu()
    assert (got - ref).abs().max().item() <= 2 ** -7 * max(1.0, ref.abs().max().item())



@pytest.mark.parametrize("dname,td,ulp", [("bf16", torch.bfloat16, 2 ** -8), ("fp16", torch.float16, 2 ** -11)])
def test_spatial_attention_level1_full_size(gpu, dname, td, ulp):
    """N = 3600 tokens (45 x 80: UNet level 1 at 720p; 3600 = 56 x 64 + 16: ragged last key tile, last query block of 16 queries), 8 heads x d = 80,
    head-major q_prescaled layout as the pipeline stores it: attn80_kernel against an fp64 softmax over every query of two (frame, head) pairs,
    deterministic, and the log-sum-exp output against torch.logsumexp."""
    from videovanish_amd import hip
    dt = hip.dtype_id(dname)
    g = torch.Generator().manual_seed(62)
    B, heads, D, Nl = 3, 8, 80, 45 * 80
    C = heads * D
    c = hip.attention_q_scale(D)
    q = _r(torch.randn(B, Nl, heads, D, generator=g) * c * 1.5, td)
    k = _r(torch.randn(B, Nl, heads, D, generator=g) * 1.5, td)
    v = _r(torch.randn(B, Nl, heads, D, generator=g), td)
    hm = torch.stack([q, k, v], 1).permute(0, 1, 3, 2, 4).contiguous().to(td).to(gpu)
    out = torch.empty(B, Nl, C, dtype=td, device=gpu)
    args = dict(B=B, heads=heads, Nq=Nl, Nkv=Nl, D=D, q_bs=3 * Nl * C, k_bs=3 * Nl * C, v_bs=3 * Nl * C, o_bs=Nl * C, q_rs=D, k_rs=D, v_rs=D, o_rs=C,
                k_off=Nl * C, v_off=2 * Nl * C, q_hs=Nl * D, k_hs=Nl * D, v_hs=Nl * D, q_prescaled=True)
    hip.attention(dt, hm, hm, hm, out, **args)
    out2 = torch.empty_like(out)
    lse = torch.empty(B, heads, Nl, dtype=torch.float32, device=gpu)
    hip.attention(dt, hm, hm, hm, out2, lse=lse, **args)
    assert torch.equal(out, out2)                                   # deterministic, and the lse output does not change the result
    for b, h in ((0, 0), (2, 5)):
        s = q[b, :, h].double() @ k[b, :, h].double().t()           # log2 units (q carries scale * log2 e)
        pr = torch.exp2(s - s.amax(-1, keepdim=True))
        ref = ((pr / pr.sum(-1, keepdim=True)) @ v[b, :, h].double()).float()
        got = out[b, :, h * D:(h + 1) * D].float().cpu()
        err = (got - ref).abs().max().item()
        print(f"attention d80 N3600 [{dname}] (b={b}, h={h}): max-abs {err:.2e} (tolerance {4 * ulp * max(1.0, ref.abs().max().item()):.2e})")
        assert err <= 4 * ulp * max(1.0, ref.abs().max().item())
        want = (torch.logsumexp(s * math.log(2.0), -1) / math.log(2.0)).float()
        assert (lse[b, h].cpu() - want).abs().max().item() <= max(2e-3, 2 * ulp)      # log2 units: the sum carries the rounding of P (one dominant key: one ulp of P)

def test_temporal_attention_full_size(gpu):
    """32 frames x 14400 pixels x 8 heads x d=40: the strided (f,hw)->(hw,f) gather at full size; 300 pixels checked."""
    from videovanish_amd import hip
    td, dt = torch.bfloat16, hip.BF16
    g = torch.Generator().manual_seed(62)
    heads, D = 8, 40
    C = heads * D
    qkv = torch.randn(FR * N, 3 * C, generator=g).to(td)
    out = torch.empty(FR * N, C, dtype=td, device=gpu)
    buf = qkv.to(gpu)
    hip.attention(dt, buf, buf, buf, out, B=N, heads=heads, Nq=FR, Nkv=FR, D=D, q_bs=3 * C, k_bs=3 * C, v_bs=3 * C, o_bs=C, q_rs=N * 3 * C,
                  k_rs=N * 3 * C, v_rs=N * 3 * C, o_rs=N * C, k_off=C, v_off=2 * C)
    pix = torch.randint(0, N, (300,), generator=g)
    x = qkv.float().reshape(FR, N, 3, heads, D)[:, pix]                       # [F,300,3,h,d]
    q, k, v = (x[:, :, i].permute(1, 2, 0, 3) for i in range(3))               # [300,h,F,d]
    ref = (torch.softmax((q @ k.transpose(-1, -2)) * D ** -0.5, -1) @ v).permute(2, 0, 1, 3)
    got = out.float().cpu().reshape(FR, N, heads, D)[:, pix]
    assert (got - ref).abs().max().item() <= 2 ** -7 * max(1.0, ref.abs().max().item())


def test_conv3_and_geglu_full_size(gpu):
    """3x3 conv 320->320 on [32,90,160,320] (+bias +fp32 residual) and the GEGLU projection on M = 460800 rows."""
    from videovanish_amd import hip, packing
    td, dt = torch.bfloat16, hip.BF16
    g = torch.Generator().manual_seed(63)
    C = 320
    x = torch.randn(FR, H, W, C, generator=g).to(td)
    w = torch.randn(C, C, 3, 3, generator=g) / math.sqrt(9 * C)
    b = torch.randn(C, generator=g)
    res = torch.randn(FR * N, C, generator=g)
    wp, K = packing.pack_conv(w, td)
    out = hip.conv_gemm(dt, x.to(gpu), wp.to(gpu), C, K, F=FR, Hin=H, Win=W, ksize=3, pad_t=1, pad_l=1, bias=b.to(gpu), res0=res.to(gpu),
                        out_dtype=torch.float32)
    f = 17
    ref = F.conv2d(x[f].float().permute(2, 0, 1)[None], _r(w, td), b, padding=1)[0].permute(1, 2, 0).reshape(N, C) + res[f * N:(f + 1) * N]
    got = out[f * N:(f + 1) * N].cpu()
    assert (got - ref).abs().max().item() <= 3e-4 * ref.abs().max().item()
    # GEGLU: [M,320] x [2560,320]^T -> [M,1280]
    M = FR * N
    a = x.reshape(M, C)
    w8 = torch.randn(8 * C, C, generator=g) / math.sqrt(C)
    b8 = torch.randn(8 * C, generator=g)
    wi, bi = packing.geglu_interleave(w8, b8)
    o = hip.conv_gemm(dt, a.to(gpu), packing.pack_matrix(wi, td, geglu=True).to(gpu), 8 * C, C, F=1, Hin=M, Win=1, bias=bi.to(gpu),
                      epilogue=hip.EPI_GEGLU, out_dtype=torch.float32)
    rows = torch.randint(0, M, (256,), generator=g)
    hfull = F.linear(a[rows].float(), _r(w8, td), b8)
    v, gate = hfull.chunk(2, -1)
    assert (o[rows.to(gpu)].cpu() - v * F.gelu(gate)).abs().max().item() <= 3e-4 * (v * F.gelu(gate)).abs().max().item()


def test_groupnorm_full_size_and_deterministic(gpu):
    from videovanish_amd import hip
    g = torch.Generator().manual_seed(64)
    C = 320
    x = torch.randn(FR, N, C, generator=g) * 2 + 0.3
    gamma, beta = torch.randn(C, generator=g), torch.randn(C, generator=g)
    xg = x.to(gpu)
    a = hip.groupnorm(hip.BF16, xg, gamma.to(gpu), beta.to(gpu), 32, 1e-5, F=FR, HW=N, silu=True, out_dtype=torch.float32)
    b = hip.groupnorm(hip.BF16, xg, gamma.to(gpu), beta.to(gpu), 32, 1e-5, F=FR, HW=N, silu=True, out_dtype=torch.float32)
    assert torch.equal(a, b)                                        # no atomics: bit-reproducible
    f = 9
    ref = F.silu(F.group_norm(x[f].t()[None], 32, gamma, beta, 1e-5))[0].t()
    assert (a.cpu().reshape(FR, N, C)[f] - ref).abs().max().item() <= 3e-4
    p = hip.groupnorm(hip.BF16, xg, gamma.to(gpu), beta.to(gpu), 32, 1e-6, F=FR, HW=N, pool_frames=True, out_dtype=torch.float32)
    xs = x[:, ::37]                                                  # pooled statistics: check mean/var of the normalised output
    pn = (p.cpu().reshape(FR, N, C) - beta) / gamma
    grp = pn.reshape(FR * N, 32, C // 32)
    assert abs(grp.mean(dim=(0, 2)).abs().max().item()) <= 1e-3 and abs(grp.var(dim=(0, 2), unbiased=False).mean().item() - 1.0) <= 1e-3


def test_denoiser_determinism_and_frame_independence(gpu):
    """Run-to-run bit reproducibility of a whole denoiser step (what makes sharded == single-GPU meaningful), and per-frame
    independence of the BrushNet branch (no temporal layers): frame 0 of a 3-frame batch == frame 0 run alone."""
    from videovanish_amd import hip
    from videovanish_amd.config import SMALL_UNET
    from videovanish_amd.nn import Ctx
    from videovanish_amd.unet import BrushNet, Denoiser
    ctx = Ctx("cuda:0", "bf16", 0)
    cfg = SMALL_UNET
    text = ctx.src.normal("text_states", (1, cfg.text_len, cfg.cross_dim))
    den = Denoiser(ctx, cfg, text)
    g = torch.Generator().manual_seed(65)
    Fr, h, w = 3, 12, 20
    lat, cond = torch.randn(Fr, h, w, 4, generator=g).to(gpu), torch.randn(Fr, h, w, 4, generator=g).to(gpu)
    mask = ((torch.rand(Fr, h * 8, w * 8, generator=g) > 0.5).to(torch.uint8) * 255).to(gpu)
    e1 = den(lat, cond, mask, 501, Fr, h, w, h * 8, w * 8)
    e2 = den(lat, cond, mask, 501, Fr, h, w, h * 8, w * 8)
    assert torch.equal(e1, e2)
    br = den.brush
    st = br.temb(501)
    x16 = hip.brushnet_input(ctx.dt, lat, cond, mask, h * 8, w * 8).view(Fr * h * w, 16)
    xa, sk_a, _ = br.run_down(x16, Fr, h, w, st)
    xb, sk_b, _ = br.run_down(x16[: h * w].contiguous(), 1, h, w, st)
    assert torch.equal(xa[: xb.shape[0]], xb)
    for (ta, _, _), (tb, _, _) in zip(sk_a, sk_b):
        assert torch.equal(ta[: tb.shape[0]], tb)


def test_image_kernels_full_size_bit_exact(gpu):
    """uint8 steps at 1280x720 against the plain-C oracle (oracle/imageops_ref.c): dilation k = 8 and k = 0, the 5x5 chamfer
    distance transform inside the feather window, the feathered composite, bilinear resize 720p <-> 540x960."""
    from oracle import imageops_c as IC
    from videovanish_amd import hip
    rng = np.random.default_rng(71)
    T, Hh, Ww = 3, 720, 1280
    masks = np.zeros((T, Hh, Ww, 3), np.uint8)
    for t in range(T):
        masks[t, 200:380, 300 + 2 * t: 620 + 2 * t, t % 3] = 255
        ys, xs = rng.integers(0, Hh, 40), rng.integers(0, Ww, 40)
        masks[t, ys, xs, 0] = 9                                           # isolated specks
    md = torch.from_numpy(masks).to(gpu)
    d8 = hip.mask_collapse_dilate(md, 8).cpu().numpy()
    for t in range(T):
        assert np.array_equal(d8[t], IC.dilate_cross(masks[t].max(2), 8)), t
    assert (hip.mask_collapse_dilate(md, 0).cpu().numpy() == 255).all()      # k = 0: to convergence = whole frame
    dt = hip.chamfer_dt(torch.from_numpy(d8).to(gpu), 4).cpu().numpy()
    ref = IC.distance_transform_l2_5(d8[0])
    sel = ref <= 4
    assert np.array_equal(dt[0][sel], ref[sel]) and (dt[0][~sel] > 4).all()
    inp = rng.integers(0, 256, (T, Hh, Ww, 3), dtype=np.uint8)
    orig = rng.integers(0, 256, (T, Hh, Ww, 3), dtype=np.uint8)
    for feather in (3.0, 0.0):
        got = hip.feather_composite(torch.from_numpy(inp).to(gpu), torch.from_numpy(orig).to(gpu), torch.from_numpy(d8).to(gpu), feather).cpu().numpy()
        for t in range(T):
            assert np.array_equal(got[t], IC.feather_composite(inp[t], orig[t], d8[t], feather)), (feather, t)
    small = hip.resize_u8(torch.from_numpy(inp).to(gpu), 540, 960).cpu().numpy()
    assert np.array_equal(small[1], IC.resize_bilinear_u8(inp[1], 960, 540))
    back = hip.resize_u8(torch.from_numpy(small).to(gpu), Hh, Ww).cpu().numpy()
    assert np.array_equal(back[2], IC.resize_bilinear_u8(small[2], Ww, Hh))


def test_bench_contract_line(gpu):
    """bench.py prints ONE JSON line with the driver's contract fields + roofline (live HIP events) + cpu_baseline (tiny arch, 2 steps)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--arch", "tiny", "--height", "128", "--width", "192", "--denoise-steps", "2",
                        "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["value"] > 0 and abs(d["value"] - 24 * 1000.0 / d["ms_per_step"]) / d["value"] < 1e-3      # 24 credited frames per chunk
    assert "workload" in d["config"] and "model" not in d["config"]
    ro = d["roofline"]
    assert ro["bound"] in ("hbm", "mfma") and ro["peak"] > 0 and abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-3 and ro["launches"] > 0
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb


def test_bench_line_with_rccl_initialised(gpu):
    """First contact with RCCL that one GPU allows: bench.py as a launcher starts it for --nproc-per-node 1 with the multi-rank code path FORCED
    (VV_BENCH_FORCE_DIST=1: init_process_group("nccl"), barriers, the max-over-ranks all_reduce, all_gather_object of the per-rank records on the GPU,
    destroy) -- the library loads, the communicator comes up under HSA_ENABLE_IPC_MODE_LEGACY=0, the collectives the N-GPU line uses run, and the line
    says which backend it saw.  The overlap exchange itself needs a second GPU (gloo / rendezvous-fabric tests in tests/test_dist_cpu.py)."""
    import json, os, socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), VV_BENCH_FORCE_DIST="1")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--arch", "tiny", "--height", "128", "--width", "192", "--denoise-steps", "2",
                        "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-kernel-events", "--no-power-trace"], capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["collective_backend"] == "nccl" and d["ranks_seen"] == 1 and d["n_gpus"] == 1
    assert d["per_rank"][0]["rank"] == 0 and d["per_rank"][0]["chunks"] == 2 and d["per_rank"][0]["owned_frames"] == 56 and d["value"] > 0


def test_fused_kernels_full_size_vs_layer_by_layer(gpu):
    """The fused level-0 kernels (spatial chain front + tail, motion module) launch 3600 blocks over 460 800 tokens at 720p but were compared with
    the oracle at <= 384 tokens only (tests/test_chain_gpu.py, tests/test_motion_gpu.py).  Here: the full size (32 frames of 90 x 160, C = 320,
    fp16) against the layer-by-layer HIP path on the same weights -- every piece of which IS checked at full size above -- with the bound taken
    from the same comparison at the small size in the same run: full-size error <= 2 x the small-size figure (+ 1e-4)."""
    from videovanish_amd import nn as vnn
    from videovanish_amd.config import UNetConfig
    from videovanish_amd.unet import sinusoidal_pos_emb
    cfg, C = UNetConfig(), 320
    ctx = vnn.Ctx("cuda:0", "fp16", 0)
    text = ctx.src.normal("text_states", (1, cfg.text_len, cfg.cross_dim))
    st = vnn.SpatialTransformer(ctx, "unet.down_blocks.0.attentions.0", C, cfg, ctx.dev(text[0], ctx.h16))
    mm = vnn.MotionModule(ctx, "unet.down_blocks.0.motion_modules.0", C, cfg, ctx.dev(sinusoidal_pos_emb(cfg.motion_max_seq, C)))
    assert st.fused is not None and mm.fused is not None
    rel = lambda a, b: ((a.float() - b.float()).abs().max() / b.float().abs().max()).item()

    def both(mod, cls, x, Fr, h, w):
        cls.FUSED = True
        a = mod(x, Fr, h, w)
        cls.FUSED = False
        try:
            b = mod(x, Fr, h, w)
        finally:
            cls.FUSED = True
        return a, b

    out = {}
    for name, mod, cls in (("spatial_chain", st, vnn.SpatialTransformer), ("motion_module", mm, vnn.MotionModule)):
        g = torch.Generator().manual_seed(71)
        xs = (torch.randn(FR * 8 * 12, C, generator=g) * 1.3 + 0.1).to(gpu)
        a, b = both(mod, cls, xs, FR, 8, 12)
        small = rel(a, b)
        xf = (torch.randn(FR * N, C, generator=g) * 1.3 + 0.1).to(gpu)
        a, b = both(mod, cls, xf, FR, H, W)
        full = rel(a, b)
        a2, _ = both(mod, cls, xf, FR, H, W)
        out[name] = (small, full)
        print(f"{name}: fused vs layer-by-layer rel max-abs {small:.2e} at 32 x 8 x 12, {full:.2e} at 32 x 90 x 160")
        assert torch.isfinite(a).all() and torch.equal(a, a2)                       # finite, run-to-run deterministic at full size
        assert full <= 2.0 * small + 1e-4, (name, small, full)
        del a, b, a2, xf
        torch.cuda.empty_cache()


def test_c3_length_clip_equals_chunkwise_blend(gpu):
    """BASELINE config c3 as stated -- 256 frames at 1280 x 720, 32 / 8 chunks = 11 chunks -- on ONE GPU through the device-resident pipeline
    (two chunks in flight), 1 DDIM step, fp16, full width: the blended fp32 pixels equal, bit for bit, the eleven chunks run one at a time and
    cross-faded in canonical chunk order with the ORACLE's plan and weights (oracle/pipeline_ref.py::chunk_plan / blend_weights).  Size-independent
    properties on top: every frame finite and inside [0, 1]; masked pixels repainted."""
    import bench
    from oracle import pipeline_ref as R
    from videovanish_amd import hip
    from videovanish_amd.config import RunConfig
    from videovanish_amd.pipeline import DiffuEraserHIP, chunk_noise
    T, Hp, Wp, chunk, overlap = 256, 720, 1280, 32, 8
    run = RunConfig(steps=1, chunk=chunk, overlap=overlap, seed=3, weight_seed=0, dtype="fp16")
    model = DiffuEraserHIP(run, "cuda:0")
    fr, mk, pr = bench.synth_clip(T, Hp, Wp, seed=77)
    fr, mk, pr = torch.from_numpy(fr).to(gpu), torch.from_numpy(mk).to(gpu), torch.from_numpy(pr).to(gpu)
    full, (lo, hi) = model.forward_device(fr, pr, mk, T, 0, steps=1, scheduler="ddim", return_float=True)
    assert (lo, hi) == (0, T) and tuple(full.shape) == (T, Hp, Wp, 3)
    plan = R.chunk_plan(T, chunk, overlap)
    assert len(plan) == 11 and plan[-1] == (T - chunk, T)
    wts = R.blend_weights(plan)
    acc = torch.zeros((T, Hp, Wp, 3), dtype=torch.float32, device=gpu)
    f = model.vae.factor
    for ci, (s, e) in enumerate(plan):
        noise = chunk_noise(run.seed, ci, (e - s, 4, Hp // f, Wp // f)).permute(0, 2, 3, 1).contiguous().to(gpu)
        dec = model.denoise_chunk(fr[s:e], pr[s:e], mk[s:e], noise, steps=1, scheduler="ddim")
        hip.decode_blend(dec.contiguous(), torch.from_numpy(np.asarray(wts[ci], np.float32)).to(gpu), acc[s:e])
        del dec
    assert torch.equal(full, acc)
    assert bool(torch.isfinite(full).all()) and float(full.min()) >= 0.0 and float(full.max()) <= 1.0
    u8 = hip.blur_compose(full, fr, mk, model.taps)
    inside = mk > 0
    assert float((u8[inside] != fr[inside]).float().mean()) > 0.5                 # the hole was repainted
