"""GPU parity of the RAFT / flow-guided propagation kernels (K9/K10) against oracle/flowprop_ref.py."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("kh,kw,stride,cin,cout,relu", [(1, 5, 1, 384, 256, False), (5, 1, 1, 384, 128, False), (7, 7, 2, 8, 64, True),
                                                        (7, 7, 1, 8, 128, True), (3, 3, 2, 96, 128, False), (1, 1, 2, 64, 96, False)])
def test_conv_rect_kernels(gpu, kh, kw, stride, cin, cout, relu):
    from videovanish_amd import hip, packing
    td, dt = torch.float16, hip.F16
    g = torch.Generator().manual_seed(21)
    Fr, H, W = 2, 13, 18
    x = torch.randn(Fr, cin, H, W, generator=g).to(td).float()
    w = (torch.randn(cout, cin, kh, kw, generator=g) / math.sqrt(cin * kh * kw)).to(td).float()
    b = torch.randn(cout, generator=g)
    ref = F.conv2d(x, w, b, stride=stride, padding=(kh // 2, kw // 2))
    if relu:
        ref = F.relu(ref)
    Ho, Wo = ref.shape[-2:]
    wp, K = packing.pack_conv(w, td)
    out = hip.conv_gemm(dt, _nhwc(x).to(td).to(gpu), wp.to(gpu), cout, K, F=Fr, Hin=H, Win=W, Hout=Ho, Wout=Wo, ksize=kh, ksize_w=kw,
                        stride=stride, pad_t=kh // 2, pad_l=kw // 2, bias=b.to(gpu), out_dtype=torch.float32,
                        act=hip.ACT_RELU if relu else hip.ACT_NONE)
    got = out.cpu().reshape(Fr, Ho, Wo, cout).permute(0, 3, 1, 2)
    assert (got - ref).abs().max().item() <= 3e-4 * max(1.0, ref.abs().max().item())


def test_instance_norm_and_out_col(gpu):
    from videovanish_amd import hip, packing
    g = torch.Generator().manual_seed(22)
    Fr, C, H, W = 3, 96, 9, 7
    x = torch.randn(Fr, C, H, W, generator=g) * 3 + 1
    ref = F.relu(F.instance_norm(x, eps=1e-5))
    ones, zeros = torch.ones(C, device=gpu), torch.zeros(C, device=gpu)
    out = hip.groupnorm(hip.F16, _nhwc(x).to(gpu), ones, zeros, C, 1e-5, F=Fr, HW=H * W, act=hip.ACT_RELU, out_dtype=torch.float32)
    got = out.cpu().reshape(Fr, H, W, C).permute(0, 3, 1, 2)
    assert (got - ref).abs().max().item() <= 2e-4
    # out_col: GEMM writes columns [64, 64+N) of a wider buffer and leaves the rest untouched
    M, K, N = 70, 64, 32
    a = torch.randn(M, K, generator=g).to(torch.float16)
    w = (torch.randn(N, K, generator=g) / 8).to(torch.float16)
    buf = torch.full((M, 128), 5.0, dtype=torch.float16, device=gpu)
    hip.conv_gemm(hip.F16, a.to(gpu), packing.pack_matrix(w.float(), torch.float16).to(gpu), N, K, F=1, Hin=M, Win=1, out=buf, out_col=64)
    r = buf.float().cpu()
    assert (r[:, :64] == 5).all() and (r[:, 96:] == 5).all()
    assert (r[:, 64:96] - a.float() @ w.float().t()).abs().max().item() <= 2e-2


def test_corr_pyramid_and_lookup(gpu):
    from oracle import flowprop_ref as FP
    from videovanish_amd import hip
    from videovanish_amd.nn import Ctx
    from videovanish_amd.raft import RAFT
    ctx = Ctx("cuda:0", "fp16", 0)
    raft = RAFT.__new__(RAFT)
    raft.ctx = ctx
    g = torch.Generator().manual_seed(23)
    h, w = 9, 12
    f1 = torch.randn(256, h, w, generator=g).to(torch.float16).float()
    f2 = torch.randn(256, h, w, generator=g).to(torch.float16).float()
    pyr_ref = FP.corr_pyramid(f1, f2)
    tok = lambda f: f.reshape(256, -1).t().contiguous().to(torch.float16).to(gpu)
    pyr = raft.corr_pyramid(tok(f1), tok(f2), h, w)
    for a, b in zip(pyr, pyr_ref):
        assert a.shape == b.shape and (a.cpu() - b).abs().max().item() <= 2e-4 * b.abs().max().item()
    coords = torch.stack([torch.rand(h * w, generator=g) * (w + 6) - 3, torch.rand(h * w, generator=g) * (h + 6) - 3], 1)
    coords[:5] = torch.tensor([[0.0, 0.0], [w - 1.0, h - 1.0], [2.5, 3.0], [-4.0, 1.0], [3.0, h + 3.5]])
    ref = FP.corr_lookup(pyr_ref, coords)
    got = hip.corr_lookup(hip.F16, [p.to(gpu) for p in pyr_ref], coords.to(gpu)).float().cpu()
    assert (got[:, 324:] == 0).all()
    assert (got[:, :324] - ref).abs().max().item() <= 2 ** -10 * max(1.0, ref.abs().max().item())


def test_gru_and_upsample_pieces(gpu):
    from oracle import flowprop_ref as FP
    from videovanish_amd import hip
    g = torch.Generator().manual_seed(24)
    M, h, w = 8 * 12, 8, 12
    zr, hh, q = torch.randn(M, 256, generator=g), torch.randn(M, 128, generator=g), torch.randn(M, 128, generator=g)
    rh = torch.empty(M, 128, dtype=torch.float16, device=gpu)
    hip.gru_rh(hip.F16, zr.to(gpu), hh.to(gpu), rh)
    assert (rh.float().cpu() - torch.sigmoid(zr[:, 128:]) * hh).abs().max().item() <= 2e-3
    hd, h16 = hh.clone().to(gpu), torch.empty(M, 128, dtype=torch.float16, device=gpu)
    hip.gru_update(hip.F16, zr.to(gpu), q.to(gpu), hd, h16)
    z = torch.sigmoid(zr[:, :128])
    assert (hd.cpu() - ((1 - z) * hh + z * torch.tanh(q))).abs().max().item() <= 1e-5
    flow = torch.randn(1, 2, h, w, generator=g) * 3
    mask = torch.randn(1, 576, h, w, generator=g)
    ref = FP.convex_upsample(flow, mask)[0].permute(1, 2, 0)
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
    coords1 = (torch.stack([xs, ys], -1) + flow[0].permute(1, 2, 0)).reshape(M, 2).contiguous()
    got = hip.convex_upsample(coords1.to(gpu), mask[0].permute(1, 2, 0).reshape(M, 576).contiguous().to(gpu), h, w).cpu()
    assert (got - ref).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item())


def _clip(T, H, W, seed):
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, (H + 16, W + 16, 3), dtype=np.uint8)
    frames = [np.ascontiguousarray(base[2 * t: 2 * t + H, t: t + W]) for t in range(T)]
    masks = []
    for t in range(T):
        m = np.zeros((H, W), np.uint8)
        m[H // 4: H // 2, W // 4 + 2 * t: W // 2 + 2 * t] = 255
        masks.append(m)
    return frames, masks


def test_propagation_bit_exact(gpu):
    """warp / consistency / fill / combine on GIVEN flows: bit-exact against the oracle."""
    from oracle import flowprop_ref as FP
    from videovanish_amd import flowprop
    T, H, W = 4, 40, 56
    frames, masks = _clip(T, H, W, 31)
    g = torch.Generator().manual_seed(32)
    # smooth-ish flows around the true shift (dx=+1, dy=+2 per frame) plus noise so that some pixels fail the consistency test
    fw = [torch.stack([torch.full((H, W), 1.0), torch.full((H, W), 2.0)]) + torch.randn(2, H, W, generator=g) * 0.3 for _ in range(T - 1)]
    bw = [-f + torch.randn(2, H, W, generator=g) * 0.3 for f in fw]
    ref, ref_filled = FP.propagate(np.stack(frames), np.stack(masks), fw, bw)
    to_dev = lambda f: f.permute(1, 2, 0).contiguous().to(gpu)
    out, filled = flowprop.propagate(torch.from_numpy(np.stack(frames)).to(gpu), torch.from_numpy(np.stack(masks)).to(gpu),
                                     [to_dev(f) for f in fw], [to_dev(f) for f in bw])
    assert ref_filled.sum() > 50
    assert np.array_equal(filled.cpu().numpy() != 0, ref_filled)
    assert np.array_equal(out.cpu().numpy(), np.stack(ref))


def test_raft_stages_vs_oracle(gpu):
    """RAFT on HIP vs the fp32 oracle (fp16 operands): encoders, correlation, first update, 3-iteration flow."""
    from oracle import flowprop_ref as FP
    from oracle.model_ref import Params
    from videovanish_amd.nn import Ctx
    from videovanish_amd.raft import RAFT
    H, W = 64, 96
    frames, _ = _clip(2, H, W, 41)
    P = Params(0)
    tr = {}
    with torch.no_grad():
        ref = FP.raft_flow(P, frames[0], frames[1], iters=3, trace=tr)
    ctx = Ctx("cuda:0", "fp16", 0)
    raft = RAFT(ctx)
    f, c, h, w = raft.features(torch.from_numpy(np.stack(frames)).to(gpu))
    f1 = f[0].cpu().t().reshape(256, h, w)
    e_f = ((f1 - tr["f1"]).abs().max() / tr["f1"].abs().max()).item()
    gt = {}
    flow = raft.flow(f[0], f[1], c[0], h, w, iters=3, trace=gt)
    e_net = (gt["net0"].cpu().t().reshape(1, 128, h, w) - tr["net0"]).abs().max().item()
    e_corr = ((gt["corr0"].cpu() - tr["corr0"]).abs().max() / tr["corr0"].abs().max()).item()
    e_d0 = ((gt["dflow0"].cpu().t().reshape(1, 2, h, w) - tr["dflow0"]).abs().max() / tr["dflow0"].abs().max()).item()
    e_flow = ((flow.cpu().permute(2, 0, 1) - ref).abs().max() / ref.abs().max()).item()
    print(f"raft[fp16]: fmap rel {e_f:.2e}, net0 abs {e_net:.2e}, corr rel {e_corr:.2e}, dflow0 rel {e_d0:.2e}, flow(3 it) rel {e_flow:.2e}")
    assert e_f <= 5e-3 and e_net <= 5e-3 and e_corr <= 5e-3 and e_d0 <= 2e-2 and e_flow <= 5e-2


def test_prior_end_to_end(gpu):
    from oracle import flowprop_ref as FP
    from videovanish_amd.flowprop import flow_propagation_prior
    T, H, W = 3, 64, 96
    frames, masks = _clip(T, H, W, 51)
    got = flow_propagation_prior(frames, masks, device="cuda:0", dtype="fp16", iters=2)
    ref = FP.flow_propagation_prior(frames, masks, iters=2)
    assert len(got) == T and all(g.shape == (H, W, 3) and g.dtype == np.uint8 for g in got)
    unm = np.stack(masks) == 0
    assert (np.stack(got)[unm] == np.stack(frames)[unm]).all()          # only hole pixels change
    diff = (np.stack(got).astype(int) - np.stack(ref).astype(int))
    frac = float((np.abs(diff) > 2).mean())
    print(f"prior e2e: fraction of pixels differing by >2 levels from the oracle: {frac:.4f}")
    assert frac <= 0.02


def test_raft_batched_pairs_match_single(gpu):
    """the update block over a stack of pairs (one launch per kernel and iteration) gives bit for bit the per-pair flows."""
    from videovanish_amd.nn import Ctx
    from videovanish_amd.raft import RAFT
    H, W = 64, 96
    frames, _ = _clip(4, H, W, 61)
    raft = RAFT(Ctx("cuda:0", "fp16", 0))
    f, c, h, w = raft.features(torch.from_numpy(np.stack(frames)).to(gpu))
    single = [raft.flow(f[t], f[t + 1], c[t], h, w, iters=3) for t in range(3)]
    batched = raft.flow_batch(f[0:3], f[1:4], c[0:3], h, w, iters=3)
    assert batched.shape == (3, H, W, 2)
    for t in range(3):
        assert torch.equal(batched[t], single[t])


def test_subvideo_propagation_matches_oracle(gpu):
    """Propainter.forward(subvideo_length=...) (reference diffuerase.py:55 passes 50): propagation per sub-video with 5 frames of
    context; here T = 13, subvideo_length = 4 -> four sub-videos.  Flows from the oracle, so the propagation must be bit-exact."""
    from oracle import flowprop_ref as FP
    from videovanish_amd import flowprop
    T, H, W = 13, 32, 48
    rng = np.random.default_rng(71)
    base = rng.integers(0, 256, (H + 2 * T, W + T, 3), dtype=np.uint8)
    frames = [np.ascontiguousarray(base[2 * t: 2 * t + H, t: t + W]) for t in range(T)]
    masks = []
    for t in range(T):
        m = np.zeros((H, W), np.uint8)
        m[H // 4: 3 * H // 4, W // 4: 3 * W // 4] = 255      # static hole + sub-pixel flows: filling it needs long-range propagation
        masks.append(m)
    assert flowprop.subvideo_ranges(T, 4, 3) == FP.subvideo_ranges(T, 4, 3) == [(0, 7, 0, 4), (1, 11, 4, 8), (5, 13, 8, 12), (9, 13, 12, 13)]
    # defaults: 10 frames of context on both sides of a sub-video (the reference passes subvideo_length=50, diffuerase.py:54)
    assert flowprop.subvideo_ranges(60, 50) == FP.subvideo_ranges(60, 50) == [(0, 60, 0, 50), (40, 60, 50, 60)] and flowprop.subvideo_ranges(50, 50) == [(0, 50, 0, 50)]
    g = torch.Generator().manual_seed(3)
    fw = [torch.randn(2, H, W, generator=g) * 0.4 for _ in range(T - 1)]
    bw = [-f + 0.05 * torch.randn(2, H, W, generator=g) for f in fw]
    ref = [None] * T
    for (s, e, lo, hi) in FP.subvideo_ranges(T, 4, 3):
        sub, _ = FP.propagate(np.stack(frames[s:e]), np.stack(masks[s:e]), fw[s:e - 1], bw[s:e - 1])
        for t in range(lo, hi):
            ref[t] = sub[t - s]
    to_dev = lambda f: f.permute(1, 2, 0).contiguous().to(gpu)
    fr, mk = torch.from_numpy(np.stack(frames)).to(gpu), torch.from_numpy(np.stack(masks)).to(gpu)
    out = torch.empty_like(fr)
    for (s, e, lo, hi) in flowprop.subvideo_ranges(T, 4, 3):
        sub, _ = flowprop.propagate(fr[s:e].contiguous(), mk[s:e].contiguous(), [to_dev(f) for f in fw[s:e - 1]], [to_dev(f) for f in bw[s:e - 1]])
        out[lo:hi] = sub[lo - s: hi - s]
    assert np.array_equal(out.cpu().numpy(), np.stack(ref))
    whole, _ = FP.propagate(np.stack(frames), np.stack(masks), fw, bw)
    assert not np.array_equal(np.stack(whole), np.stack(ref))        # the sub-video schedule really changes the result


def test_prior_with_flow_completion_runs_and_keeps_unmasked_pixels(gpu):
    """Propainter(flow_completion=True): RAFT -> recurrent flow completion -> propagation (the order of the real ProPainter).  With seeded
    random weights only the contract can be checked: shapes, pixels outside the holes untouched, holes filled, deterministic."""
    from videovanish_amd.propainter import Propainter
    T, H, W = 4, 64, 96
    frames, masks = _clip(T, H, W, 83)
    pp = Propainter(device="cuda:0", flow_completion=True)
    out = pp.forward(frames, masks, subvideo_length=50)
    assert len(out) == T and out[0].shape == (H, W, 3) and out[0].dtype == np.uint8
    for t in range(T):
        keep = masks[t] == 0
        assert np.array_equal(out[t][keep], frames[t][keep])
    # (with random RAFT / completion weights no flow is forward-backward consistent, so WHAT gets propagated cannot be asserted here;
    #  the completion network itself is checked against its oracle in tests/test_flowcomplete_gpu.py)
    again = pp.forward(frames, masks, subvideo_length=50)
    assert all(np.array_equal(a, b) for a, b in zip(out, again))


def test_full_propainter_pipeline_contract(gpu):
    """Propainter(flow_completion=True, generator=True): RAFT -> flow completion -> image propagation -> inpainting generator, the complete
    third-party pipeline the reference calls at diffuerase.py:52-57.  Seeded random weights: contract only (shapes, dtype, untouched pixels
    outside the masks, holes no longer the mean-colour fill, deterministic)."""
    from videovanish_amd.propainter import Propainter
    T, H, W = 6, 64, 96
    frames, masks = _clip(T, H, W, 91)
    pp = Propainter(device="cuda:0", flow_completion=True, generator=True)
    out = pp.forward(frames, masks, ref_stride=2, neighbor_length=4, subvideo_length=50)
    base = Propainter(device="cuda:0").forward(frames, masks, subvideo_length=50)
    assert len(out) == T and out[0].shape == (H, W, 3) and out[0].dtype == np.uint8
    for t in range(T):
        keep = masks[t] == 0
        assert np.array_equal(out[t][keep], frames[t][keep])
        assert not np.array_equal(out[t][~keep], base[t][~keep])
    again = pp.forward(frames, masks, ref_stride=2, neighbor_length=4, subvideo_length=50)
    assert all(np.array_equal(a, b) for a, b in zip(out, again))
