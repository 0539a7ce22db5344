"""ProPainter's inpainting generator: HIP path (videovanish_amd/inpaintgen.py) against the fp32 oracle
(oracle/inpaintgen_ref.py), same seeded weights and inputs, through the C ABI."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _case(t, lt, H, W, seed):
    rng = np.random.default_rng(seed)
    frames = rng.integers(0, 256, (t, H, W, 3), dtype=np.uint8)
    m_in = np.zeros((t, H, W), np.uint8)
    m_up = np.zeros((t, H, W), np.uint8)
    for f in range(t):
        m_in[f, H // 4: H // 2, W // 8 + 3 * f: W // 8 + 3 * f + W // 4] = 255           # only the left windows touch the hole: both attention paths run
        m_up[f, H // 4 + 4: H // 2 - 4, W // 8 + 3 * f + 6: W // 8 + 3 * f + W // 4 - 6] = 255
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    ff = torch.stack([torch.stack([1.5 + 0.01 * yy, -0.8 + 0.01 * xx], -1) for _ in range(lt - 1)]) + 0.2 * torch.randn(lt - 1, H, W, 2, generator=g)
    fb = -ff + 0.05 * torch.randn(lt - 1, H, W, 2, generator=g)
    return frames, m_in, m_up, ff, fb


# tolerances = 2 x the measured error (tools/n1_margins.py, profiles/r3_n1_margins.txt: fp16 1.8e-3, bf16 1.25e-2 at 2 blocks)
@pytest.mark.parametrize("dname,tol,depths", [("fp16", 3.6e-3, 2), ("bf16", 2.5e-2, 2), ("fp16", 6e-3, 8)])
def test_generator_matches_oracle(gpu, dname, tol, depths):
    from oracle import inpaintgen_ref as G
    from oracle.model_ref import Params
    from videovanish_amd import nn
    from videovanish_amd.inpaintgen import InpaintGenerator
    t, lt, H, W = 5, 3, 80, 144                 # depths = 8: the published ProPainter depth (all 8 sparse-window transformer blocks)
    frames, m_in, m_up, ff, fb = _case(t, lt, H, W, 4)
    P = Params(21)
    fr = torch.from_numpy(frames).float().permute(0, 3, 1, 2)[None] / 127.5 - 1.0
    mi = torch.from_numpy(m_in > 0).float()[None, :, None]
    mu = torch.from_numpy(m_up > 0).float()[None, :, None]
    to5 = lambda f: f.permute(0, 3, 1, 2)[None]
    with torch.no_grad():
        ref = G.generator(P, fr, to5(ff), to5(fb), mi, mu, lt, depths=depths, t_dilation=2)[0].permute(0, 2, 3, 1)      # [lt, H, W, 3], tanh range
    gen = InpaintGenerator(nn.Ctx("cuda:0", dname, 21), depths=depths, t_dilation=2)
    raw = gen.forward(torch.from_numpy(frames).to(gpu), ff.to(gpu), fb.to(gpu), torch.from_numpy(m_in).to(gpu), torch.from_numpy(m_up).to(gpu), lt)
    got = torch.tanh(raw.cpu()).reshape(lt, H, W, 3)
    err = (got - ref).abs()
    assert err.max().item() <= tol, (err.max().item(), err.mean().item())
    assert ref.std().item() > 0.05                     # a non-degenerate output


def test_generator_helper_kernels(gpu):
    import torch.nn.functional as F
    from videovanish_amd import hip
    g = torch.Generator().manual_seed(3)
    # gather_rows
    src = torch.randn(50, 24, generator=g).half()
    idx = torch.tensor([3, -1, 49, 0, 3], dtype=torch.int32)
    got = hip.gather_rows(src.to(gpu), idx.to(gpu)).cpu()
    ref = torch.stack([src[i] if i >= 0 else torch.zeros(24).half() for i in idx.tolist()])
    assert torch.equal(got, ref)
    # fold (+ normalise + GELU) against F.fold on channel-major patches
    B, C, h, w = 2, 16, 11, 14
    fh, fw = (h + 6 - 7) // 3 + 1, (w + 6 - 7) // 3 + 1
    x = torch.randn(B, fh * fw, 49, C, generator=g)                                      # tap-major
    cm = x.permute(0, 3, 2, 1).reshape(B, C * 49, fh * fw)                              # channel-major for F.fold
    ref = F.fold(cm, (h, w), (7, 7), stride=3, padding=3)
    norm = F.fold(torch.ones(B, 49, fh * fw), (h, w), (7, 7), stride=3, padding=3)
    rows = x.reshape(B * fh * fw, 49 * C).contiguous()
    got = hip.fold_patches(hip.F16, rows.to(gpu), B, fh, fw, C, h, w).cpu().view(B, h, w, C).permute(0, 3, 1, 2)
    assert (got - ref).abs().max() <= 1e-5
    got_n = hip.fold_patches(hip.F16, rows.to(gpu), B, fh, fw, C, h, w, normalise=True, gelu=True).cpu().view(B, h, w, C).permute(0, 3, 1, 2)
    assert (got_n - F.gelu(ref / norm)).abs().max() <= 1e-5
    # flow_down4 against F.interpolate
    fl = torch.randn(3, 16, 24, 2, generator=g)
    ref_d = F.interpolate(fl.permute(0, 3, 1, 2), scale_factor=0.25, mode="bilinear", align_corners=False).permute(0, 2, 3, 1) / 4.0
    assert (hip.flow_down4(fl.to(gpu)).cpu() - ref_d).abs().max() <= 1e-6
    # gen_input / gen_compose
    fr = torch.randint(0, 256, (2, 4, 6, 3), generator=g, dtype=torch.uint8)
    m1 = (torch.rand(2, 4, 6, generator=g) > 0.5).to(torch.uint8) * 255
    m2 = (torch.rand(2, 4, 6, generator=g) > 0.5).to(torch.uint8) * 255
    gi = hip.gen_input(fr.to(gpu), m1.to(gpu), m2.to(gpu)).cpu()
    assert torch.equal(gi[:, :3], fr.reshape(-1, 3).float() / 127.5 - 1.0) and torch.equal(gi[:, 3], (m1 > 0).float().reshape(-1))
    assert torch.equal(gi[:, 4], (m2 > 0).float().reshape(-1)) and gi[:, 5:].abs().max() == 0
    pred = torch.randn(48, 3, generator=g)
    acc = torch.zeros(2, 4, 6, 3, device=gpu)
    hip.gen_compose(pred.to(gpu), fr.to(gpu), m1.to(gpu), acc, True)
    img = torch.where((m1 > 0).reshape(-1, 1), torch.floor((torch.tanh(pred) + 1) * 0.5 * 255.0), fr.reshape(-1, 3).float())      # uint8 truncation
    assert (acc.cpu().reshape(-1, 3) - img).abs().max() == 0
    hip.gen_compose(pred.to(gpu) * 0.5, fr.to(gpu), m1.to(gpu), acc, False)
    img2 = torch.where((m1 > 0).reshape(-1, 1), torch.floor((torch.tanh(pred * 0.5) + 1) * 0.5 * 255.0), fr.reshape(-1, 3).float())
    assert (acc.cpu().reshape(-1, 3) - torch.floor(img * 0.5 + img2 * 0.5)).abs().max() == 0


def test_inpaint_clip_sliding_windows_match_oracle(gpu):
    """The sliding-window loop (neighbour windows of neighbor_length // 2 stride, every ref_stride-th frame as reference, uint8 mean of the
    two visits of a frame) against the oracle's restatement of ProPainter's inference loop, uint8 output."""
    from oracle import inpaintgen_ref as G
    from oracle.model_ref import Params
    from videovanish_amd import nn
    from videovanish_amd.inpaintgen import InpaintGenerator, inpaint_clip, window_schedule
    T, H, W, depths = 7, 48, 80, 2
    frames, m_in, m_up, _, _ = _case(T, T, H, W, 9)
    g = torch.Generator().manual_seed(2)
    ff = torch.randn(T - 1, H, W, 2, generator=g)
    fb = -ff + 0.05 * torch.randn(T - 1, H, W, 2, generator=g)
    assert window_schedule(T, 4, 3, 80) == G.window_schedule(T, 4, 3, 80) == [([0, 1, 2], [3, 6]), ([0, 1, 2, 3, 4], [6]), ([2, 3, 4, 5, 6], [0]), ([4, 5, 6], [0, 3])]
    updated = frames.copy()
    updated[m_up > 0] = 127
    P = Params(23)
    ref = G.inpaint_clip(P, updated, frames, ff.permute(0, 3, 1, 2), fb.permute(0, 3, 1, 2), m_in, m_up, neighbor_length=4, ref_stride=3, depths=depths)
    gen = InpaintGenerator(nn.Ctx("cuda:0", "fp16", 23), depths=depths)
    dev = lambda a: torch.from_numpy(a).to(gpu)
    got = inpaint_clip(gen, dev(updated), dev(frames), ff.to(gpu), fb.to(gpu), dev(m_in), dev(m_up), neighbor_length=4, ref_stride=3).cpu().numpy()
    assert got.shape == ref.shape == (T, H, W, 3) and got.dtype == np.uint8
    keep = m_in == 0
    assert np.array_equal(got[keep], frames[keep])
    d = np.abs(got.astype(np.int32) - ref.astype(np.int32))
    assert d.max() <= 3 and (d > 1).mean() < 0.01, (d.max(), (d > 1).mean())
