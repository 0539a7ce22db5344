"""ProPainter's recurrent flow-completion network: HIP path (videovanish_amd/flowcomplete.py) against the fp32 oracle
(oracle/flowcomplete_ref.py), same seeded weights and inputs, through the C ABI."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _case(T, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    fw = torch.stack([torch.stack([2.0 + 0.05 * yy + 0.3 * t, -1.0 + 0.04 * xx], -1) for t in range(T - 1)]) + 0.2 * torch.randn(T - 1, H, W, 2, generator=g)
    bw = -fw + 0.1 * torch.randn(T - 1, H, W, 2, generator=g)
    m = torch.zeros(T, H, W, dtype=torch.uint8)
    for t in range(T):
        m[t, H // 4: H // 2 + 2, W // 4 + t: W // 2 + t] = 255
    return fw, bw, m


# tolerances = 2 x the measured error (tools/n1_margins.py: fp16 1.0e-3, bf16 1.1e-2)
@pytest.mark.parametrize("dname,tol", [("fp16", 2e-3), ("bf16", 2.2e-2)])
def test_flow_completion_matches_oracle(gpu, dname, tol):
    from oracle import flowcomplete_ref as FC
    from oracle.model_ref import Params
    from videovanish_amd import nn
    from videovanish_amd.flowcomplete import FlowCompleteNet
    T, H, W, width, dg = 5, 32, 48, (16, 32, 64), 8
    fw, bw, m = _case(T, H, W, 3)
    P = Params(11)
    mf = (m > 0).float()[None, :, None]                                       # [1, T, 1, H, W]
    to5 = lambda f: f.permute(0, 3, 1, 2)[None]                              # [T-1,H,W,2] -> [1, T-1, 2, H, W]
    with torch.no_grad():
        rf, rb = FC.forward_bidirect_flow(P, to5(fw), to5(bw), mf, width=width, deform_groups=dg)
        cf, cb = FC.combine_flow(to5(fw), to5(bw), rf, rb, mf)
    net = FlowCompleteNet(nn.Ctx("cuda:0", dname, 11), width=width, deform_groups=dg)
    pf, pb = net.forward_bidirect_flow(fw.to(gpu), bw.to(gpu), m.to(gpu))
    back = lambda t5: t5[0].permute(0, 2, 3, 1)
    for got, ref in ((pf, rf), (pb, rb)):
        ref = back(ref)
        assert got.shape == ref.shape
        rel = ((got.cpu() - ref).abs().max() / ref.abs().max()).item()
        assert rel <= tol, rel
    gf, gb = net.complete_flows(fw.to(gpu), bw.to(gpu), m.to(gpu))
    for got, ref, src, mk in ((gf, cf, fw, m[:-1]), (gb, cb, bw, m[1:])):
        ref = back(ref)
        got = got.cpu()
        assert torch.equal(got[mk == 0], src[mk == 0])                         # outside the holes: the measured flow, untouched
        assert (got - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item())


def test_helper_kernels(gpu):
    """vv_fc_input (masking + replicate padding), vv_upsample2x_bilinear (align_corners=True), vv_flow_combine against torch."""
    import torch.nn.functional as F
    from videovanish_amd import hip
    T, H, W = 3, 8, 12
    fw, _, m = _case(T + 1, H, W, 5)
    m = m[:T]
    out = hip.fc_input(fw.to(gpu), m.to(gpu), 2).cpu()
    mf = (m > 0).float()
    ref = torch.cat([fw * (1 - mf[..., None]), mf[..., None]], -1).permute(0, 3, 1, 2)       # [T,3,H,W]
    ref = F.pad(ref, (2, 2, 2, 2), mode="replicate").permute(0, 2, 3, 1)
    assert out.shape == (T, H + 4, W + 4, 8)
    assert torch.equal(out[..., :3], ref) and out[..., 3:].abs().max() == 0
    x = torch.randn(2, 16, 5, 7, generator=torch.Generator().manual_seed(1))
    ref_u = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True).permute(0, 2, 3, 1).reshape(-1, 16)
    rows = x.permute(0, 2, 3, 1).reshape(-1, 16).contiguous()
    got32 = hip.upsample2x_bilinear(hip.F16, rows.to(gpu), 2, 5, 7).cpu()
    assert (got32 - ref_u).abs().max() <= 2e-6
    got16 = hip.upsample2x_bilinear(hip.F16, rows.half().to(gpu), 2, 5, 7).float().cpu()
    ref16 = F.interpolate(x.half().float(), scale_factor=2, mode="bilinear", align_corners=True).permute(0, 2, 3, 1).reshape(-1, 16)
    assert (got16 - ref16).abs().max() <= 2 ** -10 * ref16.abs().max()
    pred = torch.randn(T * H * W, 2, generator=torch.Generator().manual_seed(2))
    got = hip.flow_combine(pred.to(gpu), fw.to(gpu), m.to(gpu)).cpu()
    hole = (m > 0)[..., None].expand(T, H, W, 2)
    assert torch.equal(got, torch.where(hole, pred.reshape(T, H, W, 2), fw))
