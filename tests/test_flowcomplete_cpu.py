"""Oracle-level checks of the recurrent flow-completion restatement (oracle/flowcomplete_ref.py) that need no GPU: structure of the
decomposition the product uses (Conv3d (1,k,k) = per-frame conv, (3,1,1) dilation 2 = taps t-2, t, t+2), shapes, flip symmetry."""
import torch
import torch.nn.functional as F

from oracle import flowcomplete_ref as FC
from oracle.model_ref import Params


def _rand(shape, seed, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def test_p3d_decomposition_equals_conv3d():
    P = Params(2)
    x = _rand((1, 8, 6, 10, 12), 1)
    full = FC.p3d(P, "t.p3d", x, 16, 2)
    # product decomposition: per-frame 3x3 stride-2 conv + LeakyReLU, then out[t] = W0 h[t-2] + W1 h[t] + W2 h[t+2] + b
    w, b = P.conv("t.p3d.conv1.0", 8, 16, 3)
    h = torch.stack([F.leaky_relu(F.conv2d(x[:, :, t], w, b, stride=2, padding=1), 0.2) for t in range(6)], 2)
    wt, bt = FC._wt(P, "t.p3d.conv2.0", 16)
    hp = F.pad(h, (0, 0, 0, 0, 2, 2))
    out = sum(torch.einsum("oc,bcthw->bothw", wt[:, :, i], hp[:, :, 2 * i: 2 * i + 6]) for i in range(3)) + bt.view(1, -1, 1, 1, 1)
    assert full.shape == (1, 16, 6, 5, 6)
    assert (full - out).abs().max() < 1e-5


def test_complete_shapes_and_bidirectional_flip():
    P = Params(5)
    B, T, H, W = 1, 4, 16, 24
    fw, bw = _rand((B, T - 1, 2, H, W), 1), _rand((B, T - 1, 2, H, W), 2)
    m = torch.zeros(B, T, 1, H, W)
    m[:, :, :, 4:10, 6:14] = 1
    kw = dict(width=(8, 16, 32), deform_groups=4)
    with torch.no_grad():
        pf, pb = FC.forward_bidirect_flow(P, fw, bw, m, **kw)
        assert pf.shape == pb.shape == (B, T - 1, 2, H, W)
        # the backward direction is the same network on the time-reversed sequence
        direct = FC.complete(P, torch.flip(bw * (1 - m[:, 1:]), dims=[1]), torch.flip(m[:, 1:], dims=[1]), **kw)
        assert torch.equal(pb, torch.flip(direct, dims=[1]))
        cf, cb = FC.combine_flow(fw, bw, pf, pb, m)
    hole = m[:, :-1].expand_as(cf) > 0
    assert torch.equal(cf[~hole], fw[~hole]) and torch.equal(cf[hole], pf[hole])
    # the network never sees the flow inside the holes
    fw2 = fw.clone()
    fw2[hole] = 99.0
    with torch.no_grad():
        pf2, _ = FC.forward_bidirect_flow(P, fw2, bw, m, **kw)
    assert torch.equal(pf, pf2)
