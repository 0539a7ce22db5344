"""Host-side logic of the inpainting generator that needs no GPU: the index tables of the sparse window attention
(videovanish_amd/inpaintgen.py::_window_tables) against the oracle's roll / partition / pooling tensor ops, and oracle identities the
product's decomposition relies on (soft split = strided im2col + permuted linear; grouped encoder conv = dense block-sparse conv)."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import inpaintgen_ref as G
from oracle.model_ref import Params
from videovanish_amd.inpaintgen import _window_tables, token_grid


def test_window_tables_reproduce_the_oracle_key_sets():
    t, fh, fw, ws, pool, n_head = 3, 7, 12, (5, 9), (4, 4), 1
    T_ind = [1]
    tb = _window_tables(t, fh, fw, ws, pool, T_ind)
    nh, nw = tb["nh"], tb["nw"]
    assert (nh, nw, tb["ph"], tb["pw"], tb["nwin"]) == (10, 18, 2, 4, 4)
    # tag every padded token and pooled token with its own table row number, run the ORACLE's tensor ops on the tags
    tags = torch.arange(t * nh * nw, dtype=torch.float32).view(1, t, nh, nw, 1)
    part = lambda a: G.window_partition(a.contiguous(), ws, n_head).view(1, tb["nwin"], n_head, t, ws[0] * ws[1], 1)
    win = part(tags)
    valid, e = G.rolled_valid_index(ws)
    rolled = torch.cat([part(torch.roll(tags, shifts=sh, dims=(2, 3))) for sh in ((-e[0], -e[1]), (-e[0], e[1]), (e[0], -e[1]), (e[0], e[1]))], 4)[:, :, :, :, valid]
    ptags = (t * nh * nw + torch.arange(t * tb["ph"] * tb["pw"], dtype=torch.float32)).view(1, 1, 1, t, -1, 1).repeat(1, tb["nwin"], 1, 1, 1, 1)
    keys = torch.cat([win, rolled, ptags], 4)[0, :, 0][:, T_ind].reshape(tb["nwin"], -1)
    assert keys.shape == tb["k_idx"].shape
    for wi in range(tb["nwin"]):
        assert sorted(keys[wi].long().tolist()) == sorted(tb["k_idx"][wi].tolist())
        assert win[0, wi, 0].reshape(-1).long().tolist() == tb["q_idx"][wi].reshape(-1).tolist()
    # inverse table: token (f, y, x) of the unpadded grid -> its slot in the [window][frame][position] output order
    flat_q = tb["q_idx"].reshape(-1)
    for tok in (0, 5, fh * fw + 3 * fw + 11, t * fh * fw - 1):
        f, r = divmod(tok, fh * fw)
        y, x = divmod(r, fw)
        assert flat_q[tb["inv"][tok]] == (f * nh + y) * nw + x
    assert tb["pad_idx"].reshape(t, nh, nw)[1, 6, 11] == 1 * fh * fw + 6 * fw + 11 and tb["pad_idx"].reshape(t, nh, nw)[0, 7, 0] == -1


def test_soft_split_is_a_strided_im2col_with_permuted_columns():
    P = Params(1)
    C, hidden, h, w = 8, 16, 11, 14
    x = torch.randn(2, C, h, w, generator=torch.Generator().manual_seed(0))
    ref = G.soft_split(P, "ss", x, 1, hidden)
    fh, fw = token_grid(h, w)
    wgt, b = P.linear("ss.embedding", 49 * C, hidden)
    perm = torch.arange(49 * C).view(C, 49).t().reshape(-1)
    xp = F.pad(x, (3, 3, 3, 3))
    cols = torch.stack([xp[:, :, ky:ky + 3 * (fh - 1) + 1:3, kx:kx + 3 * (fw - 1) + 1:3] for ky in range(7) for kx in range(7)], 1)   # [B, 49, C, fh, fw]
    cols = cols.permute(0, 3, 4, 1, 2).reshape(2, fh * fw, 49 * C)                                                             # tap-major k * C + c
    got = F.linear(cols, wgt[:, perm], b).view(1, -1, fh, fw, hidden)
    assert (got - ref).abs().max() < 1e-5


def test_grouped_encoder_conv_equals_the_dense_block_sparse_form():
    P = Params(2)
    g, c0, cp, co = 4, 16, 24, 8                      # x0: 16 channels, previous output: 24, groups 4
    x0, prev = torch.randn(1, c0, 6, 7), torch.randn(1, cp, 6, 7)
    w, b = P.conv("enc.layers.12", (c0 + cp) // g, co, 3)
    cat = torch.cat([x0.view(1, g, -1, 6, 7), prev.view(1, g, -1, 6, 7)], 2).view(1, -1, 6, 7)
    ref = F.conv2d(cat, w, b, padding=1, groups=g)
    wd = torch.zeros(co, c0 + cp, 3, 3)
    a, p, og = c0 // g, cp // g, co // g
    for j in range(g):
        wd[j * og:(j + 1) * og, j * a:(j + 1) * a] = w[j * og:(j + 1) * og, :a]
        wd[j * og:(j + 1) * og, c0 + j * p:c0 + (j + 1) * p] = w[j * og:(j + 1) * og, a:]
    got = F.conv2d(torch.cat([x0, prev], 1), wd, b, padding=1)
    assert (got - ref).abs().max() < 1e-5


def test_generator_shapes_and_reference_frame_selection():
    P = Params(7)
    b, t, lt, H, W = 1, 4, 2, 48, 80
    g = torch.Generator().manual_seed(1)
    fr = torch.rand(b, t, 3, H, W, generator=g) * 2 - 1
    m = torch.zeros(b, t, 1, H, W); m[:, :, :, 10:30, 20:50] = 1
    ff = torch.randn(b, lt - 1, 2, H, W, generator=g)
    with torch.no_grad():
        out = G.generator(P, fr * (1 - m), ff, -ff, m, m, lt, depths=2, t_dilation=2)
    assert out.shape == (b, lt, 3, H, W) and float(out.abs().max()) <= 1.0
    assert G.get_ref_index([3, 4, 5, 6, 7], 40, 10) == [0, 10, 20, 30]
    assert G.get_ref_index([8, 9, 10, 11, 12], 40, 10) == [0, 20, 30]
