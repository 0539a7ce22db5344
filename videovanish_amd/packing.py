"""Host-side weight re-layout for the MFMA kernels: torch-convention fp32 weights -> the [Npad][Kpad] h16
K-contiguous matrices vv_conv_gemm consumes (k = (ky*ks + kx)*Cin_pad + c)."""
import torch


def _round_up(x, m):
    return (x + m - 1) // m * m


def npad_for(N, geglu=False):
    """Row padding that matches vv_conv_gemm's tile dispatch (128x160, 128x128 or 128x16 tiles)."""
    if geglu:
        return _round_up(N, 128)
    if N % 160 == 0 or N % 128 == 0:
        return N
    if N <= 64:
        return _round_up(N, 16)
    return _round_up(N, 128)


def pack_matrix(w2d, h16, geglu=False):
    """[N][K] fp32 -> zero-padded [Npad][Kpad] h16 (Kpad % 64 == 0)."""
    N, K = w2d.shape
    out = torch.zeros((npad_for(N, geglu), _round_up(K, 64)), dtype=h16)
    out[:N, :K] = w2d.to(h16)
    return out


def pack_conv(w, h16, cin_pad=None):
    """Conv2d weight [Cout][Cin][k][k] -> ([Npad][Kpad] h16, K) with k ordered (ky, kx, cin).  cin_pad: zero-pad
    the input channels (conv_in layers whose activations are stored with padded channels)."""
    cout, cin, kh, kw = w.shape
    cp = cin if cin_pad is None else cin_pad
    t = torch.zeros((cout, kh, kw, cp), dtype=torch.float32)
    t[..., :cin] = w.permute(0, 2, 3, 1)
    return pack_matrix(t.reshape(cout, kh * kw * cp), h16), kh * kw * cp


def geglu_interleave(w, b):
    """GEGLU projection [2*inner][K] (rows: values then gates) -> rows interleaved in blocks of 16
    [v0..15 | g0..15 | v16..31 | g16..31 ...] so that value and gate of one output land in the same lane."""
    two_inner, K = w.shape
    inner = two_inner // 2
    assert inner % 16 == 0
    wv, wg = w[:inner].reshape(inner // 16, 16, K), w[inner:].reshape(inner // 16, 16, K)
    wi = torch.stack([wv, wg], 1).reshape(two_inner, K)
    bv, bg = b[:inner].reshape(inner // 16, 16), b[inner:].reshape(inner // 16, 16)
    bi = torch.stack([bv, bg], 1).reshape(two_inner)
    return wi, bi
