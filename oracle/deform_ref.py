"""ORACLE (test infrastructure): modulated deformable convolution (DCNv2) and ProPainter's flow-guided deformable
alignment, fp32 torch on the CPU.

SURVEY.md row n1: the full ProPainter prior (third-party `Propainter.forward`, call site reference diffuerase.py:52-57)
aligns neighbouring-frame features with `torchvision.ops.deform_conv2d` inside its `DeformableAlignment` modules
(ProPainter `model/propainter.py`: class DeformableAlignment(ModulatedDeformConv2d), second-order, deform_groups = 16,
max_residue_magnitude = 3 in the inpainting generator; `model/recurrent_flow_completion.py` uses the same operator
without flow guidance).  PARITY UNPINNED: neither `propainter` nor `torchvision` is present in the build image
(install_videovanish.sh:78 pulls the former un-pinned); this file restates the published operator:

  out[b, co, y, x] = bias[co] + sum_{ci, ky, kx} w[co, ci, ky, kx] * m[b, g, k, y, x] * bilinear(in[b, ci], py, px)
      py = y * stride - pad + ky * dil + off[b, 2 * (g * K + k),     y, x]          g = ci // (Cin / deform_groups)
      px = x * stride - pad + kx * dil + off[b, 2 * (g * K + k) + 1, y, x]          k = ky * kw + kx

with torchvision's sampling rule (deform_conv2d_kernel.cpp, bilinear_interpolate): a sample at or beyond -1 / H (W) is 0,
otherwise the four neighbours are blended with weights (1 - lh)(1 - lw) ..., neighbours outside the image count as 0.
What pins it without the real operator: zero offsets + unit mask == F.conv2d bit for bit in structure (tested to 1e-6),
integer offsets == a conv of the shifted image, and F.grid_sample(align_corners=True, padding_mode="zeros") agrees
with the sampler wherever the sample lies inside (-1, H) x (-1, W) (tests/test_deform_cpu.py).
"""
import torch
import torch.nn.functional as F


def bilinear_zero(img, py, px):
    """img [B, C, H, W]; py / px [B, 1 or C, Ho, Wo] sampling positions -> [B, C, Ho, Wo] (torchvision's rule)."""
    B, C, H, W = img.shape
    inside = (py > -1) & (py < H) & (px > -1) & (px < W)
    y0 = torch.floor(py)
    x0 = torch.floor(px)
    lh, lw = py - y0, px - x0
    y0, x0 = y0.long(), x0.long()
    y1, x1 = y0 + 1, x0 + 1
    flat = img.reshape(B, C, H * W)

    def tap(yy, xx):
        ok = (yy >= 0) & (yy <= H - 1) & (xx >= 0) & (xx <= W - 1)
        idx = (yy.clamp(0, H - 1) * W + xx.clamp(0, W - 1)).expand(B, C, *yy.shape[2:]).reshape(B, C, -1)
        v = torch.gather(flat, 2, idx).reshape(B, C, *yy.shape[2:])
        return v * ok.to(img.dtype)

    val = ((1 - lh) * (1 - lw)) * tap(y0, x0) + ((1 - lh) * lw) * tap(y0, x1) + (lh * (1 - lw)) * tap(y1, x0) + (lh * lw) * tap(y1, x1)
    return val * inside.to(img.dtype)


def deform_columns(x, offset, mask, kh, kw, stride=1, pad=1, dil=1, deform_groups=1):
    """The deformed im2col matrix [B, Cin * K, Ho, Wo] in tap-major order (k * Cin + ci), the product kernel's layout."""
    B, C, H, W = x.shape
    Ho = (H + 2 * pad - dil * (kh - 1) - 1) // stride + 1
    Wo = (W + 2 * pad - dil * (kw - 1) - 1) // stride + 1
    K, cpg = kh * kw, C // deform_groups
    ys = (torch.arange(Ho, dtype=x.dtype) * stride - pad).view(1, 1, Ho, 1)
    xs = (torch.arange(Wo, dtype=x.dtype) * stride - pad).view(1, 1, 1, Wo)
    cols = []
    for k in range(K):
        ky, kx = divmod(k, kw)
        per_g = []
        for g in range(deform_groups):
            dy = offset[:, 2 * (g * K + k): 2 * (g * K + k) + 1]
            dx = offset[:, 2 * (g * K + k) + 1: 2 * (g * K + k) + 2]
            v = bilinear_zero(x[:, g * cpg:(g + 1) * cpg], ys + ky * dil + dy, xs + kx * dil + dx)
            if mask is not None:
                v = v * mask[:, g * K + k: g * K + k + 1]
            per_g.append(v)
        cols.append(torch.cat(per_g, 1))
    return torch.cat(cols, 1)


def deform_conv2d(x, offset, weight, bias=None, stride=1, pad=1, dil=1, mask=None):
    """torchvision.ops.deform_conv2d(x, offset, weight, bias, stride, padding, dilation, mask) for groups = 1."""
    Cout, Cin, kh, kw = weight.shape
    K = kh * kw
    dg = offset.shape[1] // (2 * K)
    col = deform_columns(x, offset, mask, kh, kw, stride, pad, dil, dg)               # [B, K*Cin, Ho, Wo]
    w = weight.permute(0, 2, 3, 1).reshape(Cout, K * Cin)                             # tap-major like the columns
    out = torch.einsum("ok,bkyx->boyx", w, col)
    return out if bias is None else out + bias.view(1, -1, 1, 1)


# ---------------------------------------------------------------------------------------------------------------------
# ProPainter DeformableAlignment (model/propainter.py): offsets and modulation mask from a 4-conv stack over
# [warped features | current feature | flows], offset = max_residue * tanh(o) + flow (flipped to (dy, dx)), mask = sigmoid.
def deformable_alignment(P, name, x, cond, flow, C, deform_groups=16, max_residue=3.0):
    """x: features to align [B, C, H, W]; cond: conditioning stack [B, Cc, H, W] (the caller concatenates warped features, the
    current feature and the flow); flow [B, 2, H, W] as (dx, dy).  Returns the aligned features [B, C, H, W].
    Weights come from P (oracle.model_ref.Params): `name.conv_offset.{0,2,4,6}` and `name.weight/.bias`."""
    K = 9
    h = cond
    chans = [C, C, C, 3 * K * deform_groups]
    for i, co in enumerate(chans):
        w, b = P.conv(f"{name}.conv_offset.{2 * i}", h.shape[1], co, 3, gain=(0.1 if i == 3 else 1.0))
        h = F.conv2d(h, w, b, padding=1)
        if i < 3:
            h = F.leaky_relu(h, 0.1)
    o1, o2, m = torch.chunk(h, 3, dim=1)
    offset = max_residue * torch.tanh(torch.cat([o1, o2], 1))
    offset = offset + flow.flip(1).repeat(1, offset.shape[1] // 2, 1, 1)
    m = torch.sigmoid(m)
    w, b = P.conv(f"{name}", C, C, 3)
    return deform_conv2d(x, offset, w, b, 1, 1, 1, m)
