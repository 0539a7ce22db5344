"""Modulated deformable convolution and ProPainter's flow-guided deformable alignment on the HIP kernels.

Reference: third-party `propainter` (model/propainter.py: class DeformableAlignment(ModulatedDeformConv2d) ->
torchvision.ops.deform_conv2d), reached from reference diffuerase.py:52-57 through `Propainter.forward`.
Layout: NHWC rows [M = B*H*W, C] like the rest of the package.  One deformable conv = `vv_deform_im2col` (gather: HBM/L2 bound)
+ `vv_conv_gemm` with ksize = 1 over K*C columns (MFMA).  No CPU fallback: everything goes through libvvhip.so."""
import torch

from . import hip, packing


class DeformConv2d:
    """torchvision.ops.deform_conv2d(x, offset, weight, bias, stride, padding, dilation, mask) for groups = 1."""

    def __init__(self, ctx, name, cin, cout, k=3, stride=1, pad=1, dil=1, deform_groups=1, bias=True, weight=None, bias_t=None):
        self.ctx, self.cin, self.cout, self.k, self.stride, self.pad, self.dil, self.dg = ctx, cin, cout, k, stride, pad, dil, deform_groups
        if weight is None:
            weight, b = ctx.src.conv(name, cin, cout, k, 1.0)
            bias_t = b if bias else None
        wp, self.K = packing.pack_conv(weight, ctx.h16, None)          # [Npad][k*k*cin], tap-major: the column order of vv_deform_im2col
        self.w = ctx.dev(wp)
        self.b = ctx.dev(bias_t.float()) if bias_t is not None else None

    def __call__(self, x, B, H, W, offset=None, mask=None, raw=None, flow=None, max_residue=0.0, out_dtype=torch.float32, res0=None):
        col, Ho, Wo = hip.deform_im2col(self.ctx.dt, x, B=B, H=H, W=W, kh=self.k, kw=self.k, stride=self.stride, pad=self.pad, dil=self.dil,
                                        deform_groups=self.dg, offset=offset, mask=mask, raw=raw, flow=flow, max_residue=max_residue)
        out = hip.conv_gemm(self.ctx.dt, col, self.w, self.cout, self.K, F=B, Hin=Ho, Win=Wo, ksize=1, bias=self.b, res0=res0, out_dtype=out_dtype)
        return out, Ho, Wo


class DeformableAlignment:
    """ProPainter DeformableAlignment: offsets and modulation from a 4-conv stack (LeakyReLU 0.1) over the conditioning features,
    offset = max_residue * tanh(.) + flow, mask = sigmoid(.), then the modulated deformable 3x3 convolution."""

    def __init__(self, ctx, name, C, cond_channels, deform_groups=16, max_residue=3.0):
        self.ctx, self.C, self.dg, self.max_residue = ctx, C, deform_groups, max_residue
        chans = [C, C, C, 27 * deform_groups]
        self.stack = []
        cin = cond_channels
        for i, co in enumerate(chans):
            w, b = ctx.src.conv(f"{name}.conv_offset.{2 * i}", cin, co, 3, 0.1 if i == 3 else 1.0)
            cpad = (cin + 7) // 8 * 8
            wp, K = packing.pack_conv(w, ctx.h16, cpad if cpad != cin else None)
            self.stack.append((ctx.dev(wp), K, ctx.dev(b.float()), co, cpad))
            cin = co
        self.dcn = DeformConv2d(ctx, name, C, C, 3, 1, 1, 1, deform_groups)

    def __call__(self, x, cond, flow, B, H, W, out_dtype=torch.float32):
        """x [M, C] features to align; cond [M, Cc] conditioning stack (h16 or fp32; Cc zero-padded to a multiple of 8 by the caller if
        needed); flow [M, 2] fp32 (dx, dy) or None."""
        h = cond
        if h.shape[-1] != self.stack[0][4]:
            raise RuntimeError(f"DeformableAlignment: cond has {h.shape[-1]} channels, the packed weights expect {self.stack[0][4]} (zero-pad to a multiple of 8)")
        for i, (w, K, b, co, _) in enumerate(self.stack):
            last = i == len(self.stack) - 1
            h = hip.conv_gemm(self.ctx.dt, h, w, co, K, F=B, Hin=H, Win=W, ksize=3, pad_t=1, pad_l=1, bias=b,
                              out_dtype=torch.float32 if last else self.ctx.h16, act=hip.ACT_NONE if last else hip.ACT_LRELU, act_slope=0.1)
        out, _, _ = self.dcn(x, B, H, W, raw=h, flow=flow, max_residue=self.max_residue, out_dtype=out_dtype)
        return out
