"""ProPainter's recurrent flow-completion network on the HIP kernels (oracle: oracle/flowcomplete_ref.py).

Reference: third-party `propainter` model/recurrent_flow_completion.py (RecurrentFlowCompleteNet.forward_bidirect_flow + combine_flow),
reached from reference diffuerase.py:52-57 through `Propainter.forward`: the RAFT flows of the masked clip are completed inside the holes
before the flow-guided propagation.  Layout: NHWC rows [T*H*W, C]; every convolution is `vv_conv_gemm`:
  Conv3d (1,k,k)              -> a 2-D conv over the T frames (F = T)
  Conv3d (3,1,1), dilation 2  -> three 1x1 GEMMs over a time-padded buffer (taps t-2, t, t+2 accumulate through the fp32 residual input)
  dilated (1,3,3) convs       -> `vv_deform_im2col` with zero offsets (its sampling grid carries the dilation) + 1x1 GEMM
  deformable alignment        -> `vv_deform_im2col` in raw mode (tanh / sigmoid of the offset stack fused) + 1x1 GEMM
  deconv                      -> `vv_upsample2x_bilinear` + conv
No CPU fallback; torch only allocates, slices and concatenates device tensors."""
import torch

from . import hip, packing
from .deform import DeformConv2d


class _Conv:
    def __init__(self, ctx, name, cin, cout, k=3, gain=1.0):
        self.ctx, self.k, self.cout = ctx, k, cout
        w, b = ctx.src.conv(name, cin, cout, k, gain)
        cpad = (cin + 7) // 8 * 8
        wp, self.K = packing.pack_conv(w, ctx.h16, cpad if cpad != cin else None)
        self.w, self.b = ctx.dev(wp), ctx.dev(b.float())

    def __call__(self, x, F, H, W, stride=1, pad=None, act=None, out_dtype=None, res0=None, x1=None, out=None):
        k = self.k
        pad = k // 2 if pad is None else pad
        Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        y = hip.conv_gemm(self.ctx.dt, x, self.w, self.cout, self.K, x1=x1, F=F, Hin=H, Win=W, Hout=Ho, Wout=Wo, ksize=k, stride=stride,
                          pad_t=pad, pad_l=pad, bias=self.b, res0=res0, out=out, out_dtype=out_dtype if out_dtype is not None else self.ctx.h16,
                          act=hip.ACT_LRELU if act else hip.ACT_NONE, act_slope=float(act or 0.0))
        return y, Ho, Wo


class _TemporalConv:
    """Conv3d (3,1,1), padding (2,0,0), dilation (2,1,1): out[t] = W0 x[t-2] + W1 x[t] + W2 x[t+2] + b."""

    def __init__(self, ctx, name, c):
        self.ctx, self.c = ctx, c
        w = ctx.src.normal(name + ".weight", (c, c, 3), std=1.0 / float(3 * c) ** 0.5)
        b = ctx.src.normal(name + ".bias", (c,), std=0.02)
        self.w = [ctx.dev(packing.pack_matrix(w[:, :, i].contiguous(), ctx.h16)) for i in range(3)]
        self.b = ctx.dev(b.float())

    def __call__(self, xpad, T, HW, act):
        """xpad: h16 [(T+4)*HW, c] with two zero frames on either side.  Returns h16 [T*HW, c] with LeakyReLU(act) applied."""
        c, dt = self.c, self.ctx.dt
        n = T * HW
        y = hip.conv_gemm(dt, xpad[0:n], self.w[0], c, c, F=1, Hin=1, Win=n, bias=self.b, out_dtype=torch.float32)
        hip.conv_gemm(dt, xpad[2 * HW:2 * HW + n], self.w[1], c, c, F=1, Hin=1, Win=n, res0=y, out=y)
        return hip.conv_gemm(dt, xpad[4 * HW:4 * HW + n], self.w[2], c, c, F=1, Hin=1, Win=n, res0=y, out_dtype=self.ctx.h16,
                             act=hip.ACT_LRELU, act_slope=act)


class _P3D:
    def __init__(self, ctx, name, cin, cout, stride):
        self.ctx, self.stride, self.cout = ctx, stride, cout
        self.conv1 = _Conv(ctx, name + ".conv1.0", cin, cout, 3)
        self.conv2 = _TemporalConv(ctx, name + ".conv2.0", cout)

    def __call__(self, x, T, H, W, act=0.2):
        s = self.stride
        Ho, Wo = (H + 2 - 3) // s + 1, (W + 2 - 3) // s + 1
        HW = Ho * Wo
        buf = torch.zeros(((T + 4) * HW, self.cout), dtype=self.ctx.h16, device=x.device)
        self.conv1(x, T, H, W, stride=s, act=0.2, out=buf[2 * HW:(T + 2) * HW])
        return self.conv2(buf, T, HW, act), Ho, Wo


class _DilatedConv:
    """(1,3,3) conv with dilation d, padding d: the deformed-im2col gather with zero offsets carries the dilation."""

    def __init__(self, ctx, name, c, dil):
        self.ctx, self.dil = ctx, dil
        self.dcn = DeformConv2d(ctx, name, c, c, 3, 1, dil, dil, deform_groups=1)
        self.zero = None

    def __call__(self, x, T, H, W, act, out_dtype):
        M = T * H * W
        if self.zero is None or self.zero.shape[0] != M:
            self.zero = torch.zeros((M, 18), dtype=torch.float32, device=x.device)
        col, _, _ = hip.deform_im2col(self.ctx.dt, x, B=T, H=H, W=W, kh=3, kw=3, stride=1, pad=self.dil, dil=self.dil, deform_groups=1, offset=self.zero)
        return hip.conv_gemm(self.ctx.dt, col, self.dcn.w, self.dcn.cout, self.dcn.K, F=T, Hin=H, Win=W, ksize=1, bias=self.dcn.b,
                             out_dtype=out_dtype, act=hip.ACT_LRELU, act_slope=act)


class _SecondOrderAlignment:
    def __init__(self, ctx, name, C, deform_groups, max_residue=5.0):
        self.ctx, self.C, self.dg, self.max_residue = ctx, C, deform_groups, max_residue
        chans = [C, C, C, 27 * deform_groups]
        self.stack, cin = [], 3 * C
        for i, co in enumerate(chans):
            self.stack.append(_Conv(ctx, f"{name}.conv_offset.{2 * i}", cin, co, 3, 0.1 if i == 3 else 1.0))
            cin = co
        self.dcn = DeformConv2d(ctx, name, 2 * C, C, 3, 1, 1, 1, deform_groups)

    def __call__(self, x2, cond, H, W):
        h = cond
        for i, conv in enumerate(self.stack):
            last = i == len(self.stack) - 1
            h, _, _ = conv(h, 1, H, W, act=None if last else 0.1, out_dtype=torch.float32 if last else self.ctx.h16)
        out, _, _ = self.dcn(x2, 1, H, W, raw=h, max_residue=self.max_residue, out_dtype=torch.float32)
        return out


class _BidirectionalPropagation:
    def __init__(self, ctx, name, C, deform_groups):
        self.ctx, self.C = ctx, C
        self.align, self.bb0, self.bb2 = {}, {}, {}
        for i, mod in enumerate(("backward_", "forward_")):
            self.align[mod] = _SecondOrderAlignment(ctx, f"{name}.deform_align.{mod}", C, deform_groups)
            self.bb0[mod] = _Conv(ctx, f"{name}.backbone.{mod}.0", (2 + i) * C, C, 3)
            self.bb2[mod] = _Conv(ctx, f"{name}.backbone.{mod}.2", C, C, 3)
        self.fusion = _Conv(ctx, f"{name}.fusion", 2 * C, C, 1)

    def __call__(self, x, T, H, W):
        """x fp32 [T*H*W, C] -> fp32 [T*H*W, C]."""
        C, HW = self.C, H * W
        spatial = [x[t * HW:(t + 1) * HW] for t in range(T)]
        feats = {}
        for mod in ("backward_", "forward_"):
            feats[mod] = []
            order = list(range(T))
            if mod == "backward_":
                order = order[::-1]
            prop = torch.zeros((HW, C), dtype=torch.float32, device=x.device)
            for i, idx in enumerate(order):
                cur = spatial[idx]
                if i > 0:
                    n2 = feats[mod][-2] if i > 1 else torch.zeros_like(prop)
                    prop = self.align[mod](torch.cat([prop, n2], 1), torch.cat([prop, cur, n2], 1), H, W)
                parts = [cur] + ([feats["backward_"][idx]] if mod == "forward_" else []) + [prop]
                h, _, _ = self.bb0[mod](torch.cat(parts, 1), 1, H, W, act=0.1)
                prop, _, _ = self.bb2[mod](h, 1, H, W, res0=prop, out_dtype=torch.float32)
                feats[mod].append(prop)
            if mod == "backward_":
                feats[mod] = feats[mod][::-1]
        bwd, fwd = torch.cat(feats["backward_"], 0), torch.cat(feats["forward_"], 0)
        out, _, _ = self.fusion(bwd, T, H, W, pad=0, x1=fwd, res0=x, out_dtype=torch.float32)
        return out


class FlowCompleteNet:
    """RecurrentFlowCompleteNet (inference path).  `complete` = its forward(); `forward_bidirect_flow` + `combine_flow` as in the reference."""

    def __init__(self, ctx, width=(32, 64, 128), deform_groups=16, name="fc"):
        self.ctx, self.width = ctx, width
        c1, c2, c3 = width
        self.down = _Conv(ctx, f"{name}.downsample.0", 3, c1, 5)
        self.e10, self.e12 = _P3D(ctx, f"{name}.encoder1.0", c1, c1, 1), _P3D(ctx, f"{name}.encoder1.2", c1, c2, 2)
        self.e20, self.e22 = _P3D(ctx, f"{name}.encoder2.0", c2, c2, 1), _P3D(ctx, f"{name}.encoder2.2", c2, c3, 2)
        self.mid = [_DilatedConv(ctx, f"{name}.mid_dilation.{2 * i}", c3, d) for i, d in enumerate((3, 2, 1))]
        self.prop = _BidirectionalPropagation(ctx, f"{name}.feat_prop_module", c3, deform_groups)
        self.d20, self.d22 = _Conv(ctx, f"{name}.decoder2.0", c3, c3), _Conv(ctx, f"{name}.decoder2.2.conv", c3, c2)
        self.d10, self.d12 = _Conv(ctx, f"{name}.decoder1.0", c2, c2), _Conv(ctx, f"{name}.decoder1.2.conv", c2, c1)
        self.u0, self.u2 = _Conv(ctx, f"{name}.upsample.0", c1, c1), _Conv(ctx, f"{name}.upsample.2.conv", c1, 2)

    def complete(self, flow, mask_u8):
        """flow fp32 [T,H,W,2] (masked inside by this call), mask u8 [T,H,W] (non-zero = hole), H, W % 8 == 0 -> fp32 [T*H*W, 2]."""
        T, H, W, _ = flow.shape
        if H % 8 or W % 8:
            raise RuntimeError(f"FlowCompleteNet: H={H}, W={W} must be multiples of 8")
        dt = self.ctx.dt
        xin = hip.fc_input(flow.contiguous(), mask_u8.contiguous(), 2)
        x, H2, W2 = self.down(xin.reshape(-1, 8), T, H + 4, W + 4, stride=2, pad=0, act=0.2)
        e1, _, _ = self.e10(x, T, H2, W2)
        e1, H4, W4 = self.e12(e1, T, H2, W2)
        e2, _, _ = self.e20(e1, T, H4, W4)
        e2, H8, W8 = self.e22(e2, T, H4, W4)
        mid = e2
        for i, conv in enumerate(self.mid):
            mid = conv(mid, T, H8, W8, 0.2, torch.float32 if i == 2 else self.ctx.h16)
        prop = self.prop(mid, T, H8, W8)
        d2, _, _ = self.d20(prop, T, H8, W8, act=0.2)
        d2, _, _ = self.d22(hip.upsample2x_bilinear(dt, d2, T, H8, W8), T, H4, W4, act=0.2, out_dtype=torch.float32)
        hip.add_inplace(dt, d2, e1)
        d1, _, _ = self.d10(d2, T, H4, W4, act=0.2)
        d1, _, _ = self.d12(hip.upsample2x_bilinear(dt, d1, T, H4, W4), T, H2, W2, act=0.2)
        up, _, _ = self.u0(d1, T, H2, W2, act=0.2)
        pred, _, _ = self.u2(hip.upsample2x_bilinear(dt, up, T, H2, W2), T, H, W, out_dtype=torch.float32)
        return pred

    def forward_bidirect_flow(self, flows_fw, flows_bw, masks_u8):
        """flows fp32 [T-1,H,W,2] (t -> t+1 / t+1 -> t), masks u8 [T,H,W] -> raw predictions (fw, bw), each fp32 [T-1,H,W,2]."""
        Tm, H, W, _ = flows_fw.shape
        pf = self.complete(flows_fw, masks_u8[:-1]).reshape(Tm, H, W, 2)
        pb = self.complete(torch.flip(flows_bw, dims=[0]), torch.flip(masks_u8[1:], dims=[0])).reshape(Tm, H, W, 2)
        return pf, torch.flip(pb, dims=[0])

    def complete_flows(self, flows_fw, flows_bw, masks_u8):
        """forward_bidirect_flow + combine_flow: predicted flow inside the holes, measured flow outside."""
        pf, pb = self.forward_bidirect_flow(flows_fw, flows_bw, masks_u8)
        return (hip.flow_combine(pf.reshape(-1, 2), flows_fw.contiguous(), masks_u8[:-1].contiguous()),
                hip.flow_combine(pb.contiguous().reshape(-1, 2), flows_bw.contiguous(), masks_u8[1:].contiguous()))
