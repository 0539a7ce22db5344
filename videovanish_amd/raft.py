"""RAFT optical flow on HIP kernels (reached by the reference through `Propainter.forward`,
diffuerase.py:52-57).  Dense parts (encoders, all-pairs correlation, update-block convolutions) are vv_conv_gemm
launches; the correlation pyramid lookup, GRU gates, convex upsampling are the K9 kernels of vv_flow.hip.
Layout: NHWC, one frame pair per update loop; encoders run batched over all frames."""
import torch

from . import hip, packing

ITERS = 20


class _Conv:
    """kh x kw convolution with optional folded per-channel affine (eval BatchNorm) and fused ReLU."""

    def __init__(self, ctx, name, cin, cout, kh, kw, cin_pad=None, affine=None, w_b=None):
        self.ctx, self.kh, self.kw, self.cout = ctx, kh, kw, cout
        if w_b is None:
            w = ctx.src.normal(name + ".weight", (cout, cin, kh, kw), std=1.0 / float(cin * kh * kw) ** 0.5)
            b = ctx.src.normal(name + ".bias", (cout,), std=0.02)
        else:
            w, b = w_b
        if affine is not None:                      # y = (conv(x) + b) * a + c  ->  fold into weights / bias
            a, c = affine
            w = w * a[:, None, None, None]
            b = b * a + c
        wp, self.K = packing.pack_conv(w, ctx.h16, cin_pad)
        self.w, self.b = ctx.dev(wp), ctx.dev(b.float())

    def __call__(self, x0, F, H, W, x1=None, stride=1, relu=False, out_dtype=torch.float32, out=None, out_col=0, scale=1.0):
        ph, pw = self.kh // 2, self.kw // 2
        Ho, Wo = (H + 2 * ph - self.kh) // stride + 1, (W + 2 * pw - self.kw) // stride + 1
        o = hip.conv_gemm(self.ctx.dt, x0, self.w, self.cout, self.K, x1=x1, F=F, Hin=H, Win=W, Hout=Ho, Wout=Wo, ksize=self.kh,
                          ksize_w=self.kw, stride=stride, pad_t=ph, pad_l=pw, bias=self.b, out_dtype=out_dtype, out=out, out_col=out_col, out_scale=scale,
                          act=hip.ACT_RELU if relu else hip.ACT_NONE)
        return o, Ho, Wo


def _bn_affine(src, name, c):
    g = src.normal(name + ".weight", (c,), 0.1, 1.0)
    be = src.normal(name + ".bias", (c,), 0.1)
    mu = src.normal(name + ".running_mean", (c,), 0.1)
    var = src.normal(name + ".running_var", (c,), 0.1, 1.0).abs() + 0.5
    a = g / torch.sqrt(var + 1e-5)
    return a, be - mu * a


class _Encoder:
    """BasicEncoder; kind 'instance' (feature net) or 'batch' (context net, BN folded into the convs)."""

    def __init__(self, ctx, name, out_dim, kind):
        self.ctx, self.kind = ctx, kind
        aff = (lambda n, c: _bn_affine(ctx.src, n, c)) if kind == "batch" else (lambda n, c: None)
        self.conv1 = _Conv(ctx, name + ".conv1", 3, 64, 7, 7, cin_pad=8, affine=aff(name + ".norm1", 64))
        self.blocks = []
        cin = 64
        for i, (planes, stride) in enumerate([(64, 1), (96, 2), (128, 2)]):
            for j, st in enumerate((stride, 1)):
                n = f"{name}.layer{i + 1}.{j}"
                blk = dict(stride=st, planes=planes,
                           c1=_Conv(ctx, n + ".conv1", cin, planes, 3, 3, affine=aff(n + ".norm1", planes)),
                           c2=_Conv(ctx, n + ".conv2", planes, planes, 3, 3, affine=aff(n + ".norm2", planes)),
                           ds=_Conv(ctx, n + ".downsample.0", cin, planes, 1, 1, affine=aff(n + ".norm3", planes)) if st != 1 else None)
                self.blocks.append(blk)
                cin = planes
        self.conv2 = _Conv(ctx, name + ".conv2", 128, out_dim, 1, 1)
        self._ones, self._zeros = {}, {}

    def _inorm(self, x, F, HW, C, relu):
        if C not in self._ones:
            self._ones[C] = torch.ones(C, dtype=torch.float32, device=self.ctx.device)
            self._zeros[C] = torch.zeros(C, dtype=torch.float32, device=self.ctx.device)
        return hip.groupnorm(self.ctx.dt, x, self._ones[C], self._zeros[C], C, 1e-5, F=F, HW=HW, act=hip.ACT_RELU if relu else hip.ACT_NONE,
                             out_dtype=torch.float32)

    def __call__(self, img8, F, H, W):
        """img8: h16 [F*H*W, 8] -> fp32 [F*h*w, out_dim], h, w."""
        inst = self.kind == "instance"
        x, H, W = self.conv1(img8, F, H, W, stride=2, relu=not inst)
        if inst:
            x = self._inorm(x, F, H * W, 64, True)
        for b in self.blocks:
            st, C = b["stride"], b["planes"]
            y, Ho, Wo = b["c1"](x, F, H, W, stride=st, relu=not inst)
            if inst:
                y = self._inorm(y, F, Ho * Wo, C, True)
            y, _, _ = b["c2"](y, F, Ho, Wo, relu=not inst)
            if inst:
                y = self._inorm(y, F, Ho * Wo, C, True)
            if b["ds"] is not None:
                xs, _, _ = b["ds"](x, F, H, W, stride=st)
                if inst:
                    xs = self._inorm(xs, F, Ho * Wo, C, False)
            else:
                xs = x
            x = hip.add_relu(xs, y)
            H, W = Ho, Wo
        o, _, _ = self.conv2(x, F, H, W)
        return o, H, W


class RAFT:
    def __init__(self, ctx):
        self.ctx = ctx
        self.fnet = _Encoder(ctx, "raft.fnet", 256, "instance")
        self.cnet = _Encoder(ctx, "raft.cnet", 256, "batch")
        pre = "raft.update"
        self.convc1 = _Conv(ctx, pre + ".encoder.convc1", 324, 256, 1, 1, cin_pad=384)
        self.convc2 = _Conv(ctx, pre + ".encoder.convc2", 256, 192, 3, 3)
        self.convf1 = _Conv(ctx, pre + ".encoder.convf1", 2, 128, 7, 7, cin_pad=8)
        self.convf2 = _Conv(ctx, pre + ".encoder.convf2", 128, 64, 3, 3)
        self.conv = _Conv(ctx, pre + ".encoder.conv", 256, 126, 3, 3)
        self.gru = []
        for tag, kh, kw in (("1", 1, 5), ("2", 5, 1)):
            wz = ctx.src.normal(f"{pre}.gru.convz{tag}.weight", (128, 384, kh, kw), std=1.0 / float(384 * kh * kw) ** 0.5)
            bz = ctx.src.normal(f"{pre}.gru.convz{tag}.bias", (128,), std=0.02)
            wr = ctx.src.normal(f"{pre}.gru.convr{tag}.weight", (128, 384, kh, kw), std=1.0 / float(384 * kh * kw) ** 0.5)
            br = ctx.src.normal(f"{pre}.gru.convr{tag}.bias", (128,), std=0.02)
            zr = _Conv(ctx, None, 384, 256, kh, kw, w_b=(torch.cat([wz, wr], 0), torch.cat([bz, br], 0)))    # z and r in one GEMM
            q = _Conv(ctx, f"{pre}.gru.convq{tag}", 384, 128, kh, kw)
            self.gru.append((zr, q))
        self.fh1 = _Conv(ctx, pre + ".flow_head.conv1", 128, 256, 3, 3)
        self.fh2 = _Conv(ctx, pre + ".flow_head.conv2", 256, 2, 3, 3)
        self.mk1 = _Conv(ctx, pre + ".mask.0", 128, 256, 3, 3)
        self.mk2 = _Conv(ctx, pre + ".mask.2", 256, 576, 1, 1)

    def features(self, frames_u8):
        """frames u8 [T,H,W,3] device -> (fmap h16 [T, h*w, 256], ctx fp32 [T, h*w, 256], h, w)."""
        T, H, W, _ = frames_u8.shape
        img8 = hip.raft_prep(self.ctx.dt, frames_u8).view(T * H * W, 8)
        f, h, w = self.fnet(img8, T, H, W)
        c, _, _ = self.cnet(img8, T, H, W)
        return f.view(T, h * w, 256), c.view(T, h * w, 256), h, w

    def corr_pyramid(self, f1_16, f2_16, h, w):
        """all-pairs correlation pyramid of P stacked pairs: f1_16 / f2_16 h16 [P*N, 256] -> 4 fp32 tensors [P*N, h_l, w_l]."""
        N = h * w
        P = f1_16.shape[0] // N
        npad = packing.npad_for(N)
        corr = torch.empty((P * N, N), dtype=torch.float32, device=f1_16.device)
        f2p = torch.zeros((npad, 256), dtype=self.ctx.h16, device=f1_16.device)
        for i in range(P):         # one 14400 x 14400 x 256 GEMM per pair (829 MB of fp32 at 720p), written into its slice
            f2p[:N] = f2_16[i * N: (i + 1) * N]
            hip.conv_gemm(self.ctx.dt, f1_16[i * N: (i + 1) * N], f2p, N, 256, F=1, Hin=N, Win=1, out=corr[i * N: (i + 1) * N], out_scale=1.0 / 16.0)
        pyr = [corr.view(P * N, h, w)]
        for _ in range(3):
            pyr.append(hip.avgpool2(pyr[-1]))
        return pyr

    def flow(self, f1, f2, cn1, h, w, iters=ITERS, trace=None):
        """f1,f2: fp32 [h*w,256] feature maps of the two frames; cn1: fp32 [h*w,256] context of frame 1.
        Returns flow 1->2, fp32 [8h, 8w, 2]."""
        return self.flow_batch(f1, f2, cn1, h, w, iters, trace)[0]

    def flow_batch(self, f1, f2, cn1, h, w, iters=ITERS, trace=None):
        """P stacked pairs: f1, f2, cn1 fp32 [P*h*w, 256] (or [P, h*w, 256]).  Every kernel of the update block runs ONCE per
        iteration over all pairs (rows = stacked h x w grids, conv F = P).  Returns flows fp32 [P, 8h, 8w, 2]."""
        ctx, dt, dev = self.ctx, self.ctx.dt, f1.device
        N = h * w
        f1, f2, cn1 = f1.reshape(-1, 256), f2.reshape(-1, 256), cn1.reshape(-1, 256)
        M = f1.shape[0]
        P = M // N
        f1_16, f2_16 = hip.pad_channels(dt, f1, 256), hip.pad_channels(dt, f2, 256)      # fp32 -> h16 MFMA operands
        pyr = self.corr_pyramid(f1_16, f2_16, h, w)
        net = torch.empty((M, 128), dtype=torch.float32, device=dev)
        net16 = torch.empty((M, 128), dtype=ctx.h16, device=dev)
        xbuf = torch.zeros((M, 256), dtype=ctx.h16, device=dev)            # [inp(128) | motion(126) | flow(2)]
        hip.raft_ctx_split(dt, cn1, net, net16, xbuf)
        ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
        coords1 = torch.stack([xs, ys], -1).reshape(N, 2).repeat(P, 1).contiguous().to(dev)
        flow8 = torch.empty((M, 8), dtype=ctx.h16, device=dev)
        corflo = torch.empty((M, 256), dtype=ctx.h16, device=dev)           # [cor(192) | flo(64)]
        rh = torch.empty((M, 128), dtype=ctx.h16, device=dev)
        mask = None
        if trace is not None:
            trace.update(net0=net.clone(), corr0=pyr[0].clone(), corr3=pyr[3].clone())
        for it in range(iters):
            look = hip.corr_lookup(dt, pyr, coords1)
            hip.raft_flow_prep(dt, coords1, w, h, flow8, xbuf)
            c1, _, _ = self.convc1(look, P, h, w, relu=True, out_dtype=ctx.h16)
            self.convc2(c1, P, h, w, relu=True, out=corflo, out_col=0)
            fl1, _, _ = self.convf1(flow8, P, h, w, relu=True, out_dtype=ctx.h16)
            self.convf2(fl1, P, h, w, relu=True, out=corflo, out_col=192)
            self.conv(corflo, P, h, w, relu=True, out=xbuf, out_col=128)
            for (zr_c, q_c) in self.gru:
                zr, _, _ = zr_c(net16, P, h, w, x1=xbuf)
                hip.gru_rh(dt, zr, net, rh)
                q, _, _ = q_c(rh, P, h, w, x1=xbuf)
                hip.gru_update(dt, zr, q, net, net16)
            d1, _, _ = self.fh1(net16, P, h, w, relu=True, out_dtype=ctx.h16)
            dflow, _, _ = self.fh2(d1, P, h, w)
            if trace is not None and it == 0:
                trace.update(lookup0=look.clone(), dflow0=dflow.clone(), net1=net.clone())
            hip.add_flow(coords1, dflow)
            if it == iters - 1:
                m1, _, _ = self.mk1(net16, P, h, w, relu=True, out_dtype=ctx.h16)
                mask, _, _ = self.mk2(m1, P, h, w, scale=0.25)
        if trace is not None:
            trace.update(coords1=coords1.clone())
        out = hip.convex_upsample(coords1, mask, h, w, F=P)
        return out if P > 1 else out[None]
