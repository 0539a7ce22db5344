"""ORACLE (test infrastructure): ProPainter's recurrent flow-completion network, fp32 torch on the CPU.

SURVEY.md row n1, first of the three learned networks of the full ProPainter prior (third-party `Propainter.forward`, call site
reference diffuerase.py:52-57): RAFT flows of the masked video are completed inside the holes BEFORE the flow-guided propagation
(`fix_flow_complete.forward_bidirect_flow` + `combine_flow` in ProPainter's inference script).  PARITY UNPINNED: `propainter` is
un-vendored, un-pinned third-party code absent from /root/reference (install_videovanish.sh:78) and its weights
(`ruffy369/propainter`, diffuerase.py:49) are not reachable; this file restates the published architecture
(ProPainter model/recurrent_flow_completion.py) with seeded synthetic weights:

  downsample   Conv3d(3 -> 32, (1,5,5), stride (1,2,2), replicate padding) + LeakyReLU(0.2)                     1/2
  encoder1     P3D(32,32) + LReLU, P3D(32,64, stride 2) + LReLU                                                 1/4
  encoder2     P3D(64,64) + LReLU, P3D(64,128, stride 2) + LReLU                                                1/8
  mid_dilation 3 x [Conv3d(128,128,(1,3,3), dilation (1,d,d)) + LReLU], d = 3, 2, 1
  feat_prop    bidirectional second-order propagation: SecondOrderDeformableAlignment (DCNv2, 16 groups, offsets 5 tanh(.),
               no flow guidance) + 2-conv backbone per direction, 1x1 fusion, residual
  decoder2     Conv2d(128,128) + LReLU, [bilinear x2 (align_corners) + Conv2d(128,64)] + LReLU,  + encoder1 skip      1/4
  decoder1     Conv2d(64,64) + LReLU, [x2 + Conv2d(64,32)] + LReLU                                                    1/2
  upsample     Conv2d(32,32) + LReLU, [x2 + Conv2d(32,2)]                                                             1/1
  P3D(ci,co,s) = Conv3d(ci,co,(1,3,3), stride (1,s,s)) + LReLU(0.2), Conv3d(co,co,(3,1,1), padding (2,0,0), dilation (2,1,1))
(the edge detector head only feeds the training loss and is not evaluated at inference.)
"""
import torch
import torch.nn.functional as F

from .deform_ref import deform_conv2d

LR = 0.2


def _w2(P, name, cin, cout, k, gain=1.0):
    return P.conv(name, cin, cout, k, gain)


def _wt(P, name, c):
    """temporal (3,1,1) kernel [c, c, 3] + bias."""
    key = name + "#t"
    if key not in P.cache:
        w = P.src.normal(name + ".weight", (c, c, 3), std=1.0 / float(3 * c) ** 0.5)
        b = P.src.normal(name + ".bias", (c,), std=0.02)
        P.cache[key] = (w, b)
    return P.cache[key]


def conv_1kk(P, name, x, cout, k=3, stride=1, pad=1, dil=1, replicate=False):
    """Conv3d with a (1,k,k) kernel on x [B, C, T, H, W]."""
    w, b = _w2(P, name, x.shape[1], cout, k)
    if replicate:
        x = F.pad(x, (pad, pad, pad, pad, 0, 0), mode="replicate")
        pad = 0
    return F.conv3d(x, w[:, :, None], b, stride=(1, stride, stride), padding=(0, pad, pad), dilation=(1, dil, dil))


def conv_t3(P, name, x):
    """Conv3d (3,1,1), padding (2,0,0), dilation (2,1,1)."""
    w, b = _wt(P, name, x.shape[1])
    return F.conv3d(x, w[:, :, :, None, None], b, padding=(2, 0, 0), dilation=(2, 1, 1))


def p3d(P, name, x, cout, stride):
    h = F.leaky_relu(conv_1kk(P, name + ".conv1.0", x, cout, 3, stride, 1), LR)
    return conv_t3(P, name + ".conv2.0", h)


def conv2(P, name, x, cout, k=3, pad=1):
    w, b = _w2(P, name, x.shape[1], cout, k)
    return F.conv2d(x, w, b, padding=pad)


def deconv(P, name, x, cout):
    x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
    return conv2(P, name + ".conv", x, cout, 3, 1)


def second_order_alignment(P, name, x, cond, C, deform_groups=16, max_residue=5.0):
    """x [B, 2C, H, W] (propagated feature | second-order feature), cond [B, 3C, H, W] -> [B, C, H, W]."""
    h = cond
    for i, co in enumerate([C, C, C, 27 * deform_groups]):
        w, b = P.conv(f"{name}.conv_offset.{2 * i}", h.shape[1], co, 3, 0.1 if i == 3 else 1.0)
        h = F.conv2d(h, w, b, padding=1)
        if i < 3:
            h = F.leaky_relu(h, 0.1)
    o1, o2, m = torch.chunk(h, 3, dim=1)
    offset = max_residue * torch.tanh(torch.cat([o1, o2], 1))
    w, b = P.conv(name, 2 * C, C, 3)
    return deform_conv2d(x, offset, w, b, 1, 1, 1, torch.sigmoid(m))


def bidirectional_propagation(P, name, x, C=128, deform_groups=16):
    """x [B, T, C, H, W] -> same shape."""
    B, T, _, H, W = x.shape
    feats = {"spatial": [x[:, i] for i in range(T)]}
    for di, mod in enumerate(("backward_", "forward_")):
        feats[mod] = []
        order = list(range(T))
        if mod == "backward_":
            order = order[::-1]
        prop = x.new_zeros(B, C, H, W)
        for i, idx in enumerate(order):
            cur = feats["spatial"][idx]
            if i > 0:
                n2 = torch.zeros_like(prop)
                if i > 1:
                    n2 = feats[mod][-2]
                cond = torch.cat([prop, cur, n2], 1)
                prop = second_order_alignment(P, f"{name}.deform_align.{mod}", torch.cat([prop, n2], 1), cond, C, deform_groups)
            feat = [cur] + [feats[k][idx] for k in feats if k not in ("spatial", mod)] + [prop]
            feat = torch.cat(feat, 1)
            h = F.leaky_relu(conv2(P, f"{name}.backbone.{mod}.0", feat, C), 0.1)
            prop = prop + conv2(P, f"{name}.backbone.{mod}.2", h, C)
            feats[mod].append(prop)
        if mod == "backward_":
            feats[mod] = feats[mod][::-1]
    outs = []
    for i in range(T):
        al = torch.cat([feats["backward_"][i], feats["forward_"][i]], 1)
        outs.append(conv2(P, f"{name}.fusion", al, C, 1, 0))
    return torch.stack(outs, 1) + x


def complete(P, masked_flows, masks, width=(32, 64, 128), deform_groups=16, name="fc"):
    """masked_flows [B, T, 2, H, W], masks [B, T, 1, H, W] (1 = hole) -> completed flow [B, T, 2, H, W]; H, W % 8 == 0."""
    c1, c2, c3 = width
    B, T, _, H, W = masked_flows.shape
    inp = torch.cat([masked_flows, masks], 2).permute(0, 2, 1, 3, 4)                      # [B, 3, T, H, W]
    x = F.leaky_relu(conv_1kk(P, f"{name}.downsample.0", inp, c1, 5, 2, 2, replicate=True), LR)
    e1 = F.leaky_relu(p3d(P, f"{name}.encoder1.0", x, c1, 1), LR)
    e1 = F.leaky_relu(p3d(P, f"{name}.encoder1.2", e1, c2, 2), LR)
    e2 = F.leaky_relu(p3d(P, f"{name}.encoder2.0", e1, c2, 1), LR)
    e2 = F.leaky_relu(p3d(P, f"{name}.encoder2.2", e2, c3, 2), LR)
    mid = e2
    for i, d in enumerate((3, 2, 1)):
        mid = F.leaky_relu(conv_1kk(P, f"{name}.mid_dilation.{2 * i}", mid, c3, 3, 1, d, d), LR)
    prop = bidirectional_propagation(P, f"{name}.feat_prop_module", mid.permute(0, 2, 1, 3, 4), c3, deform_groups)
    prop = prop.reshape(B * T, c3, H // 8, W // 8)
    e1r = e1.permute(0, 2, 1, 3, 4).reshape(B * T, c2, H // 4, W // 4)
    d2 = F.leaky_relu(conv2(P, f"{name}.decoder2.0", prop, c3), LR)
    d2 = F.leaky_relu(deconv(P, f"{name}.decoder2.2", d2, c2), LR) + e1r
    d1 = F.leaky_relu(conv2(P, f"{name}.decoder1.0", d2, c2), LR)
    d1 = F.leaky_relu(deconv(P, f"{name}.decoder1.2", d1, c1), LR)
    up = F.leaky_relu(conv2(P, f"{name}.upsample.0", d1, c1), LR)
    flow = deconv(P, f"{name}.upsample.2", up, 2)
    return flow.reshape(B, T, 2, H, W)


def forward_bidirect_flow(P, flows_fw, flows_bw, masks, **kw):
    """flows_fw / flows_bw [B, T-1, 2, H, W] (t -> t+1 / t+1 -> t), masks [B, T, 1, H, W] -> completed (fw, bw) raw predictions."""
    m_fw, m_bw = masks[:, :-1], masks[:, 1:]
    pred_fw = complete(P, flows_fw * (1 - m_fw), m_fw, **kw)
    pred_bw = complete(P, torch.flip(flows_bw * (1 - m_bw), dims=[1]), torch.flip(m_bw, dims=[1]), **kw)
    return pred_fw, torch.flip(pred_bw, dims=[1])


def combine_flow(flows_fw, flows_bw, pred_fw, pred_bw, masks):
    """predicted flow inside the holes, measured flow outside."""
    m_fw, m_bw = masks[:, :-1], masks[:, 1:]
    return pred_fw * m_fw + flows_fw * (1 - m_fw), pred_bw * m_bw + flows_bw * (1 - m_bw)
