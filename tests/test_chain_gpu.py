"""The fused tail of a level-0 spatial transformer block (csrc/vv_chain.hip: attn1 output projection, cross-attention to the 77 text tokens,
GEGLU feed-forward, proj_out, all residuals in one kernel at C = 320) against the fp32 oracle (oracle/model_ref.py::spatial_transformer) and
against the layer-by-layer HIP path on the same weights."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from videovanish_amd.config import UNetConfig


def _rel(a, b):
    return ((a - b).abs().max() / b.abs().max()).item()


@pytest.mark.parametrize("dname,tol", [("fp16", 3e-3), ("bf16", 2.5e-2)])
@pytest.mark.parametrize("Fr,H,W", [(2, 8, 8), (3, 6, 7), (1, 16, 24)])          # 128 tokens (one full block), 126 (ragged), 384
def test_fused_spatial_chain_vs_oracle_and_unfused(gpu, dname, tol, Fr, H, W):
    from oracle import model_ref as M
    from videovanish_amd import nn as vnn
    cfg = UNetConfig()
    C = 320
    name = "unet.down_blocks.0.attentions.0"
    g = torch.Generator().manual_seed(23)
    x = torch.randn(Fr, C, H, W, generator=g) * 1.3 + 0.1
    P = M.Params(0)
    text = M.text_states(P, cfg)
    with torch.no_grad():
        ref = M.spatial_transformer(P, name, x, text, cfg)
    ctx = vnn.Ctx("cuda:0", dname, 0)
    mod = vnn.SpatialTransformer(ctx, name, C, cfg, ctx.dev(text[0], ctx.h16))
    assert mod.fused is not None and mod.fused[0].shape == (462, 64, 64) and mod.fused[1].numel() == 5120
    assert mod.front[0].shape == (100, 64, 64) and mod.front[1].numel() == 960
    nhwc = lambda t: t.permute(0, 2, 3, 1).reshape(Fr * H * W, C).contiguous().to(gpu)
    back = lambda t: t.float().cpu().reshape(Fr, H, W, C).permute(0, 3, 1, 2)
    xin = nhwc(x)
    vnn.SpatialTransformer.FUSED = True
    fused = back(mod(xin, Fr, H, W))
    vnn.SpatialTransformer.FUSED = False
    try:
        plain = back(mod(xin, Fr, H, W))
    finally:
        vnn.SpatialTransformer.FUSED = True
    e_f, e_p, e_fp = _rel(fused, ref), _rel(plain, ref), _rel(fused, plain)
    print(f"spatial transformer [{dname}, {Fr}x{H}x{W}]: fused vs oracle {e_f:.2e}, layer-by-layer vs oracle {e_p:.2e}, fused vs layer-by-layer {e_fp:.2e}")
    assert torch.isfinite(fused).all()
    assert e_f <= tol and e_f <= 2.0 * e_p + 1e-4
    out16 = mod(xin, Fr, H, W, out_dtype=ctx.h16)
    assert out16.dtype == ctx.h16 and _rel(back(out16), fused) <= (2e-3 if dname == "fp16" else 1.6e-2)


def test_fused_chain_kernels_are_run_to_run_deterministic(gpu):
    """The row-split kernels hand activations between the two waves of a pair through LDS (round 5): an ordering hazard there would show as launches of the
    same inputs that differ.  30 launches of the front and of the tail at 80 blocks (more blocks than one pass over a slice of the CUs, fewer than the
    chip: both full and ragged occupancy occur), bit for bit equal; fp16 and bf16."""
    from videovanish_amd import hip, nn as vnn
    cfg = UNetConfig()
    C, Fr, H, W = 320, 4, 40, 64
    M = Fr * H * W
    g = torch.Generator().manual_seed(5)
    for dname in ("fp16", "bf16"):
        ctx = vnn.Ctx("cuda:0", dname, 0)
        text = ctx.dev(torch.randn(77, 768, generator=g), ctx.h16)
        mod = vnn.SpatialTransformer(ctx, "unet.down_blocks.0.attentions.0", C, cfg, text)
        x = torch.randn(M, C, generator=g).to(gpu)
        t0, qkv0 = hip.spatial_chain_front_c320(ctx.dt, x, mod.norm.g, mod.norm.b, mod.norm.groups, mod.norm.eps, mod.front[0], mod.front[1], F=Fr, HW=H * W)
        o = torch.randn(M, C, generator=g).to(gpu).to(ctx.h16)
        out0 = hip.spatial_chain_c320(ctx.dt, o, t0, x, mod.fused[0], mod.fused[1])
        for _ in range(30):
            t1, qkv1 = hip.spatial_chain_front_c320(ctx.dt, x, mod.norm.g, mod.norm.b, mod.norm.groups, mod.norm.eps, mod.front[0], mod.front[1], F=Fr, HW=H * W)
            out1 = hip.spatial_chain_c320(ctx.dt, o, t0, x, mod.fused[0], mod.fused[1])
            assert torch.equal(t1, t0) and torch.equal(qkv1, qkv0) and torch.equal(out1, out0)
        assert torch.isfinite(out0).all()


def test_head_major_attention_output_is_bit_identical(gpu):
    """Round 6: between the self-attention core and the fused tail the O tensor is head-major [frame][head][token][40]
    (vv_attn_params.o_hs / vv_chain_params.o_hw): the attention kernels store whole contiguous 80-byte records instead of an 80-byte slice of a 640-byte
    row.  Same values at other addresses: the attention output permuted back equals the row-major output bit for bit (d = 40 spatial kernels incl. a ragged
    token count, and the generic kernel at d = 64), and the block's result is bit-identical with either layout -- also when a 128-token block of the tail
    straddles two frames (HW = 1000) and at the bench geometry's frame size."""
    from videovanish_amd import hip, nn as vnn
    cfg = UNetConfig()
    g = torch.Generator().manual_seed(9)
    ctx = vnn.Ctx("cuda:0", "fp16", 0)
    for (B, heads, N, D) in ((2, 8, 1000, 40), (1, 8, 4096, 40), (3, 4, 200, 64), (2, 8, 900, 80)):
        C = heads * D
        qkv = (torch.randn(B, 3, heads, N, D, generator=g) * 0.5).to(gpu).to(ctx.h16)
        outs = []
        for hm in (False, True):
            o = torch.zeros((B, heads, N, D) if hm else (B * N, C), dtype=ctx.h16, device=gpu)
            hip.attention(ctx.dt, qkv, qkv, qkv, o, B=B, heads=heads, Nq=N, Nkv=N, D=D, q_bs=N * 3 * C, k_bs=N * 3 * C, v_bs=N * 3 * C, o_bs=N * C,
                          q_rs=D, k_rs=D, v_rs=D, o_rs=D if hm else C, k_off=N * C, v_off=2 * N * C, q_hs=N * D, k_hs=N * D, v_hs=N * D,
                          o_hs=N * D if hm else 0)
            outs.append(o.permute(0, 2, 1, 3).reshape(B * N, C) if hm else o)
        assert torch.isfinite(outs[0].float()).all() and torch.equal(outs[0], outs[1]), (B, heads, N, D)
    text = ctx.dev(torch.randn(77, 768, generator=g), ctx.h16)
    mod = vnn.SpatialTransformer(ctx, "unet.down_blocks.0.attentions.0", 320, cfg, text)
    for (Fr, H, W) in ((3, 25, 40), (2, 90, 160)):
        x = torch.randn(Fr * H * W, 320, generator=g).to(gpu)
        res = []
        try:
            for hm in (True, False):
                vnn.SpatialTransformer.HEAD_MAJOR_O = hm
                res.append(mod(x, Fr, H, W))
        finally:
            vnn.SpatialTransformer.HEAD_MAJOR_O = True
        assert torch.isfinite(res[0]).all() and torch.equal(res[0], res[1]), (Fr, H, W)


def test_groupnorm_partials_from_the_conv_epilogue(gpu):
    """Round 6: the 128 x 160 halo-tile 3x3 kernel leaves per-channel (sum, sum of squares) partials of its FINAL output (bias + residual
    included) for the GroupNorm that reads it next (vv_conv_params.gn_partials).  (1) finalized, they give the mean / rstd of the stored tensor (fp64 reference)
    for ragged image sizes too (rows outside the image count as zero), per frame and pooled; (2) the output itself is bit-identical with and without partials;
    (3) two launches give identical partials (fixed order, no atomics); (4) a ResBlock + spatial transformer pair computes the same result through the
    partials as through the statistics pass (to fp32 rounding of the statistics); (5) a launch that cannot honour the request is refused by name."""
    from videovanish_amd import hip, nn as vnn
    cfg = UNetConfig()
    g = torch.Generator().manual_seed(21)
    ctx = vnn.Ctx("cuda:0", "fp16", 0)
    for (Fr, H, W, C) in ((2, 90, 160, 320), (3, 45, 80, 640), (2, 37, 50, 320)):
        conv = vnn.Conv(ctx, "unet.down_blocks.0.resnets.0.conv2", C, C)
        x = (torch.randn(Fr * H * W, C, generator=g) * 0.7).to(gpu).to(ctx.h16)
        res = torch.randn(Fr * H * W, C, generator=g).to(gpu)
        plain, _, _ = conv(x, Fr, H, W, res0=res)
        out, _, _ = conv(x, Fr, H, W, res0=res, gn_partials=True)
        out2, _, _ = conv(x, Fr, H, W, res0=res, gn_partials=True)
        halo_anyway = ((H + 7) // 8) * 8 * ((W + 15) // 16) * 16 * 100 <= H * W * 115      # the dispatcher's own rule for the halo-tile kernel (vv_gemm.hip)
        if halo_anyway:
            assert torch.equal(out, plain)                     # the same kernel with the extra epilogue step: the stored tensor is bit-identical
        else:
            assert (out - plain).abs().max().item() <= 1e-5 * plain.abs().max().item()      # (37 x 50: the plain launch runs another loader / k order)
        assert torch.equal(out, out2) and torch.equal(out.vv_gn.part, out2.vv_gn.part)
        ref = out.double().view(Fr, H * W, 32, C // 32)
        for pool in (False, True):
            fin = out.vv_gn.finalize(32, 1e-6, pool_frames=pool).double()
            dims = (0, 1, 3) if pool else (1, 3)
            mean = ref.mean(dim=dims, keepdim=True).expand(Fr, 1, 32, 1).reshape(Fr, 32)
            var = ref.var(dim=dims, unbiased=False, keepdim=True).expand(Fr, 1, 32, 1).reshape(Fr, 32)
            assert (fin[..., 0] - mean).abs().max().item() <= 2e-6 * max(1.0, mean.abs().max().item()) + 2e-6
            assert ((fin[..., 1] - (var + 1e-6).rsqrt()).abs() / (var + 1e-6).rsqrt()).max().item() <= 2e-5
    # (4) the layer pair through both routes
    text = ctx.dev(torch.randn(77, 768, generator=g), ctx.h16)
    rb = vnn.ResBlock(ctx, "unet.down_blocks.0.resnets.0", 320, 320, cfg.groups, 1e-5, cfg.temb_dim, h16_mid=True)
    st = vnn.SpatialTransformer(ctx, "unet.down_blocks.0.attentions.0", 320, cfg, text)
    Fr, H, W = 2, 48, 64
    x = torch.randn(Fr * H * W, 320, generator=g).to(gpu)
    temb = torch.randn(1, cfg.temb_dim, generator=g).to(gpu)
    outs = []
    try:
        for flag in (True, False):
            vnn.ResBlock.GN_FROM_EPILOGUE = flag          # conv1 -> norm2 inside the block too, conv2 -> the transformer's GroupNorm
            y = rb(x, Fr, H, W, silu_temb=temb, want_gn=True)
            assert (getattr(y, "vv_gn", None) is not None) == flag
            outs.append(st(y, Fr, H, W))
    finally:
        vnn.ResBlock.GN_FROM_EPILOGUE = True
    rel = ((outs[0] - outs[1]).abs().max() / outs[1].abs().max()).item()
    print(f"spatial transformer through conv-epilogue GroupNorm partials vs statistics pass: rel max {rel:.2e}")
    assert torch.isfinite(outs[0]).all() and rel <= 1e-3
    # (5) not the halo-tile kernel: a 1x1 convolution cannot write partials
    lin = vnn.Conv(ctx, "unet.down_blocks.0.attentions.0.proj_in", 320, 320, k=1)
    with pytest.raises(RuntimeError, match="gn_partials"):
        lin(x.to(ctx.h16), Fr, H, W, gn_partials=True)
