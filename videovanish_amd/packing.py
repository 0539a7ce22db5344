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
    if w2d.device.type == "meta":          # shape-only construction (modelhub.manifest): nothing to pack
        return w2d
    out = torch.zeros((npad_for(N, geglu), _round_up(K, 64)), dtype=h16)
    out[:N, :K] = w2d.to(h16)
    return out


def pack_conv(w, h16, cin_pad=None):
    """Conv2d weight [Cout][Cin][k][k] -> ([Npad][Kpad] h16, K) with k ordered (ky, kx, cin).  cin_pad: zero-pad
    the input channels (conv_in layers whose activations are stored with padded channels)."""
    cout, cin, kh, kw = w.shape
    cp = cin if cin_pad is None else cin_pad
    if w.device.type == "meta":
        return w, kh * kw * cp
    t = torch.zeros((cout, kh, kw, cp), dtype=torch.float32)
    t[..., :cin] = w.permute(0, 2, 3, 1)
    return pack_matrix(t.reshape(cout, kh * kw * cp), h16), kh * kw * cp


# taps of a 3x3 kernel over a nearest-2x upsampled image that fall on the SAME source pixel, per output parity (nn.UpConv2x): rows of the 2 x 3
# matrices = the two source taps, columns = the three kernel taps they collect.  "last": parity 0 when the tap below is zero padding (odd Hv).
UPCONV2X_TAPS = {0: ((1.0, 0.0, 0.0), (0.0, 1.0, 1.0)), 1: ((1.0, 1.0, 0.0), (0.0, 0.0, 1.0)), "last": ((1.0, 0.0, 0.0), (0.0, 1.0, 0.0))}


def upconv2x_phase_weight(w, vy, vx):
    """Conv2d weight [Cout][Cin][3][3] -> the 2x2 weight [Cout][Cin][2][2] of output parity (vy, vx) of conv3x3(nearest_upsample_2x(x)) taken over
    x itself: source rows (y - 1, y) for vy = 0 / (y, y + 1) for vy = 1 (pad_t = 1 - vy), columns alike; summed in fp32 before any rounding."""
    w32 = w.float()
    return torch.einsum("ty,oiyx,sx->oits", torch.tensor(UPCONV2X_TAPS[vy], dtype=torch.float32), w32, torch.tensor(UPCONV2X_TAPS[vx], dtype=torch.float32))


def geglu_interleave(w, b):
    """GEGLU projection [2*inner][K] (rows: values then gates) -> rows interleaved in blocks of 16
    [v0..15 | g0..15 | v16..31 | g16..31 ...] so that value and gate of one output land in the same lane."""
    two_inner, K = w.shape
    if w.device.type == "meta":
        return w, b
    inner = two_inner // 2
    assert inner % 16 == 0
    wv, wg = w[:inner].reshape(inner // 16, 16, K), w[inner:].reshape(inner // 16, 16, K)
    wi = torch.stack([wv, wg], 1).reshape(two_inner, K)
    bv, bg = b[:inner].reshape(inner // 16, 16), b[inner:].reshape(inner // 16, 16)
    bi = torch.stack([bv, bg], 1).reshape(two_inner)
    return wi, bi


# ---- fused motion module (csrc/vv_motion.hip): the whole module's weights as one stream of pre-swizzled [64 x 64] h16 slabs ---------
def _perm32():
    """position p = 8 lg + e of a 32-wide MFMA k step holds logical index 16 (e >> 2) + 4 lg + (e & 3) (two 16-row accumulator tiles
    of the previous layer packed into one operand: vv_motion.hip PERM32)."""
    p = torch.arange(32)
    e, lg = p & 7, p >> 3
    return 16 * (e >> 2) + 4 * lg + (e & 3)


def _permute_k(w):
    """[N, K] (K % 32 == 0) -> columns reordered per PERM32 inside every 32-wide k step."""
    N, K = w.shape
    idx = (torch.arange(0, K, 32)[:, None] + _perm32()[None, :]).reshape(-1)
    return w[:, idx]


def _slab(block, h16):
    """[rows <= 64, 64] fp32 -> one [64, 64] h16 slab, 16-byte chunk c of row r stored at chunk c ^ (r & 7)."""
    rows = block.shape[0]
    out = torch.zeros((64, 64), dtype=torch.float32)
    out[:rows] = block
    out = out.to(h16).view(64, 8, 8)
    src = torch.arange(8)[None, :] ^ (torch.arange(64)[:, None] & 7)          # stored chunk j of row r = logical chunk j ^ (r & 7)
    return torch.gather(out, 1, src[:, :, None].expand(64, 8, 8)).reshape(64, 64)


def _dense_slabs(wp, h16, rows_per_block):
    """[N, K] (already k-permuted) -> slabs in the order (row block, k tile)."""
    N, K = wp.shape
    return [_slab(wp[r0:r0 + rows_per_block, k0:k0 + 64], h16) for r0 in range(0, N, rows_per_block) for k0 in range(0, K, 64)]


MOTION_LAYOUT = "tokens"      # stream order of vv_motion.hip's product kernel (4 waves x 32 tokens); "rowsplit": the lab kernel motion_rs_c320_kernel (VV_MOTION_FORM = 1)


def pack_motion_stream(w, h16, heads=8, layout=None):
    """w: dict of fp32 tensors of one motion module at C = 320 -- proj_in/proj_out (.w [C,C], .b), attn1/attn2 (q, k, v, o weights, o bias),
    ln1..3 (g, b), ff1 (w [8C, C], b), ff2 (w [C, 4C], b), pe [32, C].  Returns (stream [670, 64, 64] h16, params [16320] fp32) in the
    consumption order of vv_motion.hip.  layout "rowsplit" (lab kernel): per head k | v | q | Wo (the pair of waves that shares a pixel swaps its key
    tiles while the q slabs stream) and the GEGLU rows of a 64-unit chunk ordered like packing.pack_chain_stream's (row tiles (0, 1) / (2, 3) of the
    chunk's two slab groups = [value | gate] of hidden units 0..15 / 32..47 and 16..31 / 48..63)."""
    layout = layout or MOTION_LAYOUT
    assert layout in ("rowsplit", "tokens")
    if w["proj_in.w"].device.type == "meta":
        return w["proj_in.w"], w["proj_in.b"]
    C = w["proj_in.w"].shape[0]
    D = C // heads
    assert C == 320 and D == 40
    slabs = _dense_slabs(_permute_k(w["proj_in.w"]), h16, 64)
    for a in ("attn1", "attn2"):
        for h in range(heads):
            for nm in (("k", "v", "q") if layout == "rowsplit" else ("q", "k", "v")):
                wh = torch.zeros((48, C))
                wh[:D] = w[f"{a}.{nm}"][h * D:(h + 1) * D]
                slabs += _dense_slabs(_permute_k(wh), h16, 48)
            wo = torch.zeros((C, 64))
            perm = _perm32()
            wo[:, :32] = w[f"{a}.o"][:, h * D + perm]                      # k step 0: d = PERM32 (all < 32)
            p = torch.arange(32)
            e, lg = p & 7, p >> 3
            d1 = 32 + 4 * lg + e                                           # k step 1: accumulator tile 2 (d = 32..47) + a zero tile
            ok = (e < 4) & (d1 < D)
            wo[:, 32 + p[ok]] = w[f"{a}.o"][:, h * D + d1[ok]]
            slabs += _dense_slabs(wo, h16, 64)
    inner = 4 * C
    b1 = []
    for c in range(inner // 64):
        rows = []
        for i in ((0, 2, 1, 3) if layout == "rowsplit" else (0, 1, 2, 3)):
            rows += list(range(64 * c + 16 * i, 64 * c + 16 * i + 16)) + list(range(inner + 64 * c + 16 * i, inner + 64 * c + 16 * i + 16))
        rows = torch.tensor(rows)
        slabs += _dense_slabs(_permute_k(w["ff1.w"][rows]), h16, 64)
        b1.append(w["ff1.b"][rows])
        slabs += _dense_slabs(_permute_k(w["ff2.w"][:, 64 * c:64 * c + 64]), h16, 64)
    slabs += _dense_slabs(_permute_k(w["proj_out.w"]), h16, 64)
    stream = torch.stack(slabs)
    params = torch.cat([w["proj_in.b"], w["ln1.g"], w["ln1.b"], w["attn1.ob"], w["ln2.g"], w["ln2.b"], w["attn2.ob"], w["ln3.g"], w["ln3.b"],
                        torch.cat(b1), w["ff2.b"], w["proj_out.b"], w["pe"][:32].reshape(-1)]).float()
    assert stream.shape[0] == 670 and params.numel() == 16320
    return stream.contiguous(), params.contiguous()


def _head_k_columns(D):
    """column order of a per-head [rows, 64] operand whose k dimension is the head dim d (PERM32 for k step 0; k step 1: d = 32 + 4 lg + e for
    e < 4, zero elsewhere) -- the order in which vv_motion.hip / vv_chain.hip pack three 16-row accumulator tiles of a head into two k steps.
    Returns (dst columns, src d indices)."""
    perm = _perm32()
    p = torch.arange(32)
    e, lg = p & 7, p >> 3
    d1 = 32 + 4 * lg + e
    ok = (e < 4) & (d1 < D)
    return torch.cat([torch.arange(32), 32 + p[ok]]), torch.cat([perm, d1[ok]])


# ---- column-split fused kernels (round 5): weights as one PRIVATE stream of MFMA A-operand fragments per wave, read straight into registers ----
def _frag16(w2d, r0, k0):
    """fragment (16 rows r0.., 32 k k0..) of a [N, K] fp32 matrix as the v_mfma_f32_16x16x32 A operand: [64 lanes][8] with lane l = row r0 + (l & 15),
    k = k0 + 8 (l >> 4) + j (zero outside the matrix).  1 KB in h16: one global_load_dwordx4 per wave."""
    N, K = w2d.shape
    out = torch.zeros((16, 32), dtype=torch.float32)
    r1, k1 = min(N, r0 + 16), min(K, k0 + 32)
    if r1 > r0 and k1 > k0:
        out[:r1 - r0, :k1 - k0] = w2d[r0:r1, k0:k1]
    return out.reshape(16, 4, 8).permute(1, 0, 2).reshape(64, 8)          # [lg][li][j] -> lane = 16 lg + li


def _frags(w2d, row_tiles, ksteps):
    """fragments in consumption order: k step major, row tile minor.  row_tiles: first rows of the 16-row tiles."""
    return [_frag16(w2d, r0, 32 * ks) for ks in range(ksteps) for r0 in row_tiles]


CS_RING = 10      # the kernels keep this many fragments in flight per wave; every layer's fragment count is a multiple of it


def _pad_ring(fr):
    return fr + [torch.zeros((64, 8))] * (-len(fr) % CS_RING)


def pack_chain_stream_columns(w, h16, heads=8):
    """The tail of a level-0 spatial transformer block for vv_chain.hip::chain_cs_c320_kernel (column-split: wave w of a block owns output channels
    80 w .. 80 w + 79 of every layer for all 128 tokens).  Returns (stream [4 * 870, 64, 8] h16: wave w's fragments at [870 w, 870 (w + 1)), params [5120]
    fp32 = bo1 | ln2 g, b | bo2 | ln3 g, b | b1 values (1280) | b1 gates (1280) | b2 | bout).  Per wave, in order: Wo1 (50 fragments); for its two heads
    (2 w, 2 w + 1): q projection padded to 48 rows (30), K_h (10: keys x packed d), V_h^T (9 + 1), the half of Wo2 that multiplies the four heads
    (s, 2 + s, 4 + s, 6 + s) written by the four waves in that phase (25 + 5); twenty 64-unit GEGLU chunks: W1 rows [value 16 | gate 16]
    of the wave's 16 units (20) and W2 (10); proj_out (50); 10 zero fragments (the ring reads ahead)."""
    C = w["o1.w"].shape[0]
    D = C // heads
    assert C == 320 and D == 40 and heads == 8 and w["k2"].shape == (77, C) and w["v2"].shape == (77, C)
    dst, src = _head_k_columns(D)
    inner = 4 * C
    streams = []
    for wv in range(4):
        own = [80 * wv + 16 * i for i in range(5)]
        fr = _frags(w["o1.w"], own, 10)
        for s_ in range(2):
            h = 2 * wv + s_
            wq = torch.zeros((48, C))
            wq[:D] = w["q2.w"][h * D:(h + 1) * D]
            fr += _frags(wq, [0, 16, 32], 10)
            kh = torch.zeros((80, 64))                                     # K_h: rows = keys (77 -> 80), columns = d in the packed (PERM32) order
            kh[:77, dst] = w["k2"][:, h * D + src]
            fr += _frags(kh, [0, 16, 32, 48, 64], 2)
            vt = torch.zeros((48, 96))                                     # V_h^T: rows = d, columns = keys, PERM32 inside every 32-key step
            vt[:D, :77] = w["v2"][:, h * D:(h + 1) * D].t()
            fr += _pad_ring(_frags(_permute_k(vt), [0, 16, 32], 3))
            cols = torch.cat([torch.arange((2 * q + s_) * D, (2 * q + s_ + 1) * D) for q in range(4)])      # O buffer channel order: wave q's head
            fr += _pad_ring(_frags(w["o2.w"][:, cols], own, 5))
        def w1(c):
            u0 = 64 * c + 16 * wv
            rows = torch.cat([torch.arange(u0, u0 + 16), inner + torch.arange(u0, u0 + 16)])
            return _frags(w["ff1.w"][rows], [0, 16], 10)
        fr += w1(0)                                  # software-pipelined consumption order: W1 (c + 1) before W2 (c)
        for c in range(inner // 64):
            if c + 1 < inner // 64:
                fr += w1(c + 1)
            fr += _frags(w["ff2.w"][:, 64 * c:64 * c + 64], own, 2)
        fr += _frags(w["out.w"], own, 10)
        fr += [torch.zeros((64, 8))] * CS_RING
        assert len(fr) == 870, len(fr)
        streams.append(torch.stack(fr))
    stream = torch.cat(streams).to(h16)
    params = torch.cat([w["o1.b"], w["ln2.g"], w["ln2.b"], w["o2.b"], w["ln3.g"], w["ln3.b"], w["ff1.b"], w["ff2.b"], w["out.b"]]).float()
    assert stream.shape[0] == 3480 and params.numel() == 5120
    return stream.contiguous(), params.contiguous()


CHAIN_LAYOUT = "rowsplit"      # stream layout of vv_chain.hip's product kernel (chain_rs_c320_kernel); "columns" / "tokens": the lab kernels (VV_CHAIN_FORM = 2 / 0)


def pack_chain_stream(w, h16, heads=8, layout=None):
    """w: dict of fp32 tensors of the tail of one spatial transformer block at C = 320 -- o1.w/.b (attn1.to_out.0), ln2.g/.b, q2.w (attn2.to_q),
    k2 / v2 ([77, C]: the text tokens already projected by attn2.to_k / to_v), o2.w/.b (attn2.to_out.0), ln3.g/.b, ff1.w/.b ([8C, C]),
    ff2.w/.b ([C, 4C]), out.w/.b (proj_out).  Returns (stream [462, 64, 64] h16, params [5120] fp32) in the consumption order of vv_chain.hip.
    layout "rowsplit" (product): cross-attention slabs per head PAIR as q K V^T | q K V^T | Wo | Wo, and the GEGLU rows of a 64-unit chunk ordered so
    that row tiles (0, 1) / (2, 3) of its two slab groups are [value | gate] of hidden units 0..15 / 32..47 and 16..31 / 48..63: the wave that owns
    row half hf of every slab then produces exactly k step hf of the second projection."""
    layout = layout or CHAIN_LAYOUT
    if layout == "columns":
        return pack_chain_stream_columns(w, h16, heads)
    assert layout in ("rowsplit", "tokens")
    C = w["o1.w"].shape[0]
    D = C // heads
    assert C == 320 and D == 40 and w["k2"].shape == (77, C) and w["v2"].shape == (77, C)
    dst, src = _head_k_columns(D)
    slabs = _dense_slabs(_permute_k(w["o1.w"]), h16, 64)
    core, outp = [], []
    for h in range(heads):
        wh = torch.zeros((48, C))
        wh[:D] = w["q2.w"][h * D:(h + 1) * D]
        hs = _dense_slabs(_permute_k(wh), h16, 48)                         # q = Wq[head] a: 5 slabs of 48 rows
        kh = torch.zeros((128, 64))                                        # K_h: rows = keys (77 -> 80 used), columns = d in the packed order
        kh[:77, dst] = w["k2"][:, h * D + src]
        hs += [_slab(kh[0:64], h16), _slab(kh[64:80], h16)]
        vt = torch.zeros((48, 128))                                        # V_h^T: rows = d, columns = keys (PERM32 inside every 32-key step)
        vt[:D, :77] = w["v2"][:, h * D:(h + 1) * D].t()
        vt = _permute_k(vt)
        hs += [_slab(vt[:, 0:64], h16), _slab(vt[:, 64:128], h16)]
        wo = torch.zeros((C, 64))
        wo[:, dst] = w["o2.w"][:, h * D + src]
        core.append(hs)
        outp.append(_dense_slabs(wo, h16, 64))
    if layout == "rowsplit":
        for h in range(0, heads, 2):
            slabs += core[h] + core[h + 1] + outp[h] + outp[h + 1]
    else:
        for h in range(heads):
            slabs += core[h] + outp[h]
    inner = 4 * C
    b1 = []
    for c in range(inner // 64):
        rows = []
        for i in ((0, 2, 1, 3) if layout == "rowsplit" else (0, 1, 2, 3)):
            rows += list(range(64 * c + 16 * i, 64 * c + 16 * i + 16)) + list(range(inner + 64 * c + 16 * i, inner + 64 * c + 16 * i + 16))
        rows = torch.tensor(rows)
        slabs += _dense_slabs(_permute_k(w["ff1.w"][rows]), h16, 64)
        b1.append(w["ff1.b"][rows])
        slabs += _dense_slabs(_permute_k(w["ff2.w"][:, 64 * c:64 * c + 64]), h16, 64)
    slabs += _dense_slabs(_permute_k(w["out.w"]), h16, 64)
    stream = torch.stack(slabs)
    params = torch.cat([w["o1.b"], w["ln2.g"], w["ln2.b"], w["o2.b"], w["ln3.g"], w["ln3.b"], torch.cat(b1), w["ff2.b"], w["out.b"]]).float()
    assert stream.shape[0] == 462 and params.numel() == 5120
    return stream.contiguous(), params.contiguous()


def pack_chain_front_stream(w, h16):
    """w: dict of fp32 tensors of the FRONT of one spatial transformer block at C = 320 -- in.w [C, C] / in.b (proj_in), ln1.g / ln1.b, qkv.w [3C, C]
    (attn1 to_q | to_k | to_v stacked; the query rows already carry scale * log2 e).  Returns (stream [100, 64, 64] h16, params [960] fp32) in
    the consumption order of vv_chain.hip::chain_front_c320_kernel: proj_in (5 x 5 slabs), then 15 row blocks x 5 k tiles of the fused projection."""
    C = w["in.w"].shape[0]
    assert C == 320 and w["qkv.w"].shape == (3 * C, C)
    slabs = _dense_slabs(_permute_k(w["in.w"]), h16, 64) + _dense_slabs(_permute_k(w["qkv.w"]), h16, 64)
    stream = torch.stack(slabs)
    params = torch.cat([w["in.b"], w["ln1.g"], w["ln1.b"]]).float()
    assert stream.shape[0] == 100 and params.numel() == 960
    return stream.contiguous(), params.contiguous()
