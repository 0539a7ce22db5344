"""Algorithmic FLOP model of the hot path (2*MAC) -- what bench.py's roofline figures divide by."""
import math

from .config import UNetConfig, VAEConfig


def _conv3(ci, co, n): return 18.0 * ci * co * n
def _lin(ci, co, n): return 2.0 * ci * co * n
def _res(ci, co, n): return _conv3(ci, co, n) + _conv3(co, co, n) + (_lin(ci, co, n) if ci != co else 0.0)


def _spatial_tf(C, N, cfg):
    lin = 2.0 * C * C * N * (2 + 4 + 2) + _lin(C, 8 * C, N) + _lin(4 * C, C, N) + 2 * _lin(cfg.cross_dim, C, cfg.text_len)
    return lin + 4.0 * N * N * C + 4.0 * N * cfg.text_len * C


def _motion(C, N, F): return 2.0 * C * C * N * (2 + 8) + 24.0 * C * C * N + 2 * (4.0 * F * C * N)


def level_sizes(h, w, L):
    out = [(h, w)]
    for _ in range(L - 1):
        h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        out.append((h, w))
    return out


def denoise_step_per_frame(h, w, F, cfg: UNetConfig = None):
    """FLOP of one BrushNet + motion-UNet forward per frame at latent size h x w, clip length F."""
    cfg = cfg or UNetConfig()
    bo, L, lpb = cfg.block_out, len(cfg.block_out), cfg.layers_per_block
    sz = level_sizes(h, w, L)

    def backbone(motion, cin0):
        fl = _conv3(cin0, bo[0], h * w)
        cin, skip = bo[0], [bo[0]]
        for i, co in enumerate(bo):
            n = sz[i][0] * sz[i][1]
            for _ in range(lpb):
                fl += _res(cin, co, n)
                if cfg.attn_levels[i]: fl += _spatial_tf(co, n, cfg)
                if motion: fl += _motion(co, n, F)
                cin = co; skip.append(co)
            if i < L - 1:
                n2 = sz[i + 1][0] * sz[i + 1][1]
                fl += _conv3(co, co, n2); skip.append(co)
        C, n = bo[-1], sz[-1][0] * sz[-1][1]
        fl += 2 * _res(C, C, n) + _spatial_tf(C, n, cfg) + (_motion(C, n, F) if motion else 0.0)
        x = C
        for i, co in enumerate(reversed(bo)):
            lvl = L - 1 - i
            n = sz[lvl][0] * sz[lvl][1]
            for _ in range(lpb + 1):
                fl += _res(x + skip.pop(), co, n)
                if cfg.attn_levels[lvl]: fl += _spatial_tf(co, n, cfg)
                if motion: fl += _motion(co, n, F)
                x = co
            if i < L - 1:
                n2 = sz[lvl - 1][0] * sz[lvl - 1][1]
                fl += _conv3(co, co, n2)
        return fl

    unet = backbone(True, cfg.in_ch) + _conv3(bo[0], cfg.out_ch, h * w)
    brush = backbone(False, cfg.brush_in_ch)
    # zero convs (1x1) of BrushNet
    return unet + brush


def vae_per_frame(H, W, cfg: VAEConfig = None):
    """(encode FLOP, decode FLOP) per frame at image size H x W."""
    cfg = cfg or VAEConfig()
    bo, lpb = cfg.block_out, cfg.layers_per_block
    n, enc, cin = H * W, _conv3(3, bo[0], H * W), bo[0]
    for i, co in enumerate(bo):
        for _ in range(lpb):
            enc += _res(cin, co, n); cin = co
        if i < len(bo) - 1:
            n //= 4
            enc += _conv3(co, co, n)
    C = bo[-1]
    mid = 2 * _res(C, C, n) + 4 * _lin(C, C, n) + 4.0 * n * n * C
    enc += mid + _conv3(C, 2 * cfg.latent_ch, n)
    dec = _conv3(cfg.latent_ch, C, n) + mid
    cin = C
    for i, co in enumerate(reversed(bo)):
        for _ in range(lpb + 1):
            dec += _res(cin, co, n); cin = co
        if i < len(bo) - 1:
            n *= 4
            dec += _conv3(co, co, n)
    dec += _conv3(bo[0], 3, n)
    return enc, dec


def per_output_frame(H, W, F, steps, ucfg=None, vcfg=None):
    f = 2 ** (len((vcfg or VAEConfig()).block_out) - 1)
    enc, dec = vae_per_frame(H, W, vcfg)
    return steps * denoise_step_per_frame(H // f, W // f, F, ucfg) + 2 * enc + dec


def temporal_block_per_frame(h, w, F, cfg: UNetConfig = None):
    """FLOP per frame per denoise step of the 21 motion modules (GroupNorm-free count: proj in/out, 2 x (QKV + core + out),
    GEGLU FF) -- the 'fused temporal block' the north star prices against the MFMA roofline."""
    cfg = cfg or UNetConfig()
    bo, L, lpb = cfg.block_out, len(cfg.block_out), cfg.layers_per_block
    sz = level_sizes(h, w, L)
    fl = 0.0
    for i, co in enumerate(bo):
        fl += lpb * _motion(co, sz[i][0] * sz[i][1], F)
    fl += _motion(bo[-1], sz[-1][0] * sz[-1][1], F)
    for i, co in enumerate(reversed(bo)):
        lvl = L - 1 - i
        fl += (lpb + 1) * _motion(co, sz[lvl][0] * sz[lvl][1], F)
    return fl
