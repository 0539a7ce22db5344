"""cv2-free frame I/O (reference tools.py:4-45): FFV1 v3 (csrc/vv_ffv1.c) in Matroska, lossless round trips, the
reference's nearest-resize-on-write and start/max-frame semantics, and structure checks of the byte stream.
PARITY UNPINNED against ffmpeg / cv2 (absent): these tests pin self-consistency and the RFC 9043 structure only."""
import os

import numpy as np
import pytest

from videovanish_amd import frameio as FIO


def _frames(T, H, W, seed, kind):
    rng = np.random.default_rng(seed)
    if kind == "noise":
        return [rng.integers(0, 256, (H, W, 3), dtype=np.uint8) for _ in range(T)]
    if kind == "flat":            # long zero runs: exercises the run mode of the Golomb-Rice coder
        out = []
        for t in range(T):
            f = np.full((H, W, 3), 17 * t % 256, np.uint8)
            f[H // 4: H // 2, W // 3: W // 2] = (255, 0, 128)
            out.append(f)
        return out
    ys, xs = np.mgrid[0:H, 0:W]   # smooth gradients + a little noise: small residuals, adaptive k
    out = []
    for t in range(T):
        base = np.stack([(xs * 3 + t * 5) % 256, (ys * 2 + xs) % 256, (xs + ys + 7 * t) % 256], -1)
        out.append(np.clip(base + rng.integers(-2, 3, (H, W, 3)), 0, 255).astype(np.uint8))
    return out


@pytest.mark.parametrize("kind", ["noise", "flat", "smooth"])
@pytest.mark.parametrize("H,W,nv", [(16, 24, 1), (37, 53, 1), (64, 40, 3), (1, 7, 1), (9, 1, 2)])
def test_ffv1_frame_round_trip_is_lossless(kind, H, W, nv):
    f = _frames(1, H, W, 5, kind)[0]
    cfg = FIO.ffv1_config_record(nv)
    pkt = FIO.ffv1_encode(f, nv)
    assert np.array_equal(FIO.ffv1_decode(cfg, pkt, W, H), f)
    if kind == "flat" and H * W > 200:
        assert len(pkt) < H * W * 3 // 4          # the run mode actually compresses flat areas
    if kind == "noise" and H * W > 200:
        assert len(pkt) < int(H * W * 3 * 1.35)   # incompressible input costs ~9 bits per 8-bit sample, not more


def test_ffv1_stream_structure():
    f = _frames(1, 32, 48, 7, "smooth")[0]
    cfg = FIO.ffv1_config_record(2)
    pkt = FIO.ffv1_encode(f, 2)

    def crc(b):          # CRC-32 / polynomial 0x04C11DB7, MSB first, init 0 (RFC 9043 4.9.3): zero over data || crc
        c = 0
        for x in b:
            c ^= x << 24
            for _ in range(8):
                c = ((c << 1) ^ 0x04C11DB7) & 0xFFFFFFFF if c & 0x80000000 else (c << 1) & 0xFFFFFFFF
        return c
    assert crc(cfg) == 0 and len(cfg) < 64
    # two slices, each ends in [slice_size:3][error_status:1][crc:4]; walking back from the end lands exactly on byte 0
    end, n = len(pkt), 0
    while end > 0:
        size = int.from_bytes(pkt[end - 8: end - 5], "big")
        assert pkt[end - 5] == 0 and crc(pkt[end - 8 - size: end]) == 0
        end -= size + 8
        n += 1
    assert end == 0 and n == 2
    bad = bytearray(pkt)
    bad[10] ^= 0x40
    with pytest.raises(RuntimeError, match="CRC"):
        FIO.ffv1_decode(cfg, bytes(bad), 48, 32)
    with pytest.raises(RuntimeError, match="CRC"):
        FIO.ffv1_decode(cfg[:-1] + bytes([cfg[-1] ^ 1]), pkt, 48, 32)


def test_mkv_round_trip_and_tools_api(tmp_path):
    T, H, W = 7, 36, 52
    frames = _frames(T, H, W, 11, "smooth")
    path = str(tmp_path / "clip.mkv")
    FIO.write_video_frames_to_path(path, frames, 24.0, H, W)
    raw = open(path, "rb").read()
    assert raw[:4] == b"\x1a\x45\xdf\xa3" and b"matroska" in raw[:64] and b"V_FFV1" in raw[:400]
    got, fps = FIO.load_video_frames_from_path(path)
    assert abs(fps - 24.0) < 1e-3 and len(got) == T and all(np.array_equal(a, b) for a, b in zip(got, frames))
    sub, _ = FIO.load_video_frames_from_path(path, 2, 3)          # reference tools.py:17-23: skip start_frame, stop at max_frames
    assert len(sub) == 3 and np.array_equal(sub[0], frames[2]) and np.array_equal(sub[2], frames[4])
    tail, _ = FIO.load_video_frames_from_path(path, 5, -1)
    assert len(tail) == 2
    with pytest.raises(AssertionError):
        FIO.load_video_frames_from_path(path, 99, -1)             # "No frames read"
    with pytest.raises(AssertionError):
        FIO.load_video_frames_from_path(str(tmp_path / "missing.mkv"))


def test_write_resizes_mismatched_frames_with_nearest(tmp_path):
    """reference tools.py:41-42 (the reference's early-return quirk leaves frames 1.. at the model size: SURVEY a6)."""
    big = _frames(1, 40, 64, 3, "noise")[0]
    small = _frames(1, 20, 32, 4, "noise")[0]
    path = str(tmp_path / "mixed.mkv")
    FIO.write_video_frames_to_path(path, [big, small], 30.0, 40, 64)
    got, _ = FIO.load_video_frames_from_path(path)
    assert np.array_equal(got[0], big)
    assert np.array_equal(got[1], FIO.resize_nearest(small, 64, 40)) and np.array_equal(got[1][::2, ::2], small)


def test_cli_end_to_end_with_own_frame_io(tmp_path, monkeypatch):
    """diffuerase.main() with NO `tools` module importable falls back to videovanish_amd.frameio (hot path stubbed: CPU test)."""
    import sys
    import diffuerase
    monkeypatch.setitem(sys.modules, "tools", None)               # import tools -> ImportError
    T, H, W = 4, 24, 32
    frames = _frames(T, H, W, 1, "smooth")
    masks = [np.zeros((H, W, 3), np.uint8) for _ in range(T)]
    color, mask = str(tmp_path / "c.mkv"), str(tmp_path / "m.mkv")
    FIO.write_video_frames_to_path(color, frames, 25.0, H, W)
    FIO.write_video_frames_to_path(mask, masks, 25.0, H, W)
    monkeypatch.setattr(diffuerase, "run_infill_on_frames", lambda fr, mk, **kw: [255 - f for f in fr])
    monkeypatch.setattr(sys, "argv", ["diffuerase.py", "--color_video", color, "--mask_video", mask])
    diffuerase.main()
    out, fps = FIO.load_video_frames_from_path(color + "_vanished.mkv")
    assert abs(fps - 25.0) < 1e-3 and len(out) == T and all(np.array_equal(o, 255 - f) for o, f in zip(out, frames))


def test_cli_frame_io_selection_without_stub(tmp_path, monkeypatch):
    """from the repo root `import tools` SUCCEEDS (the repo's tools/ directory of bench scripts is a namespace package without
    the two I/O functions), so the CLI must test for the API, not for the import.  No sys.modules stub here."""
    import importlib
    import os
    import sys
    import diffuerase
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    monkeypatch.chdir(root)
    monkeypatch.delitem(sys.modules, "tools", raising=False)
    if root not in sys.path:
        monkeypatch.syspath_prepend(root)
    t = importlib.import_module("tools")                          # the namespace package: importable, but no frame I/O in it
    assert not hasattr(t, "load_video_frames_from_path")
    assert diffuerase._frame_io() is FIO
    T, H, W = 3, 16, 24
    frames = _frames(T, H, W, 2, "smooth")
    masks = [np.zeros((H, W, 3), np.uint8) for _ in range(T)]
    color, mask = str(tmp_path / "c.mkv"), str(tmp_path / "m.mkv")
    FIO.write_video_frames_to_path(color, frames, 24.0, H, W)
    FIO.write_video_frames_to_path(mask, masks, 24.0, H, W)
    monkeypatch.setattr(diffuerase, "run_infill_on_frames", lambda fr, mk, **kw: [f[::-1].copy() for f in fr])
    monkeypatch.setattr(sys, "argv", ["diffuerase.py", "--color_video", color, "--mask_video", mask])
    diffuerase.main()
    out, _ = FIO.load_video_frames_from_path(color + "_vanished.mkv")
    assert len(out) == T and all(np.array_equal(o, f[::-1]) for o, f in zip(out, frames))


# ---- independent fixtures: packets written by a SECOND encoder (pure Python, from RFC 9043: tests/ffv1_pyenc.py) in the forms
#      libavcodec emits by default and the C encoder cannot produce: range-coded samples, a 2 x 2 slice grid, two quantisation-table
#      sets (one with 5 context inputs), the alternative state-transition table (coder_type 2), an extra (alpha) plane, no CRC
@pytest.mark.parametrize("coder,nh,nv,alpha,ec", [(1, 2, 2, False, 1), (2, 1, 3, False, 0), (1, 3, 1, True, 1), (2, 2, 2, True, 1)])
def test_ffv1_decodes_range_coded_packets_of_an_independent_encoder(coder, nh, nv, alpha, ec):
    from tests import ffv1_pyenc as PY
    H, W = 13, 18
    rng = np.random.default_rng(coder * 10 + nh)
    f = _frames(1, H, W, 3 + coder, "smooth")[0]
    f[2:5, 3:9] = (250, 3, 128)                                   # a flat patch and a few outliers: large and zero residuals
    f[7, 11] = (0, 255, 0)
    a = rng.integers(0, 256, (H, W), dtype=np.uint8) if alpha else None
    cfg = PY.config_record(coder, nh, nv, alpha=alpha, ec=ec)
    pkt = PY.encode_frame(f, coder, nh, nv, alpha=a, ec=ec)
    assert np.array_equal(FIO.ffv1_decode(cfg, pkt, W, H), f)
    # and the noise image (every context, long symbols)
    g = _frames(1, H, W, 9, "noise")[0]
    assert np.array_equal(FIO.ffv1_decode(cfg, PY.encode_frame(g, coder, nh, nv, alpha=a, ec=ec), W, H), g)


def test_ffv1_default_state_transition_is_rfc9043_table():
    """RFC 9043 3.8.1.5 default_state_transition, first and last rows of the published listing, against the Python construction the
    fixture encoder uses (the C decoder builds the same table: a packet coded with it decodes, test above)."""
    from tests import ffv1_pyenc as PY
    t = PY.default_state_transition()
    assert t[:16] == [0, 0, 0, 0, 0, 0, 0, 0, 20, 21, 22, 23, 24, 25, 26, 27]
    assert t[16:32] == [28, 29, 30, 31, 32, 33, 34, 35, 36, 37, 37, 38, 39, 40, 41, 42]
    assert t[240:] == [241, 242, 243, 244, 245, 246, 247, 248, 248, 0, 0, 0, 0, 0, 0, 0]
    assert all(t[i] > i for i in range(8, 248)) and t[248] == 248


def test_ffv1_decoder_rejects_crafted_headers():
    """header symbols come from a user-supplied file.  Out-of-range slice positions, table-set indices, slice grids and
    over-long symbols must be refused, not used as array indices."""
    from tests import ffv1_pyenc as PY
    H, W = 8, 8
    f = _frames(1, H, W, 1, "smooth")[0]

    def packet(sx, sy, sw1, sh1, q0, q1):
        rc = PY.RangeEncoder()
        rc.put([128], 0, 1)
        st = [128] * 32
        for v in (sx, sy, sw1, sh1, q0, q1, 3, 0, 0):
            rc.put_symbol(st, v, False)
        body = rc.terminate() + bytes(64)
        sl = bytearray(body) + len(body).to_bytes(3, "big") + b"\x00"
        return bytes(sl + PY.crc32_mpeg(bytes(sl)).to_bytes(4, "big"))
    cfg = PY.config_record(1, 2, 2)
    for bad in [(2, 0, 0, 0, 0, 1), (0, 2, 0, 0, 0, 1), (0, 0, 2, 0, 0, 1), (0, 0, 0, 2, 0, 1), (0, 0, 0, 0, 2, 1), (0, 0, 0, 0, 0, 5),
                (1, 1, 1, 0, 0, 1), (0, 0, (1 << 31) - 2, 0, 0, 1), (0, 0, 0, 0, (1 << 32) - 1, 0)]:
        with pytest.raises(RuntimeError):
            FIO.ffv1_decode(cfg, packet(*bad), W, H)
    good = PY.encode_frame(f, 1, 2, 2)
    assert np.array_equal(FIO.ffv1_decode(cfg, good, W, H), f)
    with pytest.raises(RuntimeError):                             # a slice grid of 2^20 x 1 in the configuration record
        FIO.ffv1_decode(PY.config_record(1, 1 << 20, 1), good, W, H)


# ---- round 3: Golomb-Rice packets of the independent encoder, planar YCbCr streams, colour conversion, YUV4MPEG2 --------------------------
@pytest.mark.parametrize("nh,nv,alpha", [(1, 1, False), (2, 2, False), (1, 3, True)])
def test_ffv1_decodes_golomb_rice_packets_of_an_independent_encoder(nh, nv, alpha):
    """coder_type 0 (what libavcodec picks for 8-bit video and what the C encoder writes) from the SECOND encoder: run mode, adaptive Rice
    parameter, escape codes -- the C decoder is no longer checked against its own encoder only."""
    from tests import ffv1_pyenc as PY
    H, W = 14, 19
    rng = np.random.default_rng(nh + 3 * nv)
    a = rng.integers(0, 256, (H, W), dtype=np.uint8) if alpha else None
    cfg = PY.config_record(0, nh, nv, alpha=alpha)
    for kind in ("smooth", "flat", "noise"):
        f = _frames(1, H, W, 4, kind)[0]
        f[7, 11] = (0, 255, 0)                                       # an outlier: escape-coded residuals
        assert np.array_equal(FIO.ffv1_decode(cfg, PY.encode_frame(f, 0, nh, nv, alpha=a), W, H), f), kind


def _ycbcr_planes(H, W, hs, vs, seed, kind):
    rng = np.random.default_rng(seed)
    ch, cw = (H + (1 << vs) - 1) >> vs, (W + (1 << hs) - 1) >> hs
    if kind == "noise":
        return tuple(rng.integers(0, 256, s, dtype=np.uint8) for s in ((H, W), (ch, cw), (ch, cw)))
    ys, xs = np.mgrid[0:H, 0:W]
    y = np.clip(16 + (xs * 5 + ys * 3) % 220 + rng.integers(-1, 2, (H, W)), 0, 255).astype(np.uint8)
    cys, cxs = np.mgrid[0:ch, 0:cw]
    cb = np.clip(128 + 40 * np.sin(cxs / 3.0) + rng.integers(-1, 2, (ch, cw)), 0, 255).astype(np.uint8)
    cr = np.full((ch, cw), 90, np.uint8)                              # flat plane: the run mode across whole lines
    return y, cb, cr


@pytest.mark.parametrize("coder", [0, 1, 2])
@pytest.mark.parametrize("hs,vs,nh,nv,alpha", [(1, 1, 1, 1, False), (1, 1, 2, 2, False), (1, 0, 1, 2, False), (0, 0, 2, 1, True), (2, 0, 1, 1, False)])
def test_ffv1_decodes_planar_ycbcr_streams(coder, hs, vs, nh, nv, alpha):
    """colorspace_type 0 (yuv420p / 422p / 444p / 411p): planes, shared chroma states, subsampled slice grid -- packets of the independent
    encoder decode bit-exactly; odd sizes exercise the rounded-up chroma planes."""
    from tests import ffv1_pyenc as PY
    H, W = 13, 22 if (nh > 1 or nv > 1) else 21
    for kind in ("smooth", "noise"):
        y, cb, cr = _ycbcr_planes(H, W, hs, vs, 7 + coder, kind)
        a = np.random.default_rng(1).integers(0, 256, (H, W), dtype=np.uint8) if alpha else None
        cfg = PY.config_record(coder, nh, nv, alpha=alpha, colorspace=0, hshift=hs, vshift=vs)
        info = FIO.ffv1_stream_info(cfg)
        assert (info["colorspace"], info["hshift"], info["vshift"], info["alpha"], info["bits"]) == (0, hs, vs, int(alpha), 8)
        pkt = PY.encode_frame_ycbcr(y, cb, cr, coder, nh, nv, hs, vs, alpha=a)
        gy, gcb, gcr = FIO.ffv1_decode_planes(cfg, pkt, W, H)
        assert np.array_equal(gy, y) and np.array_equal(gcb, cb) and np.array_equal(gcr, cr), (kind, coder)
    # a gray stream (no chroma planes): chroma comes back neutral
    cfg = PY.config_record(coder, 1, 1, colorspace=0, chroma_planes=False)
    gy, gcb, gcr = FIO.ffv1_decode_planes(cfg, PY.encode_frame_ycbcr(y, None, None, coder, 1, 1, 0, 0), W, H)
    assert np.array_equal(gy, y) and (gcb == 128).all() and (gcr == 128).all()
    # the RGB entry point refuses a YCbCr stream instead of mis-decoding it (and ffv1_decode converts)
    assert FIO.ffv1_decode(cfg, PY.encode_frame_ycbcr(y, None, None, coder, 1, 1, 0, 0), W, H).shape == (H, W, 3)


def _ycbcr_to_rgb_ref(y, cb, cr, hs, vs, full):
    """numpy restatement of the conversion contract (include/vvio.h): bilinear chroma in 1/16 units, BT.601, 16.16 fixed point."""
    H, W = y.shape
    CH, CW = cb.shape
    X, Y = np.meshgrid(np.arange(W), np.arange(H))

    def up(c):
        c = c.astype(np.int64)
        x0 = X >> hs
        x1 = np.where((hs == 1) & (X & 1 == 1), np.minimum(x0 + 1, CW - 1), x0)
        wx1 = np.where((hs == 1) & (X & 1 == 1), 2, 0)
        wx0 = 4 - wx1
        y0 = Y >> vs
        if vs == 1:
            y1 = np.where(Y & 1 == 1, np.minimum(y0 + 1, CH - 1), np.maximum(y0 - 1, 0))
            wy0, wy1 = 3, 1
        else:
            y1, wy0, wy1 = y0, 4, 0
        return wy0 * (wx0 * c[y0, x0] + wx1 * c[y0, x1]) + wy1 * (wx0 * c[y1, x0] + wx1 * c[y1, x1])
    u, v = up(cb) - 2048, up(cr) - 2048
    ky, yoff, krv, kgu, kgv, kbu = (65536, 0, 91881, 22554, 46802, 116130) if full else (76309, 16, 104597, 25675, 53279, 132201)
    yy = 16 * ky * (y.astype(np.int64) - yoff)
    r, g, b = (yy + krv * v + (1 << 19)) >> 20, (yy - kgu * u - kgv * v + (1 << 19)) >> 20, (yy + kbu * u + (1 << 19)) >> 20
    return np.clip(np.stack([r, g, b], -1), 0, 255).astype(np.uint8)


@pytest.mark.parametrize("hs,vs", [(1, 1), (1, 0), (0, 0), (2, 0)])
@pytest.mark.parametrize("full", [False, True])
def test_ycbcr_to_rgb_host_conversion(hs, vs, full):
    H, W = 17, 23
    y, cb, cr = _ycbcr_planes(H, W, hs, vs, 3, "noise")
    got = FIO.ycbcr_to_rgb(y, cb, cr, hs, vs, full, device=False)
    assert np.array_equal(got, _ycbcr_to_rgb_ref(y, cb, cr, hs, vs, full))
    # against the BT.601 definition in floating point with the same chroma interpolation: within one level
    ch = lambda c: np.kron(c, np.ones((1 << vs, 1 << hs)))[:H, :W] if (hs, vs) == (0, 0) else None
    if (hs, vs) == (0, 0):
        Y_, U_, V_ = y.astype(np.float64), cb.astype(np.float64) - 128, cr.astype(np.float64) - 128
        if full:
            ref = np.stack([Y_ + 1.402 * V_, Y_ - 0.344136 * U_ - 0.714136 * V_, Y_ + 1.772 * U_], -1)
        else:
            ref = np.stack([1.164383 * (Y_ - 16) + 1.596027 * V_, 1.164383 * (Y_ - 16) - 0.391762 * U_ - 0.812968 * V_, 1.164383 * (Y_ - 16) + 2.017232 * U_], -1)
        assert np.abs(got.astype(np.float64) - np.clip(np.rint(ref), 0, 255)).max() <= 1


def test_ycbcr_known_colours():
    one = lambda Y, U, V, full=False: tuple(int(v) for v in FIO.ycbcr_to_rgb(np.full((2, 2), Y, np.uint8), np.full((1, 1), U, np.uint8), np.full((1, 1), V, np.uint8), 1, 1, full, device=False)[0, 0])
    assert one(16, 128, 128) == (0, 0, 0) and one(235, 128, 128) == (255, 255, 255) and one(126, 128, 128) == (128, 128, 128)
    near = lambda a, b: max(abs(x - y) for x, y in zip(a, b)) <= 1             # the 8-bit YCbCr codes of the primaries are themselves rounded
    assert near(one(81, 90, 240), (255, 0, 0)) and near(one(145, 54, 34), (0, 255, 0)) and near(one(41, 240, 110), (0, 0, 255))      # BT.601, limited range
    assert one(0, 128, 128, True) == (0, 0, 0) and one(255, 128, 128, True) == (255, 255, 255) and near(one(76, 85, 255, True), (255, 0, 0))


def test_y4m_and_ycbcr_mkv_loading(tmp_path):
    """`.y4m` (what `ffmpeg -i clip.mp4 clip.y4m` writes) and a yuv420p FFV1 track in Matroska through the reference's loader API."""
    from tests import ffv1_pyenc as PY
    H, W, T = 12, 20, 3
    planes = [_ycbcr_planes(H, W, 1, 1, 20 + t, "smooth") for t in range(T)]
    want = [_ycbcr_to_rgb_ref(y, cb, cr, 1, 1, False) for y, cb, cr in planes]
    p = str(tmp_path / "clip.y4m")
    with open(p, "wb") as f:
        f.write(b"YUV4MPEG2 W%d H%d F30000:1001 Ip A1:1 C420mpeg2\n" % (W, H))
        for y, cb, cr in planes:
            f.write(b"FRAME\n" + y.tobytes() + cb.tobytes() + cr.tobytes())
    frames, fps = FIO.load_video_frames_from_path(p)
    assert abs(fps - 30000 / 1001) < 1e-9 and len(frames) == T and all(np.array_equal(a, b) for a, b in zip(frames, want))
    part, _ = FIO.load_video_frames_from_path(p, start_frame=1, max_frames=1)
    assert len(part) == 1 and np.array_equal(part[0], want[1])
    with pytest.raises(RuntimeError, match="not supported"):
        open(str(tmp_path / "x.y4m"), "wb").write(b"YUV4MPEG2 W4 H4 F25:1 C420p10\nFRAME\n" + bytes(100))
        FIO.load_video_frames_from_path(str(tmp_path / "x.y4m"))
    mkv = str(tmp_path / "clip420.mkv")
    cfg = PY.config_record(0, 1, 2, colorspace=0, hshift=1, vshift=1)
    FIO.write_mkv_packets(mkv, W, H, cfg, [PY.encode_frame_ycbcr(y, cb, cr, 0, 1, 2, 1, 1) for y, cb, cr in planes], 25.0)
    frames, fps = FIO.load_video_frames_from_path(mkv)
    assert abs(fps - 25.0) < 1e-6 and len(frames) == T and all(np.array_equal(a, b) for a, b in zip(frames, want))


# ---- round 4: non-key frames (gop_size > 1) and FFV1 version 0 / 1 streams, against the independent Python encoder ----------------------------
def _clip_rgb(T, H, W, seed):
    """consecutive frames that share most content (what makes carried-over states differ from fresh ones)"""
    base = _frames(1, H, W, seed, "smooth")[0].astype(np.int32)
    rng = np.random.default_rng(seed)
    out = []
    for t in range(T):
        f = np.roll(base, t, axis=1) + rng.integers(-2, 3, base.shape)
        f[2:4, 1 + t:5 + t] = (255, 0, 17 * t)
        out.append(np.clip(f, 0, 255).astype(np.uint8))
    return out


@pytest.mark.parametrize("coder", [0, 1, 2])
def test_ffv1_non_key_frames_continue_from_the_previous_frame(coder):
    """version 3 with intra = 0 (an encoder running with gop_size > 1, e.g. cv2.VideoWriter's default): a frame whose key-frame bit is 0 has no
    reset -- every slice continues from the adaptive context states its predecessor left.  The stateful decoder reproduces a key / non / non /
    key / non sequence bit for bit (RGB 2 x 2 slices and yuv420p 1 x 2 slices); the stateless entry point and a decoder that has not seen a
    key frame refuse a non-key packet (-20) instead of decoding it from fresh states."""
    from tests import ffv1_pyenc as PY
    H, W, T = 12, 18, 5
    keys = [True, False, False, True, False]
    frames = _clip_rgb(T, H, W, 30 + coder)
    cfg = PY.config_record(coder, 2, 2, intra=0)
    st = {}
    pkts = [PY.encode_frame(f, coder, 2, 2, keyframe=k, persist=st) for f, k in zip(frames, keys)]
    fresh = PY.encode_frame(frames[1], coder, 2, 2, keyframe=True)
    assert pkts[1] != fresh[:len(pkts[1])]                                     # the carried-over states really change the bits
    dec = FIO.Ffv1Decoder(cfg, W, H)
    assert dec.info()["version"] == 3
    assert all(np.array_equal(dec.decode(p), f) for p, f in zip(pkts, frames))
    dec.close()
    with pytest.raises(RuntimeError, match="non-key frame"):
        FIO.Ffv1Decoder(cfg, W, H).decode(pkts[1])
    with pytest.raises(RuntimeError, match="non-key frame"):
        FIO.ffv1_decode(cfg, pkts[1], W, H)
    # planar YCbCr, Golomb-Rice / range coder alike
    planes = [_ycbcr_planes(H, W, 1, 1, 40 + t, "smooth") for t in range(T)]
    cfg = PY.config_record(coder, 1, 2, colorspace=0, hshift=1, vshift=1, intra=0)
    st = {}
    pkts = [PY.encode_frame_ycbcr(y, cb, cr, coder, 1, 2, 1, 1, keyframe=k, persist=st) for (y, cb, cr), k in zip(planes, keys)]
    dec = FIO.Ffv1Decoder(cfg, W, H)
    want = [_ycbcr_to_rgb_ref(y, cb, cr, 1, 1, False) for y, cb, cr in planes]
    assert all(np.array_equal(dec.decode(p), w) for p, w in zip(pkts, want))


def test_mkv_seek_decodes_from_the_last_key_frame_only(tmp_path, monkeypatch):
    """read_mkv(start_frame): packets before start_frame are decoded only from the last key frame at or before it -- an intra-only
    stream (this module's writer) decodes exactly the frames it returns, a key / non / non / key / non stream decodes from the key frame of the
    first wanted frame's GOP; the key-frame bit read from the packet agrees with the encoder's flag."""
    from tests import ffv1_pyenc as PY
    H, W, T = 12, 18, 6
    frames = _clip_rgb(T, H, W, 77)
    calls = []
    real = FIO.Ffv1Decoder.decode
    monkeypatch.setattr(FIO.Ffv1Decoder, "decode", lambda self, pkt: (calls.append(len(pkt)), real(self, pkt))[1])
    intra = str(tmp_path / "intra.mkv")
    FIO.write_video_frames_to_path(intra, frames, 25.0, H, W)
    part, _ = FIO.load_video_frames_from_path(intra, start_frame=4, max_frames=2)
    assert len(part) == 2 and np.array_equal(part[0], frames[4]) and np.array_equal(part[1], frames[5]) and len(calls) == 2
    keys = [True, False, False, True, False, False]
    st = {}
    pkts = [PY.encode_frame(f, 1, 1, 1, keyframe=k, persist=st) for f, k in zip(frames, keys)]
    assert [FIO._ffv1_is_key_packet(p) for p in pkts] == keys
    gop = str(tmp_path / "gop.mkv")
    FIO.write_mkv_packets(gop, W, H, PY.config_record(1, 1, 1, intra=0), pkts, 25.0, key_frames=keys)
    for start, n_dec in ((0, 2), (2, 4), (3, 2), (4, 3), (5, 3)):
        del calls[:]
        part, _ = FIO.load_video_frames_from_path(gop, start_frame=start, max_frames=2)
        want = frames[start:start + 2]
        assert len(part) == len(want) and all(np.array_equal(a, b) for a, b in zip(part, want)), start
        assert len(calls) == n_dec, (start, len(calls))


@pytest.mark.parametrize("version", [0, 1])
@pytest.mark.parametrize("coder", [0, 1, 2])
def test_ffv1_version_0_and_1_streams(version, coder, tmp_path):
    """FFV1 version 0 / 1 (what libavcodec's encoder picks for frames up to 720 x 576 [UNVERIFIED-3P]): no configuration record, the parameters and
    ONE quantisation-table set (3 or 5 context inputs) in the range-coded header of every key frame, one slice, no slice header / footer / CRC;
    key and non-key frames; RGB, yuv420p, yuv444p + alpha, gray; through the decoder object and through the Matroska loader (no CodecPrivate)."""
    from tests import ffv1_pyenc as PY
    H, W, T = 11, 17, 4
    keys = [True, False, True, False]
    frames = _clip_rgb(T, H, W, 50 + coder)
    for qset in (0, 1):
        st = {}
        pkts = [PY.encode_frame(f, coder, 1, 1, set_luma=qset, set_chroma=qset, set_alpha=qset, keyframe=k, persist=st, legacy=version) for f, k in zip(frames, keys)]
        dec = FIO.Ffv1Decoder(b"", W, H)
        with pytest.raises(RuntimeError, match="before the first key frame"):
            dec.info()
        got = [dec.decode(p) for p in pkts]
        assert all(np.array_equal(g, f) for g, f in zip(got, frames)), (version, coder, qset)
        info = dec.info()
        assert (info["version"], info["colorspace"], info["bits"]) == (version, 1, 8)
    with pytest.raises(RuntimeError, match="non-key frame"):
        FIO.Ffv1Decoder(b"", W, H).decode(pkts[1])
    # through the container: no CodecPrivate element at all; start_frame inside a GOP still decodes from the key frame on
    mkv = str(tmp_path / "v01.mkv")
    FIO.write_mkv_packets(mkv, W, H, b"", pkts, 25.0, key_frames=keys)
    got, fps = FIO.load_video_frames_from_path(mkv)
    assert len(got) == T and all(np.array_equal(g, f) for g, f in zip(got, frames))
    part, _ = FIO.load_video_frames_from_path(mkv, start_frame=1, max_frames=2)
    assert len(part) == 2 and np.array_equal(part[0], frames[1]) and np.array_equal(part[1], frames[2])
    # planar YCbCr variants
    for hs, vs, alpha, chroma in ((1, 1, False, True), (0, 0, True, True), (0, 0, False, False)):
        planes = [_ycbcr_planes(H, W, hs, vs, 60 + t, "smooth") for t in range(T)]
        a = np.random.default_rng(2).integers(0, 256, (H, W), dtype=np.uint8) if alpha else None
        st = {}
        pkts = [PY.encode_frame_ycbcr(y, cb if chroma else None, cr if chroma else None, coder, 1, 1, hs, vs, alpha=a, set_luma=1, set_chroma=1, set_alpha=1,
                                      keyframe=k, persist=st, legacy=version) for (y, cb, cr), k in zip(planes, keys)]
        dec = FIO.Ffv1Decoder(b"", W, H)
        for p, (y, cb, cr) in zip(pkts, planes):
            want = _ycbcr_to_rgb_ref(y, cb if chroma else np.full_like(cb, 128), cr if chroma else np.full_like(cr, 128), hs, vs, False)
            assert np.array_equal(dec.decode(p), want), (version, coder, hs, vs, alpha, chroma)
        assert dec.info()["colorspace"] == 0 and dec.info()["alpha"] == int(alpha)


def test_ffv1_unsupported_versions_are_named():
    """a version 4 (or unknown micro_version) configuration record is refused with its own error instead of being decoded with version 3 rules
    (ADVICE r3); version 2 -- experimental, never released -- and a version-0/1 header claiming version 2+ likewise."""
    from tests import ffv1_pyenc as PY
    for ver, micro, msg in ((4, 0, "version > 3"), (3, 7, "version > 3"), (2, 0, "version")):
        cfg = PY.config_record(1, 1, 1, version=ver, micro=micro)
        with pytest.raises(RuntimeError, match=msg):
            FIO.ffv1_stream_info(cfg)
        with pytest.raises(RuntimeError, match=msg):
            FIO.Ffv1Decoder(cfg, 8, 8)
    pkt = PY.encode_frame(np.zeros((8, 8, 3), np.uint8), 1, 1, 1, set_chroma=0, legacy=3)
    with pytest.raises(RuntimeError, match="version"):
        FIO.Ffv1Decoder(b"", 8, 8).decode(pkt)
