"""cv2-free frame I/O with the API of the reference's `tools` module (reference tools.py:4-45):

    load_video_frames_from_path(video_path, start_frame=0, max_frames=-1) -> (list of (H, W, 3) uint8 RGB frames, fps)
    write_video_frames_to_path(out_video, frames, fps, H0, W0)            -> lossless FFV1 in Matroska (.mkv)

The reference goes through cv2 (ffmpeg): `VideoWriter_fourcc(*"FFV1")` into an .mkv (tools.py:32-37), RGB<->BGR swaps
(:21,:40) and a NEAREST resize of frames whose size differs from (W0, H0) (:41-42).  Here the container is written / parsed in
Python (EBML is a few dozen lines) and the codec is the plain-C FFV1 v3 intra codec in csrc/vv_ffv1.c (libvvio.so; Golomb-Rice
coded 8-bit RGB, what ffmpeg emits for bgr0).  Frames are RGB in, RGB out: no BGR detour.

Reading supports FFV1 version 3 in Matroska (V_FFV1 or V_MS/VFW/FOURCC 'FFV1'): 8-bit RGB (what this module and cv2 write) AND planar
8-bit YCbCr (yuv420p / 422p / 444p / gray, what ffmpeg writes by default), Golomb-Rice or range coded, with or without alpha; uncompressed
RGB24/BGR24 tracks (V_UNCOMPRESSED); YUV4MPEG2 (`.y4m`); `.npy` / `.npz` frame stacks.  YCbCr is converted to RGB on the GPU when one is visible
(vv_ycbcr_to_rgb, BT.601; the same integer arithmetic runs on the host otherwise).  Other codecs (H.264, ...) raise a clear error --
decoding them is out of scope (codec I/O is either side of the hot path).
PARITY UNPINNED against ffmpeg / cv2 (absent from the build image): pinned by lossless round trips and structure checks only.
"""
import ctypes as C
import os
import struct

import numpy as np

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libvvio.so")
_lib = None


def _io():
    global _lib
    if _lib is None:
        if not os.path.isfile(_LIB_PATH):
            raise RuntimeError(f"videovanish_amd: frame I/O codec missing ({_LIB_PATH}); run videovanish_amd/csrc/build.sh")
        L = C.CDLL(_LIB_PATH)
        for name in ("vvio_abi_version", "vvio_ffv1_config_record", "vvio_ffv1_encode_frame", "vvio_ffv1_decode_frame", "vvio_ffv1_stream_info",
                     "vvio_ffv1_decode_frame_yuv", "vvio_ycbcr_to_rgb", "vvio_ffv1_decoder_open", "vvio_ffv1_decoder_info", "vvio_ffv1_decoder_decode",
                     "vvio_ffv1_decoder_close"):
            if not hasattr(L, name):
                raise RuntimeError(f"libvvio.so does not export {name}")
        L.vvio_ffv1_decoder_open.restype = C.c_void_p
        L.vvio_ffv1_decoder_open.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.vvio_ffv1_decoder_info.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.vvio_ffv1_decoder_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.vvio_ffv1_decoder_close.argtypes = [C.c_void_p]
        L.vvio_ffv1_decoder_close.restype = None
        _lib = L
    return _lib


# ---- FFV1 ----------------------------------------------------------------------------------------------------------------
def ffv1_config_record(num_v_slices=1):
    buf = (C.c_uint8 * 4096)()
    n = _io().vvio_ffv1_config_record(int(num_v_slices), buf, 4096)
    if n < 0:
        raise RuntimeError("vvio_ffv1_config_record failed")
    return bytes(buf[:n])


def ffv1_encode(frame_rgb, num_v_slices=1):
    f = np.ascontiguousarray(frame_rgb, dtype=np.uint8)
    H, W, ch = f.shape
    assert ch == 3
    cap = 2 * W * H * 3 + 4096 * num_v_slices
    out = np.empty(cap, np.uint8)
    n = _io().vvio_ffv1_encode_frame(f.ctypes.data_as(C.c_void_p), W, H, int(num_v_slices), out.ctypes.data_as(C.c_void_p), cap)
    if n < 0:
        raise RuntimeError("vvio_ffv1_encode_frame failed")
    return out[:n].tobytes()


_REASONS = {-10: "configuration record CRC mismatch", -2: "not an FFV1 version this entry point decodes (version 0 / 1 streams have no configuration record: "
                                                           "use Ffv1Decoder; version 2 was never released)",
            -26: "FFV1 version > 3 (or a micro_version beyond RFC 9043): other slice header / context rules, not decoded",
            -30: "stream parameters not known before the first key frame", -3: "bad coder_type / state-transition table",
            -6: "coded initial states are not supported", -8: "only 8-bit streams (RGB, or YCbCr with subsampling factors <= 4) are supported",
            -9: "stream colour model does not match the decoder entry point", -21: "slice header out of range", -25: "corrupt sample data",
            -13: "slice CRC mismatch", -20: "non-key frame without a preceding key frame (or with mismatching context states)"}


def ffv1_stream_info(config):
    """{colorspace (0 YCbCr / 1 RGB), chroma_planes, hshift, vshift, alpha, bits} of an FFV1 configuration record."""
    info = (C.c_int * 6)()
    cfg = (C.c_uint8 * len(config)).from_buffer_copy(config)
    r = _io().vvio_ffv1_stream_info(cfg, len(config), info)
    if r != 0:
        raise RuntimeError(f"FFV1 configuration record rejected ({r}): {_REASONS.get(r, 'malformed stream')}")
    return dict(zip(("colorspace", "chroma_planes", "hshift", "vshift", "alpha", "bits"), list(info)))


def ffv1_decode_planes(config, packet, W, H, info=None):
    """a planar YCbCr FFV1 packet -> (Y [H, W], Cb, Cr [ceil(H >> vshift), ceil(W >> hshift)]) uint8."""
    info = info or ffv1_stream_info(config)
    cw, ch = (W + (1 << info["hshift"]) - 1) >> info["hshift"], (H + (1 << info["vshift"]) - 1) >> info["vshift"]
    y, cb, cr = np.empty((H, W), np.uint8), np.empty((ch, cw), np.uint8), np.empty((ch, cw), np.uint8)
    cfg = (C.c_uint8 * len(config)).from_buffer_copy(config)
    pkt = (C.c_uint8 * len(packet)).from_buffer_copy(packet)
    r = _io().vvio_ffv1_decode_frame_yuv(cfg, len(config), pkt, len(packet), W, H, y.ctypes.data_as(C.c_void_p), cb.ctypes.data_as(C.c_void_p),
                                         cr.ctypes.data_as(C.c_void_p))
    if r != 0:
        raise RuntimeError(f"FFV1 decode failed ({r}): {_REASONS.get(r, 'malformed stream')}")
    return y, cb, cr


def ycbcr_to_rgb(y, cb, cr, hshift, vshift, full_range=False, device=None):
    """planar YCbCr ([T,] H, W + subsampled chroma) -> RGB uint8 ([T,] H, W, 3).  Default (device None / False): the host routine
    (vvio_ycbcr_to_rgb) -- file I/O never creates a HIP context by itself (under torchrun every rank loads the clip BEFORE its device is set,
    and a per-frame host -> device -> host round trip is slower than the host loop).  device = True / "cuda:N" / a torch.device: the GPU kernel
    (vv_ycbcr_to_rgb) for callers that batch a whole clip and want it there anyway.  The two are the same integer arithmetic, bit for bit
    (tests/test_frameio_cpu.py, tests/test_kernels_gpu.py)."""
    y, cb, cr = (np.ascontiguousarray(a, dtype=np.uint8) for a in (y, cb, cr))
    single = y.ndim == 2
    if single:
        y, cb, cr = y[None], cb[None], cr[None]
    T, H, W = y.shape
    use_gpu = device not in (None, False)
    if use_gpu:
        import torch
        from . import hip
        dev = torch.device("cuda", torch.cuda.current_device()) if device is True else torch.device(device)
        out = hip.ycbcr_to_rgb(torch.from_numpy(y).to(dev), torch.from_numpy(cb).to(dev), torch.from_numpy(cr).to(dev), hshift, vshift, full_range).cpu().numpy()
    else:
        out = np.empty((T, H, W, 3), np.uint8)
        for t in range(T):
            r = _io().vvio_ycbcr_to_rgb(y[t].ctypes.data_as(C.c_void_p), cb[t].ctypes.data_as(C.c_void_p), cr[t].ctypes.data_as(C.c_void_p), W, H,
                                        int(hshift), int(vshift), int(bool(full_range)), out[t].ctypes.data_as(C.c_void_p))
            if r != 0:
                raise RuntimeError("vvio_ycbcr_to_rgb failed")
    return out[0] if single else out


def ffv1_decode(config, packet, W, H):
    """one FFV1 packet -> RGB uint8 [H, W, 3]; planar YCbCr streams are decoded to planes and colour converted (BT.601 limited range)."""
    info = ffv1_stream_info(config)
    if info["colorspace"] == 0:
        y, cb, cr = ffv1_decode_planes(config, packet, W, H, info)
        return ycbcr_to_rgb(y, cb, cr, info["hshift"], info["vshift"])
    out = np.empty((H, W, 3), np.uint8)
    cfg = (C.c_uint8 * len(config)).from_buffer_copy(config)
    pkt = (C.c_uint8 * len(packet)).from_buffer_copy(packet)
    r = _io().vvio_ffv1_decode_frame(cfg, len(config), pkt, len(packet), W, H, out.ctypes.data_as(C.c_void_p))
    if r != 0:
        raise RuntimeError(f"FFV1 decode failed ({r}): {_REASONS.get(r, 'malformed stream')}")
    return out


class Ffv1Decoder:
    """The decoder of ONE stream, fed packet by packet in stream order (vvio_ffv1_decoder_*): non-key frames continue from the context states the
    previous frame left, and version 0 / 1 streams (config = b"": no configuration record) learn their parameters from the first key frame.
    decode(packet) -> RGB uint8 [H, W, 3] (planar YCbCr streams are colour converted on the host, BT.601 limited range)."""

    def __init__(self, config, W, H):
        self.W, self.H = int(W), int(H)
        st = C.c_int(0)
        cfg = (C.c_uint8 * max(1, len(config))).from_buffer_copy(config if config else b"\x00")
        self._h = _io().vvio_ffv1_decoder_open(cfg, len(config), C.byref(st))
        if not self._h:
            raise RuntimeError(f"FFV1 configuration record rejected ({st.value}): {_REASONS.get(st.value, 'malformed stream')}")

    def info(self):
        info = (C.c_int * 7)()
        r = _io().vvio_ffv1_decoder_info(self._h, info)
        if r != 0:
            raise RuntimeError(f"FFV1 stream info unavailable ({r}): {_REASONS.get(r, 'malformed stream')}")
        return dict(zip(("colorspace", "chroma_planes", "hshift", "vshift", "alpha", "bits", "version"), list(info)))

    def decode(self, packet):
        W, H = self.W, self.H
        rgb = np.empty((H, W, 3), np.uint8)
        y, cb, cr = np.empty((H, W), np.uint8), np.empty((H, W), np.uint8), np.empty((H, W), np.uint8)      # chroma: at most the luma size
        pkt = (C.c_uint8 * max(1, len(packet))).from_buffer_copy(packet if packet else b"\x00")
        r = _io().vvio_ffv1_decoder_decode(self._h, pkt, len(packet), W, H, rgb.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p),
                                           cb.ctypes.data_as(C.c_void_p), cr.ctypes.data_as(C.c_void_p))
        if r < 0:
            raise RuntimeError(f"FFV1 decode failed ({r}): {_REASONS.get(r, 'malformed stream')}")
        if r == 0:
            return rgb
        i = self.info()
        cw, ch = (W + (1 << i["hshift"]) - 1) >> i["hshift"], (H + (1 << i["vshift"]) - 1) >> i["vshift"]
        cbv, crv = cb.reshape(-1)[:cw * ch].reshape(ch, cw), cr.reshape(-1)[:cw * ch].reshape(ch, cw)
        return ycbcr_to_rgb(y, cbv, crv, i["hshift"], i["vshift"])

    def close(self):
        if getattr(self, "_h", None):
            _io().vvio_ffv1_decoder_close(self._h)
            self._h = None

    __del__ = close


# ---- YUV4MPEG2 (.y4m): uncompressed planar YCbCr, what `ffmpeg -i any.mp4 out.y4m` writes ---------------------------------------------
_Y4M_SHIFTS = {"420": (1, 1), "420jpeg": (1, 1), "420mpeg2": (1, 1), "420paldv": (1, 1), "422": (1, 0), "444": (0, 0), "mono": None}


def read_y4m(path, start_frame=0, max_frames=-1):
    """-> (list of RGB uint8 frames, fps).  8-bit 4:2:0 / 4:2:2 / 4:4:4 / mono; limited range unless the header carries XCOLORRANGE=FULL."""
    with open(path, "rb") as f:
        head = f.readline()
        if not head.startswith(b"YUV4MPEG2"):
            raise RuntimeError(f"{path}: not a YUV4MPEG2 file")
        W = H = None
        fps, cs, full = 25.0, "420", False
        for tok in head.decode("ascii", "replace").split()[1:]:
            if tok[0] == "W":
                W = int(tok[1:])
            elif tok[0] == "H":
                H = int(tok[1:])
            elif tok[0] == "F":
                n, d = tok[1:].split(":")
                fps = float(n) / max(1.0, float(d))
            elif tok[0] == "C":
                cs = tok[1:]
            elif tok == "XCOLORRANGE=FULL":
                full = True
        if not W or not H or W <= 0 or H <= 0:
            raise RuntimeError(f"{path}: missing frame size")
        if cs not in _Y4M_SHIFTS:
            raise RuntimeError(f"{path}: chroma format C{cs} is not supported (8-bit 420 / 422 / 444 / mono)")
        sh = _Y4M_SHIFTS[cs]
        hs, vs = sh if sh else (0, 0)
        cw, ch = (W + (1 << hs) - 1) >> hs, (H + (1 << vs) - 1) >> vs
        ys, cbs, crs, idx = [], [], [], 0
        while True:
            line = f.readline()
            if not line:
                break
            if not line.startswith(b"FRAME"):
                raise RuntimeError(f"{path}: frame header expected")
            n = W * H + (2 * cw * ch if sh else 0)
            buf = f.read(n)
            if len(buf) < n:
                break
            if idx >= start_frame:
                a = np.frombuffer(buf, np.uint8)
                ys.append(a[:W * H].reshape(H, W))
                if sh:
                    cbs.append(a[W * H:W * H + cw * ch].reshape(ch, cw))
                    crs.append(a[W * H + cw * ch:].reshape(ch, cw))
                else:
                    cbs.append(np.full((ch, cw), 128, np.uint8))
                    crs.append(np.full((ch, cw), 128, np.uint8))
                if max_frames > 0 and len(ys) >= max_frames:
                    break
            idx += 1
    if not ys:
        return [], fps
    rgb = ycbcr_to_rgb(np.stack(ys), np.stack(cbs), np.stack(crs), hs, vs, full)
    return [np.ascontiguousarray(r) for r in rgb], fps


# ---- EBML / Matroska -----------------------------------------------------------------------------------------------------
def _vint_size(n):
    """EBML data-size field."""
    for length in range(1, 9):
        if n < (1 << (7 * length)) - 1:
            return ((1 << (7 * length)) | n).to_bytes(length, "big")
    raise ValueError("EBML size too large")


def _el(eid, payload):
    return eid + _vint_size(len(payload)) + payload


def _uint(v):
    n = max(1, (int(v).bit_length() + 7) // 8)
    return int(v).to_bytes(n, "big")


ID_EBML, ID_SEGMENT, ID_INFO, ID_TRACKS, ID_TRACKENTRY, ID_CLUSTER = b"\x1a\x45\xdf\xa3", b"\x18\x53\x80\x67", b"\x15\x49\xa9\x66", b"\x16\x54\xae\x6b", b"\xae", b"\x1f\x43\xb6\x75"
ID_TIMECODESCALE, ID_DURATION, ID_MUXAPP, ID_WRITEAPP = b"\x2a\xd7\xb1", b"\x44\x89", b"\x4d\x80", b"\x57\x41"
ID_TRACKNUM, ID_TRACKUID, ID_TRACKTYPE, ID_FLAGLACING, ID_CODECID, ID_CODECPRIVATE, ID_DEFAULTDURATION, ID_VIDEO = b"\xd7", b"\x73\xc5", b"\x83", b"\x9c", b"\x86", b"\x63\xa2", b"\x23\xe3\x83", b"\xe0"
ID_PIXELW, ID_PIXELH, ID_COLOURSPACE = b"\xb0", b"\xba", b"\x2e\xb5\x24"
ID_TIMECODE, ID_SIMPLEBLOCK, ID_BLOCKGROUP, ID_BLOCK = b"\xe7", b"\xa3", b"\xa0", b"\xa1"


def write_mkv_packets(path, W, H, config, packets, fps, key_frames=None):
    """mux FFV1 packets (one key frame each) + their configuration record into a Matroska file (V_FFV1)."""
    packets = list(packets)
    fps = float(fps) if fps and fps > 0 else 25.0
    dur_ns = int(round(1e9 / fps))
    head = _el(ID_EBML, _el(b"\x42\x86", _uint(1)) + _el(b"\x42\xf7", _uint(1)) + _el(b"\x42\xf2", _uint(4)) + _el(b"\x42\xf3", _uint(8)) +
               _el(b"\x42\x82", b"matroska") + _el(b"\x42\x87", _uint(4)) + _el(b"\x42\x85", _uint(2)))
    info = _el(ID_INFO, _el(ID_TIMECODESCALE, _uint(1000000)) + _el(ID_MUXAPP, b"videovanish_amd.frameio") + _el(ID_WRITEAPP, b"videovanish_amd.frameio") +
               _el(ID_DURATION, struct.pack(">d", len(packets) * 1000.0 / fps)))
    video = _el(ID_VIDEO, _el(ID_PIXELW, _uint(W)) + _el(ID_PIXELH, _uint(H)))
    track = _el(ID_TRACKENTRY, _el(ID_TRACKNUM, _uint(1)) + _el(ID_TRACKUID, _uint(1)) + _el(ID_TRACKTYPE, _uint(1)) + _el(ID_FLAGLACING, _uint(0)) +
                _el(ID_CODECID, b"V_FFV1") + (_el(ID_CODECPRIVATE, config) if config else b"") + _el(ID_DEFAULTDURATION, _uint(dur_ns)) + video)
    with open(path, "wb") as f:
        f.write(head)
        f.write(ID_SEGMENT + b"\x01\xff\xff\xff\xff\xff\xff\xff")          # unknown size: clusters are streamed
        f.write(info)
        f.write(_el(ID_TRACKS, track))
        per_cluster = max(1, int(round(fps)))                               # about one second per cluster
        for c0 in range(0, len(packets), per_cluster):
            t0 = int(round(c0 * 1000.0 / fps))
            blocks = []
            for i in range(c0, min(len(packets), c0 + per_cluster)):
                rel = int(round(i * 1000.0 / fps)) - t0
                key = key_frames is None or key_frames[i]
                blocks.append(_el(ID_SIMPLEBLOCK, b"\x81" + struct.pack(">h", rel) + (b"\x80" if key else b"\x00") + packets[i]))     # track 1, key-frame flag
            f.write(_el(ID_CLUSTER, _el(ID_TIMECODE, _uint(t0)) + b"".join(blocks)))


def write_mkv_ffv1(path, frames, fps, num_v_slices=None):
    """frames: iterable of (H, W, 3) uint8 RGB arrays of one size."""
    frames = list(frames)
    H, W = frames[0].shape[:2]
    nv = num_v_slices or max(1, min(H, (H + 539) // 540))       # bands of <= 540 rows keep a slice far below the 16 MB size field
    for fr in frames:
        assert fr.shape[:2] == (H, W), "all frames of a track must have one size"
    write_mkv_packets(path, W, H, ffv1_config_record(nv), [ffv1_encode(fr, nv) for fr in frames], fps)


def _read_vint(buf, pos, is_id):
    b0 = buf[pos]
    length = 1
    mask = 0x80
    while length <= 8 and not (b0 & mask):
        length += 1
        mask >>= 1
    if length > 8:
        raise RuntimeError("malformed EBML")
    raw = buf[pos:pos + length]
    if is_id:
        return bytes(raw), pos + length
    val = int.from_bytes(raw, "big") & ((1 << (7 * length)) - 1)
    if val == (1 << (7 * length)) - 1:
        val = -1                                                            # unknown size
    return val, pos + length


def _children(buf, start, end):
    pos = start
    while pos < end:
        eid, p = _read_vint(buf, pos, True)
        size, p = _read_vint(buf, p, False)
        stop = end if size < 0 else p + size
        yield eid, p, stop
        pos = stop


def read_mkv(path, start_frame=0, max_frames=-1):
    with open(path, "rb") as f:
        buf = memoryview(f.read())
    tracks, frames = {}, []
    fps = 0.0
    want = None
    idx = 0
    pending = []                # FFV1 packets since the last key frame before start_frame
    for eid, a, b in _children(buf, 0, len(buf)):
        if eid != ID_SEGMENT:
            continue
        for sid, sa, sb in _children(buf, a, b):
            if sid == ID_TRACKS:
                for tid, ta, tb in _children(buf, sa, sb):
                    if tid != ID_TRACKENTRY:
                        continue
                    tr = {}
                    for fid, fa, fb in _children(buf, ta, tb):
                        if fid == ID_TRACKNUM: tr["num"] = int.from_bytes(buf[fa:fb], "big")
                        elif fid == ID_TRACKTYPE: tr["type"] = int.from_bytes(buf[fa:fb], "big")
                        elif fid == ID_CODECID: tr["codec"] = bytes(buf[fa:fb]).rstrip(b"\x00").decode()
                        elif fid == ID_CODECPRIVATE: tr["private"] = bytes(buf[fa:fb])
                        elif fid == ID_DEFAULTDURATION: tr["dur"] = int.from_bytes(buf[fa:fb], "big")
                        elif fid == ID_VIDEO:
                            for vid, va, vb in _children(buf, fa, fb):
                                if vid == ID_PIXELW: tr["W"] = int.from_bytes(buf[va:vb], "big")
                                elif vid == ID_PIXELH: tr["H"] = int.from_bytes(buf[va:vb], "big")
                                elif vid == ID_COLOURSPACE: tr["fourcc"] = bytes(buf[va:vb])
                    if tr.get("type") == 1 and want is None:
                        want = tr
                        tracks[tr["num"]] = tr
            elif sid == ID_CLUSTER and want is not None:
                for cid, ca, cb in _children(buf, sa, sb):
                    blk = None
                    if cid == ID_SIMPLEBLOCK:
                        blk = (ca, cb)
                    elif cid == ID_BLOCKGROUP:
                        for gid, ga, gb in _children(buf, ca, cb):
                            if gid == ID_BLOCK:
                                blk = (ga, gb)
                    if blk is None:
                        continue
                    num, p = _read_vint(buf, blk[0], False)
                    if num != want["num"]:
                        continue
                    payload = bytes(buf[p + 3: blk[1]])
                    # FFV1 streams may hold non-key frames (adaptive states continue from the previous frame), so decoding has to start at a key
                    # frame -- but only at the LAST one at or before start_frame: the packets before start_frame are buffered since the latest key
                    # frame (one GOP at most) and decoded only if the first wanted frame is not a key frame itself.  An intra-only stream (what
                    # this module's writer and `-g 1` encoders produce) therefore costs no decode per skipped frame (O(start_frame)
                    # decodes per call, O(N^2) for a clip read in windows).
                    if max_frames <= 0 or len(frames) < max_frames:
                        if idx < start_frame:
                            if _is_ffv1(want):
                                if _ffv1_is_key_packet(payload):
                                    pending = []
                                pending.append(payload)
                        else:
                            if pending and not _ffv1_is_key_packet(payload):
                                for q in pending:
                                    _decode_payload(want, q, keep=False)
                            pending = []
                            frames.append(_decode_payload(want, payload, keep=True))
                    idx += 1
                    if max_frames > 0 and len(frames) >= max_frames:
                        break
    if want is None:
        raise RuntimeError(f"{path}: no video track")
    if want.get("dur"):
        fps = 1e9 / want["dur"]
    return frames, fps


def _is_ffv1(tr):
    codec = tr.get("codec", "")
    return codec == "V_FFV1" or (codec == "V_MS/VFW/FOURCC" and tr.get("private", b"")[16:20] == b"FFV1")


def _ffv1_is_key_packet(payload):
    """The first symbol of every FFV1 frame (all versions, either sample coder) is the range-coded `keyframe` bit with the fixed initial state 128
    (RFC 9043 section 4.4; 3.8.1.1: low = the first two bytes capped at 0xFF00, range = 0xFF00): range1 = 0xFF00 * 128 >> 8 = 0x7F80 and
    the bit is 1 exactly when low >= range - range1 = 0x7F80."""
    if len(payload) < 2:
        return True
    return min((payload[0] << 8) | payload[1], 0xFF00) >= 0x7F80


def _ffv1_track_decoder(tr, config):
    if "_ffv1" not in tr:
        tr["_ffv1"] = Ffv1Decoder(config, tr["W"], tr["H"])
    return tr["_ffv1"]


def _decode_payload(tr, payload, keep=True):
    codec, W, H = tr.get("codec", ""), tr["W"], tr["H"]
    if codec == "V_FFV1":
        return _ffv1_track_decoder(tr, tr.get("private", b"")).decode(payload)       # no CodecPrivate: an FFV1 version 0 / 1 stream
    if codec == "V_MS/VFW/FOURCC":
        priv = tr.get("private", b"")
        if len(priv) >= 40 and priv[16:20] == b"FFV1":
            return _ffv1_track_decoder(tr, priv[40:]).decode(payload)               # BITMAPINFOHEADER (40 bytes) [+ FFV1 configuration record: version 3]
        raise RuntimeError(f"unsupported VFW codec {priv[16:20]!r}: this reader decodes FFV1 and uncompressed RGB only")
    if not keep:
        return None
    if codec == "V_UNCOMPRESSED":
        fourcc = tr.get("fourcc", b"RGB\x18")
        a = np.frombuffer(payload, np.uint8)[: H * W * 3].reshape(H, W, 3)
        return a[..., ::-1].copy() if fourcc.startswith(b"BGR") else a.copy()
    raise RuntimeError(f"unsupported codec {codec!r}: this reader decodes FFV1 (versions 0 / 1 / 3, 8-bit RGB or planar YCbCr) and uncompressed RGB in Matroska, and .y4m")


# ---- the reference's tools.py API ----------------------------------------------------------------------------------------
def resize_nearest(frame, W0, H0):
    """cv2.resize(f, (W0, H0), interpolation=INTER_NEAREST) restated: src index = min(floor(dst * scale), size - 1) (reference tools.py:41-42)."""
    H, W = frame.shape[:2]
    ys = np.minimum((np.arange(H0) * (H / float(H0))).astype(np.int64), H - 1)
    xs = np.minimum((np.arange(W0) * (W / float(W0))).astype(np.int64), W - 1)
    return frame[ys][:, xs]


def load_video_frames_from_path(video_path, start_frame=0, max_frames=-1):
    """reference tools.py:4-28: (list of RGB uint8 frames, fps); asserts that at least one frame was read."""
    assert os.path.isfile(video_path), f"Failed to open video: {video_path}"
    ext = os.path.splitext(video_path)[1].lower()
    if ext == ".npy":
        arr, fps = np.load(video_path), 25.0
        frames = [f for f in arr[start_frame: (start_frame + max_frames) if max_frames > 0 else None]]
    elif ext == ".npz":
        z = np.load(video_path)
        arr, fps = z["frames"], float(z["fps"]) if "fps" in z else 25.0
        frames = [f for f in arr[start_frame: (start_frame + max_frames) if max_frames > 0 else None]]
    elif ext == ".y4m":
        frames, fps = read_y4m(video_path, start_frame, max_frames)
    else:
        frames, fps = read_mkv(video_path, start_frame, max_frames)
    assert len(frames) > 0, "No frames read"
    return frames, fps


def write_video_frames_to_path(out_video, mask_frames, fps, H0, W0):
    """reference tools.py:30-45: lossless FFV1 / MKV; frames of another size are NEAREST-resized to (W0, H0) first (:41-42)."""
    out = []
    for f in mask_frames:
        f = np.asarray(f, dtype=np.uint8)
        if f.shape[0] != H0 or f.shape[1] != W0:
            f = resize_nearest(f, W0, H0)
        out.append(np.ascontiguousarray(f))
    write_mkv_ffv1(out_video, out, fps)
    print(f"[ok] wrote {len(out)} frames to {out_video}")
