"""Test infrastructure: a second, independent FFV1 ENCODER (version 3, and the version 0 / 1 bitstream) in pure Python, written from RFC 9043 (range coder 3.8.1,
default_state_transition 3.8.1.5, alternative table 3.8.1.6, configuration record 4.2, slice header 4.5, sample coding 3.8 with
the median predictor 3.3 and the context model 3.4-3.5, JPEG 2000 RCT 3.7.2, slice footer 4.8).  It emits what libavcodec's
defaults look like -- RANGE-CODED samples (coder_type 1 or 2), a num_h_slices x num_v_slices grid, two quantisation-table sets
(3 and 5 context inputs), optional extra (alpha) plane, optional CRC -- i.e. streams the C encoder in csrc/vv_ffv1.c cannot produce.
The C DECODER is checked against these packets (tests/test_frameio_cpu.py).  Slow (pure Python): tiny images only.

Round 4: `keyframe=False` + `persist` (a dict that carries the adaptive context states from frame to frame, per slice) write NON-KEY frames
(RFC 9043 4.4: key-frame bit 0, no header, states continue), `intra=0` in the configuration record; `legacy=0 | 1` writes FFV1 VERSION 0 / 1 frames
(RFC 9043 4.2 "Parameters" inside every key frame instead of a configuration record, ONE quantisation-table set, one slice, no slice header, no
slice footer, no CRC)."""
import numpy as np

# RFC 9043 3.8.1.6: the alternative state transition table (coder_type 2 streams carry it as deltas to the default table)
ALT_STATE = [
    0, 10, 10, 10, 10, 16, 16, 16, 28, 16, 16, 29, 42, 49, 20, 49,
    59, 25, 26, 26, 27, 31, 33, 33, 33, 34, 34, 37, 67, 38, 39, 39,
    40, 40, 41, 79, 43, 44, 45, 45, 48, 48, 64, 50, 51, 52, 88, 52,
    53, 74, 55, 57, 58, 58, 74, 60, 101, 61, 62, 84, 66, 66, 68, 69,
    87, 82, 71, 97, 73, 73, 82, 75, 111, 77, 94, 78, 87, 81, 83, 97,
    85, 83, 94, 86, 99, 89, 90, 99, 111, 92, 93, 134, 95, 98, 105, 98,
    105, 110, 102, 108, 102, 118, 103, 106, 106, 113, 109, 112, 114, 112, 116, 125,
    115, 116, 117, 117, 126, 119, 125, 121, 121, 123, 145, 124, 126, 131, 127, 129,
    165, 130, 132, 138, 133, 135, 145, 136, 137, 139, 146, 141, 143, 142, 144, 148,
    147, 155, 151, 149, 151, 150, 152, 157, 153, 154, 156, 168, 158, 162, 161, 160,
    172, 163, 169, 164, 166, 184, 167, 170, 177, 174, 171, 173, 182, 176, 180, 178,
    175, 189, 179, 181, 186, 183, 192, 185, 200, 187, 191, 188, 190, 197, 193, 196,
    197, 194, 195, 196, 198, 202, 199, 201, 210, 203, 207, 204, 205, 206, 208, 214,
    209, 211, 221, 212, 213, 215, 224, 216, 217, 218, 219, 220, 222, 228, 223, 225,
    226, 224, 227, 229, 240, 230, 231, 232, 233, 234, 235, 236, 238, 239, 237, 242,
    241, 243, 242, 244, 245, 246, 247, 248, 249, 250, 251, 252, 252, 253, 254, 255]


def default_state_transition():
    """RFC 9043 3.8.1.5 as the published construction (probability step 0.05, states capped at 248)."""
    one = 1 << 32
    factor = int(0.05 * (1 << 32))
    max_p = 256 - 8
    tab = [0] * 256
    p, last = one // 2, 0
    for _ in range(128):
        p8 = (256 * p + one // 2) >> 32
        if p8 <= last:
            p8 = last + 1
        if last and last < 256 and p8 <= max_p:
            tab[last] = p8
        p += ((one - p) * factor + one // 2) >> 32
        last = p8
    for i in range(256 - max_p, max_p + 1):
        if tab[i]:
            continue
        p = (i * one + 128) >> 8
        p += ((one - p) * factor + one // 2) >> 32
        p8 = (256 * p + one // 2) >> 32
        p8 = max(p8, i + 1)
        tab[i] = min(p8, max_p)
    return tab


class RangeEncoder:
    def __init__(self):
        self.low, self.range, self.out = 0, 0xFF00, bytearray()
        self.outstanding_count, self.outstanding_byte = 0, -1
        self.set_table(default_state_transition())

    def set_table(self, one_state):
        self.one = list(one_state)
        self.zero = [0] * 256
        for i in range(1, 256):
            self.zero[256 - i] = 256 - self.one[i]

    def _renorm(self):
        while self.range < 0x100:
            if self.outstanding_byte < 0:
                self.outstanding_byte = self.low >> 8
            elif self.low <= 0xFF00:
                self.out.append(self.outstanding_byte)
                self.out.extend(b"\xff" * self.outstanding_count)
                self.outstanding_count = 0
                self.outstanding_byte = self.low >> 8
            elif self.low >= 0x10000:
                self.out.append(self.outstanding_byte + 1)
                self.out.extend(b"\x00" * self.outstanding_count)
                self.outstanding_count = 0
                self.outstanding_byte = (self.low >> 8) & 0xFF
            else:
                self.outstanding_count += 1
            self.low = (self.low & 0xFF) << 8
            self.range <<= 8

    def put(self, state, idx, bit):
        r1 = (self.range * state[idx]) >> 8
        if not bit:
            self.range -= r1
            state[idx] = self.zero[state[idx]]
        else:
            self.low += self.range - r1
            self.range = r1
            state[idx] = self.one[state[idx]]
        self._renorm()

    def put_symbol(self, state, v, signed, base=0):
        if v == 0:
            self.put(state, base, 1)
            return
        a = abs(v)
        e = a.bit_length() - 1
        self.put(state, base, 0)
        for i in range(e):
            self.put(state, base + 1 + min(i, 9), 1)
        self.put(state, base + 1 + min(e, 9), 0)
        for i in range(e - 1, -1, -1):
            self.put(state, base + 22 + min(i, 9), (a >> i) & 1)
        if signed:
            self.put(state, base + 11 + min(e, 10), 1 if v < 0 else 0)

    def terminate(self):
        self.range = 0xFF
        self.low += 0xFF
        self._renorm()
        self.range = 0xFF
        self._renorm()
        return bytes(self.out)


def crc32_mpeg(b):
    c = 0
    for x in b:
        c ^= x << 24
        for _ in range(8):
            c = ((c << 1) ^ 0x04C11DB7) & 0xFFFFFFFF if c & 0x80000000 else (c << 1) & 0xFFFFFFFF
    return c


def quant_tables():
    """Two sets as libavcodec's 8-bit defaults are shaped: set 0 = 3 inputs of 11 levels, set 1 = 5 inputs (11, 11, 5, 5, 5 levels).
    (The thresholds are this file's own: any monotone table is a valid stream, the tables travel in the configuration record.)"""
    def table(edges):
        t = [0] * 256
        for i in range(128):
            t[i] = sum(1 for e in edges if i >= e)
        for i in range(1, 128):
            t[256 - i] = -t[i]
        t[128] = -t[127]
        return t
    q11, q5 = table([1, 2, 4, 7, 12]), table([1, 3])
    def scaled(t, s):
        return [s * v for v in t]
    set0 = [scaled(q11, 1), scaled(q11, 11), scaled(q11, 121), [0] * 256, [0] * 256]
    set1 = [scaled(q11, 1), scaled(q11, 11), scaled(q5, 121), scaled(q5, 605), scaled(q5, 3025)]
    return [set0, set1], [(11 ** 3 + 1) // 2, (11 * 11 * 5 * 5 * 5 + 1) // 2]


def _put_quant_table(rc, tab):
    st = [128] * 32
    last = 0
    i = 1
    while i < 128:
        if tab[i] != tab[i - 1]:
            rc.put_symbol(st, i - last - 1, False)
            last = i
        i += 1
    rc.put_symbol(st, i - last - 1, False)


def config_record(coder, nh, nv, alpha=False, ec=1, sets=None, colorspace=1, chroma_planes=True, hshift=0, vshift=0, intra=1, version=3, micro=4):
    rc = RangeEncoder()
    st = [128] * 32
    rc.put_symbol(st, version, False)      # version
    rc.put_symbol(st, micro, False)        # micro_version
    rc.put_symbol(st, coder, False)
    if coder == 2:
        d = default_state_transition()
        for i in range(1, 256):
            rc.put_symbol(st, ALT_STATE[i] - d[i], True)
    rc.put_symbol(st, colorspace, False)   # colorspace_type: 1 RGB (JPEG 2000 RCT), 0 YCbCr
    rc.put_symbol(st, 8, False)            # bits_per_raw_sample
    rc.put(st, 0, 1 if chroma_planes else 0)
    rc.put_symbol(st, hshift, False)       # log2_h_chroma_subsample
    rc.put_symbol(st, vshift, False)       # log2_v_chroma_subsample
    rc.put(st, 0, 1 if alpha else 0)       # extra_plane
    rc.put_symbol(st, nh - 1, False)
    rc.put_symbol(st, nv - 1, False)
    sets = sets or quant_tables()[0]
    rc.put_symbol(st, len(sets), False)
    for qs in sets:
        scale_tabs = []
        sc = 1
        for t in range(5):
            # tables are stored unscaled-equivalent: the run lengths only depend on where the value changes
            _put_quant_table(rc, qs[t])
    for _ in sets:
        rc.put(st, 0, 0)                   # states_coded = 0
    rc.put_symbol(st, ec, False)
    rc.put_symbol(st, intra, False)        # intra: 1 = every frame is a key frame
    body = rc.terminate()
    return body + crc32_mpeg(body).to_bytes(4, "big")


def _median(a, b, c):
    return sorted((a, b, c))[1]


# ---- Golomb-Rice sample coding (RFC 9043 3.8.2), written from the RFC: run mode + adaptive Rice parameter per context -------------------------
LOG2_RUN = [0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24]


class BitWriter:
    def __init__(self):
        self.bits = []

    def put(self, n, v):
        for i in range(n - 1, -1, -1):
            self.bits.append((v >> i) & 1)

    def tobytes(self):
        b = self.bits + [0] * (-len(self.bits) % 8)
        return bytes(int("".join(map(str, b[i:i + 8])), 2) for i in range(0, len(b), 8))


class VlcState:
    def __init__(self):
        self.drift, self.error_sum, self.bias, self.count = 0, 4, 0, 1


def _fold(v, bits):
    v &= (1 << bits) - 1
    return v - (1 << bits) if v >> (bits - 1) else v


def _put_vlc(bw, st, v, bits):
    v = _fold(v - st.bias, bits)
    k, i = 0, st.count
    while i < st.error_sum:
        k, i = k + 1, i + i
    code = v ^ (-1 if 2 * st.drift + st.count < 0 else 0)
    u = 2 * code if code >= 0 else -2 * code - 1            # signed -> unsigned: 0, -1, 1, -2, ... -> 0, 1, 2, 3, ...
    e = u >> k
    if e < 12:
        bw.put(e, 0)
        bw.put(1, 1)
        bw.put(k, u & ((1 << k) - 1))
    else:
        bw.put(12, 0)
        bw.put(bits, u - 11)
    # state update (RFC 9043 3.8.2.4)
    st.error_sum += abs(v)
    st.drift += v
    if st.count == 128:
        st.count >>= 1
        st.drift >>= 1
        st.error_sum >>= 1
    st.count += 1
    if st.drift <= -st.count:
        st.bias = max(st.bias - 1, -128)
        st.drift = max(st.drift + st.count, -st.count + 1)
    elif st.drift > 0:
        st.bias = min(st.bias + 1, 127)
        st.drift = min(st.drift - st.count, 0)


class GolombLineCoder:
    """codes the (context, difference) pairs of successive lines; the run index persists across the lines (and planes) it is used for."""

    def __init__(self, bw, bits):
        self.bw, self.bits, self.run_index = bw, bits, 0

    def _flush(self, run_count, final):
        while run_count >= 1 << LOG2_RUN[self.run_index]:
            run_count -= 1 << LOG2_RUN[self.run_index]
            self.run_index += 1
            self.bw.put(1, 1)
        return run_count

    def line(self, symbols, states):
        run_mode, run_count = False, 0
        for ctx, diff in symbols:
            if ctx == 0:
                run_mode = True
            if run_mode:
                if diff != 0:
                    run_count = self._flush(run_count, False)
                    self.bw.put(1 + LOG2_RUN[self.run_index], run_count)
                    if self.run_index:
                        self.run_index -= 1
                    run_count, run_mode = 0, False
                    if diff > 0:
                        diff -= 1
                else:
                    run_count += 1
            if not run_mode:
                _put_vlc(self.bw, states[ctx], diff, self.bits)
        if run_mode:
            run_count = self._flush(run_count, True)
            if run_count:
                self.bw.put(1, 1)


def _line_symbols(P, qs, y, w, bits):
    """(context, difference) of every sample of line y of plane P (int array [h, w]) under the border rules of RFC 9043 3.2."""
    five = any(qs[3]) or any(qs[4])
    half = 1 << (bits - 1)

    def S(yy, xx):
        if yy < 0 or xx < -1:
            return 0                                   # rows above the slice and the additional column to the left are 0
        if xx < 0:
            return S(yy - 1, 0) if yy > 0 else 0       # left border = the first column shifted down by one row, 0 on top
        if xx >= w:
            return int(P[yy, w - 1])
        return int(P[yy, xx])
    out = []
    for x in range(w):
        L, T, LT, RT = S(y, x - 1), S(y - 1, x), S(y - 1, x - 1), S(y - 1, x + 1)
        ctx = qs[0][(L - LT) & 0xFF] + qs[1][(LT - T) & 0xFF] + qs[2][(T - RT) & 0xFF]
        if five:
            ctx += qs[3][(S(y, x - 2) - L) & 0xFF] + qs[4][(S(y - 2, x) - T) & 0xFF]
        diff = int(P[y, x]) - _median(L, L + T - LT, T)
        if ctx < 0:
            ctx, diff = -ctx, -diff
        out.append((ctx, ((diff + half) & (2 * half - 1)) - half))       # folded to `bits` bits
    return out


def _legacy_header(rc, version, coder, colorspace, chroma_planes, hshift, vshift, alpha, qset):
    """versions 0 / 1: the parameters + the ONE quantisation-table set, in the frame's own range coder right behind the key-frame bit"""
    st = [128] * 32
    rc.put_symbol(st, version, False)
    rc.put_symbol(st, coder, False)
    if coder == 2:
        d = default_state_transition()
        for i in range(1, 256):
            rc.put_symbol(st, ALT_STATE[i] - d[i], True)
    rc.put_symbol(st, colorspace, False)
    if version > 0:
        rc.put_symbol(st, 8, False)        # bits_per_raw_sample (version 0 has no such field: 8 bits)
    rc.put(st, 0, 1 if chroma_planes else 0)
    rc.put_symbol(st, hshift, False)
    rc.put_symbol(st, vshift, False)
    rc.put(st, 0, 1 if alpha else 0)
    for t in range(5):
        _put_quant_table(rc, qset[t])


def _slice_header(rc, first, coder, sx, sy, qidx, keyframe=True):
    if first:
        rc.put([128], 0, 1 if keyframe else 0)        # key-frame bit
    if coder == 2:
        rc.set_table(ALT_STATE)
    st = [128] * 32
    for v in (sx, sy, 0, 0):
        rc.put_symbol(st, v, False)
    for v in qidx:
        rc.put_symbol(st, v, False)
    rc.put_symbol(st, 3, False)    # picture_structure
    rc.put_symbol(st, 0, False)
    rc.put_symbol(st, 0, False)


def _finish_slice(rc, bw, ec, legacy=None):
    if bw is not None:                                   # Golomb-Rice: the range coder only carried the header
        if legacy is None:
            rc.put([129], 0, 0)                          # (version >= 3.2 only)
        body = rc.terminate() + bw.tobytes()
    else:
        body = rc.terminate()
    if legacy is not None:
        return bytearray(body)                           # versions 0 / 1: no slice footer
    sl = bytearray(body) + len(body).to_bytes(3, "big")
    if ec:
        sl += b"\x00"
        sl += crc32_mpeg(bytes(sl)).to_bytes(4, "big")
    return sl


def encode_frame(rgb, coder, nh, nv, alpha=None, ec=1, set_luma=0, set_chroma=1, set_alpha=0, keyframe=True, persist=None, legacy=None):
    """rgb: (H, W, 3) uint8; alpha: optional (H, W) uint8.  coder 1 / 2: range-coded samples (default / custom state table); coder 0:
    Golomb-Rice.  Returns the FFV1 packet (all slices).  RGB mode codes the planes G, B-G, R-G (9 bits) line by line, interleaved."""
    sets, counts = quant_tables()
    H, W = rgb.shape[:2]
    pkt = bytearray()
    first = True
    persist = {} if persist is None else persist
    if legacy is not None:
        assert nh == nv == 1 and set_luma == set_chroma == set_alpha, "versions 0 / 1: one slice, one quantisation-table set"
    for sy in range(nv):
        for sx in range(nh):
            y0, y1 = sy * H // nv, (sy + 1) * H // nv
            x0, x1 = sx * W // nh, (sx + 1) * W // nh
            w, h = x1 - x0, y1 - y0
            rc = RangeEncoder()
            qidx = [set_luma, set_chroma] + ([set_alpha] if alpha is not None else [])
            if legacy is None:
                _slice_header(rc, first, coder, sx, sy, qidx, keyframe)
            else:
                rc.put([128], 0, 1 if keyframe else 0)
                if keyframe:
                    _legacy_header(rc, legacy, coder, 1, True, 0, 0, alpha is not None, sets[set_luma])
                if coder == 2:
                    rc.set_table(ALT_STATE)
            first = False
            plane_set = [qidx[0], qidx[1], qidx[1]] + ([qidx[2]] if alpha is not None else [])
            px = rgb[y0:y1, x0:x1].astype(np.int32)
            g, b, r = px[..., 1].copy(), px[..., 2].copy(), px[..., 0].copy()
            b -= g
            r -= g
            g += (b + r) >> 2
            b += 256
            r += 256
            planes = [g, b, r] + ([alpha[y0:y1, x0:x1].astype(np.int32)] if alpha is not None else [])
            # one state array per plane INDEX (0 luma, 1 both chroma planes, 2 alpha), even when two indices name the same table set
            nidx = 3 if alpha is not None else 2
            if keyframe or (sy, sx) not in persist:          # a key frame resets the adaptive states; a non-key frame continues from the previous frame's
                persist[(sy, sx)] = ([[VlcState() for _ in range(counts[plane_set[[0, 1, 3][i]]])] for i in range(nidx)] if coder == 0 else
                                     [[128] * (32 * counts[plane_set[[0, 1, 3][i]]]) for i in range(nidx)])
            if coder == 0:
                bw = BitWriter()
                gl = GolombLineCoder(bw, 9)
                vst = persist[(sy, sx)]
            else:
                bw = None
                sarr = persist[(sy, sx)]
            for y in range(h):
                for p, P in enumerate(planes):
                    syms = _line_symbols(P, sets[plane_set[p]], y, w, 9)
                    if coder == 0:
                        gl.line(syms, vst[(p + 1) // 2])
                    else:
                        arr = sarr[(p + 1) // 2]
                        for ctx, diff in syms:
                            rc.put_symbol(arr, diff, True, base=32 * ctx)
            pkt += _finish_slice(rc, bw, ec, legacy)
    return bytes(pkt)


def encode_frame_ycbcr(y, cb, cr, coder, nh, nv, hshift, vshift, alpha=None, ec=1, set_luma=0, set_chroma=1, set_alpha=0, keyframe=True, persist=None,
                       legacy=None):
    """planar 8-bit YCbCr (colorspace_type 0): y (H, W), cb / cr (ceil(H >> vshift), ceil(W >> hshift)) or None for a gray stream.  Per slice
    the planes are coded one after the other (Y, Cb, Cr [, alpha]); Cb and Cr share the context states of plane index 1; the run index of the
    Golomb-Rice coder restarts with every plane (RFC 9043 4.7 / 3.8.2.2)."""
    sets, counts = quant_tables()
    H, W = y.shape
    chroma = cb is not None
    pkt = bytearray()
    first = True
    persist = {} if persist is None else persist
    if legacy is not None:
        assert nh == nv == 1 and set_luma == set_chroma == set_alpha, "versions 0 / 1: one slice, one quantisation-table set"
    for sy in range(nv):
        for sx in range(nh):
            y0, y1 = sy * H // nv, (sy + 1) * H // nv
            x0, x1 = sx * W // nh, (sx + 1) * W // nh
            w, h = x1 - x0, y1 - y0
            rc = RangeEncoder()
            qidx = [set_luma] + ([set_chroma] if chroma else []) + ([set_alpha] if alpha is not None else [])
            if legacy is None:
                _slice_header(rc, first, coder, sx, sy, qidx, keyframe)
            else:
                rc.put([128], 0, 1 if keyframe else 0)
                if keyframe:
                    _legacy_header(rc, legacy, coder, 0, chroma, hshift, vshift, alpha is not None, sets[set_luma])
                if coder == 2:
                    rc.set_table(ALT_STATE)
            first = False
            cw, ch = (w + (1 << hshift) - 1) >> hshift, (h + (1 << vshift) - 1) >> vshift
            cx, cy = x0 >> hshift, y0 >> vshift
            jobs = [(y[y0:y1, x0:x1].astype(np.int32), 0, set_luma)]
            if chroma:
                jobs += [(cb[cy:cy + ch, cx:cx + cw].astype(np.int32), 1, set_chroma), (cr[cy:cy + ch, cx:cx + cw].astype(np.int32), 1, set_chroma)]
            if alpha is not None:
                jobs.append((alpha[y0:y1, x0:x1].astype(np.int32), 2, set_alpha))
            bw = BitWriter() if coder == 0 else None
            if keyframe or (sy, sx) not in persist:
                persist[(sy, sx)] = {}
            states = persist[(sy, sx)]
            for P, idx, si in jobs:
                if idx not in states:
                    states[idx] = [VlcState() for _ in range(counts[si])] if coder == 0 else [128] * (32 * counts[si])
                gl = GolombLineCoder(bw, 8) if coder == 0 else None
                for yy in range(P.shape[0]):
                    syms = _line_symbols(P, sets[si], yy, P.shape[1], 8)
                    if coder == 0:
                        gl.line(syms, states[idx])
                    else:
                        for ctx, diff in syms:
                            rc.put_symbol(states[idx], diff, True, base=32 * ctx)
            pkt += _finish_slice(rc, bw, ec, legacy)
    return bytes(pkt)
