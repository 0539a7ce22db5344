"""Test infrastructure: a second, independent FFV1 version-3 ENCODER in pure Python, written from RFC 9043 (range coder 3.8.1,
default_state_transition 3.8.1.5, alternative table 3.8.1.6, configuration record 4.2, slice header 4.5, sample coding 3.8 with
the median predictor 3.3 and the context model 3.4-3.5, JPEG 2000 RCT 3.7.2, slice footer 4.8).  It emits what libavcodec's
defaults look like -- RANGE-CODED samples (coder_type 1 or 2), a num_h_slices x num_v_slices grid, two quantisation-table sets
(3 and 5 context inputs), optional extra (alpha) plane, optional CRC -- i.e. streams the C encoder in csrc/vv_ffv1.c cannot produce.
The C DECODER is checked against these packets (tests/test_frameio_cpu.py).  Slow (pure Python): tiny images only."""
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


def config_record(coder, nh, nv, alpha=False, ec=1, sets=None):
    rc = RangeEncoder()
    st = [128] * 32
    rc.put_symbol(st, 3, False)            # version
    rc.put_symbol(st, 4, False)            # micro_version
    rc.put_symbol(st, coder, False)
    if coder == 2:
        d = default_state_transition()
        for i in range(1, 256):
            rc.put_symbol(st, ALT_STATE[i] - d[i], True)
    rc.put_symbol(st, 1, False)            # colorspace_type RGB
    rc.put_symbol(st, 8, False)            # bits_per_raw_sample
    rc.put(st, 0, 1)                       # chroma_planes
    rc.put_symbol(st, 0, False)
    rc.put_symbol(st, 0, False)
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
    rc.put_symbol(st, 1, False)            # intra
    body = rc.terminate()
    return body + crc32_mpeg(body).to_bytes(4, "big")


def _median(a, b, c):
    return sorted((a, b, c))[1]


def encode_frame(rgb, coder, nh, nv, alpha=None, ec=1, set_luma=0, set_chroma=1, set_alpha=0):
    """rgb: (H, W, 3) uint8; alpha: optional (H, W) uint8.  Returns the FFV1 packet (all slices, range-coded samples)."""
    sets, counts = quant_tables()
    H, W = rgb.shape[:2]
    pkt = bytearray()
    first = True
    for sy in range(nv):
        for sx in range(nh):
            y0, y1 = sy * H // nv, (sy + 1) * H // nv
            x0, x1 = sx * W // nh, (sx + 1) * W // nh
            w, h = x1 - x0, y1 - y0
            rc = RangeEncoder()
            if first:
                rc.put([128], 0, 1)        # keyframe
                first = False
            if coder == 2:
                rc.set_table(ALT_STATE)
            st = [128] * 32
            for v in (sx, sy, 0, 0):
                rc.put_symbol(st, v, False)
            qidx = [set_luma, set_chroma] + ([set_alpha] if alpha is not None else [])
            for v in qidx:
                rc.put_symbol(st, v, False)
            rc.put_symbol(st, 3, False)    # picture_structure
            rc.put_symbol(st, 0, False)
            rc.put_symbol(st, 0, False)
            nplanes = 4 if alpha is not None else 3
            plane_set = [qidx[0], qidx[1], qidx[1]] + ([qidx[2]] if alpha is not None else [])
            states = {}
            for s in set(plane_set):
                pass
            # one state array per CONTEXT SET index (planes 1 and 2 share theirs; luma / alpha sets are separate arrays even when they
            # name the same table set)
            sarr = [[128] * (32 * counts[plane_set[0]]), [128] * (32 * counts[plane_set[1]])]
            if alpha is not None:
                sarr.append([128] * (32 * counts[plane_set[3]]))
            px = rgb[y0:y1, x0:x1].astype(np.int32)
            g, b, r = px[..., 1].copy(), px[..., 2].copy(), px[..., 0].copy()
            b -= g
            r -= g
            g += (b + r) >> 2
            b += 256
            r += 256
            planes = [g, b, r] + ([alpha[y0:y1, x0:x1].astype(np.int32)] if alpha is not None else [])
            for y in range(h):
                for p in range(nplanes):
                    P = planes[p]
                    qs = sets[plane_set[p]]
                    five = any(qs[3]) or any(qs[4])
                    arr = sarr[(p + 1) // 2]

                    def S(yy, xx):          # RFC 9043 3.1-3.2: samples outside the slice
                        if yy < 0:
                            return 0
                        if xx < 0:
                            return S(yy - 1, 0) if yy > 0 else 0       # left border = top sample of the first column ... of the row above
                        if xx >= w:
                            return int(P[yy, w - 1])
                        return int(P[yy, xx])
                    for x in range(w):
                        L, T, LT, RT = S(y, x - 1), S(y - 1, x), S(y - 1, x - 1), S(y - 1, x + 1)
                        if y == 0:
                            T = LT = RT = 0
                        ctx = qs[0][(L - LT) & 0xFF] + qs[1][(LT - T) & 0xFF] + qs[2][(T - RT) & 0xFF]
                        if five:
                            LL = S(y, x - 2) if x >= 1 else L          # cur[-2] = cur[-1] at the left border
                            TT = S(y - 2, x) if y >= 2 else 0
                            ctx += qs[3][(LL - L) & 0xFF] + qs[4][(TT - T) & 0xFF]
                        pred = _median(L, L + T - LT, T)
                        diff = int(P[y, x]) - pred
                        if ctx < 0:
                            ctx, diff = -ctx, -diff
                        diff = ((diff + 256) & 0x1FF) - 256               # fold to 9 bits
                        rc.put_symbol(arr, diff, True, base=32 * ctx)
            body = rc.terminate()
            n = len(body)
            sl = bytearray(body) + n.to_bytes(3, "big")
            if ec:
                sl += b"\x00"
                sl += crc32_mpeg(bytes(sl)).to_bytes(4, "big")
            pkt += sl
    return bytes(pkt)
