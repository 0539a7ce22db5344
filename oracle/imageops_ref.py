"""ORACLE (test infrastructure): numpy restatement of the integer / byte image steps on the hot path.

In-tree reference steps (pinned by tests/golden/reference_intree.npz, captured from the real reference module):
  * collapse_and_dilate  -- reference diffuerase.py:27-31 (np.any over channels + scipy binary_dilation, 3x3 cross)
  * feather_alpha / composite -- reference diffuerase.py:77-112
OpenCV calls the reference makes (cv2 is absent here => restated from OpenCV's published algorithms; unpinned
against a real cv2 build):
  * resize_bilinear_u8   -- cv2.resize(f,(W0,H0)) default INTER_LINEAR, reference diffuerase.py:73
                            (opencv-python, unpinned: install_videovanish.sh:60; legacy 8-bit fixed-point path,
                             INTER_RESIZE_COEF_BITS = 11)
  * resize_nearest_u8    -- cv2.resize(..., INTER_NEAREST), reference diffuerase.py:86
  * distance_transform_l2_5 -- cv2.distanceTransform(., DIST_L2, 5), reference diffuerase.py:95-96
                            (two-pass 5x5 chamfer, 16.16 fixed point, a=1 b=1.4 c=2.1969)
Build-defined steps of the third-party compose ([UNVERIFIED-3P]):
  * gaussian_blur_21, blur_compose
"""
import numpy as np

DIST_SHIFT = 16
HV = int(round(1.0 * (1 << DIST_SHIFT)))
DIAG = int(round(1.4 * (1 << DIST_SHIFT)))
LONG = int(round(2.1969 * (1 << DIST_SHIFT)))
INIT_DIST0 = (2 ** 31 - 1) >> 2


def collapse_mask(m):
    """any(m>0) over channels -> bool (H,W).  reference diffuerase.py:29"""
    return np.any(m > 0, axis=2) if m.ndim == 3 else (m > 0)


def dilate_cross(b, iterations):
    """scipy.ndimage.binary_dilation(b, iterations=k) with the default 3x3 cross; k<1 => until convergence
    (reference diffuerase.py:30; SURVEY a2: k iterations == L1 distance <= k)."""
    b = b.copy()
    it = 0
    while True:
        n = b.copy()
        n[1:, :] |= b[:-1, :]
        n[:-1, :] |= b[1:, :]
        n[:, 1:] |= b[:, :-1]
        n[:, :-1] |= b[:, 1:]
        it += 1
        changed = bool((n != b).any())
        b = n
        if iterations >= 1 and it >= iterations:
            break
        if iterations < 1 and not changed:
            break
    return b


def collapse_and_dilate(mask_frames, k):
    """reference diffuerase.py:27-31 -> list of (H,W) uint8 in {0,255}."""
    return [dilate_cross(collapse_mask(m), k).astype(np.uint8) * 255 for m in mask_frames]


def distance_transform_l2_5(src):
    """cv2.distanceTransform(src, DIST_L2, 5): distance of every non-zero pixel to the nearest zero pixel,
    5x5 chamfer mask, two raster passes in 16.16 fixed point; returns float32."""
    H, W = src.shape
    B = 2
    tmp = np.full((H + 2 * B, W + 2 * B), INIT_DIST0, dtype=np.int64)
    fwd = [(-2, -1, LONG), (-2, 1, LONG), (-1, -2, LONG), (-1, -1, DIAG), (-1, 0, HV), (-1, 1, DIAG), (-1, 2, LONG), (0, -1, HV)]
    for i in range(H):
        row = tmp[i + B]
        for j in range(W):
            if src[i, j] == 0:
                row[j + B] = 0
            else:
                t0 = INIT_DIST0 * 2
                for dy, dx, c in fwd:
                    t = tmp[i + B + dy, j + B + dx] + c
                    if t < t0:
                        t0 = t
                row[j + B] = t0
    out = np.empty((H, W), np.float32)
    scale = np.float32(1.0 / (1 << DIST_SHIFT))
    for i in range(H - 1, -1, -1):
        for j in range(W - 1, -1, -1):
            t0 = tmp[i + B, j + B]
            if t0 > HV:
                for dy, dx, c in fwd:
                    t = tmp[i + B - dy, j + B - dx] + c
                    if t < t0:
                        t0 = t
                tmp[i + B, j + B] = t0
            t0 = min(int(t0), INIT_DIST0)
            out[i, j] = np.float32(np.float32(t0) * scale)
    return out


def chamfer_metric_fixed(dx, dy):
    """Closed form of the 5x5 chamfer path metric (16.16) between two pixels (|dx|,|dy| offsets)."""
    dx, dy = abs(int(dx)), abs(int(dy))
    if dx < dy:
        dx, dy = dy, dx
    if dx >= 2 * dy:
        return dy * LONG + (dx - 2 * dy) * HV
    return (dx - dy) * LONG + (2 * dy - dx) * DIAG


def _cv_round(x):
    return np.rint(x)  # round half to even == cvRound


def _resize_axis_tables(ssize, dsize):
    scale = float(ssize) / float(dsize)
    d = np.arange(dsize, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    lo = s < 0
    f[lo], s[lo] = 0.0, 0
    hi = s >= ssize - 1
    f[hi], s[hi] = 0.0, ssize - 1
    a0 = _cv_round((np.float32(1.0) - f) * np.float32(2048)).astype(np.int64)
    a1 = _cv_round(f * np.float32(2048)).astype(np.int64)
    s1 = np.minimum(s + 1, ssize - 1)
    return s, s1, a0, a1


def resize_bilinear_u8(img, W, H):
    """cv2.resize(img,(W,H)) INTER_LINEAR on uint8: horizontal pass to int (x2048), vertical pass
    ((b0*(S0>>4))>>16 + (b1*(S1>>4))>>16 + 2) >> 2."""
    if img.shape[0] == H and img.shape[1] == W:
        return img.copy()
    sx, sx1, ax0, ax1 = _resize_axis_tables(img.shape[1], W)
    sy, sy1, by0, by1 = _resize_axis_tables(img.shape[0], H)
    src = img.astype(np.int64)
    if src.ndim == 2:
        src = src[..., None]
    hrow = src[:, sx] * ax0[None, :, None] + src[:, sx1] * ax1[None, :, None]          # [Hs, W, C]
    s0, s1 = hrow[sy], hrow[sy1]
    out = (((by0[:, None, None] * (s0 >> 4)) >> 16) + ((by1[:, None, None] * (s1 >> 4)) >> 16) + 2) >> 2
    out = np.clip(out, 0, 255).astype(np.uint8)
    return out[..., 0] if img.ndim == 2 else out


def resize_nearest_u8(img, W, H):
    """cv2.resize(..., INTER_NEAREST): sx = min(floor(dx * (ssize/dsize)), ssize-1)."""
    ys = np.minimum(np.floor(np.arange(H) * (img.shape[0] / H)).astype(np.int64), img.shape[0] - 1)
    xs = np.minimum(np.floor(np.arange(W) * (img.shape[1] / W)).astype(np.int64), img.shape[1] - 1)
    return np.ascontiguousarray(img[ys][:, xs])


def feather_alpha(m2d_u8, feather_px, d_in=None, d_out=None):
    """reference diffuerase.py:86-103.  m2d_u8: (H,W) any non-zero = masked."""
    m_bin = np.where(m2d_u8 > 0, 255, 0).astype(np.uint8)
    if feather_px > 0:
        if d_in is None:
            d_in = distance_transform_l2_5(m_bin)
            d_out = distance_transform_l2_5(np.bitwise_not(m_bin))
        alpha = 0.5 + (d_in - d_out) / (2.0 * float(feather_px))
        return np.clip(alpha, 0.0, 1.0).astype(np.float32)
    return (m_bin > 0).astype(np.float32)


def composite(out_u8, orig_u8, alpha):
    """reference diffuerase.py:105-112: clip(rint(a*out + (1-a)*orig)) -> uint8 (float32 arithmetic)."""
    a3 = alpha[..., None]
    return np.clip(np.rint(a3 * out_u8 + (1.0 - a3) * orig_u8), 0, 255).astype(np.uint8)


def gaussian_kernel_21():
    """cv2.getGaussianKernel(21, 0): sigma = 0.3*((21-1)*0.5-1)+0.8 = 3.5, normalised, float32 taps."""
    sigma = 0.3 * ((21 - 1) * 0.5 - 1) + 0.8
    x = np.arange(21, dtype=np.float64) - 10
    k = np.exp(-(x * x) / (2 * sigma * sigma))
    return (k / k.sum()).astype(np.float32)


def gaussian_blur_21(m01):
    """Separable 21-tap blur of a float32 plane, BORDER_REFLECT_101, rows then columns, fp32 accumulation in tap
    order (build-defined stand-in for the third-party cv2.GaussianBlur(mask,(21,21),0))."""
    k = gaussian_kernel_21()
    H, W = m01.shape

    def refl(i, n):
        i = np.abs(i)
        i = np.where(i >= n, 2 * (n - 1) - i, i)
        return np.clip(i, 0, n - 1)   # tiny images: clamp after one reflection

    t = np.zeros((H, W), np.float32)
    xs = np.arange(W)
    for j in range(21):
        t = (t + k[j] * m01[:, refl(xs + j - 10, W)]).astype(np.float32)
    o = np.zeros((H, W), np.float32)
    ys = np.arange(H)
    for j in range(21):
        o = (o + k[j] * t[refl(ys + j - 10, H), :]).astype(np.float32)
    return o


def blur_compose(img01, orig_u8, m2d_u8):
    """m' = 1-(1-m)(1-blur21(m)); out = img*m' + orig/255*(1-m') -> uint8 (round half even).  fp32 throughout."""
    m = (m2d_u8 > 0).astype(np.float32)
    mb = gaussian_blur_21(m)
    mp = (np.float32(1) - (np.float32(1) - m) * (np.float32(1) - mb)).astype(np.float32)[..., None]
    o = orig_u8.astype(np.float32) * np.float32(1.0 / 255.0)
    v = (img01 * mp).astype(np.float32) + (o * (np.float32(1) - mp)).astype(np.float32)
    return np.clip(np.rint(v.astype(np.float32) * np.float32(255.0)), 0, 255).astype(np.uint8)
