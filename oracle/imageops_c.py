"""ctypes wrapper of oracle/imageops_ref.c (ORACLE: test infrastructure only)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(_HERE, "_build", "libimageops_ref.so")
        if not os.path.isfile(so):
            subprocess.check_call(["make", "-s", "-C", _HERE])
        _lib = C.CDLL(so)
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def dilate_cross(mask2d_u8, iterations):
    m = np.ascontiguousarray(np.where(mask2d_u8 > 0, 255, 0).astype(np.uint8))
    tmp = np.empty_like(m)
    lib().dilate_cross_u8(_p(m), m.shape[0], m.shape[1], int(iterations), _p(tmp))
    return m


def distance_transform_l2_5(src_u8):
    src = np.ascontiguousarray(src_u8)
    out = np.empty(src.shape, np.float32)
    lib().distance_transform_l2_5(_p(src), src.shape[0], src.shape[1], _p(out))
    return out


def feather_composite(inp, orig, mask2d, feather):
    inp, orig, mask2d = (np.ascontiguousarray(a) for a in (inp, orig, mask2d))
    out = np.empty_like(inp)
    lib().feather_composite_u8(_p(inp), _p(orig), _p(mask2d), inp.shape[0], inp.shape[1], C.c_float(feather), _p(out))
    return out


def resize_bilinear_u8(img, W, H):
    img = np.ascontiguousarray(img)
    ch = 1 if img.ndim == 2 else img.shape[2]
    out = np.empty((H, W) if img.ndim == 2 else (H, W, ch), np.uint8)
    lib().resize_bilinear_u8(_p(img), img.shape[0], img.shape[1], ch, _p(out), H, W)
    return out
