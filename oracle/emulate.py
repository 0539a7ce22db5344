"""ORACLE (test infrastructure): reduced-precision EMULATION of the HIP path inside the fp32 oracle, on CPU.

The HIP kernels round MFMA operands to bf16/fp16 (weights, the activations a GEMM reads, Q/K/V, the softmax
probabilities P) and accumulate in fp32.  `emulate(dtype, ...)` patches oracle.model_ref's primitives so that the same
roundings happen at the same places in the fp32 restatement; comparing emulated vs plain oracle output attributes the
product path's error to rounding classes / layers WITHOUT a GPU (tools/parity_emulate.py, profiles/r2_parity_*.txt), and
predicts what a split-precision (hi + lo operand) layer would buy before its kernel is written.

Classes: "w" weights, "a" GEMM input activations, "qkv" attention operands, "p" softmax probabilities.
`exact(name) -> bool | set` exempts layers by parameter name: True = the layer runs in fp32 (the 3-pass hi+lo split limit), a
set of classes = only those classes are exempt for that layer ({"a"} = activations split, weights still rounded: a 2-pass split).
Only tests/ and tools/ import this (never the product package)."""
import contextlib

import torch
import torch.nn.functional as F

from . import model_ref as M


def _r(x, dt):
    return x.to(dt).to(torch.float32)


@contextlib.contextmanager
def emulate(dtype=torch.float16, classes=("w", "a", "qkv", "p"), exact=None):
    classes = set(classes)
    exact = exact or (lambda name: False)
    orig = (M.conv2d, M.linear, M.attention)

    def conv2d(P, name, x, cout, k=3, stride=1, pad=1, gain=1.0):
        w, b = P.conv(name, x.shape[1], cout, k, gain)
        ex = exact(name)
        ex = {"w", "a"} if ex is True else (ex or ())
        if "w" in classes and "w" not in ex:
            w = _r(w, dtype)
        if "a" in classes and "a" not in ex:
            x = _r(x, dtype)
        return F.conv2d(x, w, b, stride=stride, padding=pad)

    def linear(P, name, x, cout, bias=True, gain=1.0):
        w, b = P.linear(name, x.shape[-1], cout, gain, bias)
        ex = exact(name)
        ex = {"w", "a"} if ex is True else (ex or ())
        if "w" in classes and "w" not in ex:
            w = _r(w, dtype)
        if "a" in classes and "a" not in ex:
            x = _r(x, dtype)
        return F.linear(x, w, b)

    def attention(q, k, v, heads):
        B, Nq, C = q.shape
        d = C // heads
        if "qkv" in classes:
            q, k, v = _r(q, dtype), _r(k, dtype), _r(v, dtype)
        q = q.view(B, Nq, heads, d).transpose(1, 2)
        k = k.view(B, -1, heads, d).transpose(1, 2)
        v = v.view(B, -1, heads, d).transpose(1, 2)
        s = (q @ k.transpose(-1, -2)) * (d ** -0.5)
        p = torch.exp(s - s.amax(-1, keepdim=True))
        if "p" in classes:
            p = _r(p, dtype)
        o = (p @ v) / p.sum(-1, keepdim=True)
        return o.transpose(1, 2).reshape(B, Nq, C)

    M.conv2d, M.linear, M.attention = conv2d, linear, attention
    try:
        yield
    finally:
        M.conv2d, M.linear, M.attention = orig
