"""Weight sources for the DiffuEraser hot path.

The reference constructs its models from HF hub ids (reference diffuerase.py:39-45,49); there is no network on
the build/bench machines, so BASELINE.json's configs use *random-init weights of the same architecture*.
`SyntheticWeights` derives every tensor from (seed, parameter-name) so that any consumer -- the CPU oracle, the HIP
host modules, any rank of a multi-GPU job -- materialises bit-identical fp32 weights lazily, one tensor at a time
(the full model is ~2.4 B parameters; nobody has to hold all of it in host memory at once).

Initialisation is variance preserving: W ~ N(0, gain^2 / fan_in), small biases, norm gains near 1.  BrushNet's
"zero convs" get a small non-zero gain so the branch is exercised.
"""
import zlib

import torch


class SyntheticWeights:
    def __init__(self, seed: int = 0):
        self.seed = int(seed)

    def _gen(self, name: str) -> torch.Generator:
        g = torch.Generator(device="cpu")
        g.manual_seed((zlib.crc32(name.encode()) ^ (self.seed * 0x9E3779B1)) & 0x7FFFFFFF)
        return g

    def normal(self, name: str, shape, std: float = 1.0, mean: float = 0.0) -> torch.Tensor:
        t = torch.randn(tuple(shape), generator=self._gen(name), dtype=torch.float32)
        return t * std + mean

    # ---- layer-level helpers (names follow the diffusers state-dict convention) ----
    def conv(self, name, cin, cout, k, gain=1.0):
        """Conv2d weight [cout, cin, k, k] + bias [cout]."""
        w = self.normal(name + ".weight", (cout, cin, k, k), std=gain / float(cin * k * k) ** 0.5)
        b = self.normal(name + ".bias", (cout,), std=0.02)
        return w, b

    def linear(self, name, cin, cout, gain=1.0, bias=True):
        w = self.normal(name + ".weight", (cout, cin), std=gain / float(cin) ** 0.5)
        b = self.normal(name + ".bias", (cout,), std=0.02) if bias else None
        return w, b

    def norm(self, name, c):
        g = self.normal(name + ".weight", (c,), std=0.1, mean=1.0)
        b = self.normal(name + ".bias", (c,), std=0.1)
        return g, b
