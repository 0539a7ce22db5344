"""Real-weight source: serves the same interface as weights.SyntheticWeights from diffusers-format
state dicts / safetensors files, so the HIP model can be built from the checkpoints the reference names at
diffuerase.py:41-43,49 (SD-1.5 UNet + motion adapter, BrushNet, sd-vae-ft-mse, RAFT, and the two ProPainter networks: components "fc" =
RecurrentFlowCompleteNet, "gen" = InpaintGenerator) when they are available locally.

Internal parameter names are the diffusers names prefixed by a component ("unet.", "brushnet.", "vae.", "raft."); the only
renames are listed in `map_name`.  No network access is attempted: pass local files or in-memory dicts.
"""
import re

import torch


def map_name(name):
    """internal name -> (component, key in that component's state dict), without the .weight/.bias suffix."""
    comp, key = name.split(".", 1)
    if comp == "unet":
        # diffusers UNetMotionModel: motion_modules.N.temporal_transformer.{norm,proj_in,transformer_blocks,proj_out}
        key = re.sub(r"(motion_modules\.\d+)\.", r"\1.temporal_transformer.", key)
    elif comp == "brushnet":
        if key.startswith("conv_in"):
            key = "conv_in_condition" + key[len("conv_in"):]
    elif comp == "raft":
        key = re.sub(r"^update\.", "update_block.", key)      # princeton-vl RAFT: fnet.*, cnet.*, update_block.*
    elif comp in ("fc", "gen"):
        pass      # ProPainter's own module names (recurrent_flow_completion.pth / ProPainter.pth): flowcomplete.py and inpaintgen.py use them as they are
    elif comp != "vae":
        raise KeyError(f"unknown component in parameter name {name!r}")
    return comp, key


_VAE_LEGACY_ATTN = (("to_q", "query"), ("to_k", "key"), ("to_v", "value"), ("to_out.0", "proj_attn"))


class CheckpointWeights:
    """Drop-in for SyntheticWeights backed by real tensors.  `components`: {"unet": state_dict, "brushnet": ..., "vae": ...,
    "raft": ...}; `text_states`: the [1,77,768] CLIP encoding of the empty prompt (the reference's prompt is "")."""

    def __init__(self, components, text_states=None):
        self.components = components
        self.text_states = text_states

    @classmethod
    def from_safetensors(cls, paths, text_states=None):
        from safetensors.torch import load_file
        return cls({comp: load_file(path) for comp, path in paths.items()}, text_states)

    def _get(self, name, suffix, shape):
        comp, key = map_name(name)
        if comp not in self.components:
            raise KeyError(f"no checkpoint loaded for component {comp!r} (needed by {name})")
        full = key + suffix
        sd = self.components[comp]
        if full not in sd and comp == "vae":
            # the published sd-vae-ft-mse file predates diffusers' attention refactor: query/key/value/proj_attn, stored as
            # [C, C, 1, 1] convolutions in some exports (diffusers renames them on load; a raw safetensors read does not)
            for new, old in _VAE_LEGACY_ATTN:
                if ("." + new + ".") in ("." + full):
                    alt = full.replace(new + ".", old + ".")
                    if alt in sd:
                        full = alt
                        break
        if full not in sd:
            raise KeyError(f"checkpoint of {comp!r} has no tensor {full!r} (internal name {name}{suffix})")
        t = sd[full].to(torch.float32)
        if shape is not None and tuple(t.shape) != tuple(shape):
            n_need = 1
            for d in shape:
                n_need *= int(d)
            squeezed = tuple(d for d in t.shape if d != 1) == tuple(int(d) for d in shape if d != 1)
            if t.numel() == n_need and ((len(shape) == 4 and t.dim() == 2) or (len(shape) == 2 and t.dim() == 4 and t.shape[2:] == (1, 1))):
                t = t.reshape(shape)          # linear-projection checkpoints of 1x1 convs, and 1x1-conv exports of linear projections
            elif t.numel() == n_need and t.dim() == 5 and squeezed:
                t = t.reshape(shape)          # ProPainter Conv3d kernels: (1,k,k) -> a 2-D kernel, (3,1,1) -> the three temporal taps
            else:
                raise ValueError(f"{comp}:{full} has shape {tuple(t.shape)}, the architecture needs {tuple(shape)}")
        return t.contiguous()

    def conv(self, name, cin, cout, k, gain=1.0):
        return self._get(name, ".weight", (cout, cin, k, k)), self._get(name, ".bias", (cout,))

    def linear(self, name, cin, cout, gain=1.0, bias=True):
        return self._get(name, ".weight", (cout, cin)), (self._get(name, ".bias", (cout,)) if bias else None)

    def norm(self, name, c):
        return self._get(name, ".weight", (c,)), self._get(name, ".bias", (c,))

    def normal(self, name, shape, std=1.0, mean=0.0):
        """Only non-layer tensors go through here: the text states, and RAFT tensors requested by raw suffix."""
        if name == "text_states":
            if self.text_states is None:
                raise KeyError("CheckpointWeights needs `text_states` (CLIP encoding of the empty prompt, [1,77,768])")
            return self.text_states.to(torch.float32).reshape(shape)
        for suffix in (".weight", ".bias", ".running_mean", ".running_var"):
            if name.endswith(suffix):
                return self._get(name[: -len(suffix)], suffix, shape)
        raise KeyError(f"CheckpointWeights cannot synthesise {name!r}")


class RecordingWeights:
    """Wraps a weight source and records every tensor it serves under its CHECKPOINT name (used to export a synthetic model
    in diffusers layout: tests round-trip it through CheckpointWeights)."""

    def __init__(self, src):
        self.src, self.components = src, {}

    def _rec(self, name, suffix, t):
        if t is not None:
            comp, key = map_name(name)
            self.components.setdefault(comp, {})[key + suffix] = t.clone()
        return t

    def conv(self, name, cin, cout, k, gain=1.0):
        w, b = self.src.conv(name, cin, cout, k, gain)
        return self._rec(name, ".weight", w), self._rec(name, ".bias", b)

    def linear(self, name, cin, cout, gain=1.0, bias=True):
        w, b = self.src.linear(name, cin, cout, gain, bias)
        return self._rec(name, ".weight", w), self._rec(name, ".bias", b)

    def norm(self, name, c):
        g, b = self.src.norm(name, c)
        return self._rec(name, ".weight", g), self._rec(name, ".bias", b)

    def normal(self, name, shape, std=1.0, mean=0.0):
        t = self.src.normal(name, shape, std, mean)
        if name == "text_states":
            self.text_states = t.clone()
            return t
        for suffix in (".weight", ".bias", ".running_mean", ".running_var"):
            if name.endswith(suffix):
                return self._rec(name[: -len(suffix)], suffix, t)
        return t


# ---- LoRA merge (the reference loads the PCM "2-Step" LoRA on top of the SD-1.5 UNet: diffuerase.py:37 -> ckpt="2-Step") -------

_LORA_PAIRS = ((".lora_A.weight", ".lora_B.weight"),            # peft
               (".lora.down.weight", ".lora.up.weight"),        # diffusers attention-processor format
               (".lora_down.weight", ".lora_up.weight"))        # kohya


def _kohya_key(module_key):
    return "lora_unet_" + module_key.replace(".", "_")


def merge_lora(state_dict, lora, scale=1.0, prefix="unet."):
    """Fold a LoRA into a component state dict IN PLACE: W += scale * (alpha / rank) * up @ down for every targeted layer.

    `lora` may use the peft (`<module>.lora_A/lora_B.weight`), diffusers (`<module>.lora.down/up.weight`) or kohya
    (`lora_unet_<module with _>.lora_down/up.weight` + `.alpha`) naming; an optional leading `prefix` ("unet.") is ignored.
    Linear and k x k convolution LoRAs (down k x k, up 1 x 1) are supported.  Returns the list of merged module keys;
    raises KeyError when a LoRA tensor targets a module the state dict does not have (nothing is silently dropped)."""
    mods = {k[: -len(".weight")] for k in state_dict if k.endswith(".weight")}
    kohya = {_kohya_key(m): m for m in mods}
    pairs, alphas = {}, {}
    for key, t in lora.items():
        k = key[len(prefix):] if prefix and key.startswith(prefix) else key
        if k.endswith(".alpha"):
            alphas[k[: -len(".alpha")]] = float(t)
            continue
        for down_s, up_s in _LORA_PAIRS:
            for s, slot in ((down_s, 0), (up_s, 1)):
                if k.endswith(s):
                    pairs.setdefault(k[: -len(s)], [None, None])[slot] = t
                    break
            else:
                continue
            break
        else:
            raise KeyError(f"merge_lora: unrecognised LoRA tensor name {key!r}")
    merged = []
    for base, (down, up) in pairs.items():
        if down is None or up is None:
            raise KeyError(f"merge_lora: {base!r} has only one of its down/up matrices")
        # diffusers >= 0.22 names attention projections "to_q" in the model but "to_q_lora"/"processor.to_q_lora" in old LoRAs
        mod = base.replace(".processor", "").replace("_lora", "")
        mod = kohya.get(mod, mod)
        if mod not in mods:
            raise KeyError(f"merge_lora: LoRA targets {base!r} but the checkpoint has no module {mod!r}")
        w = state_dict[mod + ".weight"]
        rank = down.shape[0]
        alpha = alphas.get(base, float(rank))
        d32, u32 = down.to(torch.float32), up.to(torch.float32)
        if w.dim() == 4:
            if u32.dim() == 2:
                u32 = u32[:, :, None, None]
            if d32.dim() == 2:
                d32 = d32[:, :, None, None]
            delta = torch.einsum("or,rikl->oikl", u32.reshape(u32.shape[0], rank), d32)     # up is 1x1
            if delta.shape != w.shape:
                delta = delta.reshape(w.shape)
        else:
            delta = u32.reshape(u32.shape[0], rank) @ d32.reshape(rank, -1)
        state_dict[mod + ".weight"] = (w.to(torch.float32) + (scale * alpha / rank) * delta).to(w.dtype)
        merged.append(mod)
    return merged
