"""Local model store for the four checkpoints the reference names (reference diffuerase.py:41-43,49): resolves a weights directory into a
`checkpoint.CheckpointWeights` for every component of the hot path, validates every tensor name and shape against the
architecture the HIP host modules build BEFORE anything is uploaded, and encodes the reference's empty prompt with the CLIP text encoder.

No network access: `root` mirrors the hub ids as directories (what `huggingface-cli download <id> --local-dir <root>/<id>` leaves behind):

    <root>/stable-diffusion-v1-5/stable-diffusion-v1-5/text_encoder/model.safetensors, tokenizer/{vocab.json, merges.txt}
    <root>/stabilityai/sd-vae-ft-mse/diffusion_pytorch_model.safetensors
    <root>/lixiaowen/diffuEraser/brushnet/diffusion_pytorch_model.safetensors, unet_main/diffusion_pytorch_model.safetensors
    <root>/ruffy369/propainter/raft-things.pth, recurrent_flow_completion.pth, ProPainter.pth
    <root>/wangfuyun/PCM_Weights/sd15/pcm_sd15_smallcfg_2step_converted.safetensors      (the "2-Step" LoRA, merged into the UNet on load)
    <root>/text_states.safetensors          (optional: a precomputed [1,77,768] encoding of "" under the key "text_states")

[UNVERIFIED-3P]: the file names inside the third-party repositories are recalled from their public layouts; a missing file is reported with
the path that was tried, a tensor that does not fit the architecture with its name and both shapes.
"""
import math
import os

import torch

from .checkpoint import CheckpointWeights, map_name, merge_lora

SD15, VAE_ID, DE_ID, PP_ID = "stable-diffusion-v1-5/stable-diffusion-v1-5", "stabilityai/sd-vae-ft-mse", "lixiaowen/diffuEraser", "ruffy369/propainter"
PCM_LORA = {"2-Step": "wangfuyun/PCM_Weights/sd15/pcm_sd15_smallcfg_2step_converted.safetensors"}
FILES = {      # component -> path below the repository directory
    "unet": (DE_ID, "unet_main/diffusion_pytorch_model.safetensors"),
    "brushnet": (DE_ID, "brushnet/diffusion_pytorch_model.safetensors"),
    "vae": (VAE_ID, "diffusion_pytorch_model.safetensors"),
    "raft": (PP_ID, "raft-things.pth"),
    "fc": (PP_ID, "recurrent_flow_completion.pth"),
    "gen": (PP_ID, "ProPainter.pth"),
}


# ---- architecture manifest: every (parameter name, shape) the host modules ask their weight source for -----------------------------------
class _ShapeWeights:
    """Weight source that records names / shapes and hands out meta tensors (no memory, no arithmetic)."""

    def __init__(self):
        self.seen = {}

    def _t(self, name, shape):
        shape = tuple(int(d) for d in shape)
        if self.seen.setdefault(name, shape) != shape:
            raise ValueError(f"{name} requested with two shapes: {self.seen[name]} and {shape}")
        return torch.empty(shape, device="meta")

    def conv(self, name, cin, cout, k, gain=1.0):
        return self._t(name + ".weight", (cout, cin, k, k)), self._t(name + ".bias", (cout,))

    def linear(self, name, cin, cout, gain=1.0, bias=True):
        return self._t(name + ".weight", (cout, cin)), (self._t(name + ".bias", (cout,)) if bias else None)

    def norm(self, name, c):
        return self._t(name + ".weight", (c,)), self._t(name + ".bias", (c,))

    def normal(self, name, shape, std=1.0, mean=0.0):
        if name == "text_states":
            return torch.empty(tuple(shape), device="meta")
        return self._t(name, shape)


class _ShapeCtx:
    device, dt, h16 = torch.device("meta"), 0, torch.bfloat16

    def __init__(self):
        self.src = _ShapeWeights()

    def dev(self, t, dtype=None):
        return t


def manifest(ucfg=None, vcfg=None, components=("unet", "brushnet", "vae")):
    """{internal parameter name: shape} of the denoiser / VAE the HIP host modules construct for these configs (CPU, meta tensors)."""
    from .config import UNetConfig, VAEConfig
    ucfg, vcfg = ucfg or UNetConfig(), vcfg or VAEConfig()
    ctx = _ShapeCtx()
    # (the packers in packing.py and nn.Linear / nn.Conv recognise meta tensors themselves: no process-global state is patched, so a model
    #  that another thread builds or runs meanwhile -- the GUI runs jobs on worker threads -- is not affected)
    from .unet import BrushNet, UNetMotion
    from .vae import VAE
    text = torch.empty((ucfg.text_len, ucfg.cross_dim), device="meta")
    if "unet" in components:
        UNetMotion(ctx, ucfg, text)
    if "brushnet" in components:
        BrushNet(ctx, ucfg, text)
    if "vae" in components:
        VAE(ctx, vcfg)
    return dict(ctx.src.seen)


def parameter_count(man, prefix, exclude=()):
    return sum(math.prod(s) for n, s in man.items() if n.startswith(prefix) and not any(e in n for e in exclude))


# ---- reading the store --------------------------------------------------------------------------------------------------------------------
def _load_state_dict(path):
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path)
    sd = torch.load(path, map_location="cpu", weights_only=True)
    if isinstance(sd, dict) and "state_dict" in sd and isinstance(sd["state_dict"], dict):
        sd = sd["state_dict"]
    return {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}      # DataParallel prefixes of the .pth releases


def _shapes(path):
    """{key: shape} of a checkpoint file without reading the tensors (safetensors); .pth files are loaded."""
    if path.endswith(".safetensors"):
        from safetensors import safe_open
        with safe_open(path, framework="pt") as f:
            return {k: tuple(f.get_slice(k).get_shape()) for k in f.keys()}
    return {k: tuple(v.shape) for k, v in _load_state_dict(path).items()}


def component_paths(root):
    return {c: os.path.join(root, repo, rel) for c, (repo, rel) in FILES.items()}


def _compatible(have, need):
    """checkpoint.CheckpointWeights._get's reshape rules: 1x1-conv <-> linear exports and squeezed Conv3d kernels."""
    have, need = tuple(int(d) for d in have), tuple(int(d) for d in need)
    if have == need:
        return True
    if math.prod(have) != math.prod(need):
        return False
    return tuple(d for d in have if d != 1) == tuple(d for d in need if d != 1)


def validate(shapes_by_component, man):
    """Every tensor of the manifest must exist in its component's file with a compatible shape.  Returns the list of problems (empty = ok)."""
    legacy = (("to_q.", "query."), ("to_k.", "key."), ("to_v.", "value."), ("to_out.0.", "proj_attn."))
    problems = []
    for name, shape in man.items():
        base, suffix = name.rsplit(".", 1)
        comp, key = map_name(base)
        have = shapes_by_component.get(comp)
        if have is None:
            problems.append(f"{comp}: no checkpoint file (needed for {name})")
            continue
        full = key + "." + suffix
        if full not in have and comp == "vae":
            for new, old in legacy:
                if ("." + new) in ("." + full) and full.replace(new, old) in have:
                    full = full.replace(new, old)
                    break
        if full not in have:
            problems.append(f"{comp}: tensor {full!r} missing (internal name {name})")
        elif not _compatible(have[full], shape):
            problems.append(f"{comp}: {full!r} has shape {have[full]}, the architecture needs {tuple(shape)}")
    return problems


def encode_empty_prompt(root, text_len=77):
    """The reference's prompt is "" (SURVEY App. D.5): one [1,77,768] CLIP encoding, computed once on the host with the checkpoint's own
    text encoder (transformers; a one-off 77-token encode), or read from <root>/text_states.safetensors."""
    pre = os.path.join(root, "text_states.safetensors")
    if os.path.isfile(pre):
        from safetensors.torch import load_file
        return load_file(pre)["text_states"].to(torch.float32)
    te, tk = os.path.join(root, SD15, "text_encoder"), os.path.join(root, SD15, "tokenizer")
    if not (os.path.isdir(te) and os.path.isdir(tk)):
        raise FileNotFoundError(f"no CLIP text encoder / tokenizer under {os.path.join(root, SD15)} and no {pre}: cannot encode the empty prompt")
    from transformers import CLIPTextModel, CLIPTokenizer
    tok = CLIPTokenizer.from_pretrained(tk, local_files_only=True)
    enc = CLIPTextModel.from_pretrained(te, local_files_only=True).eval()
    ids = tok("", padding="max_length", max_length=text_len, truncation=True, return_tensors="pt").input_ids
    with torch.no_grad():
        return enc(ids)[0].to(torch.float32)


def load(root, ckpt="2-Step", ucfg=None, vcfg=None, want=("unet", "brushnet", "vae", "raft", "fc", "gen"), text_states=None, check=True):
    """Resolve `root` into (CheckpointWeights, {"flow_completion": bool, "generator": bool}).  Denoiser / VAE files are mandatory and are
    checked tensor by tensor against the architecture manifest first (`check`); the ProPainter files are optional: whatever is present
    switches the corresponding learned stage of the prior on."""
    root = os.path.abspath(root)
    if not os.path.isdir(root):
        raise FileNotFoundError(f"weights directory {root} does not exist")
    paths = component_paths(root)
    need = [c for c in ("unet", "brushnet", "vae") if c in want]
    missing = [f"{c}: {paths[c]}" for c in need if not os.path.isfile(paths[c])]
    if missing:
        raise FileNotFoundError("weights directory is incomplete:\n  " + "\n  ".join(missing))
    if check:
        problems = validate({c: _shapes(paths[c]) for c in need}, manifest(ucfg, vcfg, need))
        if problems:
            raise ValueError(f"{len(problems)} tensors of the checkpoints under {root} do not fit the architecture:\n  " + "\n  ".join(problems[:20]))
    comps = {c: _load_state_dict(paths[c]) for c in want if os.path.isfile(paths[c])}
    lora = PCM_LORA.get(ckpt)
    if lora and "unet" in comps:
        lp = os.path.join(root, lora)
        if not os.path.isfile(lp):
            raise FileNotFoundError(f'ckpt="{ckpt}" needs the PCM LoRA at {lp} (reference diffuerase.py:37 forces "2-Step")')
        merge_lora(comps["unet"], _load_state_dict(lp))
    if text_states is None:
        text_states = encode_empty_prompt(root)
    stages = {"flow_completion": "fc" in comps, "generator": "gen" in comps and "fc" in comps}
    return CheckpointWeights(comps, text_states), stages
