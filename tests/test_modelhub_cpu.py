"""Local model store (reference diffuerase.py:41-43,49): videovanish_amd/modelhub.py resolves a weights directory to
the components of the hot path, checks every tensor name / shape against the architecture manifest before anything is loaded, merges the
PCM "2-Step" LoRA, encodes the empty prompt with the checkpoint's own CLIP text encoder, and `diffuerase.configure(weights=...)` /
$VV_WEIGHTS_DIR hand the result to the drop-in.  CPU only: synthetic stores shaped like the real files (tiny configs for the files that
are actually written; the full-size architecture is checked on shapes alone)."""
import os

import pytest
import torch

from videovanish_amd import modelhub
from videovanish_amd.checkpoint import map_name
from videovanish_amd.config import TINY_UNET, TINY_VAE, RunConfig

LORA_LAYER = "down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_q"


def _write_store(root, ucfg=TINY_UNET, vcfg=TINY_VAE, clip=True, prior=False, lora=True, drop=None, reshape=None):
    from safetensors.torch import save_file
    man = modelhub.manifest(ucfg, vcfg)
    g = torch.Generator().manual_seed(1)
    comps = {}
    for name, shape in man.items():
        base, suf = name.rsplit(".", 1)
        comp, key = map_name(base)
        comps.setdefault(comp, {})[key + "." + suf] = torch.randn(shape, generator=g) * 0.05
    if drop:
        del comps[drop[0]][drop[1]]
    if reshape:
        comps[reshape[0]][reshape[1]] = torch.zeros(reshape[2])
    paths = modelhub.component_paths(root)
    for comp, sd in comps.items():
        os.makedirs(os.path.dirname(paths[comp]), exist_ok=True)
        save_file(sd, paths[comp])
    if lora:
        C = comps["unet"][LORA_LAYER + ".weight"].shape[0]
        lp = os.path.join(root, modelhub.PCM_LORA["2-Step"])
        os.makedirs(os.path.dirname(lp), exist_ok=True)
        save_file({f"unet.{LORA_LAYER}.lora_A.weight": torch.ones(2, C) * 0.5, f"unet.{LORA_LAYER}.lora_B.weight": torch.ones(C, 2) * 0.25}, lp)
    if clip:
        import json
        from transformers import CLIPTextConfig, CLIPTextModel, CLIPTokenizer
        tk = os.path.join(root, modelhub.SD15, "tokenizer")
        os.makedirs(tk, exist_ok=True)
        json.dump({"<|startoftext|>": 0, "<|endoftext|>": 1, "a</w>": 2}, open(os.path.join(tk, "vocab.json"), "w"))
        open(os.path.join(tk, "merges.txt"), "w").write("#version: 0.2\n")
        CLIPTokenizer(os.path.join(tk, "vocab.json"), os.path.join(tk, "merges.txt")).save_pretrained(tk)
        cfg = CLIPTextConfig(vocab_size=3, hidden_size=ucfg.cross_dim, intermediate_size=2 * ucfg.cross_dim, num_hidden_layers=1, num_attention_heads=2,
                             max_position_embeddings=77, bos_token_id=0, eos_token_id=1, pad_token_id=1)
        torch.manual_seed(3)
        CLIPTextModel(cfg).save_pretrained(os.path.join(root, modelhub.SD15, "text_encoder"))
    if prior:
        for comp in ("raft", "fc", "gen"):
            os.makedirs(os.path.dirname(paths[comp]), exist_ok=True)
            torch.save({"module.some.layer.weight": torch.zeros(2, 2)}, paths[comp])
    return comps


def test_full_size_manifest_equals_the_published_checkpoints():
    """The manifest of the DEFAULT configs (what a real store is validated against): parameter counts of the published files."""
    man = modelhub.manifest()
    assert modelhub.parameter_count(man, "unet.", exclude=("motion_modules",)) == 859_520_964      # SD-1.5 UNet2DConditionModel
    assert modelhub.parameter_count(man, "vae.") == 83_653_863                                      # sd-vae-ft-mse
    assert len([n for n in man if n.startswith("brushnet.brushnet_") and n.endswith(".weight")]) == 12 + 1 + 15


def test_validate_names_every_mismatch():
    man = modelhub.manifest()
    shapes = {}
    for name, shape in man.items():
        base, suf = name.rsplit(".", 1)
        comp, key = map_name(base)
        shapes.setdefault(comp, {})[key + "." + suf] = tuple(shape)
    assert modelhub.validate(shapes, man) == []
    # accepted variants: the published VAE file's pre-refactor attention names, 1x1-conv exports of linear projections
    v = dict(shapes["vae"])
    for new, old in (("to_q", "query"), ("to_k", "key"), ("to_v", "value"), ("to_out.0", "proj_attn")):
        for suf in ("weight", "bias"):
            k = f"decoder.mid_block.attentions.0.{new}.{suf}"
            s = v.pop(k)
            v[f"decoder.mid_block.attentions.0.{old}.{suf}"] = s + (1, 1) if suf == "weight" else s
    assert modelhub.validate({**shapes, "vae": v}, man) == []
    # rejected: a missing tensor, a wrong shape, a missing component -- each named
    u = dict(shapes["unet"])
    del u["conv_in.weight"]
    u["conv_out.bias"] = (5,)
    probs = modelhub.validate({"unet": u, "vae": shapes["vae"]}, man)
    assert any("'conv_in.weight' missing" in p for p in probs) and any("'conv_out.bias' has shape (5,)" in p for p in probs)
    assert any(p.startswith("brushnet: no checkpoint file") for p in probs)


def test_load_resolves_a_store(tmp_path):
    root = str(tmp_path / "store")
    comps = _write_store(root, prior=True)
    w, stages = modelhub.load(root, ckpt="2-Step", ucfg=TINY_UNET, vcfg=TINY_VAE)
    assert stages == {"flow_completion": True, "generator": True}
    assert set(w.components) == {"unet", "brushnet", "vae", "raft", "fc", "gen"}
    assert "some.layer.weight" in w.components["raft"]                       # DataParallel prefix of the .pth releases stripped
    man = modelhub.manifest(TINY_UNET, TINY_VAE)
    for name, shape in man.items():                                           # every tensor the host modules will ask for is served, right shape
        base, suf = name.rsplit(".", 1)
        assert tuple(w._get(base, "." + suf, shape).shape) == tuple(shape)
    # the PCM LoRA was merged: W += (alpha / rank) * up @ down with alpha = rank
    C = comps["unet"][LORA_LAYER + ".weight"].shape[0]
    delta = (torch.ones(C, 2) * 0.25) @ (torch.ones(2, C) * 0.5)
    assert torch.allclose(w.components["unet"][LORA_LAYER + ".weight"], comps["unet"][LORA_LAYER + ".weight"] + delta, atol=1e-6)
    # the empty prompt: [1, 77, cross_dim] from the store's own text encoder, deterministic
    assert tuple(w.text_states.shape) == (1, 77, TINY_UNET.cross_dim) and torch.isfinite(w.text_states).all()
    assert torch.equal(w.text_states, modelhub.encode_empty_prompt(root))


def test_load_fails_loudly(tmp_path):
    with pytest.raises(FileNotFoundError, match="does not exist"):
        modelhub.load(str(tmp_path / "nope"))
    root = str(tmp_path / "a")
    _write_store(root, clip=False, lora=False)
    with pytest.raises(FileNotFoundError, match="PCM LoRA"):
        modelhub.load(root, ckpt="2-Step", ucfg=TINY_UNET, vcfg=TINY_VAE, text_states=torch.zeros(1, 7, 64))
    with pytest.raises(FileNotFoundError, match="cannot encode the empty prompt"):
        modelhub.load(root, ckpt="50-Step", ucfg=TINY_UNET, vcfg=TINY_VAE)
    os.remove(modelhub.component_paths(root)["vae"])
    with pytest.raises(FileNotFoundError, match="sd-vae-ft-mse"):
        modelhub.load(root, ckpt="50-Step", ucfg=TINY_UNET, vcfg=TINY_VAE, text_states=torch.zeros(1, 7, 64))
    root = str(tmp_path / "b")
    _write_store(root, clip=False, lora=False, drop=("brushnet", "brushnet_mid_block.weight"), reshape=("unet", "conv_out.bias", (9,)))
    with pytest.raises(ValueError) as e:
        modelhub.load(root, ckpt="50-Step", ucfg=TINY_UNET, vcfg=TINY_VAE, text_states=torch.zeros(1, 7, 64))
    assert "brushnet_mid_block.weight" in str(e.value) and "conv_out.bias" in str(e.value) and "(9,)" in str(e.value)
    # a full-size architecture against a tiny store: refused before any tensor is read
    root = str(tmp_path / "c")
    _write_store(root, clip=False, lora=False)
    with pytest.raises(ValueError, match="do not fit the architecture"):
        modelhub.load(root, ckpt="50-Step", text_states=torch.zeros(1, 77, 768))


def test_configure_hands_the_store_to_the_drop_in(tmp_path, monkeypatch):
    """diffuerase.configure(weights=dir) and $VV_WEIGHTS_DIR: the drop-in resolves the reference's four model ids to local files."""
    import diffuerase
    root = str(tmp_path / "store")
    _write_store(root, prior=True)
    run = RunConfig(unet=TINY_UNET, vae=TINY_VAE)
    try:
        diffuerase.configure(run=run, weights=root)
        w, stages = diffuerase._resolve_weights("2-Step")
        assert set(w.components) >= {"unet", "brushnet", "vae", "raft"} and stages == {"flow_completion": True, "generator": True}
        assert diffuerase._resolve_weights("2-Step")[0] is w                           # resolved once
        diffuerase.configure(run=run, weights=root, prior={"generator": False})        # an explicit prior= wins
        assert diffuerase._resolve_weights("2-Step")[1] == {"flow_completion": True, "generator": False}
        diffuerase.configure(run=run)
        assert diffuerase._resolve_weights("2-Step") == (None, {})                    # no store: seeded random init, learned prior stages off
        monkeypatch.setenv("VV_WEIGHTS_DIR", root)
        diffuerase.configure(run=run)
        assert set(diffuerase._resolve_weights("2-Step")[0].components) >= {"unet", "brushnet", "vae"}
    finally:
        monkeypatch.delenv("VV_WEIGHTS_DIR", raising=False)
        diffuerase.configure(None)
