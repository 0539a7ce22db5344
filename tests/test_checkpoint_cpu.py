"""CPU tests of the real-weight source: name mapping to the diffusers / RAFT checkpoint layouts and the
CheckpointWeights <-> SyntheticWeights round trip (the GPU round trip is in tests/test_model_gpu.py)."""
import pytest
import torch

from videovanish_amd.checkpoint import CheckpointWeights, RecordingWeights, map_name
from videovanish_amd.weights import SyntheticWeights


def test_name_mapping():
    assert map_name("unet.down_blocks.0.resnets.1.conv1") == ("unet", "down_blocks.0.resnets.1.conv1")
    assert map_name("unet.down_blocks.2.motion_modules.1.transformer_blocks.0.attn1.to_q") == \
        ("unet", "down_blocks.2.motion_modules.1.temporal_transformer.transformer_blocks.0.attn1.to_q")
    assert map_name("unet.mid_block.motion_modules.0.proj_in") == ("unet", "mid_block.motion_modules.0.temporal_transformer.proj_in")
    assert map_name("brushnet.conv_in") == ("brushnet", "conv_in_condition")
    assert map_name("brushnet.brushnet_up_blocks.14") == ("brushnet", "brushnet_up_blocks.14")
    assert map_name("vae.decoder.mid_block.attentions.0.to_q") == ("vae", "decoder.mid_block.attentions.0.to_q")
    assert map_name("raft.update.gru.convz1") == ("raft", "update_block.gru.convz1")
    assert map_name("raft.cnet.layer2.0.downsample.0") == ("raft", "cnet.layer2.0.downsample.0")
    with pytest.raises(KeyError):
        map_name("clip.text_model.x")


def test_round_trip_and_errors(tmp_path):
    rec = RecordingWeights(SyntheticWeights(3))
    w, b = rec.conv("unet.down_blocks.0.resnets.0.conv1", 8, 16, 3)
    lw, lb = rec.linear("unet.down_blocks.0.motion_modules.0.proj_in", 16, 16)
    g, be = rec.norm("vae.encoder.conv_norm_out", 16)
    rm = rec.normal("raft.cnet.norm1.running_mean", (4,), 0.1)
    ts = rec.normal("text_states", (1, 7, 8))
    assert "down_blocks.0.motion_modules.0.temporal_transformer.proj_in.weight" in rec.components["unet"]
    from safetensors.torch import save_file
    paths = {}
    for comp, sd in rec.components.items():
        paths[comp] = str(tmp_path / f"{comp}.safetensors")
        save_file(sd, paths[comp])
    ck = CheckpointWeights.from_safetensors(paths, text_states=rec.text_states)
    w2, b2 = ck.conv("unet.down_blocks.0.resnets.0.conv1", 8, 16, 3)
    assert torch.equal(w, w2) and torch.equal(b, b2)
    lw2, lb2 = ck.linear("unet.down_blocks.0.motion_modules.0.proj_in", 16, 16)
    assert torch.equal(lw, lw2) and torch.equal(lb, lb2)
    assert torch.equal(ck.norm("vae.encoder.conv_norm_out", 16)[0], g)
    assert torch.equal(ck.normal("raft.cnet.norm1.running_mean", (4,)), rm) and torch.equal(ck.normal("text_states", (1, 7, 8)), ts)
    with pytest.raises(ValueError):
        ck.conv("unet.down_blocks.0.resnets.0.conv1", 8, 32, 3)            # architecture / checkpoint shape mismatch
    with pytest.raises(KeyError):
        ck.conv("unet.down_blocks.0.resnets.0.conv2", 16, 16, 3)           # tensor missing from the checkpoint
    with pytest.raises(KeyError):
        ck.conv("brushnet.conv_in", 9, 16, 3)                              # component not loaded
    with pytest.raises(KeyError):
        CheckpointWeights(rec.components).normal("text_states", (1, 7, 8))  # text states must be supplied


def test_merge_lora_formats():
    """PCM-LoRA style merge: peft, diffusers and kohya naming give W + scale*alpha/rank * up@down."""
    from videovanish_amd.checkpoint import merge_lora
    g = torch.Generator().manual_seed(5)
    base = {"down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_q.weight": torch.randn(16, 16, generator=g),
            "down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_q.bias": torch.randn(16, generator=g),
            "down_blocks.0.resnets.0.conv1.weight": torch.randn(8, 4, 3, 3, generator=g),
            "mid_block.attentions.0.proj_in.weight": torch.randn(16, 16, 1, 1, generator=g)}
    q, cv, pi = (k for k in base if k.endswith(".weight"))
    dq, uq = torch.randn(4, 16, generator=g), torch.randn(16, 4, generator=g)
    dc, uc = torch.randn(2, 4, 3, 3, generator=g), torch.randn(8, 2, 1, 1, generator=g)
    dp, up = torch.randn(4, 16, generator=g), torch.randn(16, 4, generator=g)
    want_q = base[q] + 0.5 * (uq @ dq)
    want_c = base[cv] + 0.5 * (8.0 / 2) * torch.einsum("or,rikl->oikl", uc[:, :, 0, 0], dc)
    want_p = base[pi] + 0.5 * (up @ dp)[:, :, None, None]
    lora = {"unet." + q[:-7] + ".lora_A.weight": dq, "unet." + q[:-7] + ".lora_B.weight": uq,                       # peft
            "lora_unet_" + cv[:-7].replace(".", "_") + ".lora_down.weight": dc,                                      # kohya conv + alpha
            "lora_unet_" + cv[:-7].replace(".", "_") + ".lora_up.weight": uc,
            "lora_unet_" + cv[:-7].replace(".", "_") + ".alpha": torch.tensor(8.0),
            pi[:-7] + ".lora.down.weight": dp, pi[:-7] + ".lora.up.weight": up}                                      # diffusers, linear LoRA on a 1x1 conv
    sd = {k: v.clone() for k, v in base.items()}
    merged = merge_lora(sd, lora, scale=0.5)
    assert sorted(merged) == sorted([q[:-7], cv[:-7], pi[:-7]])
    assert torch.allclose(sd[q], want_q, atol=1e-6) and torch.allclose(sd[cv], want_c, atol=1e-5) and torch.allclose(sd[pi], want_p, atol=1e-6)
    assert torch.equal(sd[q[:-7] + ".bias"], base[q[:-7] + ".bias"])
    # a merged layer behaves like base + LoRA branch
    x = torch.randn(3, 16, generator=g)
    assert torch.allclose(x @ sd[q].t(), x @ base[q].t() + 0.5 * (x @ dq.t()) @ uq.t(), atol=1e-5)
    with pytest.raises(KeyError):
        merge_lora({k: v.clone() for k, v in base.items()}, {"unet.up_blocks.9.to_q.lora_A.weight": dq, "unet.up_blocks.9.to_q.lora_B.weight": uq})
    with pytest.raises(KeyError):
        merge_lora({k: v.clone() for k, v in base.items()}, {"unet." + q[:-7] + ".lora_A.weight": dq})
    with pytest.raises(KeyError):
        merge_lora({k: v.clone() for k, v in base.items()}, {"something.else": dq})


def test_vae_legacy_attention_names():
    """sd-vae-ft-mse as published uses query/key/value/proj_attn (some exports as [C,C,1,1]); vae.py asks for to_q/.../to_out.0."""
    g = torch.Generator().manual_seed(9)
    sd = {}
    for old in ("query", "key", "value", "proj_attn"):
        sd[f"decoder.mid_block.attentions.0.{old}.weight"] = torch.randn(8, 8, 1, 1, generator=g)
        sd[f"decoder.mid_block.attentions.0.{old}.bias"] = torch.randn(8, generator=g)
    ck = CheckpointWeights({"vae": sd})
    for new, old in (("to_q", "query"), ("to_k", "key"), ("to_v", "value"), ("to_out.0", "proj_attn")):
        w, b = ck.linear(f"vae.decoder.mid_block.attentions.0.{new}", 8, 8)
        assert w.shape == (8, 8) and torch.equal(w, sd[f"decoder.mid_block.attentions.0.{old}.weight"].reshape(8, 8))
        assert torch.equal(b, sd[f"decoder.mid_block.attentions.0.{old}.bias"])
    with pytest.raises(KeyError):
        ck.linear("vae.encoder.mid_block.attentions.0.to_q", 8, 8)


def test_propainter_state_dicts_load_into_the_n1_networks():
    """ProPainter's own checkpoints (recurrent_flow_completion.pth -> component "fc", ProPainter.pth -> "gen") use the module names that
    flowcomplete.py / inpaintgen.py ask for; their Conv3d kernels are 5-D ((1,k,k) spatial, (3,1,1) temporal) and are folded to the 2-D / tap
    forms on load.  Build a state dict in that layout from the oracle's parameters and read it back through CheckpointWeights."""
    from oracle.model_ref import Params
    from oracle import flowcomplete_ref as FC
    P = Params(4)
    g = torch.Generator().manual_seed(0)
    fw = torch.randn(1, 2, 2, 16, 16, generator=g)
    m = torch.zeros(1, 2, 1, 16, 16); m[..., 4:9, 5:11] = 1
    with torch.no_grad():
        FC.complete(P, fw * (1 - m), m, width=(8, 16, 32), deform_groups=4)                 # materialises every parameter the network uses
    sd = {}
    for key, val in P.cache.items():
        w, b = val
        if key.endswith("#t"):                                   # temporal taps [c, c, 3] -> Conv3d (3,1,1)
            name = key[:-2]
            sd[name[3:] + ".weight"], sd[name[3:] + ".bias"] = w[:, :, :, None, None].clone(), b.clone()
        elif ".conv1.0" in key or ".downsample.0" in key or ".mid_dilation." in key:   # Conv3d (1,k,k)
            sd[key[3:] + ".weight"], sd[key[3:] + ".bias"] = w[:, :, None].clone(), b.clone()
        else:
            sd[key[3:] + ".weight"], sd[key[3:] + ".bias"] = w.clone(), b.clone()
    assert sd["downsample.0.weight"].shape == (8, 3, 1, 5, 5) and sd["encoder1.0.conv2.0.weight"].shape == (8, 8, 3, 1, 1)
    ck = CheckpointWeights({"fc": sd})
    w2, b2 = ck.conv("fc.downsample.0", 3, 8, 5)
    assert torch.equal(w2, P.cache["fc.downsample.0"][0]) and torch.equal(b2, P.cache["fc.downsample.0"][1])
    wt = ck.normal("fc.encoder1.0.conv2.0.weight", (8, 8, 3))
    assert torch.equal(wt, P.cache["fc.encoder1.0.conv2.0#t"][0])
    wd, _ = ck.conv("fc.feat_prop_module.deform_align.backward_", 64, 32, 3)
    assert torch.equal(wd, P.cache["fc.feat_prop_module.deform_align.backward_"][0])
    with pytest.raises(ValueError):
        ck.conv("fc.downsample.0", 3, 8, 3)                      # a kernel of the wrong size is refused, not reshaped
    assert map_name("gen.transformers.transformer.3.attention.query") == ("gen", "transformers.transformer.3.attention.query")
