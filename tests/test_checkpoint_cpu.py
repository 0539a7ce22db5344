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
