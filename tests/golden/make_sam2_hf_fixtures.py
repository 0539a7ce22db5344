#!/usr/bin/env python3
"""Generate golden vectors for the SAM 2 oracle from an INDEPENDENT published implementation: `transformers.models.sam2_video`
(Hugging Face transformers, build container only -- nothing of `transformers` travels to the GPU box).

The reference's masking step instantiates the `sam2` package (reference sam2_masker.py:13,84-150), which is absent from /root/reference.
`oracle/sam2_ref.py` restates that package; this script pins the restatement against the second public implementation of the same network:

  1. a small, structurally complete configuration (PIN_SAM2 below: windowed / global / q-pooling Hiera blocks, top-down FPN level, RoPE memory
     attention with object-pointer tokens, two-way decoder with high-resolution features, dynamic multimask) is instantiated as
     `Sam2VideoModel`;
  2. its parameters are OVERWRITTEN with the name-seeded synthetic weights the oracle and the HIP model use (`Sam2Weights(PIN_SAM2, SEED)`),
     through an explicit name map (`hf_name`): the map must cover the HF state dict exactly, so it also pins the parameter manifest
     (names, shapes, the [4, D] point-embedding split, the conv-transpose layouts) against the HF module tree;
  3. the HF modules are run on seeded inputs and inputs + outputs are stored in `sam2_hf_vectors.npz` (no weights, no source text).

tests/test_sam2_cpu.py::test_oracle_matches_transformers_sam2_vectors re-runs `oracle/sam2_ref.py` on the stored inputs (fp32, <= 2e-5).

    python tests/golden/make_sam2_hf_fixtures.py
"""
import os
import re
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sam2_hf_vectors.npz")

from videovanish_amd.sam2_config import Sam2Config  # noqa: E402
from videovanish_amd.sam2_weights import Sam2Weights, manifest  # noqa: E402

SEED = 7
# mem_dim must be 64: transformers hard-codes kv_in_dim = 64 in Sam2VideoMemoryAttentionLayer
PIN_SAM2 = Sam2Config(image_size=128, embed_dim=32, num_heads=1, stages=(1, 2, 3, 1), global_att_blocks=(4,),
                      window_pos_embed_bkg_spatial_size=(3, 3), window_spec=(8, 4, 4, 2), d_model=128, mem_dim=64, mem_attn_layers=2,
                      mem_attn_ff=256, dec_heads=2, dec_mlp=256, mask_in_chans=8)


def hf_config(cfg):
    from transformers.models.sam2.configuration_sam2 import Sam2HieraDetConfig, Sam2VisionConfig
    from transformers.models.sam2_video.configuration_sam2_video import (Sam2VideoConfig, Sam2VideoMaskDecoderConfig,
                                                                         Sam2VideoPromptEncoderConfig)
    S, D = cfg.image_size, cfg.d_model
    dims = list(cfg.stage_dims)
    heads = [cfg.num_heads * 2 ** i for i in range(len(cfg.stages))]
    bb = Sam2HieraDetConfig(hidden_size=cfg.embed_dim, num_attention_heads=cfg.num_heads, image_size=[S, S],
                            window_positional_embedding_background_size=list(cfg.window_pos_embed_bkg_spatial_size),
                            blocks_per_stage=list(cfg.stages), embed_dim_per_stage=dims, num_attention_heads_per_stage=heads,
                            window_size_per_stage=list(cfg.window_spec), global_attention_blocks=list(cfg.global_att_blocks),
                            num_query_pool_stages=cfg.q_pool)
    vc = Sam2VisionConfig(backbone_config=bb, backbone_channel_list=dims[::-1], backbone_feature_sizes=[[S // 4, S // 4], [S // 8, S // 8], [S // 16, S // 16]],
                          fpn_hidden_size=D, fpn_top_down_levels=list(cfg.fpn_top_down_levels), num_feature_levels=3)
    pe = Sam2VideoPromptEncoderConfig(hidden_size=D, image_size=S, patch_size=16, mask_input_channels=cfg.mask_in_chans, num_point_embeddings=4)
    md = Sam2VideoMaskDecoderConfig(hidden_size=D, mlp_dim=cfg.dec_mlp, num_hidden_layers=cfg.dec_depth, num_attention_heads=cfg.dec_heads,
                                    attention_downsample_rate=cfg.dec_downsample, num_multimask_outputs=cfg.num_multimask, iou_head_depth=3,
                                    iou_head_hidden_dim=D, dynamic_multimask_via_stability=True,
                                    dynamic_multimask_stability_delta=cfg.stability_delta, dynamic_multimask_stability_thresh=cfg.stability_thresh)
    fs = cfg.feat_size
    return Sam2VideoConfig(vision_config=vc, prompt_encoder_config=pe, mask_decoder_config=md, image_size=S, num_maskmem=cfg.num_maskmem,
                           sigmoid_scale_for_mem_enc=cfg.sigmoid_scale_for_mem_enc, sigmoid_bias_for_mem_enc=cfg.sigmoid_bias_for_mem_enc,
                           multimask_min_pt_num=cfg.multimask_min_pt_num, multimask_max_pt_num=cfg.multimask_max_pt_num,
                           max_object_pointers_in_encoder=cfg.max_obj_ptrs_in_encoder, memory_attention_hidden_size=D,
                           memory_attention_num_layers=cfg.mem_attn_layers, memory_attention_feed_forward_hidden_size=cfg.mem_attn_ff,
                           memory_attention_rope_theta=int(cfg.rope_theta), memory_attention_rope_feat_sizes=[fs, fs],
                           memory_encoder_hidden_size=D, memory_encoder_output_channels=cfg.mem_dim, mask_downsampler_embed_dim=D,
                           memory_fuser_num_layers=cfg.fuser_layers, memory_fuser_embed_dim=D, memory_fuser_intermediate_dim=4 * D)


_MLP3 = {"0": "proj_in", "1": "layers.0", "2": "proj_out"}
_RULES = [
    (r"^image_encoder\.trunk\.patch_embed\.proj\.", "vision_encoder.backbone.patch_embed.projection."),
    (r"^image_encoder\.trunk\.(pos_embed(?:_window)?)$", r"vision_encoder.backbone.\1"),
    (r"^image_encoder\.trunk\.blocks\.(\d+)\.norm([12])\.", r"vision_encoder.backbone.blocks.\1.layer_norm\2."),
    (r"^image_encoder\.trunk\.blocks\.(\d+)\.mlp\.layers\.0\.", r"vision_encoder.backbone.blocks.\1.mlp.proj_in."),
    (r"^image_encoder\.trunk\.blocks\.(\d+)\.mlp\.layers\.1\.", r"vision_encoder.backbone.blocks.\1.mlp.proj_out."),
    (r"^image_encoder\.trunk\.blocks\.", "vision_encoder.backbone.blocks."),
    (r"^image_encoder\.neck\.convs\.(\d+)\.conv\.", r"vision_encoder.neck.convs.\1."),
    (r"^memory_attention\.layers\.(\d+)\.(self_attn|cross_attn_image)\.out_proj\.", r"memory_attention.layers.\1.\2.o_proj."),
    (r"^memory_attention\.layers\.(\d+)\.norm(\d)\.", r"memory_attention.layers.\1.layer_norm\2."),
    (r"^memory_attention\.norm\.", "memory_attention.layer_norm."),
    (r"^memory_encoder\.mask_downsampler\.encoder\.12\.", "memory_encoder.mask_downsampler.final_conv."),
    (r"^memory_encoder\.mask_downsampler\.encoder\.(0|3|6|9)\.", lambda m: f"memory_encoder.mask_downsampler.layers.{int(m.group(1)) // 3}.conv."),
    (r"^memory_encoder\.mask_downsampler\.encoder\.(1|4|7|10)\.", lambda m: f"memory_encoder.mask_downsampler.layers.{int(m.group(1)) // 3}.layer_norm."),
    (r"^memory_encoder\.pix_feat_proj\.", "memory_encoder.feature_projection."),
    (r"^memory_encoder\.fuser\.layers\.(\d+)\.dwconv\.", r"memory_encoder.memory_fuser.layers.\1.depthwise_conv."),
    (r"^memory_encoder\.fuser\.layers\.(\d+)\.norm\.", r"memory_encoder.memory_fuser.layers.\1.layer_norm."),
    (r"^memory_encoder\.fuser\.layers\.(\d+)\.pwconv([12])\.", r"memory_encoder.memory_fuser.layers.\1.pointwise_conv\2."),
    (r"^memory_encoder\.fuser\.layers\.(\d+)\.gamma$", r"memory_encoder.memory_fuser.layers.\1.scale"),
    (r"^memory_encoder\.out_proj\.", "memory_encoder.projection."),
    (r"^sam_prompt_encoder\.not_a_point_embed\.", "prompt_encoder.not_a_point_embed."),
    (r"^sam_prompt_encoder\.no_mask_embed\.", "prompt_encoder.no_mask_embed."),
    (r"^sam_prompt_encoder\.mask_downscaling\.0\.", "prompt_encoder.mask_embed.conv1."),
    (r"^sam_prompt_encoder\.mask_downscaling\.1\.", "prompt_encoder.mask_embed.layer_norm1."),
    (r"^sam_prompt_encoder\.mask_downscaling\.3\.", "prompt_encoder.mask_embed.conv2."),
    (r"^sam_prompt_encoder\.mask_downscaling\.4\.", "prompt_encoder.mask_embed.layer_norm2."),
    (r"^sam_prompt_encoder\.mask_downscaling\.6\.", "prompt_encoder.mask_embed.conv3."),
    (r"^sam_mask_decoder\.transformer\.layers\.(\d+)\.(self_attn|cross_attn_token_to_image|cross_attn_image_to_token)\.out_proj\.",
     r"mask_decoder.transformer.layers.\1.\2.o_proj."),
    (r"^sam_mask_decoder\.transformer\.layers\.(\d+)\.mlp\.layers\.0\.", r"mask_decoder.transformer.layers.\1.mlp.proj_in."),
    (r"^sam_mask_decoder\.transformer\.layers\.(\d+)\.mlp\.layers\.1\.", r"mask_decoder.transformer.layers.\1.mlp.proj_out."),
    (r"^sam_mask_decoder\.transformer\.layers\.(\d+)\.norm(\d)\.", r"mask_decoder.transformer.layers.\1.layer_norm\2."),
    (r"^sam_mask_decoder\.transformer\.final_attn_token_to_image\.out_proj\.", "mask_decoder.transformer.final_attn_token_to_image.o_proj."),
    (r"^sam_mask_decoder\.transformer\.norm_final_attn\.", "mask_decoder.transformer.layer_norm_final_attn."),
    (r"^sam_mask_decoder\.output_upscaling\.0\.", "mask_decoder.upscale_conv1."),
    (r"^sam_mask_decoder\.output_upscaling\.1\.", "mask_decoder.upscale_layer_norm."),
    (r"^sam_mask_decoder\.output_upscaling\.3\.", "mask_decoder.upscale_conv2."),
    (r"^sam_mask_decoder\.output_hypernetworks_mlps\.(\d+)\.layers\.(\d)\.", lambda m: f"mask_decoder.output_hypernetworks_mlps.{m.group(1)}.{_MLP3[m.group(2)]}."),
    (r"^sam_mask_decoder\.(iou_prediction_head|pred_obj_score_head)\.layers\.(\d)\.", lambda m: f"mask_decoder.{m.group(1)}.{_MLP3[m.group(2)]}."),
    (r"^sam_mask_decoder\.", "mask_decoder."),
    (r"^maskmem_tpos_enc$", "memory_temporal_positional_encoding"),
    (r"^no_mem_embed$", "no_memory_embedding"),
    (r"^no_mem_pos_enc$", "no_memory_positional_encoding"),
    (r"^no_obj_ptr$", "no_object_pointer"),
    (r"^no_obj_embed_spatial$", "occlusion_spatial_embedding_parameter"),
    (r"^obj_ptr_proj\.layers\.(\d)\.", lambda m: f"object_pointer_proj.{_MLP3[m.group(1)]}."),
    (r"^obj_ptr_tpos_proj\.", "temporal_positional_encoding_projection_layer."),
]


def hf_name(name):
    """published (`sam2.1_hiera_large.pt`) tensor name -> transformers `Sam2VideoModel` tensor name (None: handled separately)."""
    if name.startswith("sam_prompt_encoder.point_embeddings.") or name == "sam_prompt_encoder.pe_layer.positional_encoding_gaussian_matrix":
        return None
    for pat, rep in _RULES:
        new, n = re.subn(pat, rep, name)
        if n:
            return new
    return name          # memory_attention q/k/v projections, linear1/2, mask_downsample: same name on both sides


def load_into_hf(model, W, cfg):
    sd = model.state_dict()
    new, used = {}, set()
    for name in manifest(cfg):
        hn = hf_name(name)
        if hn is None:
            continue
        assert hn in sd, f"{name} -> {hn}: no such tensor in the transformers model"
        t = W.get(name)
        assert tuple(t.shape) == tuple(sd[hn].shape), f"{name} {tuple(t.shape)} vs {hn} {tuple(sd[hn].shape)}"
        new[hn] = t
        used.add(hn)
    new["prompt_encoder.point_embed.weight"] = torch.cat([W.get(f"sam_prompt_encoder.point_embeddings.{i}.weight") for i in range(4)], 0)
    g = W.get("sam_prompt_encoder.pe_layer.positional_encoding_gaussian_matrix")
    new["prompt_encoder.shared_embedding.positional_embedding"] = g
    new["shared_image_embedding.positional_embedding"] = g
    missing = sorted(set(sd) - set(new))
    assert not missing, f"transformers tensors the published manifest does not provide: {missing}"
    model.load_state_dict(new, strict=True)
    return len(new)


def main():
    from transformers.models.sam2_video.modeling_sam2_video import Sam2VideoModel
    cfg = PIN_SAM2
    torch.manual_seed(0)
    hf = Sam2VideoModel(hf_config(cfg)).eval().float()
    W = Sam2Weights(cfg, SEED)
    n = load_into_hf(hf, W, cfg)
    S, D, fs, Mm = cfg.image_size, cfg.d_model, cfg.feat_size, cfg.mem_dim
    rng = np.random.default_rng(SEED)
    out = {}
    with torch.no_grad():
        # ---- 1. image encoder: Hiera trunk + FPN neck + conv_s0 / conv_s1 on a normalised image (the oracle's preprocess is its own arithmetic)
        frame = rng.integers(0, 256, (S, S, 3), dtype=np.uint8)
        mean, std = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1), torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)
        pix = ((torch.from_numpy(frame).float().permute(2, 0, 1) / 255.0 - mean) / std)[None]
        vo = hf.get_image_features(pix, return_dict=True)
        sizes = [(S // 4, S // 4), (S // 8, S // 8), (fs, fs)]
        fpn = [f.permute(1, 2, 0).reshape(1, -1, *sz) for f, sz in zip(vo.fpn_hidden_states, sizes)]
        pos_top = vo.fpn_position_encoding[2].permute(1, 2, 0).reshape(1, -1, fs, fs)
        out.update(frame=frame, enc_s0=fpn[0].numpy(), enc_s1=fpn[1].numpy(), enc_top=fpn[2].numpy(), enc_pos=pos_top.numpy())
        # ---- 2. SAM heads on a conditioning frame (no memory): one click (multimask), box as two corner points + a negative click, and a
        #         click together with the previous low-resolution logits as a mask prompt
        emb = [fpn[0], fpn[1], fpn[2] + hf.no_memory_embedding.view(1, -1, 1, 1)]
        cases = {"click": (np.float32([[40.0, 70.0]]), np.int32([1]), None, True),
                 "box": (np.float32([[20.0, 30.0], [100.0, 90.0], [64.5, 12.25]]), np.int32([2, 3, 0]), None, False)}
        r = hf._single_frame_forward(input_points=torch.tensor(cases["click"][0])[None, None], input_labels=torch.tensor(cases["click"][1])[None, None],
                                     image_embeddings=emb, multimask_output=True)
        prev = torch.clamp(r.pred_masks, -32.0, 32.0)
        cases["reprompt"] = (np.float32([[40.0, 70.0], [90.0, 20.0]]), np.int32([1, 0]), prev.numpy(), False)
        for k, (pts, lab, mask, multi) in cases.items():
            r = hf._single_frame_forward(input_points=torch.tensor(pts)[None, None], input_labels=torch.tensor(lab)[None, None],
                                         input_masks=None if mask is None else torch.tensor(mask), image_embeddings=emb, multimask_output=multi)
            out.update({f"sam_{k}_points": pts, f"sam_{k}_labels": lab, f"sam_{k}_multimask": np.int32(multi), f"sam_{k}_masks": r.pred_masks.numpy(),
                        f"sam_{k}_ptr": r.object_pointer.numpy(), f"sam_{k}_obj": r.object_score_logits.numpy(), f"sam_{k}_iou": r.iou_scores.numpy()})
            if mask is not None:
                out[f"sam_{k}_mask_in"] = mask
        # ---- 2b. a caller-supplied MASK as the frame's output (SAM2Base._use_mask_as_output): logits -10 / +10, antialiased low-resolution copy, the
        #          object pointer from the SAM heads prompted with mask_downsample(mask) on the RAW top-level features; an empty mask: "no object"
        yy, xx = np.mgrid[0:S, 0:S]
        blob = (((yy - 0.42 * S) ** 2 / (0.20 * S) ** 2 + (xx - 0.55 * S) ** 2 / (0.31 * S) ** 2) <= 1.0).astype(np.float32)
        blob[int(0.40 * S):int(0.46 * S), int(0.50 * S):int(0.58 * S)] = 0.0                    # a hole
        for tag, mk in (("blob", blob), ("empty", np.zeros((S, S), np.float32))):
            mt = torch.tensor(mk)[None, None]
            r = hf._use_mask_as_output(backbone_features=fpn[2], high_res_features=[fpn[0], fpn[1]], mask_inputs=mt)
            # the pointer: transformers calls its SAM heads here with their default multimask_output=True (the best-IoU token); the `sam2` package the
            # reference imports calls _forward_sam_heads with ITS default multimask_output=False (the single-mask token).  The vectors follow the
            # package: the same transformers heads with the flag spelled out, then the published mixing rule of _use_mask_as_output.
            p1 = hf._single_frame_forward(input_masks=hf.mask_downsample(mt), image_embeddings=[fpn[0], fpn[1], fpn[2]], multimask_output=False).object_pointer
            lam = float(mk.max() > 0)
            ptr = lam * p1 + (1.0 - lam) * hf.no_object_pointer
            out.update({f"mask_{tag}_in": mk, f"mask_{tag}_low": r.pred_masks.numpy(), f"mask_{tag}_ptr": ptr.numpy(),
                        f"mask_{tag}_ptr_transformers_default": r.object_pointer.numpy(), f"mask_{tag}_obj": r.object_score_logits.numpy()})
        # ---- 3. memory encoder, fp32 (the HF wrapper `_encode_new_memory` then rounds the features to bfloat16: stored too, looser check)
        low = torch.tensor(out["sam_click_masks"]).reshape(1, 1, 4 * fs, 4 * fs)
        high = torch.nn.functional.interpolate(low, size=(S, S), mode="bilinear", align_corners=False)
        for tag, from_pts in (("pts", True), ("trk", False)):
            m = (high > 0).float() if from_pts else torch.sigmoid(high)
            m = m * cfg.sigmoid_scale_for_mem_enc + cfg.sigmoid_bias_for_mem_enc
            f32, pe = hf.memory_encoder(fpn[2], m)
            obj = torch.tensor([[3.0 if from_pts else -2.0]])            # second case: occluded object -> no_obj_embed_spatial is added
            f16, _ = hf._encode_new_memory(vo.fpn_hidden_states[2], high, obj, from_pts)
            out.update({f"mem_{tag}_feat": f32.numpy(), f"mem_{tag}_pos": pe.numpy(), f"mem_{tag}_obj": obj.numpy(),
                        f"mem_{tag}_wrapped_bf16": f16.float().permute(1, 2, 0).reshape(1, Mm, fs, fs).numpy()})
        out["mem_low_res_in"] = low.numpy()
        # ---- 4. a tracked frame: memory selection + temporal encodings + object pointers + RoPE memory attention.  Frame 9 of 12, conditioning
        #         frame 0, frames 1..8 tracked (random memories: this stage only mixes them)
        T, cur = 12, 9
        g = torch.Generator().manual_seed(SEED)
        bank = {t: {"maskmem_features": torch.randn(1, Mm, fs, fs, generator=g) * 0.5, "maskmem_pos_enc": torch.randn(1, Mm, fs, fs, generator=g) * 0.5,
                    "obj_ptr": torch.randn(1, D, generator=g) * 0.5} for t in range(cur)}
        flat = lambda x: x.flatten(2).permute(2, 0, 1).contiguous()

        def hf_entry(o):
            return {"maskmem_features": flat(o["maskmem_features"]), "maskmem_pos_enc": flat(o["maskmem_pos_enc"]), "object_pointer": o["obj_ptr"]}

        session = types.SimpleNamespace(dtype=torch.float32, output_dict_per_obj={0: {
            "cond_frame_outputs": {0: hf_entry(bank[0])}, "non_cond_frame_outputs": {t: hf_entry(bank[t]) for t in range(1, cur)}}})
        cond = hf._prepare_memory_conditioned_features(inference_session=session, frame_idx=cur, obj_idx=0, is_initial_conditioning_frame=False,
                                                        current_vision_features=vo.fpn_hidden_states[2], current_vision_positional_embeddings=vo.fpn_position_encoding[2],
                                                        num_total_frames=T, track_in_reverse_time=False, streaming=False)
        out.update(trk_frame_idx=np.int32(cur), trk_num_frames=np.int32(T), trk_out=cond.numpy(),
                   trk_mem=np.stack([bank[t]["maskmem_features"].numpy() for t in range(cur)]),
                   trk_mem_pos=np.stack([bank[t]["maskmem_pos_enc"].numpy() for t in range(cur)]),
                   trk_ptr=np.stack([bank[t]["obj_ptr"].numpy() for t in range(cur)]))
    out["seed"] = np.int32(SEED)
    np.savez_compressed(OUT, **out)
    print(f"{n} tensors loaded into transformers.Sam2VideoModel; wrote {OUT} ({os.path.getsize(OUT) / 1e6:.2f} MB, {len(out)} arrays)")


if __name__ == "__main__":
    main()
