"""Architecture of the SAM 2.1 video predictor the reference's masking step instantiates.

Reference call site: sam2_masker.py:19-20 (`configs/sam2.1/sam2.1_hiera_l.yaml`, `sam2.1_hiera_large.pt`), :88 (`build_sam2_video_predictor`).
The `sam2` package itself is third-party and absent from /root/reference (github.com/calledit/sam2_numpy_frames, un-pinned): the values
below restate the published SAM 2.1 Hiera-L configuration and the hydra overrides `build_sam2_video_predictor` applies
(dynamic multimask via stability, binarised click masks for the memory encoder, hole filling) -- [UNVERIFIED-3P], the parameter count of the
manifest built from them (videovanish_amd/sam2_weights.py) equals the published 224.4 M (tests/test_sam2_cpu.py).
"""
from dataclasses import dataclass
from typing import Tuple


@dataclass(frozen=True)
class Sam2Config:
    image_size: int = 1024
    # ---- Hiera trunk (hieradet.py)
    embed_dim: int = 144
    num_heads: int = 2
    stages: Tuple[int, ...] = (2, 6, 36, 4)
    global_att_blocks: Tuple[int, ...] = (23, 33, 43)
    window_pos_embed_bkg_spatial_size: Tuple[int, int] = (7, 7)
    window_spec: Tuple[int, ...] = (8, 4, 16, 8)
    q_pool: int = 3
    # ---- FPN neck (image_encoder.py)
    d_model: int = 256
    fpn_top_down_levels: Tuple[int, ...] = (2, 3)
    scalp: int = 1
    # ---- memory (memory_attention.py, memory_encoder.py, sam2_base.py)
    mem_dim: int = 64
    num_maskmem: int = 7
    mem_attn_layers: int = 4
    mem_attn_ff: int = 2048
    rope_theta: float = 10000.0
    fuser_layers: int = 2
    max_obj_ptrs_in_encoder: int = 16
    sigmoid_scale_for_mem_enc: float = 20.0
    sigmoid_bias_for_mem_enc: float = -10.0
    binarize_mask_from_pts_for_mem_enc: bool = True     # build_sam2_video_predictor override
    # ---- SAM heads (prompt_encoder.py, mask_decoder.py, transformer.py)
    dec_depth: int = 2
    dec_heads: int = 8
    dec_mlp: int = 2048
    dec_downsample: int = 2
    mask_in_chans: int = 16
    num_multimask: int = 3
    stability_delta: float = 0.05                       # dynamic_multimask_via_stability (video predictor override)
    stability_thresh: float = 0.98
    multimask_min_pt_num: int = 0
    multimask_max_pt_num: int = 1
    fill_hole_area: int = 8                             # video predictor override; needs connected components (see sam2_predictor.py)

    @property
    def feat_size(self):          # side of the stride-16 feature map the memory / decoder work on
        return self.image_size // 16

    @property
    def stage_dims(self):
        return tuple(self.embed_dim * 2 ** i for i in range(len(self.stages)))


def hiera_blocks(cfg):
    """Per-block (dim, dim_out, heads, window_size, q_stride) exactly as Hiera.__init__ derives them (hieradet.py): the window size
    lags the stage change by one block, pooling blocks are the first block of stages 2..q_pool+1."""
    depth = sum(cfg.stages)
    stage_ends = [sum(cfg.stages[:i]) - 1 for i in range(1, len(cfg.stages) + 1)]
    q_pool_blocks = [x + 1 for x in stage_ends[:-1]][:cfg.q_pool]
    out, dim, heads, cur_stage = [], cfg.embed_dim, cfg.num_heads, 1
    for i in range(depth):
        dim_out = dim
        window = cfg.window_spec[cur_stage - 1]
        if i in cfg.global_att_blocks:
            window = 0
        if i - 1 in stage_ends:
            dim_out, heads, cur_stage = dim * 2, heads * 2, cur_stage + 1
        out.append(dict(dim=dim, dim_out=dim_out, heads=heads, window=window, q_stride=2 if i in q_pool_blocks else 0))
        dim = dim_out
    return out, stage_ends


# structurally complete small configurations for parity tests (every block kind appears: windowed / global / pooling blocks, the top-down
# FPN level, memory attention with RoPE + object pointers, the two-way decoder with high-resolution features)
TINY_SAM2 = Sam2Config(image_size=128, embed_dim=32, num_heads=1, stages=(1, 2, 3, 1), global_att_blocks=(4,),
                       window_pos_embed_bkg_spatial_size=(3, 3), window_spec=(8, 4, 4, 2), d_model=64, mem_dim=16, mem_attn_layers=2,
                       mem_attn_ff=128, dec_heads=2, dec_mlp=128, mask_in_chans=8)
# head dim 72 in the trunk (padded to 80 on the MFMA path), d_model 256 in the memory attention (single head of 256)
SMALL_SAM2 = Sam2Config(image_size=256, embed_dim=72, num_heads=1, stages=(1, 2, 2, 1), global_att_blocks=(4,),
                        window_pos_embed_bkg_spatial_size=(5, 5), window_spec=(8, 4, 8, 4), d_model=256, mem_dim=64, mem_attn_layers=1,
                        mem_attn_ff=512, dec_mlp=512)


# ----------------------------------------------------------------------------------------------------------------
# memory selection of SAM2Base._prepare_memory_conditioned_features (sam2_base.py): pure bookkeeping, shared by every implementation
# of the arithmetic.  Returns ([(t_pos, out)], [(signed frame distance, out)]) for one frame of one object.
# ----------------------------------------------------------------------------------------------------------------
def select_memories(cfg, frame_idx, output_dict, num_frames, track_in_reverse=False):
    cond = output_dict["cond_frame_outputs"]
    non_cond = output_dict["non_cond_frame_outputs"]
    assert len(cond) > 0, "tracking needs at least one conditioning frame"
    selected, unselected = dict(cond), {}            # max_cond_frames_in_attn = -1: all conditioning frames are used
    t_pos_and_prevs = [(0, out) for out in selected.values()]
    stride = 1                                       # memory_temporal_stride_for_eval
    for t_pos in range(1, cfg.num_maskmem):
        t_rel = cfg.num_maskmem - t_pos
        if t_rel == 1:
            prev = frame_idx + t_rel if track_in_reverse else frame_idx - t_rel
        elif not track_in_reverse:
            prev = ((frame_idx - 2) // stride) * stride - (t_rel - 2) * stride
        else:
            prev = -(-(frame_idx + 2) // stride) * stride + (t_rel - 2) * stride
        out = non_cond.get(prev, None)
        if out is None:
            out = unselected.get(prev, None)
        t_pos_and_prevs.append((t_pos, out))
    mems = [(t_pos, out) for t_pos, out in t_pos_and_prevs if out is not None]
    # object pointers: conditioning frames in the past (only_obj_ptrs_in_the_past_for_eval), then up to max-1 recent frames
    max_ptrs = min(num_frames, cfg.max_obj_ptrs_in_encoder)
    sign = -1 if track_in_reverse else 1
    ptrs = [((frame_idx - t) * sign, out) for t, out in selected.items() if (t >= frame_idx if track_in_reverse else t <= frame_idx)]
    for t_diff in range(1, max_ptrs):
        t = frame_idx + t_diff if track_in_reverse else frame_idx - t_diff
        if t < 0 or t >= num_frames:
            break
        out = non_cond.get(t, unselected.get(t, None))
        if out is not None:
            ptrs.append((t_diff, out))
    return mems, ptrs, max_ptrs
