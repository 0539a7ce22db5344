"""Parameter manifest + weight sources of the SAM 2.1 video predictor (reference sam2_masker.py:19-20,88).

`manifest(cfg)` lists every tensor of the published state dict (names as `sam2.1_hiera_large.pt` stores them under "model") with its shape;
`Sam2Weights` serves them either from that checkpoint (validated name by name, shape by shape, BEFORE anything is used) or -- there is no
network on the build / bench machines -- as name-seeded synthetic tensors that the CPU oracle and the HIP host modules materialise
bit-identically (same scheme as videovanish_amd/weights.py).  [UNVERIFIED-3P] names; the manifest of the default config has the published
224.4 M parameters.
"""
import torch

from .sam2_config import Sam2Config, hiera_blocks
from .weights import SyntheticWeights


def manifest(cfg: Sam2Config = Sam2Config()):
    """name -> (shape, kind); kind: w = weight (fan-in scaled), b = bias, g = norm gain, e = embedding / free parameter, s = layer scale."""
    m = {}

    def lin(n, cin, cout, bias=True):
        m[n + ".weight"] = ((cout, cin), "w")
        if bias:
            m[n + ".bias"] = ((cout,), "b")

    def conv(n, cin, cout, k, groups=1):
        m[n + ".weight"] = ((cout, cin // groups, k, k), "w")
        m[n + ".bias"] = ((cout,), "b")

    def convT(n, cin, cout, k):
        m[n + ".weight"] = ((cin, cout, k, k), "wt")
        m[n + ".bias"] = ((cout,), "b")

    def norm(n, c):
        m[n + ".weight"] = ((c,), "g")
        m[n + ".bias"] = ((c,), "b")

    def attn(n, dim, internal, kv_dim=None):
        kv_dim = dim if kv_dim is None else kv_dim
        lin(n + ".q_proj", dim, internal)
        lin(n + ".k_proj", kv_dim, internal)
        lin(n + ".v_proj", kv_dim, internal)
        lin(n + ".out_proj", internal, dim)

    def mlp(n, cin, hidden, cout, layers):
        dims = [cin] + [hidden] * (layers - 1) + [cout]
        for j in range(layers):
            lin(f"{n}.layers.{j}", dims[j], dims[j + 1])

    E, D, Mm = cfg.embed_dim, cfg.d_model, cfg.mem_dim
    T = "image_encoder.trunk."
    conv(T + "patch_embed.proj", 3, E, 7)
    m[T + "pos_embed"] = ((1, E) + tuple(cfg.window_pos_embed_bkg_spatial_size), "e")
    m[T + "pos_embed_window"] = ((1, E, cfg.window_spec[0], cfg.window_spec[0]), "e")
    blocks, stage_ends = hiera_blocks(cfg)
    for i, b in enumerate(blocks):
        n = f"{T}blocks.{i}"
        norm(n + ".norm1", b["dim"])
        lin(n + ".attn.qkv", b["dim"], 3 * b["dim_out"])
        lin(n + ".attn.proj", b["dim_out"], b["dim_out"])
        norm(n + ".norm2", b["dim_out"])
        mlp(n + ".mlp", b["dim_out"], 4 * b["dim_out"], b["dim_out"], 2)
        if b["dim"] != b["dim_out"]:
            lin(n + ".proj", b["dim"], b["dim_out"])
    for j, c in enumerate(reversed(cfg.stage_dims)):
        conv(f"image_encoder.neck.convs.{j}.conv", c, D, 1)
    for i in range(cfg.mem_attn_layers):
        n = f"memory_attention.layers.{i}"
        attn(n + ".self_attn", D, D)
        attn(n + ".cross_attn_image", D, D, kv_dim=Mm)
        lin(n + ".linear1", D, cfg.mem_attn_ff)
        lin(n + ".linear2", cfg.mem_attn_ff, D)
        for k in (1, 2, 3):
            norm(f"{n}.norm{k}", D)
    norm("memory_attention.norm", D)
    c = 1
    for j in range(4):                              # MaskDownSampler(kernel 3, stride 2, padding 1, total_stride 16)
        conv(f"memory_encoder.mask_downsampler.encoder.{3 * j}", c, c * 4, 3)
        norm(f"memory_encoder.mask_downsampler.encoder.{3 * j + 1}", c * 4)
        c *= 4
    conv("memory_encoder.mask_downsampler.encoder.12", c, D, 1)
    conv("memory_encoder.pix_feat_proj", D, D, 1)
    for i in range(cfg.fuser_layers):
        n = f"memory_encoder.fuser.layers.{i}"
        conv(n + ".dwconv", D, D, 7, groups=D)
        norm(n + ".norm", D)
        lin(n + ".pwconv1", D, 4 * D)
        lin(n + ".pwconv2", 4 * D, D)
        m[n + ".gamma"] = ((D,), "s")
    conv("memory_encoder.out_proj", D, Mm, 1)
    P = "sam_prompt_encoder."
    m[P + "pe_layer.positional_encoding_gaussian_matrix"] = ((2, D // 2), "e")
    for i in range(4):
        m[f"{P}point_embeddings.{i}.weight"] = ((1, D), "e")
    m[P + "not_a_point_embed.weight"] = ((1, D), "e")
    mic = cfg.mask_in_chans
    conv(P + "mask_downscaling.0", 1, mic // 4, 2)
    norm(P + "mask_downscaling.1", mic // 4)
    conv(P + "mask_downscaling.3", mic // 4, mic, 2)
    norm(P + "mask_downscaling.4", mic)
    conv(P + "mask_downscaling.6", mic, D, 1)
    m[P + "no_mask_embed.weight"] = ((1, D), "e")
    Q = "sam_mask_decoder."
    for i in range(cfg.dec_depth):
        n = f"{Q}transformer.layers.{i}"
        attn(n + ".self_attn", D, D)
        attn(n + ".cross_attn_token_to_image", D, D // cfg.dec_downsample)
        attn(n + ".cross_attn_image_to_token", D, D // cfg.dec_downsample)
        mlp(n + ".mlp", D, cfg.dec_mlp, D, 2)
        for k in (1, 2, 3, 4):
            norm(f"{n}.norm{k}", D)
    attn(Q + "transformer.final_attn_token_to_image", D, D // cfg.dec_downsample)
    norm(Q + "transformer.norm_final_attn", D)
    m[Q + "iou_token.weight"] = ((1, D), "e")
    m[Q + "mask_tokens.weight"] = ((cfg.num_multimask + 1, D), "e")
    m[Q + "obj_score_token.weight"] = ((1, D), "e")
    convT(Q + "output_upscaling.0", D, D // 4, 2)
    norm(Q + "output_upscaling.1", D // 4)
    convT(Q + "output_upscaling.3", D // 4, D // 8, 2)
    conv(Q + "conv_s0", D, D // 8, 1)
    conv(Q + "conv_s1", D, D // 4, 1)
    for i in range(cfg.num_multimask + 1):
        mlp(f"{Q}output_hypernetworks_mlps.{i}", D, D, D // 8, 3)
    mlp(Q + "iou_prediction_head", D, D, cfg.num_multimask + 1, 3)
    mlp(Q + "pred_obj_score_head", D, D, 1, 3)
    m["maskmem_tpos_enc"] = ((cfg.num_maskmem, 1, 1, Mm), "e")
    m["no_mem_embed"] = ((1, 1, D), "e")
    m["no_mem_pos_enc"] = ((1, 1, D), "e")
    m["no_obj_ptr"] = ((1, D), "e")
    m["no_obj_embed_spatial"] = ((1, Mm), "e")
    mlp("obj_ptr_proj", D, D, D, 3)
    lin("obj_ptr_tpos_proj", D, Mm)
    conv("mask_downsample", 1, 1, 4)
    return m


def parameter_count(man):
    n = 0
    for shape, _ in man.values():
        k = 1
        for s in shape:
            k *= s
        n += k
    return n


class Sam2Weights:
    """get(name) -> fp32 CPU tensor of the manifest's shape.  Synthetic (seeded by name) unless a checkpoint state dict is given."""

    def __init__(self, cfg: Sam2Config = Sam2Config(), seed=0, state_dict=None):
        self.cfg, self.man = cfg, manifest(cfg)
        self.src = SyntheticWeights(seed)
        self.sd = state_dict
        if state_dict is not None:
            problems = []
            for name, (shape, _) in self.man.items():
                if name not in state_dict:
                    problems.append(f"'{name}' missing")
                elif tuple(state_dict[name].shape) != tuple(shape):
                    problems.append(f"'{name}' has shape {tuple(state_dict[name].shape)}, the architecture needs {tuple(shape)}")
            if problems:
                raise ValueError("SAM 2 checkpoint does not fit the architecture: " + "; ".join(problems[:12]) +
                                 (f" (+{len(problems) - 12} more)" if len(problems) > 12 else ""))

    @classmethod
    def from_checkpoint(cls, path, cfg: Sam2Config = Sam2Config()):
        """`sam2.1_hiera_large.pt` as published: a pickled {"model": state_dict} (reference sam2_masker.py:19)."""
        import os
        if not os.path.isfile(path):
            raise FileNotFoundError(f"SAM 2 checkpoint {path} does not exist")
        sd = torch.load(path, map_location="cpu", weights_only=True)
        sd = sd.get("model", sd)
        return cls(cfg, state_dict={k: v.float() for k, v in sd.items()})

    def get(self, name):
        shape, kind = self.man[name]
        if self.sd is not None:
            return self.sd[name].float().contiguous()
        if kind == "w":
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            return self.src.normal(name, shape, std=float(fan_in) ** -0.5)
        if kind == "wt":                                  # ConvTranspose2d [cin, cout, k, k]: every output pixel sees cin inputs
            return self.src.normal(name, shape, std=float(shape[0]) ** -0.5)
        if kind == "b":
            if name == "sam_mask_decoder.pred_obj_score_head.layers.2.bias":
                return self.src.normal(name, shape, std=0.02, mean=3.0)     # synthetic models "see" their object (score > 0), so masks / memories are exercised
            return self.src.normal(name, shape, std=0.02)
        if kind == "g":
            return self.src.normal(name, shape, std=0.1, mean=1.0)
        if kind == "s":
            return self.src.normal(name, shape, std=0.05, mean=0.3)
        return self.src.normal(name, shape, std=0.5)       # embeddings / free parameters
