"""ORACLE (test infrastructure, not product code): fp32 CPU PyTorch restatement of the SAM 2.1 video-predictor arithmetic
(reference call sites sam2_masker.py:88 `build_sam2_video_predictor`, :93 `init_state`, :122/:135
`add_new_points_or_box`, :147 `propagate_in_video`).

Only tests/ and tools/ benchmarks' baseline legs may import this file.  The `sam2` package is third-party, absent from /root/reference and un-pinned
(github.com/calledit/sam2_numpy_frames); the reference holds no tests or vectors for it.  PINNED SINCE ROUND 4 against an independent published
implementation of the same network, `transformers.models.sam2_video` (Hugging Face): tests/golden/make_sam2_hf_fixtures.py loads the SAME name-seeded
weights into `Sam2VideoModel` through an explicit name map and stores its inputs / outputs stage by stage; tests/test_sam2_cpu.py::
test_oracle_matches_transformers_sam2_vectors re-runs this file on them: <= 5e-7 of the output range on the image encoder, the prompted frame (click /
box + negative click / click + mask prompt), the memory encoder and a tracked frame (memory selection, temporal encodings, object pointers, RoPE
memory attention).  Not covered by that pin (no counterpart runs offline): the numpy-frames fork's own frame loading, `fill_holes`.  What follows restates
the published SAM 2.1 modules (hieradet.py, image_encoder.py, position_encoding.py, memory_attention.py, memory_encoder.py,
sam/prompt_encoder.py, sam/mask_decoder.py, sam/transformer.py, sam2_base.py, utils/misc.py::fill_holes_in_mask_scores) [UNVERIFIED-3P];
the parameter manifest these functions consume has the published 224.4 M parameters (tests/test_sam2_cpu.py).
Deviation, stated: the predictor upstream stores memory features in bfloat16 (it runs under bf16 autocast); this oracle is the fp32 ideal.

Model interface (the same five methods videovanish_amd/sam2_model.py implements on the HIP kernels; videovanish_amd/sam2_predictor.py
drives either): encode_image, track_step, encode_memory_from_low_res, fill_holes, masks_to_video_res.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from oracle import imageops_ref
from videovanish_amd.sam2_config import Sam2Config, hiera_blocks, select_memories
from videovanish_amd.sam2_weights import Sam2Weights

NO_OBJ_SCORE = -1024.0
IMG_MEAN = (0.485, 0.456, 0.406)
IMG_STD = (0.229, 0.224, 0.225)


def window_partition(x, ws):
    B, H, W, C = x.shape
    ph, pw = (ws - H % ws) % ws, (ws - W % ws) % ws
    if ph or pw:
        x = F.pad(x, (0, 0, 0, pw, 0, ph))
    Hp, Wp = H + ph, W + pw
    x = x.view(B, Hp // ws, ws, Wp // ws, ws, C)
    return x.permute(0, 1, 3, 2, 4, 5).reshape(-1, ws, ws, C), (Hp, Wp)


def window_unpartition(win, ws, pad_hw, hw):
    Hp, Wp = pad_hw
    H, W = hw
    B = win.shape[0] // (Hp * Wp // ws // ws)
    x = win.reshape(B, Hp // ws, Wp // ws, ws, ws, -1).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, -1)
    return x[:, :H, :W, :]


def sine_pos_2d(num_pos_feats, h, w, temperature=10000.0):
    """PositionEmbeddingSine(num_pos_feats, normalize=True, scale=2 pi): [1, num_pos_feats, h, w]."""
    npf = num_pos_feats // 2
    y = torch.arange(1, h + 1, dtype=torch.float32).view(h, 1).expand(h, w)
    x = torch.arange(1, w + 1, dtype=torch.float32).view(1, w).expand(h, w)
    eps, scale = 1e-6, 2 * math.pi
    y = y / (h + eps) * scale
    x = x / (w + eps) * scale
    dim_t = temperature ** (2 * (torch.arange(npf, dtype=torch.float32) // 2) / npf)
    px, py = x[:, :, None] / dim_t, y[:, :, None] / dim_t
    px = torch.stack((px[:, :, 0::2].sin(), px[:, :, 1::2].cos()), dim=3).flatten(2)
    py = torch.stack((py[:, :, 0::2].sin(), py[:, :, 1::2].cos()), dim=3).flatten(2)
    return torch.cat((py, px), dim=2).permute(2, 0, 1)[None]


def sine_pe_1d(pos, dim, temperature=10000.0):
    """get_1d_sine_pe (sam2_utils.py)."""
    pe_dim = dim // 2
    dim_t = temperature ** (2 * (torch.arange(pe_dim, dtype=torch.float32) // 2) / pe_dim)
    e = pos.unsqueeze(-1) / dim_t
    return torch.cat([e.sin(), e.cos()], dim=-1)


def axial_cis(dim, end_x, end_y, theta=10000.0):
    """compute_axial_cis (position_encoding.py): complex [end_x * end_y, dim / 2]."""
    fr = 1.0 / (theta ** (torch.arange(0, dim, 4)[: dim // 4].float() / dim))
    t = torch.arange(end_x * end_y, dtype=torch.float32)
    tx, ty = (t % end_x).float(), torch.div(t, end_x, rounding_mode="floor").float()
    fx, fy = torch.outer(tx, fr), torch.outer(ty, fr)
    return torch.cat([torch.polar(torch.ones_like(fx), fx), torch.polar(torch.ones_like(fy), fy)], dim=-1)


def apply_rope(x, cis, repeat=False):
    """x [B, heads, N, d]; cis [n, d / 2]; with repeat the table is tiled along the sequence (rope_k_repeat)."""
    if x.shape[-2] == 0:
        return x
    xc = torch.view_as_complex(x.float().reshape(*x.shape[:-1], -1, 2))
    if repeat and xc.shape[-2] != cis.shape[0]:
        cis = cis.repeat(xc.shape[-2] // cis.shape[0], 1)
    return torch.view_as_real(xc * cis[None, None]).flatten(3)


class OracleSam2:
    def __init__(self, cfg: Sam2Config = Sam2Config(), weights=None, seed=0):
        self.cfg = cfg
        self.W = weights if weights is not None else Sam2Weights(cfg, seed)
        self._c = {}
        self.blocks, self.stage_ends = hiera_blocks(cfg)
        fs = cfg.feat_size
        self.rope = axial_cis(cfg.d_model, fs, fs, cfg.rope_theta)
        self.device = torch.device("cpu")

    # ---- parameters ------------------------------------------------------------------------------------------------
    def w(self, name):
        if name not in self._c:
            self._c[name] = self.W.get(name)
        return self._c[name]

    def lin(self, x, n):
        return F.linear(x, self.w(n + ".weight"), self.w(n + ".bias"))

    def ln(self, x, n, eps=1e-5):
        return F.layer_norm(x, x.shape[-1:], self.w(n + ".weight"), self.w(n + ".bias"), eps)

    def ln2d(self, x, n, eps=1e-6):
        u = x.mean(1, keepdim=True)
        s = (x - u).pow(2).mean(1, keepdim=True)
        x = (x - u) / torch.sqrt(s + eps)
        return self.w(n + ".weight")[None, :, None, None] * x + self.w(n + ".bias")[None, :, None, None]

    def conv(self, x, n, stride=1, padding=0, groups=1):
        return F.conv2d(x, self.w(n + ".weight"), self.w(n + ".bias"), stride=stride, padding=padding, groups=groups)

    def mlp(self, x, n, layers, sigmoid=False):
        for j in range(layers):
            x = self.lin(x, f"{n}.layers.{j}")
            if j < layers - 1:
                x = F.relu(x)
        return torch.sigmoid(x) if sigmoid else x

    # ---- image encoder (hieradet.py, image_encoder.py) -------------------------------------------------------------
    def preprocess(self, frame_u8):
        """uint8 RGB [H, W, 3] -> normalised [1, 3, S, S].  Resize = cv2.INTER_LINEAR semantics (what a numpy-frame loader does)."""
        S = self.cfg.image_size
        img = imageops_ref.resize_bilinear_u8(np.ascontiguousarray(frame_u8), S, S) if frame_u8.shape[:2] != (S, S) else frame_u8
        x = torch.from_numpy(np.ascontiguousarray(img)).float().permute(2, 0, 1) / 255.0
        mean, std = torch.tensor(IMG_MEAN).view(3, 1, 1), torch.tensor(IMG_STD).view(3, 1, 1)
        return ((x - mean) / std)[None]

    def _pos_embed(self, h, w):
        T = "image_encoder.trunk."
        win = self.w(T + "pos_embed_window")
        pe = F.interpolate(self.w(T + "pos_embed"), size=(h, w), mode="bicubic")
        pe = pe + win.tile([x // y for x, y in zip(pe.shape, win.shape)])
        return pe.permute(0, 2, 3, 1)

    def _attention(self, x, n, heads, pool):
        B, H, W, _ = x.shape
        qkv = self.lin(x, n + ".qkv").reshape(B, H * W, 3, heads, -1)
        q, k, v = torch.unbind(qkv, 2)
        if pool:
            q = F.max_pool2d(q.reshape(B, H, W, -1).permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1)
            H, W = q.shape[1:3]
            q = q.reshape(B, H * W, heads, -1)
        o = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2))
        return self.lin(o.transpose(1, 2).reshape(B, H, W, -1), n + ".proj")

    def _block(self, x, i, b):
        n = f"image_encoder.trunk.blocks.{i}"
        shortcut = x
        x = self.ln(x, n + ".norm1", 1e-6)
        if b["dim"] != b["dim_out"]:
            shortcut = self.lin(x, n + ".proj")
            if b["q_stride"]:
                shortcut = F.max_pool2d(shortcut.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1)
        ws = b["window"]
        if ws > 0:
            H, W = x.shape[1], x.shape[2]
            x, pad_hw = window_partition(x, ws)
        x = self._attention(x, n + ".attn", b["heads"], bool(b["q_stride"]))
        if b["q_stride"]:
            ws = b["window"] // 2
            H, W = shortcut.shape[1:3]
            pad_hw = (H + (ws - H % ws) % ws, W + (ws - W % ws) % ws)
        if b["window"] > 0:
            x = window_unpartition(x, ws, pad_hw, (H, W))
        x = shortcut + x
        h = F.gelu(self.lin(self.ln(x, n + ".norm2", 1e-6), n + ".mlp.layers.0"))
        return x + self.lin(h, n + ".mlp.layers.1")

    def trunk(self, img):
        x = self.conv(img, "image_encoder.trunk.patch_embed.proj", stride=4, padding=3).permute(0, 2, 3, 1)
        x = x + self._pos_embed(x.shape[1], x.shape[2])
        outs = []
        for i, b in enumerate(self.blocks):
            x = self._block(x, i, b)
            if i in self.stage_ends:
                outs.append(x.permute(0, 3, 1, 2))
        return outs

    def neck(self, xs):
        n = len(xs) - 1
        out, prev = [None] * len(xs), None
        for i in range(n, -1, -1):
            lat = self.conv(xs[i], f"image_encoder.neck.convs.{n - i}.conv")
            if i in self.cfg.fpn_top_down_levels and prev is not None:
                prev = lat + F.interpolate(prev, scale_factor=2.0, mode="nearest")
            else:
                prev = lat
            out[i] = prev
        return out

    def encode_image(self, frame_u8):
        """-> {"fpn": [s0 (D/8 ch, stride 4), s1 (D/4 ch, stride 8), top (D ch, stride 16)], "pos": sine encoding of the top level}
        (SAM2Base.forward_image: conv_s0 / conv_s1 are applied here once per frame)."""
        cfg = self.cfg
        feats = self.neck(self.trunk(self.preprocess(frame_u8)))
        if cfg.scalp > 0:
            feats = feats[:-cfg.scalp]
        feats = feats[-3:]
        s0 = self.conv(feats[0], "sam_mask_decoder.conv_s0")
        s1 = self.conv(feats[1], "sam_mask_decoder.conv_s1")
        top = feats[2]
        return {"fpn": [s0, s1, top], "pos": sine_pos_2d(cfg.d_model, top.shape[2], top.shape[3])}

    # ---- memory attention (memory_attention.py, sam/transformer.py::RoPEAttention) ---------------------------------
    def _rope_attn(self, q, k, v, n, num_k_exclude_rope=0, repeat=False):
        q, k, v = self.lin(q, n + ".q_proj"), self.lin(k, n + ".k_proj"), self.lin(v, n + ".v_proj")
        q, k, v = q[:, None], k[:, None], v[:, None]                    # one head
        q = apply_rope(q, self.rope)
        nk = k.shape[-2] - num_k_exclude_rope
        k = torch.cat([apply_rope(k[:, :, :nk], self.rope, repeat=repeat), k[:, :, nk:]], dim=2)
        o = F.scaled_dot_product_attention(q, k, v)
        return self.lin(o[:, 0], n + ".out_proj")

    def memory_attention(self, curr, curr_pos, memory, memory_pos, num_obj_ptr_tokens):
        """curr / curr_pos [1, HW, D]; memory / memory_pos [1, Nm, mem_dim] -> [1, HW, D]."""
        out = curr + 0.1 * curr_pos
        for i in range(self.cfg.mem_attn_layers):
            n = f"memory_attention.layers.{i}"
            t2 = self.ln(out, n + ".norm1")
            out = out + self._rope_attn(t2, t2, t2, n + ".self_attn")
            t2 = self.ln(out, n + ".norm2")
            out = out + self._rope_attn(t2, memory + memory_pos, memory, n + ".cross_attn_image", num_obj_ptr_tokens, repeat=True)
            t2 = self.ln(out, n + ".norm3")
            out = out + self.lin(F.relu(self.lin(t2, n + ".linear1")), n + ".linear2")
        return self.ln(out, "memory_attention.norm")

    # ---- memory encoder (memory_encoder.py) ------------------------------------------------------------------------
    def memory_encoder(self, pix_feat, masks):
        n = "memory_encoder.mask_downsampler.encoder"
        m = masks
        for j in range(4):
            m = F.gelu(self.ln2d(self.conv(m, f"{n}.{3 * j}", stride=2, padding=1), f"{n}.{3 * j + 1}"))
        m = self.conv(m, f"{n}.12")
        x = self.conv(pix_feat, "memory_encoder.pix_feat_proj") + m
        for i in range(self.cfg.fuser_layers):
            f = f"memory_encoder.fuser.layers.{i}"
            h = self.ln2d(self.conv(x, f + ".dwconv", padding=3, groups=x.shape[1]), f + ".norm").permute(0, 2, 3, 1)
            h = self.lin(F.gelu(self.lin(h, f + ".pwconv1")), f + ".pwconv2") * self.w(f + ".gamma")
            x = x + h.permute(0, 3, 1, 2)
        x = self.conv(x, "memory_encoder.out_proj")
        return x, sine_pos_2d(self.cfg.mem_dim, x.shape[2], x.shape[3])

    # ---- SAM heads (prompt_encoder.py, mask_decoder.py, transformer.py) --------------------------------------------
    def _pe_encoding(self, coords01):
        c = (2 * coords01 - 1) @ self.w("sam_prompt_encoder.pe_layer.positional_encoding_gaussian_matrix")
        c = 2 * math.pi * c
        return torch.cat([torch.sin(c), torch.cos(c)], dim=-1)

    def dense_pe(self):
        fs = self.cfg.feat_size
        g = (torch.arange(fs, dtype=torch.float32) + 0.5) / fs
        xy = torch.stack([g.view(1, fs).expand(fs, fs), g.view(fs, 1).expand(fs, fs)], dim=-1)
        return self._pe_encoding(xy).permute(2, 0, 1)[None]

    def prompt_encoder(self, coords, labels, mask_logits):
        """coords [1, P, 2] in image_size pixels, labels [1, P] (-1 pad, 0 neg, 1 pos, 2 / 3 box corners); mask_logits [1,1,4fs,4fs] or None."""
        P = "sam_prompt_encoder."
        S = float(self.cfg.image_size)
        pts = torch.cat([coords + 0.5, torch.zeros(1, 1, 2)], dim=1)            # boxes arrive as points: always padded
        lab = torch.cat([labels, -torch.ones(1, 1)], dim=1)
        e = self._pe_encoding(pts / S)
        e[lab == -1] = 0.0
        e[lab == -1] += self.w(P + "not_a_point_embed.weight")
        for i in range(4):
            e[lab == i] += self.w(f"{P}point_embeddings.{i}.weight")
        fs = self.cfg.feat_size
        if mask_logits is not None:
            d = F.gelu(self.ln2d(self.conv(mask_logits, P + "mask_downscaling.0", stride=2), P + "mask_downscaling.1"))
            d = F.gelu(self.ln2d(self.conv(d, P + "mask_downscaling.3", stride=2), P + "mask_downscaling.4"))
            dense = self.conv(d, P + "mask_downscaling.6")
        else:
            dense = self.w(P + "no_mask_embed.weight").reshape(1, -1, 1, 1).expand(1, -1, fs, fs)
        return e, dense

    def _attn(self, q, k, v, n, heads):
        q, k, v = self.lin(q, n + ".q_proj"), self.lin(k, n + ".k_proj"), self.lin(v, n + ".v_proj")
        sep = lambda t: t.reshape(t.shape[0], t.shape[1], heads, -1).transpose(1, 2)
        o = F.scaled_dot_product_attention(sep(q), sep(k), sep(v))
        return self.lin(o.transpose(1, 2).reshape(q.shape[0], q.shape[1], -1), n + ".out_proj")

    def two_way_transformer(self, src, pos_src, tokens):
        """src / pos_src [1, D, h, w]; tokens [1, Nt, D] -> (queries [1, Nt, D], keys [1, hw, D])."""
        T = "sam_mask_decoder.transformer."
        H = self.cfg.dec_heads
        keys, key_pe = src.flatten(2).permute(0, 2, 1), pos_src.flatten(2).permute(0, 2, 1)
        queries, query_pe = tokens, tokens
        for i in range(self.cfg.dec_depth):
            n = f"{T}layers.{i}"
            if i == 0:
                queries = self._attn(queries, queries, queries, n + ".self_attn", H)
            else:
                q = queries + query_pe
                queries = queries + self._attn(q, q, queries, n + ".self_attn", H)
            queries = self.ln(queries, n + ".norm1")
            q, k = queries + query_pe, keys + key_pe
            queries = self.ln(queries + self._attn(q, k, keys, n + ".cross_attn_token_to_image", H), n + ".norm2")
            queries = self.ln(queries + self.lin(F.relu(self.lin(queries, n + ".mlp.layers.0")), n + ".mlp.layers.1"), n + ".norm3")
            q, k = queries + query_pe, keys + key_pe
            keys = self.ln(keys + self._attn(k, q, queries, n + ".cross_attn_image_to_token", H), n + ".norm4")
        q, k = queries + query_pe, keys + key_pe
        queries = self.ln(queries + self._attn(q, k, keys, T + "final_attn_token_to_image", H), T + "norm_final_attn")
        return queries, keys

    def mask_decoder(self, image_embeddings, sparse, dense, high_res, multimask_output):
        Q = "sam_mask_decoder."
        nm = self.cfg.num_multimask + 1
        out_tokens = torch.cat([self.w(Q + "obj_score_token.weight"), self.w(Q + "iou_token.weight"), self.w(Q + "mask_tokens.weight")], dim=0)
        tokens = torch.cat((out_tokens[None], sparse), dim=1)
        src = image_embeddings + dense
        b, c, h, w = src.shape
        hs, keys = self.two_way_transformer(src, self.dense_pe(), tokens)
        iou_token_out, mask_tokens_out = hs[:, 1, :], hs[:, 2:2 + nm, :]
        src = keys.transpose(1, 2).view(b, c, h, w)
        s0, s1 = high_res
        up = F.conv_transpose2d(src, self.w(Q + "output_upscaling.0.weight"), self.w(Q + "output_upscaling.0.bias"), stride=2)
        up = F.gelu(self.ln2d(up + s1, Q + "output_upscaling.1"))
        up = F.gelu(F.conv_transpose2d(up, self.w(Q + "output_upscaling.3.weight"), self.w(Q + "output_upscaling.3.bias"), stride=2) + s0)
        hyper = torch.stack([self.mlp(mask_tokens_out[:, i, :], f"{Q}output_hypernetworks_mlps.{i}", 3) for i in range(nm)], dim=1)
        b, c, h, w = up.shape
        masks = (hyper @ up.view(b, c, h * w)).view(b, -1, h, w)
        iou = self.mlp(iou_token_out, Q + "iou_prediction_head", 3, sigmoid=True)
        obj = self.mlp(hs[:, 0, :], Q + "pred_obj_score_head", 3)
        self.last_decoder = {"masks": masks.clone(), "iou": iou.clone()}          # for tests: every candidate the selection below chooses from
        if multimask_output:
            masks, iou, tok = masks[:, 1:], iou[:, 1:], mask_tokens_out[:, 1:]
        else:                                                   # dynamic_multimask_via_stability
            flat = masks[:, 0:1].flatten(-2)
            d = self.cfg.stability_delta
            ai, au = (flat > d).sum(-1).float(), (flat > -d).sum(-1).float()
            self.last_decoder["stability"] = float(torch.where(au > 0, ai / au, torch.ones_like(au)))
            stable = torch.where(au > 0, ai / au, torch.ones_like(au)) >= self.cfg.stability_thresh
            best = torch.argmax(iou[:, 1:], dim=-1)
            bm, bi = masks[:, 1:][torch.arange(b), best].unsqueeze(1), iou[:, 1:][torch.arange(b), best].unsqueeze(1)
            masks = torch.where(stable[..., None, None].expand_as(masks[:, 0:1]), masks[:, 0:1], bm)
            iou = torch.where(stable.expand_as(iou[:, 0:1]), iou[:, 0:1], bi)
            tok = mask_tokens_out[:, 0:1]
        return masks, iou, tok, obj

    def sam_heads(self, pix_feat, high_res, point_inputs, mask_inputs, multimask_output):
        """SAM2Base._forward_sam_heads -> (low_res_masks [1,1,4fs,4fs], obj_ptr [1,D], object_score_logits [1,1])."""
        if point_inputs is not None:
            coords, labels = point_inputs["point_coords"].float(), point_inputs["point_labels"].float()
        else:
            coords, labels = torch.zeros(1, 1, 2), -torch.ones(1, 1)
        sparse, dense = self.prompt_encoder(coords, labels, mask_inputs)
        masks, ious, tokens, obj = self.mask_decoder(pix_feat, sparse, dense, high_res, multimask_output)
        appearing = obj > 0
        masks = torch.where(appearing[:, None, None], masks, torch.full_like(masks, NO_OBJ_SCORE))
        token = tokens[:, 0]
        if multimask_output:
            best = torch.argmax(ious, dim=-1)
            masks = masks[torch.arange(1), best].unsqueeze(1)
            token = tokens[torch.arange(1), best]
        ptr = self.mlp(token, "obj_ptr_proj", 3)
        lam = appearing.float()
        ptr = lam * ptr + (1 - lam) * self.w("no_obj_ptr")
        return masks, ptr, obj

    # ---- SAM2Base.track_step / _encode_new_memory ------------------------------------------------------------------
    def _memory_conditioned(self, frame_idx, is_init_cond_frame, feats, output_dict, num_frames, track_in_reverse):
        cfg = self.cfg
        top, pos = feats["fpn"][2], feats["pos"]
        B, C, H, W = top.shape
        cur = top.flatten(2).permute(0, 2, 1)                                  # [1, HW, C]
        if is_init_cond_frame:
            return (cur + self.w("no_mem_embed")).permute(0, 2, 1).view(B, C, H, W)
        mems, ptrs, max_ptrs = select_memories(cfg, frame_idx, output_dict, num_frames, track_in_reverse)
        mem, mem_pos = [], []
        for t_pos, prev in mems:
            mem.append(prev["maskmem_features"].flatten(2).permute(0, 2, 1))
            mem_pos.append(prev["maskmem_pos_enc"].flatten(2).permute(0, 2, 1) + self.w("maskmem_tpos_enc")[cfg.num_maskmem - t_pos - 1])
        n_ptr_tokens = 0
        if ptrs:
            pos_list = torch.tensor([p for p, _ in ptrs], dtype=torch.float32)
            obj_ptrs = torch.stack([o["obj_ptr"] for _, o in ptrs], dim=0)     # [n, 1, C]
            obj_pos = self.lin(sine_pe_1d(pos_list / (max_ptrs - 1), C), "obj_ptr_tpos_proj")      # [n, mem_dim]
            split = C // cfg.mem_dim
            obj_ptrs = obj_ptrs.reshape(-1, 1, split, cfg.mem_dim).permute(0, 2, 1, 3).flatten(0, 1)   # [n * split, 1, mem_dim]
            obj_pos = obj_pos.repeat_interleave(split, dim=0)
            mem.append(obj_ptrs.permute(1, 0, 2))
            mem_pos.append(obj_pos[None])
            n_ptr_tokens = obj_ptrs.shape[0]
        out = self.memory_attention(cur, pos.flatten(2).permute(0, 2, 1), torch.cat(mem, dim=1), torch.cat(mem_pos, dim=1), n_ptr_tokens)
        return out.permute(0, 2, 1).view(B, C, H, W)

    def use_multimask(self, is_init_cond_frame, point_inputs):
        n = 0 if point_inputs is None else point_inputs["point_labels"].shape[1]
        return self.cfg.multimask_min_pt_num <= n <= self.cfg.multimask_max_pt_num        # multimask_output_in_sam and ..._for_tracking are on

    def use_mask_as_output(self, feats, mask_inputs):
        """SAM2Base._use_mask_as_output (use_mask_input_as_output_without_sam: on in the 2.1 configurations): a caller-supplied binary mask
        [1, 1, S, S] IS the frame's output -- logits -10 / +10 at the image size, antialiased bilinear to the low resolution -- and the SAM heads run only
        for the object pointer, prompted with the learned 4x4 / stride-4 `mask_downsample` of the mask on the frame's RAW top-level features (no
        memory conditioning); whether the object appears is read off the mask, not off the decoder's score."""
        fs = self.cfg.feat_size
        m = mask_inputs.float()
        high = m * 20.0 - 10.0
        low = F.interpolate(high, size=(4 * fs, 4 * fs), mode="bilinear", align_corners=False, antialias=True)
        md = self.conv(m, "mask_downsample", stride=4)
        _, ptr, _ = self.sam_heads(feats["fpn"][2], feats["fpn"][:2], None, md, False)
        lam = (m.flatten(1) > 0).any(dim=1, keepdim=True).float()
        ptr = lam * ptr + (1 - lam) * self.w("no_obj_ptr")
        return low, ptr, 20.0 * lam - 10.0

    def track_step(self, frame_idx, is_init_cond_frame, feats, point_inputs, output_dict, num_frames, track_in_reverse=False,
                   run_mem_encoder=True, prev_sam_mask_logits=None, mask_inputs=None):
        if mask_inputs is not None:
            masks, ptr, obj = self.use_mask_as_output(feats, mask_inputs)
            out = {"pred_masks": masks, "obj_ptr": ptr, "object_score_logits": obj, "maskmem_features": None, "maskmem_pos_enc": None}
            if run_mem_encoder:
                out["maskmem_features"], out["maskmem_pos_enc"] = self.encode_memory_from_low_res(feats, masks, obj, point_inputs is not None)      # upstream: is_mask_from_pts = (point_inputs is not None)
            return out
        pix = self._memory_conditioned(frame_idx, is_init_cond_frame, feats, output_dict, num_frames, track_in_reverse)
        masks, ptr, obj = self.sam_heads(pix, feats["fpn"][:2], point_inputs, prev_sam_mask_logits,
                                         self.use_multimask(is_init_cond_frame, point_inputs))
        out = {"pred_masks": masks, "obj_ptr": ptr, "object_score_logits": obj, "maskmem_features": None, "maskmem_pos_enc": None}
        if run_mem_encoder:
            out["maskmem_features"], out["maskmem_pos_enc"] = self.encode_memory_from_low_res(feats, masks, obj, point_inputs is not None)
        return out

    def encode_memory_from_low_res(self, feats, pred_masks, object_score_logits, is_mask_from_pts):
        cfg = self.cfg
        S = cfg.image_size
        high = F.interpolate(pred_masks, size=(S, S), mode="bilinear", align_corners=False)
        if cfg.binarize_mask_from_pts_for_mem_enc and is_mask_from_pts:
            m = (high > 0).float()
        else:
            m = torch.sigmoid(high)
        m = m * cfg.sigmoid_scale_for_mem_enc + cfg.sigmoid_bias_for_mem_enc
        f, pos = self.memory_encoder(feats["fpn"][2], m)
        appearing = (object_score_logits > 0).float()
        f = f + (1 - appearing[..., None, None]) * self.w("no_obj_embed_spatial")[..., None, None].expand(*f.shape)
        return f, pos

    # ---- utils/misc.py::fill_holes_in_mask_scores, SAM2VideoPredictor._get_orig_video_res_output --------------------
    def fill_holes(self, pred_masks):
        """background components (8-connected) of area <= fill_hole_area become foreground (score 0.1)."""
        from scipy import ndimage
        a = self.cfg.fill_hole_area
        if a <= 0:
            return pred_masks
        m = pred_masks.clone()
        bg = (m[0, 0] <= 0).numpy()
        lab, n = ndimage.label(bg, structure=np.ones((3, 3), dtype=bool))
        areas = np.bincount(lab.ravel())
        hole = (lab > 0) & (areas[lab] <= a)
        m[0, 0][torch.from_numpy(hole)] = 0.1
        return m

    def masks_to_video_res(self, pred_masks, H, W):
        if tuple(pred_masks.shape[-2:]) == (H, W):
            return pred_masks
        return F.interpolate(pred_masks, size=(H, W), mode="bilinear", align_corners=False)

    def clamp_prev_logits(self, pred_masks):
        return torch.clamp(pred_masks, -32.0, 32.0)

    def to_numpy(self, t):
        return t.detach().cpu().numpy()
