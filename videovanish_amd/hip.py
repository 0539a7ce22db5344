"""ctypes binding of libvvhip.so (include/vvhip.h).  PyTorch is used only for device memory and streams.

There is NO fallback: if the shared library is missing or a launcher reports an error, a RuntimeError is raised
(the GUI turns exceptions into a dialog -- reference videovanish.py:121-128,1341-1343).
"""
import ctypes as C
import os

import torch

BF16, F16, F32, U8 = 0, 1, 2, 3
SPLIT3 = 16      # vv_groupnorm out_dtype: the K-concatenated split-precision operand (see split3)
EPI_NONE, EPI_GEGLU = 0, 1
ACT_NONE, ACT_SILU, ACT_RELU, ACT_LRELU, ACT_GELU, ACT_SIGMOID = 0, 1, 2, 3, 4, 5
ABI_VERSION = 10
_DT = {"bf16": BF16, "fp16": F16}
_TORCH_H16 = {BF16: torch.bfloat16, F16: torch.float16}

PROFILE_TAG = ""   # prefix added to profile keys (nn.MotionModule sets "motion:" so bench.py can price the temporal block)
PROFILE = None   # bench.py sets this to a list: every MFMA-kernel launch is then bracketed by HIP events on the launch stream


class _Prof:
    """Bracket one launch with events on torch's current stream (the stream the kernel is launched on)."""

    def __init__(self, key, flops, nbytes):
        self.rec = None
        if PROFILE is not None:
            self.rec = [PROFILE_TAG + key, flops, nbytes, torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)]

    def __enter__(self):
        if self.rec is not None:
            self.rec[3].record()
        return self

    def __exit__(self, *a):
        if self.rec is not None:
            self.rec[4].record()
            PROFILE.append(self.rec)
        return False


# the in-tree product build; lab tools that A/B another build of the library set hip._LIB_PATH before the first launch (tools/bench_*.py).  No
# environment variable is read anywhere in this module: behaviour is selected by API only.
_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libvvhip.so")
PROFILE_SHAPES = False   # bench.py --profile-shapes: profile keys of the GEMM / attention launches carry their M, N, K (tools/shape_table.py)
_lib = None


class ConvParams(C.Structure):
    _fields_ = [("in0", C.c_void_p), ("in1", C.c_void_p), ("in_dtype", C.c_int32), ("C0", C.c_int32), ("C1", C.c_int32),
                ("F", C.c_int32), ("Hin", C.c_int32), ("Win", C.c_int32), ("Hv", C.c_int32), ("Wv", C.c_int32),
                ("Hout", C.c_int32), ("Wout", C.c_int32), ("ksize", C.c_int32), ("stride", C.c_int32),
                ("pad_t", C.c_int32), ("pad_l", C.c_int32), ("weight", C.c_void_p), ("N", C.c_int32), ("K", C.c_int32),
                ("Kpad", C.c_int32), ("Npad", C.c_int32), ("bias", C.c_void_p), ("rowvec", C.c_void_p),
                ("res0", C.c_void_p), ("res1", C.c_void_p), ("res_dtype", C.c_int32), ("out", C.c_void_p),
                ("out_dtype", C.c_int32), ("ldo", C.c_int32), ("epilogue", C.c_int32), ("out_scale", C.c_float), ("ksize_w", C.c_int32),
                ("act", C.c_int32), ("split_heads", C.c_int32), ("split_dim", C.c_int32), ("split_tokens", C.c_int32),
                ("tile_hint", C.c_int32), ("act_slope", C.c_float),
                ("sc_oh", C.c_int32), ("sc_ow", C.c_int32), ("sc_sy", C.c_int32), ("sc_sx", C.c_int32), ("sc_oy", C.c_int32), ("sc_ox", C.c_int32),
                ("gn_partials", C.c_void_p)]


class DeformParams(C.Structure):
    _fields_ = [("x", C.c_void_p), ("x_dtype", C.c_int32), ("offset", C.c_void_p), ("mask", C.c_void_p), ("raw", C.c_void_p),
                ("flow", C.c_void_p), ("max_residue", C.c_float), ("col", C.c_void_p), ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("C", C.c_int32), ("kh", C.c_int32), ("kw", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32), ("dil", C.c_int32),
                ("deform_groups", C.c_int32), ("Ho", C.c_int32), ("Wo", C.c_int32)]


class GroupNormParams(C.Structure):
    _fields_ = [("in0", C.c_void_p), ("in1", C.c_void_p), ("in_dtype", C.c_int32), ("C0", C.c_int32), ("C1", C.c_int32),
                ("F", C.c_int32), ("HW", C.c_int32), ("groups", C.c_int32), ("pool_frames", C.c_int32), ("eps", C.c_float),
                ("gamma", C.c_void_p), ("beta", C.c_void_p), ("silu", C.c_int32), ("stats_ws", C.c_void_p),
                ("out", C.c_void_p), ("out_dtype", C.c_int32)]


class MotionParams(C.Structure):
    _fields_ = [("x", C.c_void_p), ("res1", C.c_void_p), ("out", C.c_void_p), ("out_dtype", C.c_int32), ("stream", C.c_void_p),
                ("params", C.c_void_p), ("gn_affine", C.c_void_p), ("C", C.c_int32), ("F", C.c_int32), ("heads", C.c_int32), ("HW", C.c_int32),
                ("n_slabs", C.c_int32), ("n_params", C.c_int32)]


class ChainParams(C.Structure):
    _fields_ = [("o", C.c_void_p), ("t_in", C.c_void_p), ("x", C.c_void_p), ("res1", C.c_void_p), ("out", C.c_void_p), ("out_dtype", C.c_int32),
                ("stream", C.c_void_p), ("params", C.c_void_p), ("M", C.c_int64), ("C", C.c_int32), ("heads", C.c_int32), ("text_len", C.c_int32),
                ("n_slabs", C.c_int32), ("n_params", C.c_int32), ("layout", C.c_int32), ("o_hw", C.c_int32)]


CHAIN_LAYOUT_IDS = {"tokens": 0, "rowsplit": 1, "columns": 2}      # VV_CHAIN_LAYOUT_* (vvhip.h)


class ChainFrontParams(C.Structure):
    _fields_ = [("x", C.c_void_p), ("gn_affine", C.c_void_p), ("t_out", C.c_void_p), ("qkv", C.c_void_p), ("stream", C.c_void_p), ("params", C.c_void_p),
                ("M", C.c_int64), ("HW", C.c_int32), ("C", C.c_int32), ("heads", C.c_int32), ("n_slabs", C.c_int32), ("n_params", C.c_int32)]


class AttnParams(C.Structure):
    _fields_ = [("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p), ("o", C.c_void_p),
                ("q_bs", C.c_int64), ("k_bs", C.c_int64), ("v_bs", C.c_int64), ("o_bs", C.c_int64),
                ("q_rs", C.c_int64), ("k_rs", C.c_int64), ("v_rs", C.c_int64), ("o_rs", C.c_int64),
                ("B", C.c_int32), ("heads", C.c_int32), ("Nq", C.c_int32), ("Nkv", C.c_int32), ("D", C.c_int32),
                ("scale", C.c_float), ("q_hs", C.c_int64), ("k_hs", C.c_int64), ("v_hs", C.c_int64), ("q_prescaled", C.c_int32), ("lse", C.c_void_p), ("o_hs", C.c_int64)]


EXPORTS = ["vv_abi_version", "vv_last_error", "vv_device_count", "vv_device_name", "vv_conv_gemm", "vv_groupnorm_nsplit",
           "vv_groupnorm", "vv_layernorm", "vv_attention", "vv_axpby_f32", "vv_silu_f32", "vv_sched_step", "vv_add_inplace",
           "vv_mask_collapse_dilate", "vv_resize_bilinear_u8", "vv_resize_nearest_u8", "vv_feather_composite", "vv_chamfer_dt",
           "vv_preprocess", "vv_brushnet_input", "vv_pad_channels", "vv_decode_blend", "vv_blur_compose",
           "vv_avgpool2_f32", "vv_corr_lookup", "vv_raft_ctx_split", "vv_raft_flow_prep", "vv_gru_rh", "vv_gru_update", "vv_add_flow",
           "vv_add_relu_f32", "vv_convex_upsample", "vv_fb_valid", "vv_deform_im2col", "vv_fc_input", "vv_upsample2x_bilinear", "vv_flow_combine", "vv_gather_rows", "vv_fold_patches", "vv_flow_down4", "vv_gen_compose", "vv_gen_input", "vv_prop_fill", "vv_prop_combine", "vv_masked_sum_u8", "vv_u8_to_f32", "vv_u8_is_zero",
           "vv_raft_prep", "vv_split_f32", "vv_pad_channels_f32", "vv_window_average",
           "vv_groupnorm_stats", "vv_gn_affine", "vv_conv_gn_partial_blocks", "vv_gn_finalize_partials", "vv_groupnorm_apply_fin", "vv_motion_module_c320", "vv_split3", "vv_spatial_chain_c320", "vv_spatial_chain_front_c320", "vv_gn_affine_frames",
           # SAM 2 (row n4)
           "vv_u8_normalize", "vv_layernorm_ex", "vv_maxpool2x2", "vv_rope_apply", "vv_dwconv", "vv_pixel_shuffle2", "vv_resize_bilinear_f32",
           "vv_mask_mem_input", "vv_act", "vv_prompt_points", "vv_sine_pe_1d", "vv_sam_select", "vv_sam_pick", "vv_select_f32",
           "vv_add_rowvec_unless", "vv_clamp_f32", "vv_fill_holes", "vv_hyper_masks", "vv_attention_merge", "vv_ycbcr_to_rgb", "vv_sam2_maskdown"]


def lib():
    """Load libvvhip.so (once).  Raises RuntimeError when it has not been built -- never falls back."""
    global _lib
    if _lib is None:
        if not os.path.isfile(_LIB_PATH):
            raise RuntimeError(f"videovanish_amd: HIP extension missing ({_LIB_PATH}); run videovanish_amd/csrc/build.sh "
                               "or __graft_entry__.build() -- there is no CPU fallback")
        L = C.CDLL(_LIB_PATH)
        L.vv_last_error.restype = C.c_char_p
        for name in EXPORTS:
            if not hasattr(L, name):
                raise RuntimeError(f"libvvhip.so does not export {name}")
        v = L.vv_abi_version()
        if v != ABI_VERSION:
            raise RuntimeError(f"libvvhip.so ABI version {v} != {ABI_VERSION}")
        _lib = L
    return _lib


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed ({rc}): {lib().vv_last_error().decode()}")


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def dt_of(t):
    return {torch.bfloat16: BF16, torch.float16: F16, torch.float32: F32, torch.uint8: U8}[t.dtype]


def h16(dtype):
    return _TORCH_H16[dtype]


def dtype_id(name):
    return _DT[name]


def _need_cuda(*ts):
    cur = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("videovanish_amd.hip: tensors must live on the GPU (no CPU fallback)")
        if not t.is_contiguous():
            raise RuntimeError("videovanish_amd.hip: tensors must be contiguous")
        if cur is None:
            cur = torch.cuda.current_device()
        if t.device.index != cur:
            # kernels are launched on the CURRENT device's stream: a tensor on another GPU would be a cross-device pointer
            raise RuntimeError(f"videovanish_amd.hip: tensor on {t.device} but the current device is cuda:{cur} "
                               "(one process per GPU: torch.cuda.set_device(LOCAL_RANK) before building the model)")


# ----------------------------------------------------------------------------------------------------------------
def conv_gemm(dtype, x0, weight, N, K, *, x1=None, F=1, Hin=1, Win=1, Hv=None, Wv=None, Hout=None, Wout=None, ksize=1,
              stride=1, pad_t=0, pad_l=0, bias=None, rowvec=None, res0=None, res1=None, out=None, out_dtype=None,
              epilogue=EPI_NONE, out_scale=1.0, C0=None, C1=0, ksize_w=0, act=ACT_NONE, out_col=0, split_heads=0, split_dim=0, split_tokens=0, tile_hint=0,
              act_slope=0.0, scatter=None, gn_partials=False):
    """Launch vv_conv_gemm.  gn_partials: the layer also leaves per-channel (sum, sum of squares) partials of its output for the GroupNorm that reads it next
    (vv_conv_params.gn_partials; only the 128 x 160 halo-tile 3x3 kernel can): the returned tensor carries them as `out.vv_gn`.  x0/x1: NHWC activations ([F,Hin,Win,C] or any shape with C last); weight: [Npad,Kpad] h16.
    scatter = (OH, OW, sy, sx, oy, ox): row (f, y, x) of this launch goes to row (f*OH + y*sy + oy)*OW + x*sx + ox of `out` (required; residuals
    are read at the same rows) -- the four parity launches of a convolution over a nearest-2x upsampled image (nn.UpConv2x)."""
    _need_cuda(x0, x1, weight, bias, rowvec, res0, res1, out)      # out_col: write into columns [out_col, out_col+N) of `out`
    Hv = Hin if Hv is None else Hv
    Wv = Win if Wv is None else Wv
    Hout = Hv if Hout is None else Hout
    Wout = Wv if Wout is None else Wout
    C0 = x0.shape[-1] if C0 is None else C0
    if x1 is not None:
        C1 = x1.shape[-1]
    M = F * Hout * Wout
    nout = N // 2 if epilogue == EPI_GEGLU else N
    if res0 is not None and res1 is not None and res0.dtype != res1.dtype:
        raise RuntimeError(f"vv_conv_gemm: res0 ({res0.dtype}) and res1 ({res1.dtype}) must share one dtype (one res_dtype field covers both)")
    res_dt = dt_of(res0) if res0 is not None else (dt_of(res1) if res1 is not None else F32)
    sc = (0, 0, 0, 0, 0, 0) if scatter is None else tuple(int(v) for v in scatter)
    if scatter is not None and (out is None or out.numel() < F * sc[0] * sc[1] * nout):
        raise RuntimeError("vv_conv_gemm: a scattered store needs the caller's [F*OH*OW, N] output tensor")
    if out is None:
        od = h16(dtype) if out_dtype is None else out_dtype
        out = torch.empty((M, nout), dtype=od, device=x0.device)
    p = ConvParams(in0=x0.data_ptr(), in1=x1.data_ptr() if x1 is not None else 0, in_dtype=dt_of(x0), C0=C0, C1=C1, F=F, Hin=Hin,
                   Win=Win, Hv=Hv, Wv=Wv, Hout=Hout, Wout=Wout, ksize=ksize, stride=stride, pad_t=pad_t, pad_l=pad_l,
                   weight=weight.data_ptr(), N=N, K=K, Kpad=weight.shape[1], Npad=weight.shape[0],
                   bias=bias.data_ptr() if bias is not None else 0, rowvec=rowvec.data_ptr() if rowvec is not None else 0,
                   res0=res0.data_ptr() if res0 is not None else 0, res1=res1.data_ptr() if res1 is not None else 0,
                   res_dtype=res_dt, out=out.data_ptr() + out_col * out.element_size(), out_dtype=dt_of(out),
                   ldo=out.shape[-1],
                   epilogue=epilogue, out_scale=out_scale, ksize_w=ksize_w, act=act, split_heads=split_heads, split_dim=split_dim,
                   split_tokens=split_tokens, tile_hint=tile_hint, act_slope=act_slope,
                   sc_oh=sc[0], sc_ow=sc[1], sc_sy=sc[2], sc_sx=sc[3], sc_oy=sc[4], sc_ox=sc[5])
    if gn_partials:
        nblk = lib().vv_conv_gn_partial_blocks(Hout, Wout)
        part = torch.empty((F, nblk, N, 2), dtype=torch.float32, device=x0.device)
        p.gn_partials = part.data_ptr()
        out.vv_gn = GNPartials(part, nblk, F, Hout * Wout, N)
    if PROFILE is not None:
        Npad = weight.shape[0]
        # mirror of launch_t() in vv_gemm.hip (label only): LDS-DMA loaders prefer the 128x128 tile (4 blocks per CU) when N allows
        dma = x0.dtype != torch.float32 and C0 % 64 == 0 and C1 % 64 == 0 and weight.shape[1] == K
        lin = dma and ksize == 1 and stride == 1 and C1 == 0 and Hv == Hin and Wv == Win and Hout == Hin and Wout == Win
        tile = "128x128" if epilogue == EPI_GEGLU else (
            "128x160" if Npad % 160 == 0 else ("128x128" if Npad % 128 == 0 else "128x16"))
        # mirror of vv_gemm256_try() in vv_gemm256.hip (label only): the long-k / wide-N shapes run on the 256-row kernels
        if tile_hint != 1 and dma and Hv == Hin and Wv == Win and ksize * (ksize_w or ksize) <= 9 and (
                (Npad % 320 == 0 and epilogue != EPI_GEGLU) or Npad % 256 == 0):
            n256 = Npad % 256 == 0
            form = 0
            if lin:
                if K >= 5120 and Npad % 320 == 0 and Npad < 3840 and epilogue != EPI_GEGLU:
                    form = 1
                elif n256 and ((epilogue == EPI_GEGLU and K >= 1280) or (K >= 1280 and Npad >= 3840) or K >= 5120):
                    form = 3
                elif K >= 640:
                    form = 1
            elif ksize == 3 and stride == 1 and K >= 5760 and M <= 65536 and Npad % 320 == 0:
                form = 1
            bn = 256 if form == 3 else (320 if (Npad % 320 == 0 and epilogue != EPI_GEGLU) else 256)
            if form and ((M + 255) // 256) * (Npad // bn) >= 400:
                tile = f"256x{bn}" + ("p8" if form == 3 else "")
        if PROFILE_SHAPES:
            tile = f"M{M},N{N},K{K}|" + tile
        key = f"conv_gemm[{tile},{'f32in' if x0.dtype == torch.float32 else 'h16in'},k{ksize}{'x%d' % ksize_w if ksize_w and ksize_w != ksize else ''}]"
        es = x0.element_size()
        nbytes = F * Hin * Win * (C0 + C1) * es + N * K * 2 + M * nout * out.element_size() + sum(
            M * N * r.element_size() for r in (res0, res1) if r is not None)
        with _Prof(key, 2.0 * M * N * K, nbytes):
            _check(lib().vv_conv_gemm(C.byref(p), dtype, _stream()), "vv_conv_gemm")
        return out
    _check(lib().vv_conv_gemm(C.byref(p), dtype, _stream()), "vv_conv_gemm")
    return out


class GNPartials:
    """per-channel (sum, sum of squares) partials [F, nblk, C, 2] of a tensor, written by the epilogue of the layer that produced it (conv_gemm(gn_partials=True))"""

    def __init__(self, part, nblk, F, HW, C):
        self.part, self.nblk, self.F, self.HW, self.C = part, nblk, F, HW, C

    def finalize(self, groups, eps, pool_frames=False):
        """-> fin [F, groups, 2] (mean, rstd): vv_gn_finalize_partials (double accumulation, fixed order)"""
        fin = torch.empty((self.F, groups, 2), dtype=torch.float32, device=self.part.device)
        _check(lib().vv_gn_finalize_partials(_p(self.part), self.F, self.nblk, self.C, self.HW, groups, _f(eps), int(bool(pool_frames)), _p(fin), _stream()),
               "vv_gn_finalize_partials")
        return fin


def groupnorm(dtype, x0, gamma, beta, groups, eps, *, x1=None, F, HW, silu=False, pool_frames=False, out_dtype=None, act=None, partials=None):
    """partials (GNPartials of x0, one source): the statistics pass over HBM is skipped -- (mean, rstd) come from the producer's partial sums."""
    _need_cuda(x0, x1, gamma, beta)
    C0 = x0.shape[-1]
    C1 = x1.shape[-1] if x1 is not None else 0
    Ctot = C0 + C1
    nsplit = lib().vv_groupnorm_nsplit(HW, Ctot)
    ws = torch.empty(F * (nsplit + 1) * groups * 2, dtype=torch.float32, device=x0.device)
    if out_dtype == "split3":          # [M, 3C] h16 = [hi | lo * 2^4 | hi * 2^-10]: feeds a split-precision GEMM directly (no fp32 round trip)
        out = torch.empty((F * HW, 3 * Ctot), dtype=h16(dtype), device=x0.device)
        odt = SPLIT3
    else:
        out = torch.empty((F * HW, Ctot), dtype=h16(dtype) if out_dtype is None else out_dtype, device=x0.device)
        odt = dt_of(out)
    p = GroupNormParams(in0=x0.data_ptr(), in1=x1.data_ptr() if x1 is not None else 0, in_dtype=dt_of(x0), C0=C0, C1=C1, F=F, HW=HW,
                        groups=groups, pool_frames=int(pool_frames), eps=eps, gamma=gamma.data_ptr(), beta=beta.data_ptr(),
                        silu=int(act) if act is not None else int(silu), stats_ws=ws.data_ptr(), out=out.data_ptr(), out_dtype=odt)
    if partials is not None:
        assert x1 is None and (partials.F, partials.HW, partials.C) == (F, HW, C0), "GroupNorm partials do not describe this tensor"
        fin = partials.finalize(groups, eps, pool_frames)
        with _Prof("groupnorm", 0.0, F * HW * Ctot * (x0.element_size() + (6 if odt == SPLIT3 else out.element_size()))):
            _check(lib().vv_groupnorm_apply_fin(C.byref(p), _p(fin), dtype, _stream()), "vv_groupnorm_apply_fin")
        return out
    with _Prof("groupnorm", 0.0, F * HW * Ctot * (2 * x0.element_size() + (6 if odt == SPLIT3 else out.element_size()))):
        _check(lib().vv_groupnorm(C.byref(p), dtype, _stream()), "vv_groupnorm")
    return out


def motion_module_c320(dtype, x, stream_w, params, gamma, beta, groups, eps, *, F, HW, res1=None, out_dtype=torch.float32):
    """The fused motion module (vv_motion.hip): clip-pooled GroupNorm statistics -> per-channel affine -> ONE kernel for the whole block."""
    _need_cuda(x, stream_w, params, gamma, beta, res1)
    Cc = x.shape[-1]
    nsplit = lib().vv_groupnorm_nsplit(HW, Cc)
    ws = torch.empty(F * (nsplit + 1) * groups * 2, dtype=torch.float32, device=x.device)
    gp = GroupNormParams(in0=x.data_ptr(), in1=0, in_dtype=dt_of(x), C0=Cc, C1=0, F=F, HW=HW, groups=groups, pool_frames=1, eps=eps,
                         gamma=gamma.data_ptr(), beta=beta.data_ptr(), silu=0, stats_ws=ws.data_ptr(), out=0, out_dtype=F32)
    with _Prof("groupnorm", 0.0, F * HW * Cc * x.element_size()):
        _check(lib().vv_groupnorm_stats(C.byref(gp), dtype, _stream()), "vv_groupnorm_stats")
    aff = torch.empty((2, Cc), dtype=torch.float32, device=x.device)
    fin = ws.data_ptr() + F * nsplit * groups * 2 * 4
    _check(lib().vv_gn_affine(C.c_void_p(fin), _p(gamma), _p(beta), Cc, groups, _p(aff), _stream()), "vv_gn_affine")
    out = torch.empty((F * HW, Cc), dtype=out_dtype, device=x.device)
    mp = MotionParams(x=x.data_ptr(), res1=res1.data_ptr() if res1 is not None else 0, out=out.data_ptr(), out_dtype=dt_of(out),
                      stream=stream_w.data_ptr(), params=params.data_ptr(), gn_affine=aff.data_ptr(), C=Cc, F=F, heads=8, HW=HW,
                      n_slabs=stream_w.shape[0], n_params=params.numel())
    with _Prof("motion_module_fused[c320]", 2.0 * 22 * Cc * Cc * F * HW + 8.0 * F * Cc * F * HW, F * HW * Cc * (x.element_size() * 2 + out.element_size())):
        _check(lib().vv_motion_module_c320(C.byref(mp), dtype, _stream()), "vv_motion_module_c320")
    return out


def layernorm(dtype, x, gamma, beta, pe=None, rows_per_frame=1):
    _need_cuda(x, gamma, beta, pe)
    M, Cc = x.shape
    out = torch.empty((M, Cc), dtype=h16(dtype), device=x.device)
    with _Prof("layernorm", 0.0, M * Cc * 6):
        _check(lib().vv_layernorm(_p(x), M, Cc, _p(gamma), _p(beta), _p(pe), rows_per_frame, _p(out), dtype, _stream()), "vv_layernorm")
    return out


def attention_q_scale(D):
    """factor a caller folds into its query projection for vv_attention(q_prescaled=1): softmax scale * log2(e)."""
    return float(D) ** -0.5 * 1.4426950408889634


def attention(dtype, q, k, v, out, *, B, heads, Nq, Nkv, D, q_bs, k_bs, v_bs, o_bs, q_rs, k_rs, v_rs, o_rs, q_off=0, k_off=0, v_off=0,
              q_hs=0, k_hs=0, v_hs=0, q_prescaled=False, scale=None, lse=None, o_hs=0):
    """q/k/v/out: h16 tensors (any shape); element offsets *_off select a column block inside a fused QKV buffer.
    q_prescaled: q already carries D**-0.5 * log2(e) (attention_q_scale(D) folded into the query projection)."""
    _need_cuda(q, k, v, out)
    es = 2
    p = AttnParams(q=q.data_ptr() + q_off * es, k=k.data_ptr() + k_off * es, v=v.data_ptr() + v_off * es, o=out.data_ptr(),
                   q_bs=q_bs, k_bs=k_bs, v_bs=v_bs, o_bs=o_bs, q_rs=q_rs, k_rs=k_rs, v_rs=v_rs, o_rs=o_rs, B=B, heads=heads, Nq=Nq,
                   Nkv=Nkv, D=D, scale=float(D) ** -0.5 if scale is None else float(scale), q_hs=q_hs, k_hs=k_hs, v_hs=v_hs, q_prescaled=1 if q_prescaled else 0, lse=lse.data_ptr() if lse is not None else 0, o_hs=o_hs)
    kind = "temporal" if (Nq <= 32 and Nkv <= 32) else ("cross" if Nkv < 128 and Nq != Nkv else "spatial")
    if PROFILE_SHAPES:
        kind = f"B{B},N{Nq}|" + kind
    with _Prof(f"attention[{kind},d{D}]", 4.0 * B * heads * Nq * Nkv * D, 2 * B * heads * D * (2 * Nq + 2 * Nkv)):
        _check(lib().vv_attention(C.byref(p), dtype, _stream()), "vv_attention")
    return out


def deform_im2col(dtype, x, *, B, H, W, kh=3, kw=3, stride=1, pad=1, dil=1, deform_groups=1, offset=None, mask=None, raw=None, flow=None,
                  max_residue=0.0):
    """Deformed, modulated im2col matrix [M, kh*kw*C] (h16) of an NHWC tensor x [B*H*W, C]: see vv_deform_im2col in include/vvhip.h.
    Either (offset [M, 2*dg*K], mask [M, dg*K] | None) or raw [M, 3*dg*K] (+ flow [M, 2]) -- the DeformableAlignment front end."""
    _need_cuda(x, offset, mask, raw, flow)
    Cc = x.shape[-1]
    Ho = (H + 2 * pad - dil * (kh - 1) - 1) // stride + 1
    Wo = (W + 2 * pad - dil * (kw - 1) - 1) // stride + 1
    col = torch.empty((B * Ho * Wo, kh * kw * Cc), dtype=h16(dtype), device=x.device)
    for t in (offset, mask, raw, flow):
        if t is not None and t.dtype != torch.float32:
            raise RuntimeError("vv_deform_im2col: offset / mask / raw / flow must be fp32")
    p = DeformParams(x=x.data_ptr(), x_dtype=dt_of(x), offset=_p(offset), mask=_p(mask), raw=_p(raw), flow=_p(flow), max_residue=max_residue,
                     col=col.data_ptr(), B=B, H=H, W=W, C=Cc, kh=kh, kw=kw, stride=stride, pad=pad, dil=dil, deform_groups=deform_groups, Ho=Ho, Wo=Wo)
    with _Prof("deform_im2col", 0.0, col.numel() * 2 * 5):
        _check(lib().vv_deform_im2col(C.byref(p), dtype, _stream()), "vv_deform_im2col")
    return col, Ho, Wo


def fc_input(flow, mask_u8, pad):
    """flow fp32 [T,H,W,2] + mask u8 [T,H,W] -> fp32 [T, H+2pad, W+2pad, 8] = (flow*(1-m) | m | 0..) replicate-padded."""
    _need_cuda(flow, mask_u8)
    T, H, W, _ = flow.shape
    out = torch.empty((T, H + 2 * pad, W + 2 * pad, 8), dtype=torch.float32, device=flow.device)
    _check(lib().vv_fc_input(_p(flow), _p(mask_u8), T, H, W, pad, _p(out), _stream()), "vv_fc_input")
    return out


def upsample2x_bilinear(dtype, x, B, H, W):
    """NHWC rows [B*H*W, C] (h16 or fp32) -> [B*2H*2W, C], bilinear, align_corners=True."""
    _need_cuda(x)
    Cc = x.shape[-1]
    out = torch.empty((B * 4 * H * W, Cc), dtype=x.dtype, device=x.device)
    _check(lib().vv_upsample2x_bilinear(_p(x), dt_of(x), B, H, W, Cc, _p(out), dtype, _stream()), "vv_upsample2x_bilinear")
    return out


def flow_combine(pred, flow, mask_u8):
    """pred fp32 [N, ld>=2], flow fp32 [..., 2] with N pixels, mask u8 [N] -> fp32 like flow: pred in the hole, flow outside."""
    _need_cuda(pred, flow, mask_u8)
    out = torch.empty_like(flow)
    _check(lib().vv_flow_combine(_p(pred), pred.shape[-1], _p(flow), _p(mask_u8), C.c_int64(mask_u8.numel()), _p(out), _stream()), "vv_flow_combine")
    return out


def gather_rows(src, idx):
    """out[i] = src[idx[i]] (idx int32 on the device, < 0 -> zero row); rows must be multiples of 16 bytes."""
    _need_cuda(src, idx)
    out = torch.empty((idx.numel(), src.shape[-1]), dtype=src.dtype, device=src.device)
    _check(lib().vv_gather_rows(_p(src), _p(idx), C.c_int64(idx.numel()), src.shape[-1] * src.element_size(), _p(out), _stream()), "vv_gather_rows")
    return out


def fold_patches(dtype, x, B, fh, fw, Cc, h, w, k=7, stride=3, pad=3, normalise=False, gelu=False, out_dtype=torch.float32):
    """tap-major patch rows [B*fh*fw, k*k*C] -> [B*h*w, C] (F.fold; optional overlap normalisation and GELU)."""
    _need_cuda(x)
    out = torch.empty((B * h * w, Cc), dtype=out_dtype, device=x.device)
    _check(lib().vv_fold_patches(_p(x), dt_of(x), B, fh, fw, Cc, h, w, k, stride, pad, int(normalise), int(gelu), _p(out), dt_of(out), dtype, _stream()),
           "vv_fold_patches")
    return out


def flow_down4(flow):
    """fp32 [T,H,W,2] -> fp32 [T,H/4,W/4,2] (bilinear 1/4, align_corners=False, values / 4)."""
    _need_cuda(flow)
    T, H, W, _ = flow.shape
    out = torch.empty((T, H // 4, W // 4, 2), dtype=torch.float32, device=flow.device)
    _check(lib().vv_flow_down4(_p(flow), T, H, W, _p(out), _stream()), "vv_flow_down4")
    return out


def gen_input(frames_u8, mask_in_u8, mask_up_u8):
    """u8 [T,H,W,3] + two u8 [T,H,W] masks -> fp32 [T*H*W, 8] encoder input rows."""
    _need_cuda(frames_u8, mask_in_u8, mask_up_u8)
    n = mask_in_u8.numel()
    out = torch.empty((n, 8), dtype=torch.float32, device=frames_u8.device)
    _check(lib().vv_gen_input(_p(frames_u8), _p(mask_in_u8), _p(mask_up_u8), C.c_int64(n), _p(out), _stream()), "vv_gen_input")
    return out


def gen_compose(pred, ori_u8, mask_u8, acc, first):
    _need_cuda(pred, ori_u8, mask_u8, acc)
    _check(lib().vv_gen_compose(_p(pred), pred.shape[-1], _p(ori_u8), _p(mask_u8), C.c_int64(mask_u8.numel()), _p(acc), int(first), _stream()), "vv_gen_compose")
    return acc


def axpby(x, y, ca, cb, out=None):
    _need_cuda(x, y, out)
    out = torch.empty_like(x) if out is None else out
    _check(lib().vv_axpby_f32(_p(x), _p(y), C.c_float(ca), C.c_float(cb), _p(out), C.c_int64(x.numel()), _stream()), "vv_axpby_f32")
    return out


def sched_step(x, eps, z, sa_t, sb_t, c_x0, c_eps, c_z=0.0, out=None):
    _need_cuda(x, eps, z, out)
    out = torch.empty_like(x) if out is None else out
    _check(lib().vv_sched_step(_p(x), _p(eps), _p(z), C.c_float(sa_t), C.c_float(sb_t), C.c_float(c_x0), C.c_float(c_eps), C.c_float(c_z),
                               _p(out), C.c_int64(x.numel()), _stream()), "vv_sched_step")
    return out


def silu(x):
    _need_cuda(x)
    out = torch.empty_like(x)
    _check(lib().vv_silu_f32(_p(x), _p(out), C.c_int64(x.numel()), _stream()), "vv_silu_f32")
    return out


def add_inplace(dtype, x, y):
    _need_cuda(x, y)
    assert x.dtype == torch.float32 and x.numel() == y.numel()
    _check(lib().vv_add_inplace(_p(x), _p(y), dt_of(y), C.c_int64(x.numel()), dtype, _stream()), "vv_add_inplace")
    return x


def mask_collapse_dilate(masks, iters):
    """masks [T,H,W,ch] u8 -> [T,H,W] u8 {0,255}  (reference diffuerase.py:27-31)."""
    _need_cuda(masks)
    if masks.dim() == 3:
        masks = masks[..., None].contiguous()
    T, H, W, ch = masks.shape
    out = torch.empty((T, H, W), dtype=torch.uint8, device=masks.device)
    tmp = torch.empty_like(out)
    flags = torch.empty(T, dtype=torch.int32, device=masks.device)
    with _Prof("mask_collapse_dilate", 0.0, T * H * W * (ch + 1)):
        _check(lib().vv_mask_collapse_dilate(_p(masks), T, H, W, ch, int(iters), _p(out), _p(tmp), _p(flags), _stream()), "vv_mask_collapse_dilate")
    return out


def resize_u8(src, Hd, Wd, mode="bilinear"):
    _need_cuda(src)
    s4 = src if src.dim() == 4 else src[..., None]
    T, Hs, Ws, ch = s4.shape
    dst = torch.empty((T, Hd, Wd, ch), dtype=torch.uint8, device=src.device)
    fn = lib().vv_resize_bilinear_u8 if mode == "bilinear" else lib().vv_resize_nearest_u8
    with _Prof("resize_u8", 0.0, src.numel() + dst.numel()):
        _check(fn(_p(src), T, Hs, Ws, ch, _p(dst), Hd, Wd, _stream()), "vv_resize_u8")
    return dst if src.dim() == 4 else dst[..., 0]


def feather_composite(inpainted, orig, mask2d, feather_px):
    _need_cuda(inpainted, orig, mask2d)
    T, H, W, _ = inpainted.shape
    out = torch.empty_like(inpainted)
    with _Prof("feather_composite", 0.0, T * H * W * (3 + 3 + 1 + 3)):
        _check(lib().vv_feather_composite(_p(inpainted), _p(orig), _p(mask2d), T, H, W, C.c_float(feather_px), _p(out), _stream()), "vv_feather_composite")
    return out


def chamfer_dt(bin_u8, R):
    _need_cuda(bin_u8)
    T, H, W = bin_u8.shape
    out = torch.empty((T, H, W), dtype=torch.float32, device=bin_u8.device)
    _check(lib().vv_chamfer_dt(_p(bin_u8), T, H, W, R, _p(out), _stream()), "vv_chamfer_dt")
    return out


def preprocess(dtype, frames, mask2d, want_img=True, want_masked=True):
    _need_cuda(frames, mask2d)
    T, H, W, _ = frames.shape
    img = torch.empty((T, H, W, 8), dtype=h16(dtype), device=frames.device) if want_img else None
    msk = torch.empty((T, H, W, 8), dtype=h16(dtype), device=frames.device) if want_masked else None
    _check(lib().vv_preprocess(_p(frames), _p(mask2d), T, H, W, _p(img), _p(msk), dtype, _stream()), "vv_preprocess")
    return img, msk


def brushnet_input(dtype, lat, cond, mask2d, H, W):
    _need_cuda(lat, cond, mask2d)
    F, h, w, _ = lat.shape
    out = torch.empty((F, h, w, 16), dtype=h16(dtype), device=lat.device)
    _check(lib().vv_brushnet_input(_p(lat), _p(cond), _p(mask2d), F, h, w, H, W, _p(out), dtype, _stream()), "vv_brushnet_input")
    return out


def pad_channels(dtype, x, cpad, scale=1.0):
    _need_cuda(x)
    cin = x.shape[-1]
    rows = x.numel() // cin
    out = torch.empty(x.shape[:-1] + (cpad,), dtype=h16(dtype), device=x.device)
    _check(lib().vv_pad_channels(_p(x), C.c_int64(rows), cin, cpad, C.c_float(scale), _p(out), dtype, _stream()), "vv_pad_channels")
    return out


def window_average(value, count):
    """value fp32 [T, ...], count fp32 [T] -> value / count per frame."""
    _need_cuda(value, count)
    T = value.shape[0]
    out = torch.empty_like(value)
    _check(lib().vv_window_average(_p(value), _p(count), T, C.c_int64(value.numel() // T), _p(out), _stream()), "vv_window_average")
    return out


def pad_channels_f32(x, cpad, scale=1.0):
    _need_cuda(x)
    cin = x.shape[-1]
    rows = x.numel() // cin
    out = torch.empty(x.shape[:-1] + (cpad,), dtype=torch.float32, device=x.device)
    _check(lib().vv_pad_channels_f32(_p(x), C.c_int64(rows), cin, cpad, C.c_float(scale), _p(out), _stream()), "vv_pad_channels_f32")
    return out


def split_f32(dtype, x, lo_scale):
    """fp32 tensor -> (hi, lo) h16 tensors of the same shape: hi = h16(x), lo = h16((x - hi) * lo_scale)."""
    _need_cuda(x)
    assert x.dtype == torch.float32
    hi = torch.empty(x.shape, dtype=h16(dtype), device=x.device)
    lo = torch.empty(x.shape, dtype=h16(dtype), device=x.device)
    _check(lib().vv_split_f32(_p(x), C.c_int64(x.numel()), C.c_float(lo_scale), _p(hi), _p(lo), dtype, _stream()), "vv_split_f32")
    return hi, lo


def _packing():
    from . import packing
    return packing


def spatial_chain_c320(dtype, o, t_in, x, stream_w, params, *, res1=None, out_dtype=torch.float32, layout=None, o_hw=0):
    """The fused tail of a level-0 spatial transformer block (vv_chain.hip): attn1 out-proj + residual, cross-attention to the text tokens,
    GEGLU feed-forward, proj_out + block residual in ONE kernel.  o: h16 [M,320] (o_hw = 0), or head-major [M / o_hw, 8, o_hw, 40] as vv_attention writes it
    with o_hs = o_hw * 40; t_in, x (, res1): fp32 [M,320]."""
    _need_cuda(o, t_in, x, stream_w, params, res1)
    M, Cc = t_in.shape
    assert o.numel() == M * Cc and x.shape == (M, Cc) and o.dtype == h16(dtype) and t_in.dtype == x.dtype == torch.float32
    assert (o.shape == (M, Cc)) if o_hw == 0 else (M % o_hw == 0 and o.is_contiguous())
    out = torch.empty((M, Cc), dtype=out_dtype, device=x.device)
    cp = ChainParams(o=o.data_ptr(), t_in=t_in.data_ptr(), x=x.data_ptr(), res1=res1.data_ptr() if res1 is not None else 0, out=out.data_ptr(),
                     out_dtype=dt_of(out), stream=stream_w.data_ptr(), params=params.data_ptr(), M=M, C=Cc, heads=8, text_len=77,
                     n_slabs=stream_w.shape[0], n_params=params.numel(), layout=CHAIN_LAYOUT_IDS[layout or _packing().CHAIN_LAYOUT], o_hw=int(o_hw))
    flops = 2.0 * M * Cc * Cc * (1 + 1 + 1 + 12 + 1) + 4.0 * M * 77 * Cc
    with _Prof("spatial_chain_fused[c320]", flops, M * Cc * (2 + 4 + 4 + out.element_size())):
        _check(lib().vv_spatial_chain_c320(C.byref(cp), dtype, _stream()), "vv_spatial_chain_c320")
    return out


def spatial_chain_front_c320(dtype, x, gamma, beta, groups, eps, stream_w, params, *, F, HW, partials=None):
    """The fused front of a level-0 spatial transformer block (vv_chain.hip): per-frame GroupNorm statistics -> per-channel affine -> ONE kernel for
    GroupNorm apply + proj_in + LayerNorm + the fused q|k|v projection.  Returns (t fp32 [M,320] = the block's residual stream, qkv h16 head-major
    [F][3][8][HW][40])."""
    _need_cuda(x, gamma, beta, stream_w, params)
    M, Cc = x.shape
    assert M == F * HW and x.dtype == torch.float32
    aff = torch.empty((F, 2, Cc), dtype=torch.float32, device=x.device)
    if partials is not None:      # the producing convolution left per-channel partial sums of x (conv_gemm(gn_partials=True)): no statistics pass over HBM
        assert (partials.F, partials.HW, partials.C) == (F, HW, Cc), "GroupNorm partials do not describe this tensor"
        fin_t = partials.finalize(groups, eps)
        fin = fin_t.data_ptr()
    else:
        nsplit = lib().vv_groupnorm_nsplit(HW, Cc)
        ws = torch.empty(F * (nsplit + 1) * groups * 2, dtype=torch.float32, device=x.device)
        gp = GroupNormParams(in0=x.data_ptr(), in1=0, in_dtype=dt_of(x), C0=Cc, C1=0, F=F, HW=HW, groups=groups, pool_frames=0, eps=eps,
                             gamma=gamma.data_ptr(), beta=beta.data_ptr(), silu=0, stats_ws=ws.data_ptr(), out=0, out_dtype=F32)
        with _Prof("groupnorm", 0.0, F * HW * Cc * x.element_size()):
            _check(lib().vv_groupnorm_stats(C.byref(gp), dtype, _stream()), "vv_groupnorm_stats")
        fin = ws.data_ptr() + F * nsplit * groups * 2 * 4
    _check(lib().vv_gn_affine_frames(C.c_void_p(fin), _p(gamma), _p(beta), Cc, groups, F, _p(aff), _stream()), "vv_gn_affine_frames")
    t = torch.empty((M, Cc), dtype=torch.float32, device=x.device)
    qkv = torch.empty((F, 3, 8, HW, Cc // 8), dtype=h16(dtype), device=x.device)
    fp = ChainFrontParams(x=x.data_ptr(), gn_affine=aff.data_ptr(), t_out=t.data_ptr(), qkv=qkv.data_ptr(), stream=stream_w.data_ptr(),
                          params=params.data_ptr(), M=M, HW=HW, C=Cc, heads=8, n_slabs=stream_w.shape[0], n_params=params.numel())
    with _Prof("spatial_chain_front_fused[c320]", 2.0 * M * Cc * Cc * 4, M * Cc * (4 + 4 + 6)):
        _check(lib().vv_spatial_chain_front_c320(C.byref(fp), dtype, _stream()), "vv_spatial_chain_front_c320")
    return t, qkv


def split3(dtype, x):
    """[M, C] fp32 (or h16) -> [M, 3C] h16 = [hi | (x - hi) * 2^4 | hi * 2^-10]: the A operand of a K-concatenated split-precision GEMM (vv_split3)."""
    _need_cuda(x)
    assert x.dim() == 2 and x.is_contiguous()
    M, Cc = x.shape
    out = torch.empty((M, 3 * Cc), dtype=h16(dtype), device=x.device)
    with _Prof("split3", 0.0, x.numel() * (x.element_size() + 6)):
        _check(lib().vv_split3(_p(x), dt_of(x), C.c_int64(M), Cc, _p(out), dtype, _stream()), "vv_split3")
    return out


def decode_blend(dec, w, acc):
    """dec [T,H,W,ld] fp32 ; w [T] fp32 ; acc [T,H,W,3] fp32 updated in place."""
    _need_cuda(dec, w, acc)
    T, H, W, ld = dec.shape
    _check(lib().vv_decode_blend(_p(dec), ld, _p(w), T, C.c_int64(H * W), _p(acc), _stream()), "vv_decode_blend")
    return acc


def blur_compose(pix01, orig, mask2d, taps21):
    _need_cuda(pix01, orig, mask2d)
    T, H, W, _ = pix01.shape
    tmp = torch.empty((T, H, W), dtype=torch.float32, device=pix01.device)
    out = torch.empty((T, H, W, 3), dtype=torch.uint8, device=pix01.device)
    taps = (C.c_float * 21)(*[float(x) for x in taps21])
    _check(lib().vv_blur_compose(_p(pix01), _p(orig), _p(mask2d), T, H, W, taps, _p(tmp), _p(out), _stream()), "vv_blur_compose")
    return out


# ---- K9/K10: RAFT / flow-guided propagation kernels ----------------------------------------------------------------
def avgpool2(x):
    """[N,h,w] fp32 -> [N,h//2,w//2]."""
    _need_cuda(x)
    N, h, w = x.shape
    out = torch.empty((N, h // 2, w // 2), dtype=torch.float32, device=x.device)
    with _Prof("avgpool2[corr pyramid]", 0.0, x.numel() * 5):
        _check(lib().vv_avgpool2_f32(_p(x), C.c_int64(N), h, w, _p(out), _stream()), "vv_avgpool2_f32")
    return out


def corr_lookup(dtype, pyr, coords, cpad=384):
    """pyr: 4 fp32 tensors [N,h_l,w_l]; coords [N,2] fp32 (x,y) -> h16 [N,cpad] (324 real channels)."""
    _need_cuda(*pyr, coords)
    N, h, w = pyr[0].shape
    out = torch.empty((N, cpad), dtype=h16(dtype), device=coords.device)
    with _Prof("corr_lookup", 0.0, N * (4 * 100 * 4 + 324 * 2 + 8)):
        _check(lib().vv_corr_lookup(_p(pyr[0]), _p(pyr[1]), _p(pyr[2]), _p(pyr[3]), h, w, _p(coords), C.c_int64(N), cpad, _p(out), dtype, _stream()),
               "vv_corr_lookup")
    return out


def raft_ctx_split(dtype, cn, net, net16, xbuf):
    _need_cuda(cn, net, net16, xbuf)
    with _Prof("raft_elementwise", 0.0, cn.numel() * 4 + net.numel() * 6 + xbuf.shape[0] * 256):
        _check(lib().vv_raft_ctx_split(_p(cn), C.c_int64(cn.shape[0]), _p(net), _p(net16), _p(xbuf), dtype, _stream()), "vv_raft_ctx_split")


def raft_flow_prep(dtype, coords1, w, h, flow8, xbuf):
    _need_cuda(coords1, flow8, xbuf)
    with _Prof("raft_elementwise", 0.0, coords1.numel() * 4 + flow8.numel() * 2 + coords1.shape[0] * 4):
        _check(lib().vv_raft_flow_prep(_p(coords1), C.c_int64(coords1.shape[0]), w, h, _p(flow8), _p(xbuf), dtype, _stream()), "vv_raft_flow_prep")


def gru_rh(dtype, zr, h, rh):
    _need_cuda(zr, h, rh)
    with _Prof("raft_elementwise", 0.0, h.numel() * (2 + 4 + 2)):
        _check(lib().vv_gru_rh(_p(zr), _p(h), C.c_int64(h.shape[0]), _p(rh), dtype, _stream()), "vv_gru_rh")


def gru_update(dtype, zr, q, h, h16_out):
    _need_cuda(zr, q, h, h16_out)
    with _Prof("raft_elementwise", 0.0, h.numel() * (4 + 2 + 4 + 4 + 2)):
        _check(lib().vv_gru_update(_p(zr), _p(q), C.c_int64(h.shape[0]), _p(h), _p(h16_out), dtype, _stream()), "vv_gru_update")


def add_flow(coords1, dflow):
    _need_cuda(coords1, dflow)
    with _Prof("raft_elementwise", 0.0, coords1.numel() * 12):
        _check(lib().vv_add_flow(_p(coords1), _p(dflow), dflow.shape[-1], C.c_int64(coords1.shape[0]), _stream()), "vv_add_flow")


def add_relu(a, b):
    _need_cuda(a, b)
    out = torch.empty_like(a)
    with _Prof("add_relu", 0.0, a.numel() * 12):
        _check(lib().vv_add_relu_f32(_p(a), _p(b), _p(out), C.c_int64(a.numel()), _stream()), "vv_add_relu_f32")
    return out


def convex_upsample(coords1, mask, h, w, F=1):
    """coords1 [F*h*w, 2], mask [F*h*w, 576] -> flow [8h, 8w, 2] (F = 1) or [F, 8h, 8w, 2]."""
    _need_cuda(coords1, mask)
    out = torch.empty((F, 8 * h, 8 * w, 2), dtype=torch.float32, device=coords1.device)
    with _Prof("convex_upsample", 0.0, mask.numel() * mask.element_size() + coords1.numel() * 4 + out.numel() * 4):
        _check(lib().vv_convex_upsample(_p(coords1), _p(mask), F, h, w, _p(out), _stream()), "vv_convex_upsample")
    return out[0] if F == 1 else out


def fb_valid(f_ab, f_ba):
    _need_cuda(f_ab, f_ba)
    H, W, _ = f_ab.shape
    out = torch.empty((H, W), dtype=torch.uint8, device=f_ab.device)
    with _Prof("fb_valid[bilinear warp of the backward flow]", 0.0, H * W * (8 + 8 + 1)):
        _check(lib().vv_fb_valid(_p(f_ab), _p(f_ba), H, W, _p(out), _stream()), "vv_fb_valid")
    return out


def prop_fill(cur_t, cur_nb, known_t, known_nb, valid, flow, filled_t):
    _need_cuda(cur_t, cur_nb, known_t, known_nb, valid, flow, filled_t)
    H, W, _ = cur_t.shape
    with _Prof("prop_fill[bilinear warp]", 0.0, H * W * (3 + 3 + 2 + 1 + 8 + 3)):
        _check(lib().vv_prop_fill(_p(cur_t), _p(cur_nb), _p(known_t), _p(known_nb), _p(valid), _p(flow), H, W, _p(filled_t), _stream()), "vv_prop_fill")


def prop_combine(orig, a, b, fa, fb, hole, mean3):
    _need_cuda(orig, a, b, fa, fb, hole, mean3)
    H, W, _ = orig.shape
    out = torch.empty((H, W, 3), dtype=torch.uint8, device=orig.device)
    filled = torch.empty((H, W), dtype=torch.uint8, device=orig.device)
    with _Prof("prop_combine", 0.0, H * W * (3 * 3 + 3 + 3 + 1)):
        _check(lib().vv_prop_combine(_p(orig), _p(a), _p(b), _p(fa), _p(fb), _p(hole), H, W, _p(mean3), _p(out), _p(filled), _stream()), "vv_prop_combine")
    return out, filled


def masked_sum_u8(frame, hole):
    """-> int64 tensor [4] (sum r, g, b over non-hole pixels, count) on the device."""
    _need_cuda(frame, hole)
    sums = torch.empty(4, dtype=torch.int64, device=frame.device)
    _check(lib().vv_masked_sum_u8(_p(frame), _p(hole), C.c_int64(hole.numel()), _p(sums), _stream()), "vv_masked_sum_u8")
    return sums


def u8_to_f32(x):
    _need_cuda(x)
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    _check(lib().vv_u8_to_f32(_p(x), _p(out), C.c_int64(x.numel()), _stream()), "vv_u8_to_f32")
    return out


def u8_is_zero(x):
    _need_cuda(x)
    out = torch.empty_like(x)
    _check(lib().vv_u8_is_zero(_p(x), _p(out), C.c_int64(x.numel()), _stream()), "vv_u8_is_zero")
    return out


def raft_prep(dtype, img):
    """u8 [..., 3] -> h16 [..., 8] scaled to [-1,1]."""
    _need_cuda(img)
    out = torch.empty(img.shape[:-1] + (8,), dtype=h16(dtype), device=img.device)
    with _Prof("raft_prep", 0.0, img.numel() + out.numel() * 2):
        _check(lib().vv_raft_prep(_p(img), C.c_int64(img.numel() // 3), _p(out), dtype, _stream()), "vv_raft_prep")
    return out


# ---- SAM 2 (include/vvhip.h "SAM 2" section) ---------------------------------------------------------------------
def _f(x):
    return C.c_float(float(x))


def u8_normalize(dtype, img_u8, mean, std, cpad=8, s2d=1):
    """uint8 [H, W, 3] -> h16 [(H/s2d)*(W/s2d), cpad] = ((x / 255) - mean) / std; s2d > 1: space-to-depth blocks, channel (dy*s2d + dx)*3 + c."""
    _need_cuda(img_u8)
    H, W = img_u8.shape[:2]
    out = torch.empty(((H // s2d) * (W // s2d), cpad), dtype=h16(dtype), device=img_u8.device)
    m = (C.c_float * 3)(*[float(v) for v in mean])
    s = (C.c_float * 3)(*[1.0 / float(v) for v in std])
    _check(lib().vv_u8_normalize(_p(img_u8), H, W, m, s, _p(out), cpad, s2d, dtype, _stream()), "vv_u8_normalize")
    return out


def layernorm_ex(dtype, x, gamma, beta, eps, act=ACT_NONE, out_dtype=None, cpad=None):
    """LayerNorm over the last dim of fp32 [M, C] (any C) + optional activation; out_dtype None = h16, torch.float32 = fp32;
    cpad: output row length (zero padded channels, for a following conv whose input channels are padded)."""
    _need_cuda(x, gamma, beta)
    M, Cc = x.shape
    cp = Cc if cpad is None else cpad
    out = torch.empty((M, cp), dtype=h16(dtype) if out_dtype is None else out_dtype, device=x.device)
    _check(lib().vv_layernorm_ex(_p(x), C.c_int64(M), Cc, _p(gamma), _p(beta), _f(eps), act, _p(out), dt_of(out), cp, dtype, _stream()), "vv_layernorm_ex")
    return out


def maxpool2x2(x, B, H, W, Cc=None, in_bs=0, x_off=0):
    """B images [H, W, C] (fp32 or h16; image b starts x_off + b * in_bs elements into x, in_bs = 0: contiguous) -> [B*(H/2)*(W/2), C]."""
    _need_cuda(x)
    Cc = x.shape[-1] if Cc is None else Cc
    out = torch.empty((B * (H // 2) * (W // 2), Cc), dtype=x.dtype, device=x.device)
    _check(lib().vv_maxpool2x2(C.c_void_p(x.data_ptr() + x_off * x.element_size()), dt_of(x), B, H, W, Cc, C.c_int64(in_bs), _p(out), _stream()),
           "vv_maxpool2x2")
    return out


def rope_apply(dtype, x, rows_rope, cos_sin, D, col0=0, ld=None):
    """in place: rows < rows_rope of the h16 matrix x, columns col0 .. col0 + D, rotated pairwise by cos_sin [n, D/2, 2] (row r uses r % n)."""
    _need_cuda(x, cos_sin)
    _check(lib().vv_rope_apply(_p(x), C.c_int64(rows_rope), x.shape[-1] if ld is None else ld, col0, D, _p(cos_sin), cos_sin.shape[0], dtype,
                               _stream()), "vv_rope_apply")
    return x


def dwconv(x, H, W, w, bias):
    _need_cuda(x, w, bias)
    out = torch.empty_like(x)
    _check(lib().vv_dwconv(_p(x), H, W, x.shape[-1], _p(w), _p(bias), w.shape[-1], _p(out), _stream()), "vv_dwconv")
    return out


def pixel_shuffle2(dtype, y, bias, h, w, add=None, act=ACT_NONE, out_dtype=torch.float32):
    _need_cuda(y, bias, add)
    Cc = y.shape[-1] // 4
    out = torch.empty((4 * h * w, Cc), dtype=out_dtype, device=y.device)
    _check(lib().vv_pixel_shuffle2(_p(y), _p(bias), _p(add), h, w, Cc, act, _p(out), dt_of(out), dtype, _stream()), "vv_pixel_shuffle2")
    return out


def resize_bilinear_f32(src, Hs, Ws, Hd, Wd):
    """fp32 [Hs*Ws, C] -> [Hd*Wd, C], F.interpolate(mode="bilinear", align_corners=False)."""
    _need_cuda(src)
    Cc = src.shape[-1]
    out = torch.empty((Hd * Wd, Cc), dtype=torch.float32, device=src.device)
    _check(lib().vv_resize_bilinear_f32(_p(src), Hs, Ws, Cc, _p(out), Hd, Wd, _stream()), "vv_resize_bilinear_f32")
    return out


def mask_mem_input(dtype, logits, binarize, scale, bias):
    _need_cuda(logits)
    n = logits.numel()
    out = torch.empty((n, 8), dtype=h16(dtype), device=logits.device)
    _check(lib().vv_mask_mem_input(_p(logits), C.c_int64(n), int(bool(binarize)), _f(scale), _f(bias), _p(out), dtype, _stream()), "vv_mask_mem_input")
    return out


def act_inplace(x, act):
    _need_cuda(x)
    _check(lib().vv_act(_p(x), dt_of(x), C.c_int64(x.numel()), act, _stream()), "vv_act")
    return x


def prompt_points(coords, labels, inv_size, gauss, table):
    _need_cuda(coords, labels, gauss, table)
    P, D = labels.numel(), table.shape[-1]
    out = torch.empty((P, D), dtype=torch.float32, device=coords.device)
    _check(lib().vv_prompt_points(_p(coords), _p(labels), P, _f(inv_size), _p(gauss), _p(table), D, _p(out), _stream()), "vv_prompt_points")
    return out


def sine_pe_1d(pos, dim, temperature=10000.0):
    _need_cuda(pos)
    out = torch.empty((pos.numel(), dim), dtype=torch.float32, device=pos.device)
    _check(lib().vv_sine_pe_1d(_p(pos), pos.numel(), dim, _f(temperature), _p(out), _stream()), "vv_sine_pe_1d")
    return out


def sam_select(masks, iou, obj_logit, multimask, delta, thresh):
    _need_cuda(masks, iou, obj_logit)
    sel = torch.empty(4, dtype=torch.int32, device=masks.device)
    _check(lib().vv_sam_select(_p(masks), masks.shape[-1], masks.shape[0], _p(iou), _p(obj_logit), int(bool(multimask)), _f(delta), _f(thresh), _p(sel),
                               _stream()), "vv_sam_select")
    return sel


def sam_pick(masks, sel, no_obj_score):
    _need_cuda(masks, sel)
    out = torch.empty(masks.shape[-1], dtype=torch.float32, device=masks.device)
    _check(lib().vv_sam_pick(_p(masks), masks.shape[-1], _p(sel), _f(no_obj_score), _p(out), _stream()), "vv_sam_pick")
    return out


def select_f32(a, b, flag):
    _need_cuda(a, b, flag)
    out = torch.empty_like(a)
    _check(lib().vv_select_f32(_p(a), _p(b), _p(flag), C.c_int64(a.numel()), _p(out), _stream()), "vv_select_f32")
    return out


def add_rowvec_unless(x, vec, score):
    """x[m] += vec unless score[0] > 0 (score: fp32 on the device)."""
    _need_cuda(x, vec, score)
    _check(lib().vv_add_rowvec_unless(_p(x), _p(vec), _p(score), C.c_int64(x.shape[0]), x.shape[1], _stream()), "vv_add_rowvec_unless")
    return x


def clamp_f32(x, lo, hi):
    _need_cuda(x)
    out = torch.empty_like(x)
    _check(lib().vv_clamp_f32(_p(x), C.c_int64(x.numel()), _f(lo), _f(hi), _p(out), _stream()), "vv_clamp_f32")
    return out


def fill_holes(mask, H, W, max_area):
    """in place on fp32 [H*W]."""
    _need_cuda(mask)
    ws = torch.empty(3 * H * W, dtype=torch.int32, device=mask.device)
    _check(lib().vv_fill_holes(_p(mask), H, W, max_area, _p(ws), _stream()), "vv_fill_holes")
    return mask


def hyper_masks(hyper, up):
    """masks [nm, HW] = hyper [nm, C] @ up [HW, C]^T, fp32."""
    _need_cuda(hyper, up)
    nm, Cc = hyper.shape
    out = torch.empty((nm, up.shape[0]), dtype=torch.float32, device=up.device)
    _check(lib().vv_hyper_masks(_p(hyper), _p(up), up.shape[0], Cc, nm, _p(out), _stream()), "vv_hyper_masks")
    return out


def attention_split_kv(dtype, q, k, v, out, *, heads, Nq, Nkv, D, S, q_rs, k_rs, v_rs, o_rs, q_hs=0, k_hs=0, v_hs=0, scale=None):
    """one batch of `heads` heads over a LONG key sequence, the keys split into S equal chunks that run as S batches of vv_attention (S x the blocks)
    and are merged through their log-sum-exps (vv_attention_merge).  Nkv % S == 0.  q / k / v / out: row-major [N, heads * D]-style matrices."""
    if Nkv % S:
        raise RuntimeError(f"attention_split_kv: Nkv={Nkv} is not a multiple of S={S}")
    chunk = Nkv // S
    parts = torch.empty((S, Nq, o_rs), dtype=h16(dtype), device=q.device)
    lse = torch.empty((S, heads, Nq), dtype=torch.float32, device=q.device)
    attention(dtype, q, k, v, parts, B=S, heads=heads, Nq=Nq, Nkv=chunk, D=D, q_bs=0, k_bs=chunk * k_rs, v_bs=chunk * v_rs, o_bs=Nq * o_rs,
              q_rs=q_rs, k_rs=k_rs, v_rs=v_rs, o_rs=o_rs, q_hs=q_hs, k_hs=k_hs, v_hs=v_hs, scale=scale, lse=lse)
    _check(lib().vv_attention_merge(_p(parts), _p(lse), S, heads, Nq, D, o_rs, _p(out), dtype, _stream()), "vv_attention_merge")
    return out


def ycbcr_to_rgb(y, cb, cr, hshift, vshift, full_range=False):
    """planar uint8 YCbCr on the device (y [T, H, W], cb / cr [T, ceil(H >> vshift), ceil(W >> hshift)]) -> RGB uint8 [T, H, W, 3] (row n3)."""
    _need_cuda(y, cb, cr)
    T, H, W = y.shape
    out = torch.empty((T, H, W, 3), dtype=torch.uint8, device=y.device)
    _check(lib().vv_ycbcr_to_rgb(_p(y), _p(cb), _p(cr), T, H, W, int(hshift), int(vshift), int(bool(full_range)), _p(out), _stream()), "vv_ycbcr_to_rgb")
    return out


def sam2_maskdown(dtype, logits, lo, S, binarize, scale, bias, l1, l2, eps=1e-6):
    """fused front of the SAM 2 memory encoder's mask path (vv_sam2_maskdown): logits fp32 [lo*lo] -> h16 [(S/4)^2, 16].  l1 / l2: (w, b, gamma, beta)."""
    _need_cuda(logits, *l1, *l2)
    mid = torch.empty(((S // 2) ** 2, 8), dtype=h16(dtype), device=logits.device)
    out = torch.empty(((S // 4) ** 2, 16), dtype=h16(dtype), device=logits.device)
    _check(lib().vv_sam2_maskdown(_p(logits), lo, S, int(bool(binarize)), _f(scale), _f(bias), _p(l1[0]), _p(l1[1]), _p(l1[2]), _p(l1[3]), _p(l2[0]), _p(l2[1]),
                                  _p(l2[2]), _p(l2[3]), _f(eps), _p(mid), _p(out), dtype, _stream()), "vv_sam2_maskdown")
    return out
