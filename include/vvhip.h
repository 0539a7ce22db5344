/* vvhip.h -- C ABI of libvvhip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the VideoVanish
 * DiffuEraser hot path.
 *
 * The reference (calledit/VideoVanish) has no FFI of its own: its hot path `run_infill_on_frames`
 * (reference diffuerase.py:20-114) calls third-party PyTorch modules (`DiffuEraser.forward`,
 * diffuerase.py:62-67; `Propainter.forward`, diffuerase.py:52-57) whose arithmetic torch dispatches to
 * cuDNN / SDPA library kernels.  Each entry point below replaces one such library-op class on that path
 * (SURVEY.md section 2.1 table, section 8b "C-ABI the replacement must export"); the reference-side call that
 * ends up in it is cited per function.  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc'd; the Python host passes tensor.data_ptr()) unless it is
 *     named host_*;  no ownership is transferred, nothing is allocated or freed inside;
 *   - every launcher is asynchronous on `stream` (a hipStream_t passed as void*; 0 = default stream) and is
 *     safe to capture into a hipGraph (no sync, no allocation);
 *   - return value: 0 = launched, negative = VV_E_* (nothing launched); vv_last_error() gives the message of the
 *     calling thread's last failure; no C++ exception ever crosses this boundary;
 *   - activations are frames-major NHWC: [F][H][W][C] == a row-major [M = F*H*W][C] matrix;
 *   - "h16" = the 16-bit MFMA operand type selected by `dtype` (VV_BF16 or VV_F16); accumulation, norm
 *     statistics, softmax and the residual trunk are fp32.
 *
 * Deliberate deviation from the sketch in SURVEY.md section 8(b): the ABI is STATELESS.  There is no
 * vv_init / vv_destroy / vv_ctx -- nothing here owns a device, a stream, a workspace or a communicator, so
 * there is nothing a context would hold: the caller passes the stream and every workspace per call, and the
 * error string is thread-local (vv_last_error).  There is no vv_exchange_overlap either: the overlap exchange
 * at blend time is the host's torch.distributed point-to-point (backend "nccl" = RCCL over xGMI;
 * videovanish_amd/pipeline.py), the only place the path communicates, and it moves whole decoded frames
 * that no kernel of this library needs to see in flight.
 */
#ifndef VVHIP_H
#define VVHIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VV_ABI_VERSION 10

enum { VV_BF16 = 0, VV_F16 = 1, VV_F32 = 2, VV_U8 = 3,
       VV_SPLIT3 = 16 /* vv_groupnorm out_dtype only: the K-concatenated split-precision operand [rows][3C] of vv_split3, in the operand dtype */ };
enum { VV_OK = 0, VV_E_ARG = -1, VV_E_UNSUPPORTED = -2, VV_E_LAUNCH = -3 };
enum { VV_EPI_NONE = 0, VV_EPI_GEGLU = 1 };
enum { VV_ACT_NONE = 0, VV_ACT_SILU = 1, VV_ACT_RELU = 2, VV_ACT_LRELU = 3 /* x > 0 ? x : act_slope * x */, VV_ACT_GELU = 4 /* exact (erf) */, VV_ACT_SIGMOID = 5 };

int vv_abi_version(void);
const char* vv_last_error(void);
/* Number of HIP devices visible / name of device `dev` (host-side probe; fails loudly with VV_E_LAUNCH). */
int vv_device_count(void);
int vv_device_name(int dev, char* host_buf, int buflen);

/* ------------------------------------------------------------------------------------------------------------
 * K1/K6  implicit-GEMM convolution / linear layer on MFMA (v_mfma_f32_16x16x32_{bf16,f16}).
 *   out[m][n] = epi( sum_k A[m][k] * W[n][k] + bias[n] + rowvec[f(m)][n] + res0[m][n] + res1[m][n] )
 * A is gathered on the fly (im2col in LDS) from one or two NHWC sources (channel concat), optionally through a
 * nearest-neighbour resize (fused upsample) and with zero padding;  k = (ky*ks + kx)*Cin + c.
 * Replaces: torch conv2d / linear inside UNet+BrushNet ResBlocks, transformer projections, GEGLU feed-forward,
 * VAE convs (all reached from reference diffuerase.py:62-67 -> DiffuEraser.forward).
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct {
    const void* in0;      /* NHWC source 0 */
    const void* in1;      /* NHWC source 1 (channel concat after in0) or NULL */
    int32_t in_dtype;     /* VV_BF16/VV_F16 (must equal dtype) or VV_F32 (converted while staging) */
    int32_t C0, C1;       /* channels of in0 / in1; Cin = C0 + C1; both multiples of 8 */
    int32_t F;            /* frames (outermost dim) */
    int32_t Hin, Win;     /* stored spatial size of the sources */
    int32_t Hv, Wv;       /* virtual size after nearest resize (== Hin,Win when no resize) */
    int32_t Hout, Wout;   /* output spatial size; M = F*Hout*Wout */
    int32_t ksize;        /* kernel height: 1, 2, 3, 5 or 7 (2: explicit pad_t / pad_l, e.g. stride 2 without padding) */
    int32_t stride;       /* 1 or 2 */
    int32_t pad_t, pad_l; /* top/left zero padding (bottom/right implied by bounds) */
    const void* weight;   /* h16 [Npad][Kpad], Kpad % 64 == 0, rows >= N are zero, Npad % tileN == 0 */
    int32_t N;            /* real output channels */
    int32_t K;            /* real reduction length = ksize*ksize*Cin */
    int32_t Kpad;         /* leading dimension of weight */
    int32_t Npad;         /* padded rows of weight */
    const float* bias;    /* [N] or NULL */
    const float* rowvec;  /* [F][N] per-frame additive vector (time embedding) or NULL */
    const void* res0;     /* [M][N] residual or NULL */
    const void* res1;     /* [M][N] second residual or NULL */
    int32_t res_dtype;    /* VV_F32 or h16, both residuals */
    void* out;            /* [M][Nout], Nout = N (or N/2 for GEGLU), leading dimension ldo */
    int32_t out_dtype;    /* VV_F32 or h16 */
    int32_t ldo;
    int32_t epilogue;     /* VV_EPI_* ; GEGLU expects weight rows interleaved in blocks of 16: [v0..15 g0..15 v16..] */
    float out_scale;      /* multiplies the accumulated product+bias before residuals (1.0f = none) */
    int32_t ksize_w;      /* kernel width (0 = same as ksize); k = (ky*ksize_w + kx)*Cin + c */
    int32_t act;          /* VV_ACT_NONE, VV_ACT_RELU or VV_ACT_LRELU (slope act_slope) applied last (after residuals) */
    int32_t split_heads;  /* > 0: head-major store for a fused QKV projection (h16 out, no residuals): column n = (which, head, d) with
                             N = 3 * split_heads * split_dim, row m = (b, token) with split_tokens rows per b;
                             out[(((b*3 + which)*split_heads + head)*split_tokens + token)*split_dim + d]; ldo is ignored */
    int32_t split_dim;    /* head dim (multiple of 4) */
    int32_t split_tokens; /* rows per batch element: row m = b*split_tokens + token; NEGATIVE: -split_tokens tokens per batch element
                             with token-major rows, m = token*(M/tokens) + b (temporal attention: token = frame, b = pixel) */
    int32_t tile_hint;    /* 0 = automatic tiling; 1 = 128-row tiles only; 2 / 3 = the 256-row tile kernel (2-phase / 8-phase form) whenever
                             the shape is eligible (h16 sources with channel counts % 64 == 0, <= 9 taps, no fused resize,
                             Npad % 320 or % 256 == 0; 8-phase: Npad % 256 == 0) */
    float act_slope;      /* VV_ACT_LRELU: negative-side slope (ProPainter: 0.1 in the alignment offset stacks, 0.2 in the encoders) */
    /* ABI 9 -- output scatter (sc_oh > 0): row m = (f, y, x) of this launch's Hout x Wout grid is stored at (and its residuals are read from) row
       (f * sc_oh + y * sc_sy + sc_oy) * sc_ow + x * sc_sx + sc_ox of an [F][sc_oh][sc_ow][ldo] tensor.  One 3x3 convolution over a nearest-2x
       upsampled image is four 2x2 convolutions over the source image, one per output parity, with the taps that fall on the same source pixel
       summed (4/9 of the work): each of the four launches scatters with sc_sy = sc_sx = 2, (sc_oy, sc_ox) = its parity.  Not with GEGLU /
       split_heads / rowvec; runs on the 128-row tiles whatever tile_hint says. */
    int32_t sc_oh, sc_ow, sc_sy, sc_sx, sc_oy, sc_ox;
    /* ABI 10 -- GroupNorm statistics out of the producing layer (round 6; optional, NULL = off): per-channel partial sums of the layer's FINAL output (after
       bias, scale and residual) for the GroupNorm that reads it next, so that GroupNorm needs no statistics pass of its own over HBM.
       gn_partials: fp32 [F][nblk][N][2] = (sum, sum of squares) over the rows of one row block, nblk = vv_conv_gn_partial_blocks(Hout, Wout) row blocks per
       frame (a row block = the 64 rows one wave of an 8 x 16-pixel tile owns: it never straddles a frame).  Every slot is written by exactly one wave in a fixed
       order (deterministic; rows outside the image contribute zero); vv_gn_finalize_partials sums them in double.  Honoured ONLY by the 128 x 160 halo-tile 3x3
       kernel with the staged fp32 epilogue (3x3, stride 1, h16 source, fp32 output, at most the fp32 residual, no rowvec / act / scatter, N % 160 == 0): any
       other launch with gn_partials != NULL fails with VV_E_UNSUPPORTED rather than leave the buffer unwritten. */
    float* gn_partials;
} vv_conv_params;
int vv_conv_gemm(const vv_conv_params* host_p, int dtype, void* stream);
int vv_conv_gn_partial_blocks(int Hout, int Wout);
/* (mean, rstd) [F][groups][2] -- the layout vv_gn_affine / vv_gn_affine_frames and vv_groupnorm_apply_fin read -- from per-channel partials [F][nblk][C][2];
   pool_frames = 1: one (mean, rstd) per group over the whole clip, replicated per frame.  Double accumulation in a fixed order. */
int vv_gn_finalize_partials(const float* partials, int F, int nblk, int C, int HW, int groups, float eps, int pool_frames, float* fin, void* stream);


/* ------------------------------------------------------------------------------------------------------------
 * K2  GroupNorm (+SiLU) and LayerNorm (+positional embedding), fp32 statistics, h16 output.
 * Replaces torch group_norm / layer_norm / silu in every ResBlock, transformer and motion module.
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct {
    const void* in0; const void* in1;   /* NHWC sources (channel concat) */
    int32_t in_dtype;                   /* VV_F32 or h16 */
    int32_t C0, C1;                     /* multiples of 4 */
    int32_t F, HW;                      /* frames, pixels per frame */
    int32_t groups;
    int32_t pool_frames;                /* 1: statistics pooled over all F frames (motion module norm) */
    float eps;
    const float* gamma; const float* beta;   /* [C] */
    int32_t silu;                       /* activation after the affine: VV_ACT_NONE / VV_ACT_SILU / VV_ACT_RELU */
    float* stats_ws;                    /* workspace: F * (nsplit + 1) * groups * 2 floats, nsplit = vv_groupnorm_nsplit() */
    void* out; int32_t out_dtype;       /* [F*HW][C] h16 (or fp32) */
} vv_groupnorm_params;
int vv_groupnorm_nsplit(int HW, int C);
/* the APPLY pass of vv_groupnorm alone, statistics taken from `fin` ([F][groups][2] mean, rstd) instead of a statistics pass (p.stats_ws is not used) */
int vv_groupnorm_apply_fin(const vv_groupnorm_params* host_p, const float* fin, int dtype, void* stream);
int vv_groupnorm(const vv_groupnorm_params* host_p, int dtype, void* stream);

/* statistics half of vv_groupnorm only (out / gamma / beta unused): leaves (mean, rstd) per (frame, group) in
 * stats_ws[(F * nsplit + f) * groups * 2 ...]; consumed by vv_gn_affine for the fused motion module */
int vv_groupnorm_stats(const vv_groupnorm_params* host_p, int dtype, void* stream);
/* out[0][c] = rstd_g * gamma_c, out[1][c] = beta_c - mean_g * out[0][c]  (mean_rstd: [groups][2] device floats) */
int vv_gn_affine(const float* mean_rstd, const float* gamma, const float* beta, int C, int groups, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * K5 fused: a whole AnimateDiff temporal transformer ("motion module": GroupNorm apply, proj_in, 2 x (LayerNorm + positional
 * embedding, per-head QKV, attention over the F frames of a pixel, out projection), LayerNorm + GEGLU feed-forward, proj_out,
 * residuals) in ONE kernel for C = 320, F = 32, 8 heads: one wave owns one pixel's 32 tokens end to end (trunk and activations in
 * registers), the weights stream through an LDS ring.  Replaces the 17 launches of the unfused block (reference diffuerase.py:62-67
 * -> DiffuEraser.forward -> UNetMotionModel motion modules; SURVEY row a5.5 / north_star "temporal attention").
 *   stream : the module's weights as 670 pre-swizzled [64 x 64] h16 slabs in consumption order (packing.pack_motion_stream)
 *   params : fp32 block (biases, LayerNorm affine, GEGLU bias in chunk order, positional table), n_params floats
 *   gn_affine : [2][320] per-call GroupNorm scale / shift (vv_groupnorm_stats + vv_gn_affine)
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct {
    const float* x;         /* fp32 [F*HW][320] trunk input (frames-major) */
    const float* res1;      /* optional second residual, same shape, or NULL */
    void* out;              /* [F*HW][320] fp32 or h16 */
    int32_t out_dtype;
    const void* stream;     /* h16 weight slabs */
    const float* params;
    const float* gn_affine;
    int32_t C, F, heads, HW;
    int32_t n_slabs, n_params;
} vv_motion_params;
int vv_motion_module_c320(const vv_motion_params* host_p, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Fused tail of a spatial Transformer2D block at C = 320, 8 heads, 77 text tokens (the level-0 blocks of the UNet and of BrushNet):
 *   t = t_in + Wo1 o + bo1;  t += Wo2 CrossAttn(LN2(t), text K/V) + bo2;  t += FF(LN3(t));  out = x [+ res1] + Wout t + bout
 * (diffusers BasicTransformerBlock after its self-attention core + Transformer2DModel.proj_out) in ONE kernel per 128 tokens:
 * replaces 9 launches of vv_conv_gemm / vv_layernorm / vv_attention and their intermediates in HBM.  o: h16 [M][320] (the self-attention core's
 * output), t_in: fp32 [M][320] (proj_in output = the block's residual stream), x: fp32 [M][320] (the Transformer2D input), res1: optional fp32.
 * stream / params: packing.pack_chain_stream (462 slabs of [64][64] h16 incl. the per-head text K and V^T; 5120 floats).
 * ABI 10 -- `layout` names the ORDER of the slabs inside the stream; it must be the one the library's kernel consumes (VV_CHAIN_LAYOUT_ROWSPLIT for
 * the product build), else VV_E_ARG: the three orders have the same slab and parameter counts, so a size check cannot tell them apart (a stream packed
 * the pre-round-5 way would compute wrong activations silently).  Order of VV_CHAIN_LAYOUT_ROWSPLIT, every slab [64 rows][64 k] h16, k pre-permuted
 * (packing._permute_k) and pre-swizzled for the LDS ring:
 *   25 slabs  Wo1 (5 row blocks x 5 k tiles)
 *   4 x head pair (h, h+1):  for each of the two heads: 5 slabs Wq2[head] (40 -> 48 rows) | 2 slabs text K_h (77 keys -> 64 + 16 rows) | 2 slabs text
 *             V_h^T (40 -> 48 rows x 128 keys);  then 5 slabs Wo2[:, head h] and 5 slabs Wo2[:, head h+1] (k = the head's 40 channels -> 64)
 *   20 x 64-unit GEGLU chunk c:  10 slabs W1 rows [value | gate] of the chunk's hidden-unit row tiles in the order (0, 2, 1, 3) (row tile i = units
 *             64 c + 16 i .. + 16: 16 value rows then their 16 gate rows), then 5 slabs W2[:, 64 c .. 64 c + 64]
 *   25 slabs  Wout (proj_out)
 * params (fp32): bo1 | ln2.g | ln2.b | bo2 | ln3.g | ln3.b | b1 in the W1 row order above (2560) | b2 | bout.
 * ------------------------------------------------------------------------------------------------------------ */
#define VV_CHAIN_LAYOUT_TOKENS   0   /* lab kernels only (-DVV_CHAIN_FORM=0): per head q K V^T Wo, GEGLU row tiles (0, 1, 2, 3) -- the order of ABI <= 9 before round 5 */
#define VV_CHAIN_LAYOUT_ROWSPLIT 1   /* the product kernel */
#define VV_CHAIN_LAYOUT_COLUMNS  2   /* lab kernel only (-DVV_CHAIN_FORM=2): per-wave fragment streams, 3480 slabs */
typedef struct {
    const void* o; const float* t_in; const float* x; const float* res1;
    void* out; int32_t out_dtype;
    const void* stream; const float* params;
    int64_t M;
    int32_t C, heads, text_len;
    int32_t n_slabs, n_params;
    int32_t layout;                   /* VV_CHAIN_LAYOUT_* of `stream` (ABI 10) */
    int32_t o_hw;                     /* ABI 10: 0 = o is row-major [M][320]; > 0 = o is HEAD-MAJOR [M / o_hw][8][o_hw][40] (tokens per frame = o_hw;
                                         what vv_attention writes with o_hs = o_hw * 40) */
} vv_chain_params;
int vv_spatial_chain_c320(const vv_chain_params* host_p, int dtype, void* stream);

/* Front half of the same block (everything before the self-attention core): t = Win GN(x) + bin (GroupNorm APPLY with per-frame statistics,
 * proj_in); qkv = Wqkv LN1(t), stored head-major [frame][q|k|v][head][token][40] (what vv_attention reads with q_rs = 40, q_hs = HW * 40; the
 * query rows of the packed weights carry scale * log2 e: q_prescaled = 1).  gn_affine: [F][2][320] per-frame scale / shift from
 * vv_groupnorm_stats + vv_gn_affine_frames.  stream / params: packing.pack_chain_front_stream (100 slabs; 960 floats).
 * Replaces vv_groupnorm's apply pass, two vv_conv_gemm and one vv_layernorm of nn.SpatialTransformer. */
typedef struct {
    const float* x; const float* gn_affine;
    float* t_out; void* qkv;
    const void* stream; const float* params;
    int64_t M; int32_t HW;
    int32_t C, heads;
    int32_t n_slabs, n_params;
} vv_chain_front_params;
int vv_spatial_chain_front_c320(const vv_chain_front_params* host_p, int dtype, void* stream);
/* (mean, rstd) [F][groups][2] of a per-frame GroupNorm -> per-channel scale / shift [F][2][C] */
int vv_gn_affine_frames(const float* mean_rstd, const float* gamma, const float* beta, int C, int groups, int F, float* out, void* stream);

/* LayerNorm over the last dim of a [M][C] fp32 matrix, eps 1e-5; out = LN(x)*gamma+beta (+ pe[(m / rows_per_frame)][c]). */
int vv_layernorm(const float* x, int M, int C, const float* gamma, const float* beta, const float* pe,
                 int rows_per_frame, void* out, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * K3/K4/K5  fused attention: online-softmax flash attention on MFMA with LDS-staged K/V tiles
 * (S^T = K Q^T so every lane owns one query column; V read through ds_read_b64_tr_b16).
 *   element (b, h, i, c) of q lives at q[b*q_bs + h*q_hs + i*q_rs + c], q_hs = D when 0  (same for k, v; o: o[b*o_bs + h*D + i*o_rs + c])
 * spatial self-attention : b = frame, i = pixel (rows of the fused QKV GEMM output, row stride 3C)
 * cross-attention        : K/V have Nkv = 77 rows, kv batch stride 0 (one text tensor for all frames)
 * temporal attention     : b = pixel, i = frame  (q_bs = 3C, q_rs = HW*3C): the (f,hw)->(hw,f) gather is fused
 * VAE mid-block attention: heads = 1, D = 512
 * Replaces torch scaled_dot_product_attention in Transformer2D / motion modules / VAE.
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct {
    const void* q; const void* k; const void* v; void* o;   /* h16 */
    int64_t q_bs, k_bs, v_bs, o_bs;   /* batch strides (elements) */
    int64_t q_rs, k_rs, v_rs, o_rs;   /* row strides (elements) */
    int32_t B, heads, Nq, Nkv, D;     /* D in {32,40,64,80,160,512} */
    float scale;                      /* softmax scale (D^-1/2) */
    int64_t q_hs, k_hs, v_hs;         /* head strides (elements) of q / k / v; 0 = D (heads side by side inside a row).  The
                                         head-major QKV layout written by vv_conv_gemm's split_heads store uses
                                         hs = Nq*D, rs = D: a head's K/V rows are contiguous 2*D-byte records */
    int32_t q_prescaled;              /* 1: q already holds scale * log2(e) * Q (folded into the query projection weights by the
                                         caller, so the single h16 rounding of q covers it); the kernel then skips its own scaling.
                                         The d = 40 spatial kernel subtracts the softmax reference maximum on the matrix pipe and
                                         needs Q in that form: with q_prescaled = 0 it rescales (and re-rounds) Q itself */
    float* lse;                       /* optional (ABI 8) [B][heads][Nq] fp32: log2 of sum_k 2^(scale log2e q.k) over THIS call's keys.  With it a
                                         long key sequence is split over the batch index (q_bs = 0, k_bs = v_bs = chunk * row stride) and the
                                         partial outputs are merged by vv_attention_merge: more blocks for one long head (SAM 2 memory attention:
                                         4096 queries x 28736 keys, one head of 256).  Generic kernel only (not the d = 40 spatial form). */
    int64_t o_hs;                     /* ABI 10: head stride (elements) of o; 0 = D (heads side by side inside a row of o_rs elements: o[b*o_bs + i*o_rs + h*D + c]).
                                         HEAD-MAJOR output o[b][head][token][D] (o_hs = Nq*D, o_rs = D, o_bs = heads*Nq*D): every wave stores whole
                                         contiguous 2*D-byte records, 64 tokens = 64 * 2*D contiguous bytes -- in the row layout a head's 80-byte (D = 40)
                                         slice of a 640-byte row is a partial 64-byte HBM burst on both sides (measured 1.6x write traffic).  The
                                         level-0 spatial blocks use it: vv_spatial_chain_c320 reads that layout with vv_chain_params.o_hw > 0. */
} vv_attn_params;
int vv_attention(const vv_attn_params* host_p, int dtype, void* stream);
/* out[q][h*D + c] = sum_s w_s o_parts[s][q][h*D + c] / sum_s w_s, w_s = 2^(lse[s][h][q] - max_s lse): merges S partial attention results
 * (o_parts: h16 [S][Nq][ld], lse: [S][heads][Nq]) into out (h16 [Nq][ld]) */
int vv_attention_merge(const void* o_parts, const float* lse, int S, int heads, int Nq, int D, int ld, void* out, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * K12  scheduler / latent elementwise (fp32).  Replaces scheduler.add_noise / scheduler.step.
 * ------------------------------------------------------------------------------------------------------------ */
/* out = ca*x + cb*y   (add_noise: ca=sqrt(abar), cb=sqrt(1-abar);  generic axpby, in place allowed) */
int vv_axpby_f32(const float* x, const float* y, float ca, float cb, float* out, int64_t n, void* stream);
/* DDIM / TCD update: x0 = (x - sb_t*eps)/sa_t ; out = c_x0*x0 + c_eps*eps + c_z*z (z may be NULL) */
int vv_sched_step(const float* x, const float* eps, const float* z, float sa_t, float sb_t, float c_x0, float c_eps,
                  float c_z, float* out, int64_t n, void* stream);
/* out = x * sigmoid(x) (time-embedding MLP activation), fp32 */
int vv_silu_f32(const float* x, float* out, int64_t n, void* stream);
/* trunk += addend (fp32 += fp32 or h16) */
int vv_add_inplace(float* x, const void* y, int y_dtype, int64_t n, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * K11  uint8 image steps on the path (integer / fixed-point; bit-exact against the oracle).
 * ------------------------------------------------------------------------------------------------------------ */
/* reference diffuerase.py:27-31: any(mask>0 over `ch` channels) then `iters` 3x3-cross dilations (iters<1: fill
 * whole frame iff any pixel set).  masks [T][H][W][ch] u8 -> out [T][H][W] u8 in {0,255}; tmp same size as out. */
int vv_mask_collapse_dilate(const uint8_t* masks, int T, int H, int W, int ch, int iters, uint8_t* out, uint8_t* tmp,
                            int32_t* flags /* [T] workspace */, void* stream);
/* cv2.resize INTER_LINEAR / INTER_NEAREST on uint8 (reference diffuerase.py:73,86), T images [Hs][Ws][ch]. */
int vv_resize_bilinear_u8(const uint8_t* src, int T, int Hs, int Ws, int ch, uint8_t* dst, int Hd, int Wd, void* stream);
int vv_resize_nearest_u8(const uint8_t* src, int T, int Hs, int Ws, int ch, uint8_t* dst, int Hd, int Wd, void* stream);
/* planar 8-bit YCbCr -> RGB24 on the GPU (row n3: colour conversion of decoded frames; replaces the libswscale conversion behind cv2.VideoCapture,
 * reference tools.py:17-21): y [T][H][W], cb / cr [T][ceil(H >> vshift)][ceil(W >> hshift)] -> rgb [T][H][W][3].  BT.601 matrix, limited (16..235) or
 * full range, MPEG-2 4:2:0 chroma siting with bilinear chroma interpolation, integer arithmetic identical to vvio_ycbcr_to_rgb (include/vvio.h). */
int vv_ycbcr_to_rgb(const uint8_t* y, const uint8_t* cb, const uint8_t* cr, int T, int H, int W, int hshift, int vshift, int full_range, uint8_t* rgb,
                    void* stream);
/* reference diffuerase.py:77-112: alpha from the 5x5-chamfer distance transforms of mask / inverse mask (16.16 fixed
 * point, evaluated in a (2R+1)^2 window, R = ceil(feather_px): alpha saturates beyond), then
 * out = clip(rint(alpha*inp + (1-alpha)*orig)).  feather_px <= 0: hard composite. */
int vv_feather_composite(const uint8_t* inpainted, const uint8_t* orig, const uint8_t* mask2d, int T, int H, int W,
                         float feather_px, uint8_t* out, void* stream);
/* windowed 5x5 chamfer distance transform: distance to the nearest ZERO pixel where it is <= R, else a large
 * constant ((INT_MAX>>2)/65536); fp32 */
int vv_chamfer_dt(const uint8_t* bin, int T, int H, int W, int R, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * K9/K10  RAFT correlation lookup + recurrent-update pieces + convex upsampling, bilinear warp, forward/backward
 * consistency and flow-guided fill (reference diffuerase.py:52-57 -> Propainter.forward).
 * The dense parts of RAFT (encoders, the all-pairs correlation f1^T f2 / 16, update-block convs) run on vv_conv_gemm.
 * Layouts: feature maps / flows are NHWC ([h][w][C]); coords are absolute (x, y) pixel positions.
 * ------------------------------------------------------------------------------------------------------------ */
int vv_avgpool2_f32(const float* in, int64_t N, int h, int w, float* out, void* stream);          /* [N][h][w] -> [N][h/2][w/2] */
/* out[n][l*81 + i*9 + j] = bilinear(corr_l[n], x/2^l + i-4, y/2^l + j-4), zeros outside; channels 324..cpad-1 = 0 (h16) */
int vv_corr_lookup(const float* l0, const float* l1, const float* l2, const float* l3, int h, int w, const float* coords,
                   int64_t N, int cpad, void* out, int dtype, void* stream);
/* context encoder output cn [M][256] -> net = tanh(cn[:, :128]) (fp32 + h16), relu(cn[:, 128:]) -> xbuf[:, 0:128] (h16, ld 256) */
int vv_raft_ctx_split(const float* cn, int64_t M, float* net, void* net16, void* xbuf, int dtype, void* stream);
/* flow = coords1 - grid -> flow8 (h16 [M][8]) and xbuf[:, 254:256] */
int vv_raft_flow_prep(const float* coords1, int64_t M, int w, int h, void* flow8, void* xbuf, int dtype, void* stream);   /* M rows = stacked h x w grids (batched pairs) */
int vv_gru_rh(const float* zr, const float* h, int64_t M, void* rh, int dtype, void* stream);        /* rh = sigmoid(zr[:,128:]) * h */
int vv_gru_update(const float* zr, const float* q, int64_t M, float* h, void* h16, int dtype, void* stream); /* h = (1-z)h + z tanh(q) */
int vv_add_flow(float* coords1, const float* dflow, int ld, int64_t M, void* stream);              /* coords1 += dflow[:, 0:2] */
int vv_add_relu_f32(const float* a, const float* b, float* out, int64_t n, void* stream);
int vv_convex_upsample(const float* coords1, const float* mask, int F, int h, int w, float* out, void* stream);   /* F stacked grids -> flow [F][8h][8w][2] */
int vv_fb_valid(const float* f_ab, const float* f_ba, int H, int W, uint8_t* valid, void* stream);

/* Modulated deformable convolution (DCNv2), SURVEY 8(f) row n1: ProPainter's DeformableAlignment modules call
 * torchvision.ops.deform_conv2d(x, offset, weight, bias, stride, padding, dilation, mask) (third-party `propainter`, reached from
 * reference diffuerase.py:52-57).  vv_deform_im2col gathers the deformed, modulated im2col matrix
 *     col[m][k * C + c] = mask[m][g * K + k] * bilinear(x[b, :, :, c], y * stride - pad + ky * dil + dy, x * stride - pad + kx * dil + dx)
 * (m = (b, y, x) output pixel, k = ky * kw + kx, g = c / (C / deform_groups), (dy, dx) = offset[m][2 (g K + k)], [.. + 1]; torchvision's
 * sampling rule: a sample at or beyond -1 / H (W) is 0, neighbours outside the image count as 0) in the conv weights' own k order, and
 * the convolution itself is then vv_conv_gemm with ksize = 1 over K * C input channels.
 * raw != NULL: the DeformableAlignment front end is fused in -- raw is the conv_offset output [M][3 * deform_groups * K] = (o1 | o2 | m):
 * offset = max_residue * tanh(cat(o1, o2)) + flow (flow [M][2] as (dx, dy), may be NULL), mask = sigmoid(m). */
typedef struct {
    const void* x;            /* [B][H][W][C] NHWC, h16 (= dtype) or fp32 */
    int32_t x_dtype;
    const float* offset;      /* [M][2 * deform_groups * K] or NULL when raw is given */
    const float* mask;        /* [M][deform_groups * K] or NULL (no modulation) */
    const float* raw;         /* [M][3 * deform_groups * K] or NULL */
    const float* flow;        /* [M][2] or NULL (raw mode only) */
    float max_residue;        /* raw mode only */
    void* col;                /* [M][K * C] h16 */
    int32_t B, H, W, C, kh, kw, stride, pad, dil, deform_groups, Ho, Wo;
} vv_deform_params;
int vv_deform_im2col(const vv_deform_params* host_p, int dtype, void* stream);

/* Helpers of ProPainter's recurrent flow-completion network (third-party model/recurrent_flow_completion.py, reached from reference
 * diffuerase.py:52-57 through Propainter.forward -> forward_bidirect_flow / combine_flow).
 * vv_fc_input: network input rows (flow * (1 - m) | m | 0 0 0 0 0) fp32 [T][H + 2 pad][W + 2 pad][8], replicate-padded (the first Conv3d has
 *   padding_mode = 'replicate'); flow fp32 [T][H][W][2], mask u8 [T][H][W] (non-zero = hole).
 * vv_upsample2x_bilinear: F.interpolate(scale_factor = 2, mode = 'bilinear', align_corners = True) on NHWC ([B][H][W][C] -> [B][2H][2W][C],
 *   h16 -> h16 or fp32 -> fp32; C % 8 == 0) -- the resize inside ProPainter's `deconv` blocks.
 * vv_flow_combine: combine_flow: out = pred inside the hole, the measured flow outside; pred fp32 rows of ld_pred floats (first two used). */
int vv_fc_input(const float* flow, const uint8_t* mask, int T, int H, int W, int pad, float* out, void* stream);
int vv_upsample2x_bilinear(const void* x, int x_dtype, int B, int H, int W, int C, void* out, int dtype, void* stream);
int vv_flow_combine(const float* pred, int ld_pred, const float* flow, const uint8_t* mask, int64_t npx, float* out, void* stream);

/* Helpers of ProPainter's inpainting generator (third-party model/propainter.py + model/modules/sparse_transformer.py, reached from reference
 * diffuerase.py:52-57 through Propainter.forward with ref_stride / neighbor_length of diffuerase.py:53-54).
 * vv_gather_rows: out[i] = src[idx[i]] (rows of row_bytes, a multiple of 16; idx < 0 -> zeros): window_partition, the rolled neighbour-window
 *   keys, the pooled global keys and the inverse scatter of the sparse window attention are all row gathers with host-built index tables.
 * vv_fold_patches: F.fold of tap-major patch rows [B * fh * fw][k * k * C] onto [B][h][w][C] (overlap-add in gather form); normalise = divide by
 *   the overlap count, gelu = GELU on the result (the fold / normalise / [unfold] / GELU of FusionFeedForward: GELU commutes with the unfold
 *   gather, which is vv_deform_im2col with zero offsets); also SoftComp's fold.
 * vv_flow_down4: F.interpolate(flow, scale_factor = 1/4, 'bilinear', align_corners = False) / 4 (flows for the 1/4-resolution features).
 * vv_gen_compose: acc = first ? img : (acc + img) / 2 with img = ((tanh(pred) + 1) / 2 * 255) inside the hole and the original frame outside
 *   (the comp_frames update of ProPainter's inference loop; pred fp32 rows of ld_pred floats, ori u8 RGB, mask u8, acc fp32 RGB). */
int vv_gather_rows(const void* src, const int32_t* idx, int64_t n, int row_bytes, void* out, void* stream);
int vv_fold_patches(const void* x, int x_dtype, int B, int fh, int fw, int C, int h, int w, int k, int stride, int pad, int normalise, int gelu,
                    void* out, int out_dtype, int dtype, void* stream);
int vv_flow_down4(const float* flow, int T, int H, int W, float* out, void* stream);
/* vv_gen_input: encoder input rows (frame / 127.5 - 1 | mask_in | mask_updated | 0 0 0) fp32 [npx][8] from u8 RGB frames and u8 masks. */
int vv_gen_input(const uint8_t* frames, const uint8_t* mask_in, const uint8_t* mask_updated, int64_t npx, float* out, void* stream);
int vv_gen_compose(const float* pred, int ld_pred, const uint8_t* ori, const uint8_t* mask, int64_t npx, float* acc, int first, void* stream);
/* fill the unknown pixels of frame t (cur_t [H][W][3] fp32, in place) from neighbour nb warped by `flow` (t -> nb) */
int vv_prop_fill(float* cur_t, const float* cur_nb, uint8_t* known_t, const uint8_t* known_nb, const uint8_t* valid,
                 const float* flow, int H, int W, uint8_t* filled_t, void* stream);
int vv_prop_combine(const float* orig, const float* a, const float* b, const uint8_t* fa, const uint8_t* fb, const uint8_t* hole,
                    int H, int W, const float* mean3, uint8_t* out, uint8_t* filled, void* stream);
int vv_masked_sum_u8(const uint8_t* frame, const uint8_t* hole, int64_t npix, unsigned long long* sums4, void* stream);
int vv_u8_to_f32(const uint8_t* in, float* out, int64_t n, void* stream);
int vv_u8_is_zero(const uint8_t* in, uint8_t* out, int64_t n, void* stream);                     /* known map = (hole == 0) */
int vv_raft_prep(const uint8_t* img, int64_t npix, void* out8, int dtype, void* stream);           /* u8 RGB -> h16 [..][8], 2x/255-1 */

/* model-side pre/post */
/* frames u8 [T][H][W][3], mask u8 [T][H][W] -> img (x/127.5-1) and masked img (img*(1-m)) as h16 NHWC with C=8
 * (channels 3..7 zero) ready for the VAE conv_in; either output may be NULL. */
int vv_preprocess(const uint8_t* frames, const uint8_t* mask2d, int T, int H, int W, void* img, void* masked, int dtype,
                  void* stream);
/* BrushNet input: [lat(4) | cond(4) | mask_nearest(1) | 0 x7] -> h16 [F][h][w][16]; mask sampled at (y*fy, x*fx) */
int vv_brushnet_input(const float* lat, const float* cond, const uint8_t* mask2d, int F, int h, int w, int H, int W,
                      void* out16, int dtype, void* stream);
/* fp32 NHWC [..][cin] -> h16 [..][cpad] (zero padded channels), used for conv_in of latents */
int vv_pad_channels(const float* x, int64_t rows, int cin, int cpad, float scale, void* out, int dtype, void* stream);
/* reference temporal windowing: out[f][..] = value[f][..] / count[f], count = windows that covered frame f */
int vv_window_average(const float* value, const float* count, int frames, int64_t per_frame, float* out, void* stream);
/* same, fp32 output (input of a split-precision layer) */
int vv_pad_channels_f32(const float* x, int64_t rows, int cin, int cpad, float scale, float* out, void* stream);
/* split precision operands for the 3-pass "precise" convolutions (VAE decoder): hi = h16(x), lo = h16((x - hi) * lo_scale);
 * x*w ~= hi*wh + (lo*wh)/lo_scale + hi*wl/w_scale, each product one vv_conv_gemm launch accumulated through res0/out_scale */
int vv_split_f32(const float* x, int64_t n, float lo_scale, void* hi16, void* lo16, int dtype, void* stream);
/* the K-concatenated form used by the split-precision ("precise") layers since ABI 7: x [rows][C] (fp32, or h16 = its own hi part) ->
 * out [rows][3C] h16 = [ hi | (x - hi) * 2^4 | hi * 2^-10 ].  With the weights packed as [ wh | wh * 2^-4 | (w - wh) * 2^10 ] over 3C input
 * channels ONE vv_conv_gemm launch accumulates hi*wh + lo*wh + hi*wl in fp32 (was: three launches through res0 / out_scale) */
int vv_split3(const void* x, int x_dtype, int64_t rows, int C, void* out16, int dtype, void* stream);
/* decoded [T][H][W][ld] fp32 (first 3 channels) -> pix01 = clamp(x/2+0.5,0,1) blended into acc:
 * acc = acc*(1-w[t]) + pix*w[t]  (separately rounded fp32 products, no FMA contraction) */
int vv_decode_blend(const float* dec, int ld, const float* w, int T, int64_t HW, float* acc, void* stream);
/* m' = 1-(1-m)(1-blur21(m)); out = rint(255*(pix*m' + orig/255*(1-m'))) u8.  host_taps21: the 21 normalised
 * Gaussian taps (sigma 3.5, cv2.getGaussianKernel(21,0)) as fp32 in HOST memory; tmp: T*H*W floats (device). */
int vv_blur_compose(const float* pix01, const uint8_t* orig, const uint8_t* mask2d, int T, int H, int W,
                    const float* host_taps21, float* tmp, uint8_t* out, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * SAM 2.1 video predictor (reference sam2_masker.py:88-150 drives the third-party `sam2` package): the kernels of that
 * path beyond vv_conv_gemm / vv_attention / vv_gather_rows (videovanish_amd/csrc/vv_sam2.hip; host side: videovanish_amd/sam2_model.py).
 * All activations NHWC ([rows][C]); "h16" = the dtype argument (VV_BF16 / VV_F16).
 * ------------------------------------------------------------------------------------------------------------ */
/* uint8 RGB [H][W][3] -> h16: ((x / 255) - mean[c]) * istd[c] (mean3 / istd3: HOST pointers).  s2d = 1: out [H*W][cpad], channels >= 3 zero.
 * s2d > 1 (space-to-depth): out [(H/s2d)*(W/s2d)][cpad], channel (dy*s2d + dx)*3 + c = pixel (s2d*Y + dy, s2d*X + dx): a k x k stride-s2d convolution
 * becomes a stride-1 convolution over blocks (the patch embedding of the Hiera trunk: 7x7 stride 4 -> 2x2 over 48 channels) */
int vv_u8_normalize(const uint8_t* src, int H, int W, const float* mean3, const float* istd3, void* out, int cpad, int s2d, int dtype, void* stream);
/* LayerNorm over the last dim of fp32 [M][C] (any C), given eps, optional VV_ACT_* after the affine, out h16 or fp32 (LayerNorm2d of NCHW
 * modules is this on NHWC rows) */
int vv_layernorm_ex(const float* x, int64_t M, int C, const float* gamma, const float* beta, float eps, int act, void* out, int out_dtype,
                    int cpad /* output row length >= C (0 = C): columns past C are written as zeros */, int dtype, void* stream);
/* 2x2 / stride-2 max pooling of [B][H][W][C] (fp32 or h16, even H and W); in_bs: elements between input batches (0 = H*W*C) */
int vv_maxpool2x2(const void* x, int x_dtype, int B, int H, int W, int C, int64_t in_bs, void* out, void* stream);
/* rotary encoding in place on an h16 matrix: for rows r < rows_rope the column pairs (col0 + 2i, col0 + 2i + 1), i < D/2, are rotated by
 * cos_sin[(r % n_table)][i] = (cos, sin)   (compute_axial_cis / apply_rotary_enc; r % n_table = rope_k_repeat) */
int vv_rope_apply(void* x, int64_t rows_rope, int ld, int col0, int D, const float* cos_sin, int n_table, int dtype, void* stream);
/* depthwise k x k convolution (zero padding k/2), fp32 [H][W][C], weights [C][k][k] */
int vv_dwconv(const float* x, int H, int W, int C, const float* w, const float* bias, int k, float* out, void* stream);
/* tail of ConvTranspose2d(kernel 2, stride 2): y [h*w][4*C] with columns (dy, dx, c) -> out [2h*2w][C] = act(y + bias (+ add)) */
int vv_pixel_shuffle2(const float* y, const float* bias, const float* add, int h, int w, int C, int act, void* out, int out_dtype, int dtype,
                      void* stream);
/* bilinear resize of fp32 [Hs][Ws][C] with torch semantics (align_corners = False, no antialias) */
int vv_resize_bilinear_f32(const float* src, int Hs, int Ws, int C, float* dst, int Hd, int Wd, void* stream);
/* mask logits -> input of the memory encoder's mask downsampler: (binarize ? (x > 0) : sigmoid(x)) * scale + bias, h16 [n][8] (channel 0) */
int vv_mask_mem_input(const float* logits, int64_t n, int binarize, float scale, float bias, void* out, int dtype, void* stream);
/* in-place activation (VV_ACT_RELU / GELU / SIGMOID / SILU) on fp32 or h16 */
int vv_act(void* x, int x_dtype, int64_t n, int act, void* stream);
/* prompt encoder: out[p] = label < 0 ? table[0] : pe((coords[p] + 0.5) * inv_size) + table[label + 1]; pe(c) = [sin | cos](2 pi (2c - 1) @ gauss);
 * table [5][D] = not_a_point_embed, point_embeddings 0..3 */
int vv_prompt_points(const float* coords, const int32_t* labels, int P, float inv_size, const float* gauss, const float* table, int D, float* out,
                     void* stream);
/* get_1d_sine_pe: out [n][dim] = [sin(pos / t_j) | cos(pos / t_j)] */
int vv_sine_pe_1d(const float* pos, int n, int dim, float temperature, float* out, void* stream);
/* decoder output selection: sel[0] = mask index (multimask: argmax of iou[1..]; else mask 0 if its stability score >= thresh, else that argmax),
 * sel[1] = (obj_logit > 0), sel[2] = output-token index for the object pointer */
int vv_sam_select(const float* masks, int HW, int nm, const float* iou, const float* obj_logit, int multimask, float delta, float thresh,
                  int32_t* sel, void* stream);
/* out [HW] = sel[1] ? masks[sel[0]] : no_obj_score */
int vv_sam_pick(const float* masks, int HW, const int32_t* sel, float no_obj_score, float* out, void* stream);
/* out = flag[0] ? a : b ;  x[m][c] += vec[c] unless score[0] > 0 (no_obj_embed_spatial of the memory encoder) ;  out = clamp(x, lo, hi) */
int vv_select_f32(const float* a, const float* b, const int32_t* flag, int64_t n, float* out, void* stream);
int vv_add_rowvec_unless(float* x, const float* vec, const float* score, int64_t M, int C, void* stream);
int vv_clamp_f32(const float* x, int64_t n, float lo, float hi, float* out, void* stream);
/* memory encoder, mask path, layers 0-4 of MaskDownSampler fused with what feeds them (sam2_base.py::_encode_new_memory): low-res logits [lo][lo]
 * -> bilinear to [S][S] (torch semantics) -> (binarize ? x > 0 : sigmoid) * scale + bias -> conv 3x3 / 2 (1 -> 4) + LayerNorm2d + GELU -> mid h16
 * [(S/2)^2][8] -> conv 3x3 / 2 (4 -> 16) + LayerNorm2d + GELU -> out h16 [(S/4)^2][16].  Weights fp32 [Cout][Cin][3][3]; the S x S intermediate
 * never exists.  Replaces vv_resize_bilinear_f32 + vv_mask_mem_input + 2 x (vv_conv_gemm + vv_layernorm_ex) of the layer-by-layer path. */
int vv_sam2_maskdown(const float* logits, int lo, int S, int binarize, float scale, float bias, const float* w1, const float* b1, const float* g1,
                     const float* be1, const float* w2, const float* b2, const float* g2, const float* be2, float eps, void* mid, void* out, int dtype,
                     void* stream);
/* masks [nm][HW] = hyper [nm][C] @ up [HW][C]^T (fp32; the mask decoder's hypernetwork product), nm <= 8 */
int vv_hyper_masks(const float* hyper, const float* up, int HW, int C, int nm, float* masks, void* stream);
/* fill_holes_in_mask_scores: 8-connected components of (mask <= 0) with area <= max_area are set to 0.1; ws: 3*H*W int32 */
int vv_fill_holes(float* mask, int H, int W, int max_area, int32_t* ws, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VVHIP_H */
