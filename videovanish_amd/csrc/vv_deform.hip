// Deformed im2col for modulated deformable convolution (DCNv2; row n1 of SURVEY 8f: ProPainter's DeformableAlignment calls
// torchvision.ops.deform_conv2d).  out[b,co,y,x] = bias + sum_{ci,k} w[co,ci,k] * m[g,k] * bilinear(in[ci], p_k + off[g,k]) is
// computed as  col = gather(in, off, m)  [M][K*C] h16, tap-major like the conv weights' k order  ->  vv_conv_gemm (1x1, K = kh*kw*C).
// HBM/L2-gather bound: NHWC makes every corner of a (pixel, group, tap) sample one contiguous 8-channel vector; one lane produces one
// 16-byte chunk of a col row, consecutive lanes consecutive chunks (stores fully coalesced, ~4 x 16 B gathered per 16 B written).
#include "vv_common.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void deform_im2col_kernel(const vv_deform_params p, const int64_t nchunk, const int chunks_per_pixel) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= nchunk) return;
    const int64_t m = gid / chunks_per_pixel;
    const int idx = (int)(gid - m * chunks_per_pixel);            // chunk inside the col row: k * (C/8) + c8
    const int C8 = p.C >> 3, K = p.kh * p.kw, cpg = p.C / p.deform_groups;
    const int k = idx / C8, c8 = idx - k * C8, g = (c8 * 8) / cpg;
    const int HoWo = p.Ho * p.Wo;
    const int b = (int)(m / HoWo), r = (int)(m - (int64_t)b * HoWo), oy = r / p.Wo, ox = r - oy * p.Wo;
    const int ky = k / p.kw, kx = k - ky * p.kw;
    const int j = g * K + k;
    float dy, dx, mk = 1.0f;
    if (p.raw) {
        // ProPainter DeformableAlignment: raw = conv_offset output [M][3*dg*K] = (o1 | o2 | mask); offset = max_residue * tanh(cat(o1, o2))
        // + flow flipped to (dy, dx); mask = sigmoid
        const float* rw = p.raw + m * (int64_t)(3 * p.deform_groups * K);
        dy = p.max_residue * tanhf(rw[2 * j]);
        dx = p.max_residue * tanhf(rw[2 * j + 1]);
        if (p.flow) { dy += p.flow[2 * m + 1]; dx += p.flow[2 * m]; }
        mk = 1.0f / (1.0f + __expf(-rw[2 * p.deform_groups * K + j]));
    } else {
        const float* of = p.offset + m * (int64_t)(2 * p.deform_groups * K);
        dy = of[2 * j]; dx = of[2 * j + 1];
        if (p.mask) mk = p.mask[m * (int64_t)(p.deform_groups * K) + j];
    }
    const float py = (float)(oy * p.stride - p.pad + ky * p.dil) + dy;
    const float px = (float)(ox * p.stride - p.pad + kx * p.dil) + dx;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    if (py > -1.0f && py < (float)p.H && px > -1.0f && px < (float)p.W) {          // torchvision: a sample at or beyond -1 / H is 0
        const float fy = floorf(py), fx = floorf(px);
        const int y0 = (int)fy, x0 = (int)fx;
        const float lh = py - fy, lw = px - fx;
        const float wgt[4] = {(1.f - lh) * (1.f - lw), (1.f - lh) * lw, lh * (1.f - lw), lh * lw};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int yy = y0 + (t >> 1), xx = x0 + (t & 1);
            if (yy < 0 || yy > p.H - 1 || xx < 0 || xx > p.W - 1) continue;           // neighbours outside the image count as 0
            const int64_t off = (((int64_t)b * p.H + yy) * p.W + xx) * p.C + c8 * 8;
            float v[8];
            if (p.x_dtype == VV_F32) {
                const float4* s = (const float4*)((const float*)p.x + off);
                *(float4*)&v[0] = s[0]; *(float4*)&v[4] = s[1];
            } else {
                unpack8<T>(*(const uint4*)((const unsigned short*)p.x + off), v);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += wgt[t] * v[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] *= mk;
    *(uint4*)((unsigned short*)p.col + gid * 8) = pack8<T>(acc);
}

}  // namespace

extern "C" int vv_deform_im2col(const vv_deform_params* pp, int dtype, void* stream) {
    if (!pp) VV_FAIL(VV_E_ARG, "vv_deform_im2col: null params");
    const vv_deform_params& p = *pp;
    if (dtype != VV_BF16 && dtype != VV_F16) VV_FAIL(VV_E_ARG, "vv_deform_im2col: bad dtype");
    if (!p.x || !p.col || (!p.raw && !p.offset)) VV_FAIL(VV_E_ARG, "vv_deform_im2col: null pointer (x, col and one of raw / offset are required)");
    if (p.x_dtype != VV_F32 && p.x_dtype != dtype) VV_FAIL(VV_E_ARG, "vv_deform_im2col: x_dtype mismatch");
    if (p.B <= 0 || p.H <= 0 || p.W <= 0 || p.Ho <= 0 || p.Wo <= 0) VV_FAIL(VV_E_ARG, "vv_deform_im2col: empty problem");
    if (p.kh <= 0 || p.kw <= 0 || p.stride <= 0 || p.dil <= 0 || p.pad < 0) VV_FAIL(VV_E_ARG, "vv_deform_im2col: bad kernel geometry");
    if (p.deform_groups <= 0 || p.C <= 0 || p.C % p.deform_groups || (p.C / p.deform_groups) % 8)
        VV_FAIL(VV_E_ARG, "vv_deform_im2col: C=%d must split into deform_groups=%d groups of a multiple of 8 channels", p.C, p.deform_groups);
    if (p.Ho != (p.H + 2 * p.pad - p.dil * (p.kh - 1) - 1) / p.stride + 1 || p.Wo != (p.W + 2 * p.pad - p.dil * (p.kw - 1) - 1) / p.stride + 1)
        VV_FAIL(VV_E_ARG, "vv_deform_im2col: Ho/Wo do not match the geometry");
    const int cpp = p.kh * p.kw * (p.C / 8);
    const int64_t nchunk = (int64_t)p.B * p.Ho * p.Wo * cpp;
    const int64_t nblk = (nchunk + 255) / 256;
    if (nblk > 0x7fffffff) VV_FAIL(VV_E_ARG, "vv_deform_im2col: grid too large");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == VV_BF16) hipLaunchKernelGGL(deform_im2col_kernel<BF16>, dim3((unsigned)nblk), dim3(256), 0, st, p, nchunk, cpp);
    else hipLaunchKernelGGL(deform_im2col_kernel<F16>, dim3((unsigned)nblk), dim3(256), 0, st, p, nchunk, cpp);
    VV_CHECK_LAUNCH("vv_deform_im2col");
    return VV_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Helpers of ProPainter's recurrent flow-completion network (row n1; oracle/flowcomplete_ref.py; videovanish_amd/flowcomplete.py).
namespace {

// network input: (flow * (1 - m) | m | 0 x 5) per pixel, replicate-padded by `pad` pixels (first Conv3d: padding_mode = 'replicate')
__global__ __launch_bounds__(256) void fc_input_kernel(const float* __restrict__ flow, const uint8_t* __restrict__ mask, int T, int H, int W, int pad,
                                                      float* __restrict__ out) {
    const int Hp = H + 2 * pad, Wp = W + 2 * pad;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (int64_t)T * Hp * Wp) return;
    const int t = (int)(gid / (Hp * Wp)), r = (int)(gid - (int64_t)t * Hp * Wp), y = r / Wp, x = r - y * Wp;
    const int cy = min(max(y - pad, 0), H - 1), cx = min(max(x - pad, 0), W - 1);
    const int64_t src = ((int64_t)t * H + cy) * W + cx;
    const float m = mask[src] ? 1.0f : 0.0f;
    float4* o = (float4*)(out + gid * 8);
    o[0] = make_float4(flow[2 * src] * (1.0f - m), flow[2 * src + 1] * (1.0f - m), m, 0.f);
    o[1] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// F.interpolate(scale_factor=2, mode='bilinear', align_corners=True) on NHWC; one lane = 8 channels of one output pixel
template <typename T, bool F32>
__global__ __launch_bounds__(256) void upsample2x_kernel(const void* __restrict__ xin, int B, int H, int W, int C, void* __restrict__ out) {
    const int C8 = C >> 3, Ho = 2 * H, Wo = 2 * W;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (int64_t)B * Ho * Wo * C8) return;
    const int c8 = (int)(gid % C8);
    const int64_t px = gid / C8;
    const int b = (int)(px / (Ho * Wo)), r = (int)(px - (int64_t)b * Ho * Wo), oy = r / Wo, ox = r - oy * Wo;
    const float sy = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f, sx = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
    const float fy = sy * (float)oy, fx = sx * (float)ox;
    const int y0 = (int)fy, x0 = (int)fx, y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const float wgt[4] = {(1.f - ly) * (1.f - lx), (1.f - ly) * lx, ly * (1.f - lx), ly * lx};
    const int ys[4] = {y0, y0, y1, y1}, xs[4] = {x0, x1, x0, x1};
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int64_t off = (((int64_t)b * H + ys[t]) * W + xs[t]) * C + c8 * 8;
        float v[8];
        if (F32) { const float4* s = (const float4*)((const float*)xin + off); *(float4*)&v[0] = s[0]; *(float4*)&v[4] = s[1]; }
        else unpack8<T>(*(const uint4*)((const unsigned short*)xin + off), v);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += wgt[t] * v[e];
    }
    const int64_t o = px * C + c8 * 8;
    if (F32) { float4* d = (float4*)((float*)out + o); d[0] = *(float4*)&acc[0]; d[1] = *(float4*)&acc[4]; }
    else *(uint4*)((unsigned short*)out + o) = pack8<T>(acc);
}

__global__ __launch_bounds__(256) void flow_combine_kernel(const float* __restrict__ pred, int ldp, const float* __restrict__ flow, const uint8_t* __restrict__ mask,
                                                          int64_t npx, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npx) return;
    const bool hole = mask[i] != 0;
    out[2 * i] = hole ? pred[i * ldp] : flow[2 * i];
    out[2 * i + 1] = hole ? pred[i * ldp + 1] : flow[2 * i + 1];
}

}  // namespace

extern "C" int vv_fc_input(const float* flow, const uint8_t* mask, int T, int H, int W, int pad, float* out, void* stream) {
    if (!flow || !mask || !out) VV_FAIL(VV_E_ARG, "vv_fc_input: null pointer");
    if (T <= 0 || H <= 0 || W <= 0 || pad < 0) VV_FAIL(VV_E_ARG, "vv_fc_input: bad shape");
    const int64_t n = (int64_t)T * (H + 2 * pad) * (W + 2 * pad);
    hipLaunchKernelGGL(fc_input_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, flow, mask, T, H, W, pad, out);
    VV_CHECK_LAUNCH("vv_fc_input");
    return VV_OK;
}

extern "C" int vv_upsample2x_bilinear(const void* x, int x_dtype, int B, int H, int W, int C, void* out, int dtype, void* stream) {
    if (!x || !out) VV_FAIL(VV_E_ARG, "vv_upsample2x_bilinear: null pointer");
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8) VV_FAIL(VV_E_ARG, "vv_upsample2x_bilinear: C=%d must be a multiple of 8", C);
    if (dtype != VV_BF16 && dtype != VV_F16) VV_FAIL(VV_E_ARG, "vv_upsample2x_bilinear: bad dtype");
    if (x_dtype != VV_F32 && x_dtype != dtype) VV_FAIL(VV_E_ARG, "vv_upsample2x_bilinear: x_dtype mismatch");
    const int64_t n = (int64_t)B * 4 * H * W * (C / 8);
    const dim3 grid((unsigned)((n + 255) / 256)), blk(256);
    hipStream_t st = (hipStream_t)stream;
    if (x_dtype == VV_F32) hipLaunchKernelGGL((upsample2x_kernel<F16, true>), grid, blk, 0, st, x, B, H, W, C, out);
    else if (dtype == VV_BF16) hipLaunchKernelGGL((upsample2x_kernel<BF16, false>), grid, blk, 0, st, x, B, H, W, C, out);
    else hipLaunchKernelGGL((upsample2x_kernel<F16, false>), grid, blk, 0, st, x, B, H, W, C, out);
    VV_CHECK_LAUNCH("vv_upsample2x_bilinear");
    return VV_OK;
}

extern "C" int vv_flow_combine(const float* pred, int ld_pred, const float* flow, const uint8_t* mask, int64_t npx, float* out, void* stream) {
    if (!pred || !flow || !mask || !out) VV_FAIL(VV_E_ARG, "vv_flow_combine: null pointer");
    if (npx <= 0 || ld_pred < 2) VV_FAIL(VV_E_ARG, "vv_flow_combine: bad shape");
    hipLaunchKernelGGL(flow_combine_kernel, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pred, ld_pred, flow, mask, npx, out);
    VV_CHECK_LAUNCH("vv_flow_combine");
    return VV_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Helpers of ProPainter's inpainting generator (row n1; oracle/inpaintgen_ref.py; videovanish_amd/inpaintgen.py).
namespace {

// out[i] = src[idx[i]] for rows of `row16` 16-byte chunks; idx < 0 -> zero row (window partition / rolled / pooled key gathers)
__global__ __launch_bounds__(256) void gather_rows_kernel(const uint4* __restrict__ src, const int32_t* __restrict__ idx, int64_t n, int row16, uint4* __restrict__ out) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= n * row16) return;
    const int64_t r = gid / row16;
    const int c = (int)(gid - r * row16);
    const int32_t s = idx[r];
    out[gid] = s >= 0 ? src[(int64_t)s * row16 + c] : make_uint4(0, 0, 0, 0);
}

// F.fold of tap-major patch rows x [B * fh * fw][K * C] (k = ky * kw + kx) onto the [B][h][w][C] grid (gather form: a pixel sums the <= ceil(k/s)^2
// patches that cover it), optionally divided by the overlap count and passed through GELU (fusion feed-forward), one lane = 8 channels
template <typename T, bool F32IN, bool F32OUT>
__global__ __launch_bounds__(256) void fold_kernel(const void* __restrict__ xin, int B, int fh, int fw, int C, int h, int w, int kk, int st, int pd,
                                                   int normalise, int gelu, void* __restrict__ out) {
    const int C8 = C >> 3;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (int64_t)B * h * w * C8) return;
    const int c8 = (int)(gid % C8);
    const int64_t px = gid / C8;
    const int b = (int)(px / (h * w)), r = (int)(px - (int64_t)b * h * w), y = r / w, x = r - y * w;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    int cnt = 0;
    // patches (py, px) with py * st - pd <= y < py * st - pd + kk
    const int py_lo = max(0, (y + pd - kk + st) / st), py_hi = min(fh - 1, (y + pd) / st);
    const int px_lo = max(0, (x + pd - kk + st) / st), px_hi = min(fw - 1, (x + pd) / st);
    for (int py = py_lo; py <= py_hi; ++py)
        for (int pxx = px_lo; pxx <= px_hi; ++pxx) {
            const int ky = y + pd - py * st, kx = x + pd - pxx * st;
            if (ky < 0 || ky >= kk || kx < 0 || kx >= kk) continue;
            const int64_t off = (((int64_t)b * fh + py) * fw + pxx) * (int64_t)(kk * kk * C) + (int64_t)(ky * kk + kx) * C + c8 * 8;
            float v[8];
            if (F32IN) { const float4* s = (const float4*)((const float*)xin + off); *(float4*)&v[0] = s[0]; *(float4*)&v[4] = s[1]; }
            else unpack8<T>(*(const uint4*)((const unsigned short*)xin + off), v);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += v[e];
            ++cnt;
        }
    if (normalise && cnt > 0) {
        const float inv = 1.0f / (float)cnt;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] *= inv;
    }
    if (gelu) {
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = gelu_f(acc[e]);
    }
    const int64_t o = px * C + c8 * 8;
    if (F32OUT) { float4* d = (float4*)((float*)out + o); d[0] = *(float4*)&acc[0]; d[1] = *(float4*)&acc[4]; }
    else *(uint4*)((unsigned short*)out + o) = pack8<T>(acc);
}

// F.interpolate(flow, scale_factor = 1/4, bilinear, align_corners = False) / 4: the mean of the 2 x 2 block at (4y+1..4y+2, 4x+1..4x+2), divided by 4
__global__ __launch_bounds__(256) void flow_down4_kernel(const float* __restrict__ f, int T, int H, int W, float* __restrict__ out) {
    const int h = H / 4, w = W / 4;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (int64_t)T * h * w) return;
    const int t = (int)(gid / (h * w)), r = (int)(gid - (int64_t)t * h * w), y = r / w, x = r - y * w;
    const float2* s = (const float2*)f + ((int64_t)t * H + 4 * y + 1) * W + 4 * x + 1;
    const float2 a = s[0], b = s[1], c = s[W], d = s[W + 1];
    ((float2*)out)[gid] = make_float2(((a.x + b.x) * 0.5f * 0.5f + (c.x + d.x) * 0.5f * 0.5f) * 0.25f, ((a.y + b.y) * 0.5f * 0.5f + (c.y + d.y) * 0.5f * 0.5f) * 0.25f);
}

// generator output -> composite: img = uint8((tanh(x) + 1) / 2 * 255) inside the hole, the original outside; second visit of a frame: uint8 mean of both
__global__ __launch_bounds__(256) void gen_compose_kernel(const float* __restrict__ pred, int ldp, const uint8_t* __restrict__ ori, const uint8_t* __restrict__ mask,
                                                         int64_t npx, float* __restrict__ acc, int first) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npx) return;
    const bool hole = mask[i] != 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        // the reference's uint8 arithmetic: the prediction is truncated to uint8, the mean of two visits again
        const float v = hole ? floorf(fminf(fmaxf((tanhf(pred[i * ldp + c]) + 1.0f) * 0.5f * 255.0f, 0.f), 255.f)) : (float)ori[3 * i + c];
        acc[3 * i + c] = first ? v : floorf(acc[3 * i + c] * 0.5f + v * 0.5f);
    }
}

}  // namespace

extern "C" int vv_gather_rows(const void* src, const int32_t* idx, int64_t n, int row_bytes, void* out, void* stream) {
    if (!src || !idx || !out) VV_FAIL(VV_E_ARG, "vv_gather_rows: null pointer");
    if (n <= 0 || row_bytes <= 0 || row_bytes % 16) VV_FAIL(VV_E_ARG, "vv_gather_rows: row_bytes=%d must be a positive multiple of 16", row_bytes);
    const int64_t tot = n * (row_bytes / 16);
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint4*)src, idx, n, row_bytes / 16, (uint4*)out);
    VV_CHECK_LAUNCH("vv_gather_rows");
    return VV_OK;
}

extern "C" int vv_fold_patches(const void* x, int x_dtype, int B, int fh, int fw, int C, int h, int w, int k, int stride, int pad, int normalise, int gelu,
                               void* out, int out_dtype, int dtype, void* stream) {
    if (!x || !out) VV_FAIL(VV_E_ARG, "vv_fold_patches: null pointer");
    if (dtype != VV_BF16 && dtype != VV_F16) VV_FAIL(VV_E_ARG, "vv_fold_patches: bad dtype");
    if ((x_dtype != VV_F32 && x_dtype != dtype) || (out_dtype != VV_F32 && out_dtype != dtype)) VV_FAIL(VV_E_ARG, "vv_fold_patches: dtype mismatch");
    if (B <= 0 || C <= 0 || C % 8 || k <= 0 || stride <= 0 || pad < 0) VV_FAIL(VV_E_ARG, "vv_fold_patches: bad geometry (C=%d must be a multiple of 8)", C);
    if (fh != (h + 2 * pad - k) / stride + 1 || fw != (w + 2 * pad - k) / stride + 1) VV_FAIL(VV_E_ARG, "vv_fold_patches: patch grid does not match the output size");
    const int64_t n = (int64_t)B * h * w * (C / 8);
    const dim3 grid((unsigned)((n + 255) / 256)), blk(256);
    hipStream_t st = (hipStream_t)stream;
#define FOLD(TT, FI, FO) hipLaunchKernelGGL((fold_kernel<TT, FI, FO>), grid, blk, 0, st, x, B, fh, fw, C, h, w, k, stride, pad, normalise, gelu, out)
    const bool fi = x_dtype == VV_F32, fo = out_dtype == VV_F32;
    if (dtype == VV_BF16) { if (fi && fo) FOLD(BF16, true, true); else if (fi) FOLD(BF16, true, false); else if (fo) FOLD(BF16, false, true); else FOLD(BF16, false, false); }
    else { if (fi && fo) FOLD(F16, true, true); else if (fi) FOLD(F16, true, false); else if (fo) FOLD(F16, false, true); else FOLD(F16, false, false); }
#undef FOLD
    VV_CHECK_LAUNCH("vv_fold_patches");
    return VV_OK;
}

extern "C" int vv_flow_down4(const float* flow, int T, int H, int W, float* out, void* stream) {
    if (!flow || !out) VV_FAIL(VV_E_ARG, "vv_flow_down4: null pointer");
    if (T <= 0 || H <= 0 || W <= 0 || H % 4 || W % 4) VV_FAIL(VV_E_ARG, "vv_flow_down4: H=%d W=%d must be multiples of 4", H, W);
    const int64_t n = (int64_t)T * (H / 4) * (W / 4);
    hipLaunchKernelGGL(flow_down4_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, flow, T, H, W, out);
    VV_CHECK_LAUNCH("vv_flow_down4");
    return VV_OK;
}

extern "C" int vv_gen_compose(const float* pred, int ld_pred, const uint8_t* ori, const uint8_t* mask, int64_t npx, float* acc, int first, void* stream) {
    if (!pred || !ori || !mask || !acc) VV_FAIL(VV_E_ARG, "vv_gen_compose: null pointer");
    if (npx <= 0 || ld_pred < 3) VV_FAIL(VV_E_ARG, "vv_gen_compose: bad shape");
    hipLaunchKernelGGL(gen_compose_kernel, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pred, ld_pred, ori, mask, npx, acc, first);
    VV_CHECK_LAUNCH("vv_gen_compose");
    return VV_OK;
}

namespace {
// encoder input rows: (frame / 127.5 - 1 | mask_in | mask_updated | 0 0 0)
__global__ __launch_bounds__(256) void gen_input_kernel(const uint8_t* __restrict__ fr, const uint8_t* __restrict__ m_in, const uint8_t* __restrict__ m_up,
                                                       int64_t npx, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npx) return;
    float4* o = (float4*)(out + i * 8);
    o[0] = make_float4((float)fr[3 * i] / 127.5f - 1.0f, (float)fr[3 * i + 1] / 127.5f - 1.0f, (float)fr[3 * i + 2] / 127.5f - 1.0f, m_in[i] ? 1.0f : 0.0f);
    o[1] = make_float4(m_up[i] ? 1.0f : 0.0f, 0.f, 0.f, 0.f);
}
}  // namespace

extern "C" int vv_gen_input(const uint8_t* frames, const uint8_t* mask_in, const uint8_t* mask_updated, int64_t npx, float* out, void* stream) {
    if (!frames || !mask_in || !mask_updated || !out) VV_FAIL(VV_E_ARG, "vv_gen_input: null pointer");
    if (npx <= 0) VV_FAIL(VV_E_ARG, "vv_gen_input: empty input");
    hipLaunchKernelGGL(gen_input_kernel, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, (hipStream_t)stream, frames, mask_in, mask_updated, npx, out);
    VV_CHECK_LAUNCH("vv_gen_input");
    return VV_OK;
}
