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
