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
