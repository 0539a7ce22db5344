// K12 + model-side pre/post elementwise kernels (HBM-bound, vectorised where the layout allows).
// Floating-point contraction is OFF in this file: the blend / scheduler arithmetic is specified as separately
// rounded fp32 operations so that single-GPU, multi-GPU and the CPU oracle agree bit for bit.
#include "vv_common.h"
#pragma clang fp contract(off)

namespace {

constexpr int EB = 256;
inline dim3 grid_for(int64_t n, int per = 1) {
    int64_t b = (n + (int64_t)EB * per - 1) / ((int64_t)EB * per);
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return dim3((unsigned)b);
}

__global__ void axpby_kernel(const float* x, const float* y, float ca, float cb, float* out, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)EB + threadIdx.x; i < n; i += (int64_t)gridDim.x * EB) out[i] = ca * x[i] + cb * y[i];
}

__global__ void sched_step_kernel(const float* x, const float* eps, const float* z, float sa_t, float sb_t, float c_x0, float c_eps,
                                  float c_z, float* out, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)EB + threadIdx.x; i < n; i += (int64_t)gridDim.x * EB) {
        const float e = eps[i];
        const float x0 = (x[i] - sb_t * e) / sa_t;
        float v = c_x0 * x0 + c_eps * e;
        if (z) v = v + c_z * z[i];
        out[i] = v;
    }
}

__global__ void silu_kernel(const float* x, float* out, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)EB + threadIdx.x; i < n; i += (int64_t)gridDim.x * EB) out[i] = silu_f(x[i]);
}

template <typename T>
__global__ void add_inplace_kernel(float* x, const void* y, int y_f32, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)EB + threadIdx.x; i < n; i += (int64_t)gridDim.x * EB)
        x[i] = x[i] + (y_f32 ? ((const float*)y)[i] : T::to_f32(((const unsigned short*)y)[i]));
}

template <typename T>
__global__ void preprocess_kernel(const uint8_t* frames, const uint8_t* mask, int64_t npix, unsigned short* img, unsigned short* masked) {
    for (int64_t i = blockIdx.x * (int64_t)EB + threadIdx.x; i < npix; i += (int64_t)gridDim.x * EB) {
        float v[8], w[8];
        const float keep = (mask && mask[i] > 0) ? 0.f : 1.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) { v[c] = (float)frames[i * 3 + c] / 127.5f - 1.0f; w[c] = v[c] * keep; }
#pragma unroll
        for (int c = 3; c < 8; ++c) { v[c] = 0.f; w[c] = 0.f; }
        if (img) *(uint4*)(img + i * 8) = pack8<T>(v);
        if (masked) *(uint4*)(masked + i * 8) = pack8<T>(w);
    }
}

template <typename T>
__global__ void brushnet_input_kernel(const float* lat, const float* cond, const uint8_t* mask, int F, int h, int w, int H, int W, unsigned short* out) {
    const int64_t n = (int64_t)F * h * w;
    for (int64_t i = blockIdx.x * (int64_t)EB + threadIdx.x; i < n; i += (int64_t)gridDim.x * EB) {
        const int x = (int)(i % w); const int64_t r = i / w; const int y = (int)(r % h); const int f = (int)(r / h);
        const int ys = (int)(((int64_t)y * H) / h), xs = (int)(((int64_t)x * W) / w);
        float v[16];
        const float4 a = *(const float4*)(lat + i * 4), b = *(const float4*)(cond + i * 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        v[8] = mask[((int64_t)f * H + ys) * W + xs] > 0 ? 1.f : 0.f;
#pragma unroll
        for (int c = 9; c < 16; ++c) v[c] = 0.f;
        *(uint4*)(out + i * 16) = pack8<T>(v);
        *(uint4*)(out + i * 16 + 8) = pack8<T>(v + 8);
    }
}

template <typename T>
__global__ void pad_channels_kernel(const float* x, int64_t rows, int cin, int cpad, float scale, unsigned short* out) {
    const int64_t n = rows * cpad;
    for (int64_t i = blockIdx.x * (int64_t)EB + threadIdx.x; i < n; i += (int64_t)gridDim.x * EB) {
        const int c = (int)(i % cpad); const int64_t r = i / cpad;
        out[i] = T::from_f32(c < cin ? x[r * cin + c] * scale : 0.f);
    }
}

__global__ void decode_blend_kernel(const float* dec, int ld, const float* w, int T, int64_t HW, float* acc) {
    const int64_t n = (int64_t)T * HW;
    for (int64_t i = blockIdx.x * (int64_t)EB + threadIdx.x; i < n; i += (int64_t)gridDim.x * EB) {
        const float wt = w[i / HW];
        const float om = 1.0f - wt;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float p = dec[i * ld + c] / 2.0f + 0.5f;
            p = fminf(fmaxf(p, 0.f), 1.f);
            const float a = acc[i * 3 + c] * om;
            const float b = p * wt;
            acc[i * 3 + c] = a + b;
        }
    }
}

}  // namespace

extern "C" int vv_axpby_f32(const float* x, const float* y, float ca, float cb, float* out, int64_t n, void* stream) {
    if (!x || !y || !out || n < 0) VV_FAIL(VV_E_ARG, "vv_axpby_f32: bad args");
    if (n == 0) return VV_OK;
    hipLaunchKernelGGL(axpby_kernel, grid_for(n), dim3(EB), 0, (hipStream_t)stream, x, y, ca, cb, out, n);
    VV_CHECK_LAUNCH("vv_axpby_f32");
    return VV_OK;
}

extern "C" int vv_sched_step(const float* x, const float* eps, const float* z, float sa_t, float sb_t, float c_x0, float c_eps, float c_z,
                             float* out, int64_t n, void* stream) {
    if (!x || !eps || !out || n < 0) VV_FAIL(VV_E_ARG, "vv_sched_step: bad args");
    if (n == 0) return VV_OK;
    hipLaunchKernelGGL(sched_step_kernel, grid_for(n), dim3(EB), 0, (hipStream_t)stream, x, eps, z, sa_t, sb_t, c_x0, c_eps, c_z, out, n);
    VV_CHECK_LAUNCH("vv_sched_step");
    return VV_OK;
}

extern "C" int vv_silu_f32(const float* x, float* out, int64_t n, void* stream) {
    if (!x || !out || n < 0) VV_FAIL(VV_E_ARG, "vv_silu_f32: bad args");
    if (n == 0) return VV_OK;
    hipLaunchKernelGGL(silu_kernel, grid_for(n), dim3(EB), 0, (hipStream_t)stream, x, out, n);
    VV_CHECK_LAUNCH("vv_silu_f32");
    return VV_OK;
}

extern "C" int vv_add_inplace(float* x, const void* y, int y_dtype, int64_t n, int dtype, void* stream) {
    if (!x || !y || n < 0) VV_FAIL(VV_E_ARG, "vv_add_inplace: bad args");
    if (y_dtype != VV_F32 && y_dtype != dtype) VV_FAIL(VV_E_ARG, "vv_add_inplace: y_dtype mismatch");
    if (n == 0) return VV_OK;
    if (dtype == VV_BF16) hipLaunchKernelGGL(add_inplace_kernel<BF16>, grid_for(n), dim3(EB), 0, (hipStream_t)stream, x, y, y_dtype == VV_F32, n);
    else if (dtype == VV_F16) hipLaunchKernelGGL(add_inplace_kernel<F16>, grid_for(n), dim3(EB), 0, (hipStream_t)stream, x, y, y_dtype == VV_F32, n);
    else VV_FAIL(VV_E_ARG, "vv_add_inplace: bad dtype");
    VV_CHECK_LAUNCH("vv_add_inplace");
    return VV_OK;
}

extern "C" int vv_preprocess(const uint8_t* frames, const uint8_t* mask2d, int T, int H, int W, void* img, void* masked, int dtype, void* stream) {
    if (!frames || T <= 0 || H <= 0 || W <= 0 || (!img && !masked)) VV_FAIL(VV_E_ARG, "vv_preprocess: bad args");
    if (masked && !mask2d) VV_FAIL(VV_E_ARG, "vv_preprocess: masked output needs a mask");
    const int64_t n = (int64_t)T * H * W;
    if (dtype == VV_BF16) hipLaunchKernelGGL(preprocess_kernel<BF16>, grid_for(n), dim3(EB), 0, (hipStream_t)stream, frames, mask2d, n, (unsigned short*)img, (unsigned short*)masked);
    else if (dtype == VV_F16) hipLaunchKernelGGL(preprocess_kernel<F16>, grid_for(n), dim3(EB), 0, (hipStream_t)stream, frames, mask2d, n, (unsigned short*)img, (unsigned short*)masked);
    else VV_FAIL(VV_E_ARG, "vv_preprocess: bad dtype");
    VV_CHECK_LAUNCH("vv_preprocess");
    return VV_OK;
}

extern "C" int vv_brushnet_input(const float* lat, const float* cond, const uint8_t* mask2d, int F, int h, int w, int H, int W, void* out16,
                                 int dtype, void* stream) {
    if (!lat || !cond || !mask2d || !out16 || F <= 0 || h <= 0 || w <= 0) VV_FAIL(VV_E_ARG, "vv_brushnet_input: bad args");
    const int64_t n = (int64_t)F * h * w;
    if (dtype == VV_BF16) hipLaunchKernelGGL(brushnet_input_kernel<BF16>, grid_for(n), dim3(EB), 0, (hipStream_t)stream, lat, cond, mask2d, F, h, w, H, W, (unsigned short*)out16);
    else if (dtype == VV_F16) hipLaunchKernelGGL(brushnet_input_kernel<F16>, grid_for(n), dim3(EB), 0, (hipStream_t)stream, lat, cond, mask2d, F, h, w, H, W, (unsigned short*)out16);
    else VV_FAIL(VV_E_ARG, "vv_brushnet_input: bad dtype");
    VV_CHECK_LAUNCH("vv_brushnet_input");
    return VV_OK;
}

// reference windowing (third-party DiffuEraser pipeline: per step noise_pred = value / count over the overlapping temporal windows)
__global__ void window_average_kernel(const float* value, const float* count, int64_t per_frame, int64_t n, float* out) {
    for (int64_t i = blockIdx.x * (int64_t)EB + threadIdx.x; i < n; i += (int64_t)gridDim.x * EB) out[i] = value[i] / count[i / per_frame];
}

extern "C" int vv_window_average(const float* value, const float* count, int frames, int64_t per_frame, float* out, void* stream) {
    if (!value || !count || !out || frames <= 0 || per_frame <= 0) VV_FAIL(VV_E_ARG, "vv_window_average: bad args");
    const int64_t n = (int64_t)frames * per_frame;
    hipLaunchKernelGGL(window_average_kernel, grid_for(n), dim3(EB), 0, (hipStream_t)stream, value, count, per_frame, n, out);
    VV_CHECK_LAUNCH("vv_window_average");
    return VV_OK;
}

__global__ void pad_channels_f32_kernel(const float* x, int64_t rows, int cin, int cpad, float scale, float* out) {
    const int64_t n = rows * cpad;
    for (int64_t i = blockIdx.x * (int64_t)EB + threadIdx.x; i < n; i += (int64_t)gridDim.x * EB) {
        const int c = (int)(i % cpad);
        out[i] = c < cin ? x[(i / cpad) * cin + c] * scale : 0.f;
    }
}

extern "C" int vv_pad_channels_f32(const float* x, int64_t rows, int cin, int cpad, float scale, float* out, void* stream) {
    if (!x || !out || rows <= 0 || cin <= 0 || cpad < cin) VV_FAIL(VV_E_ARG, "vv_pad_channels_f32: bad args");
    hipLaunchKernelGGL(pad_channels_f32_kernel, grid_for(rows * cpad), dim3(EB), 0, (hipStream_t)stream, x, rows, cin, cpad, scale, out);
    VV_CHECK_LAUNCH("vv_pad_channels_f32");
    return VV_OK;
}

// split precision (hi + lo): hi = h16(x), lo = h16((x - f32(hi)) * lo_scale).  x ~= hi + lo / lo_scale to ~2^-19 (fp16) relative.
template <typename T>
__global__ void split_f32_kernel(const float* x, int64_t n4, float lo_scale, unsigned short* hi, unsigned short* lo) {
    for (int64_t i = blockIdx.x * (int64_t)EB + threadIdx.x; i < n4; i += (int64_t)gridDim.x * EB) {
        const float4 v = ((const float4*)x)[i];
        const float a[4] = {v.x, v.y, v.z, v.w};
        unsigned short h[4], l[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            h[c] = T::from_f32(a[c]);
            l[c] = T::from_f32((a[c] - T::to_f32(h[c])) * lo_scale);
        }
        ((uint2*)hi)[i] = make_uint2(h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16));
        ((uint2*)lo)[i] = make_uint2(l[0] | ((unsigned)l[1] << 16), l[2] | ((unsigned)l[3] << 16));
    }
}

extern "C" int vv_split_f32(const float* x, int64_t n, float lo_scale, void* hi, void* lo, int dtype, void* stream) {
    if (!x || !hi || !lo || n <= 0 || (n & 3)) VV_FAIL(VV_E_ARG, "vv_split_f32: bad args (n must be a positive multiple of 4)");
    if (dtype == VV_BF16) hipLaunchKernelGGL(split_f32_kernel<BF16>, grid_for(n / 4), dim3(EB), 0, (hipStream_t)stream, x, n / 4, lo_scale, (unsigned short*)hi, (unsigned short*)lo);
    else if (dtype == VV_F16) hipLaunchKernelGGL(split_f32_kernel<F16>, grid_for(n / 4), dim3(EB), 0, (hipStream_t)stream, x, n / 4, lo_scale, (unsigned short*)hi, (unsigned short*)lo);
    else VV_FAIL(VV_E_ARG, "vv_split_f32: bad dtype");
    VV_CHECK_LAUNCH("vv_split_f32");
    return VV_OK;
}

// split precision, K-concatenated form: out[row] = [ hi | (x - hi) * 2^4 | hi * 2^-10 ] (three C-channel groups, h16): against weights packed as
// [ wh | wh * 2^-4 | (w - wh) * 2^10 ] ONE GEMM over 3 C input channels accumulates hi*wh + lo*wh + hi*wl in its fp32 accumulator (the power-of-two
// scales only keep the small parts out of the fp16 subnormal range).  x: fp32 or h16 (h16 is its own hi: the lo group is zero).
template <typename T, typename IN>
__global__ void split3_kernel(const IN* x, int64_t rows, int C, unsigned short* out) {
    const int c4 = C / 4;
    const int64_t n4 = rows * c4;
    for (int64_t i = blockIdx.x * (int64_t)EB + threadIdx.x; i < n4; i += (int64_t)gridDim.x * EB) {
        const int64_t row = i / c4;
        const int c = (int)(i - row * c4) * 4;
        float a[4];
        if constexpr (sizeof(IN) == 4) { const float4 v = *(const float4*)(x + row * C + c); a[0] = v.x; a[1] = v.y; a[2] = v.z; a[3] = v.w; }
        else { const uint2 v = *(const uint2*)(x + row * C + c); a[0] = T::to_f32(v.x & 0xffff); a[1] = T::to_f32(v.x >> 16); a[2] = T::to_f32(v.y & 0xffff); a[3] = T::to_f32(v.y >> 16); }
        unsigned short h[4], l[4], g[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            h[e] = T::from_f32(a[e]);
            const float hf = T::to_f32(h[e]);
            l[e] = T::from_f32((a[e] - hf) * 16.0f);
            g[e] = T::from_f32(hf * 0.0009765625f);
        }
        unsigned short* o = out + row * 3 * C + c;
        *(uint2*)o = make_uint2(h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16));
        *(uint2*)(o + C) = make_uint2(l[0] | ((unsigned)l[1] << 16), l[2] | ((unsigned)l[3] << 16));
        *(uint2*)(o + 2 * C) = make_uint2(g[0] | ((unsigned)g[1] << 16), g[2] | ((unsigned)g[3] << 16));
    }
}

extern "C" int vv_split3(const void* x, int x_dtype, int64_t rows, int C, void* out, int dtype, void* stream) {
    if (!x || !out || rows <= 0 || C <= 0 || (C & 3)) VV_FAIL(VV_E_ARG, "vv_split3: bad args (C must be a positive multiple of 4)");
    if (dtype != VV_BF16 && dtype != VV_F16) VV_FAIL(VV_E_ARG, "vv_split3: bad dtype");
    if (x_dtype != VV_F32 && x_dtype != dtype) VV_FAIL(VV_E_ARG, "vv_split3: input must be fp32 or the operand dtype");
    const auto g = grid_for(rows * (C / 4));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == VV_BF16) {
        if (x_dtype == VV_F32) hipLaunchKernelGGL((split3_kernel<BF16, float>), g, dim3(EB), 0, st, (const float*)x, rows, C, (unsigned short*)out);
        else hipLaunchKernelGGL((split3_kernel<BF16, unsigned short>), g, dim3(EB), 0, st, (const unsigned short*)x, rows, C, (unsigned short*)out);
    } else {
        if (x_dtype == VV_F32) hipLaunchKernelGGL((split3_kernel<F16, float>), g, dim3(EB), 0, st, (const float*)x, rows, C, (unsigned short*)out);
        else hipLaunchKernelGGL((split3_kernel<F16, unsigned short>), g, dim3(EB), 0, st, (const unsigned short*)x, rows, C, (unsigned short*)out);
    }
    VV_CHECK_LAUNCH("vv_split3");
    return VV_OK;
}

extern "C" int vv_pad_channels(const float* x, int64_t rows, int cin, int cpad, float scale, void* out, int dtype, void* stream) {
    if (!x || !out || rows <= 0 || cin <= 0 || cpad < cin) VV_FAIL(VV_E_ARG, "vv_pad_channels: bad args");
    const int64_t n = rows * cpad;
    if (dtype == VV_BF16) hipLaunchKernelGGL(pad_channels_kernel<BF16>, grid_for(n), dim3(EB), 0, (hipStream_t)stream, x, rows, cin, cpad, scale, (unsigned short*)out);
    else if (dtype == VV_F16) hipLaunchKernelGGL(pad_channels_kernel<F16>, grid_for(n), dim3(EB), 0, (hipStream_t)stream, x, rows, cin, cpad, scale, (unsigned short*)out);
    else VV_FAIL(VV_E_ARG, "vv_pad_channels: bad dtype");
    VV_CHECK_LAUNCH("vv_pad_channels");
    return VV_OK;
}

extern "C" int vv_decode_blend(const float* dec, int ld, const float* w, int T, int64_t HW, float* acc, void* stream) {
    if (!dec || !w || !acc || T <= 0 || HW <= 0 || ld < 3) VV_FAIL(VV_E_ARG, "vv_decode_blend: bad args");
    hipLaunchKernelGGL(decode_blend_kernel, grid_for((int64_t)T * HW), dim3(EB), 0, (hipStream_t)stream, dec, ld, w, T, HW, acc);
    VV_CHECK_LAUNCH("vv_decode_blend");
    return VV_OK;
}
