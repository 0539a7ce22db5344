// K11: uint8 image steps of the in-tree part of the path (reference diffuerase.py:27-31, 69-112) and the
// third-party compose.  Integer / fixed-point work: results are bit-exact against oracle/.
// HBM-bound byte kernels: one thread per pixel, coalesced along x.
#include "vv_common.h"
#pragma clang fp contract(off)

namespace {

constexpr int EB = 256;
inline dim3 grid_for(int64_t n) {
    int64_t b = (n + EB - 1) / EB;
    if (b > 16384) b = 16384;
    if (b < 1) b = 1;
    return dim3((unsigned)b);
}

// ---- mask collapse + dilation (reference diffuerase.py:29-30) ------------------------------------------------
__global__ void collapse_kernel(const uint8_t* m, int64_t npix_per, int T, int ch, uint8_t* out, int32_t* flags) {
    const int64_t n = npix_per * T;
    for (int64_t i = blockIdx.x * (int64_t)EB + threadIdx.x; i < n; i += (int64_t)gridDim.x * EB) {
        bool any = false;
        for (int c = 0; c < ch; ++c) any |= m[i * ch + c] > 0;
        out[i] = any ? 255 : 0;
        if (any) atomicOr(&flags[i / npix_per], 1);
    }
}
__global__ void dilate_cross_kernel(const uint8_t* in, uint8_t* out, int T, int H, int W) {
    const int64_t n = (int64_t)T * H * W;
    for (int64_t i = blockIdx.x * (int64_t)EB + threadIdx.x; i < n; i += (int64_t)gridDim.x * EB) {
        const int x = (int)(i % W); const int y = (int)((i / W) % H);
        uint8_t v = in[i];
        if (y > 0) v |= in[i - W];
        if (y < H - 1) v |= in[i + W];
        if (x > 0) v |= in[i - 1];
        if (x < W - 1) v |= in[i + 1];
        out[i] = v;
    }
}
__global__ void fill_by_flag_kernel(uint8_t* out, const int32_t* flags, int64_t npix_per, int T) {
    const int64_t n = npix_per * T;
    for (int64_t i = blockIdx.x * (int64_t)EB + threadIdx.x; i < n; i += (int64_t)gridDim.x * EB) out[i] = flags[i / npix_per] ? 255 : 0;
}
__global__ void copy_u8_kernel(const uint8_t* in, uint8_t* out, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)EB + threadIdx.x; i < n; i += (int64_t)gridDim.x * EB) out[i] = in[i];
}

// ---- cv2.resize on uint8 (legacy fixed-point INTER_LINEAR; INTER_NEAREST) -------------------------------------
__device__ __forceinline__ void lin_coef(int d, int ssize, int dsize, int& s0, int& s1, int& a0, int& a1) {
    const double scale = (double)ssize / (double)dsize;
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
    a0 = __float2int_rn((1.0f - f) * 2048.0f);
    a1 = __float2int_rn(f * 2048.0f);
    s0 = s; s1 = min(s + 1, ssize - 1);
}
__global__ void resize_bilinear_kernel(const uint8_t* src, int T, int Hs, int Ws, int ch, uint8_t* dst, int Hd, int Wd) {
    const int64_t n = (int64_t)T * Hd * Wd;
    for (int64_t i = blockIdx.x * (int64_t)EB + threadIdx.x; i < n; i += (int64_t)gridDim.x * EB) {
        const int x = (int)(i % Wd); const int y = (int)((i / Wd) % Hd); const int t = (int)(i / ((int64_t)Wd * Hd));
        int x0, x1, ax0, ax1, y0, y1, by0, by1;
        lin_coef(x, Ws, Wd, x0, x1, ax0, ax1);
        lin_coef(y, Hs, Hd, y0, y1, by0, by1);
        const uint8_t* r0 = src + ((int64_t)t * Hs + y0) * Ws * ch;
        const uint8_t* r1 = src + ((int64_t)t * Hs + y1) * Ws * ch;
        for (int c = 0; c < ch; ++c) {
            const int h0 = r0[x0 * ch + c] * ax0 + r0[x1 * ch + c] * ax1;
            const int h1 = r1[x0 * ch + c] * ax0 + r1[x1 * ch + c] * ax1;
            int v = (((by0 * (h0 >> 4)) >> 16) + ((by1 * (h1 >> 4)) >> 16) + 2) >> 2;
            v = v < 0 ? 0 : (v > 255 ? 255 : v);
            dst[i * ch + c] = (uint8_t)v;
        }
    }
}
__global__ void resize_nearest_kernel(const uint8_t* src, int T, int Hs, int Ws, int ch, uint8_t* dst, int Hd, int Wd) {
    const int64_t n = (int64_t)T * Hd * Wd;
    const double sy = (double)Hs / Hd, sx = (double)Ws / Wd;
    for (int64_t i = blockIdx.x * (int64_t)EB + threadIdx.x; i < n; i += (int64_t)gridDim.x * EB) {
        const int x = (int)(i % Wd); const int y = (int)((i / Wd) % Hd); const int t = (int)(i / ((int64_t)Wd * Hd));
        const int ys = min((int)floor(y * sy), Hs - 1), xs = min((int)floor(x * sx), Ws - 1);
        for (int c = 0; c < ch; ++c) dst[i * ch + c] = src[(((int64_t)t * Hs + ys) * Ws + xs) * ch + c];
    }
}

// ---- 5x5 chamfer distance (16.16 fixed point; a=1, b=1.4, c=2.1969), windowed closed form ---------------------
constexpr int C_HV = 65536, C_DIAG = 91750, C_LONG = 143976;
constexpr int DIST_BIG = (0x7fffffff >> 2);
__device__ __forceinline__ int chamfer_fixed(int dx, int dy) {
    dx = dx < 0 ? -dx : dx; dy = dy < 0 ? -dy : dy;
    if (dx < dy) { const int tmp = dx; dx = dy; dy = tmp; }
    return dx >= 2 * dy ? dy * C_LONG + (dx - 2 * dy) * C_HV : (dx - dy) * C_LONG + (2 * dy - dx) * C_DIAG;
}
// distance (fixed) from (x,y) to the nearest pixel whose "is-zero" predicate holds, searched in a (2R+1)^2 window.
// want_nonzero=false: nearest pixel with bin==0; true: nearest pixel with bin!=0 (== zero pixel of the inverse)
__device__ __forceinline__ int window_dist(const uint8_t* img, int H, int W, int x, int y, int R, bool want_nonzero) {
    int best = DIST_BIG;
    for (int dy = -R; dy <= R; ++dy) {
        const int yy = y + dy;
        if (yy < 0 || yy >= H) continue;
        for (int dx = -R; dx <= R; ++dx) {
            const int xx = x + dx;
            if (xx < 0 || xx >= W) continue;
            const bool nz = img[(int64_t)yy * W + xx] > 0;
            if (nz == want_nonzero) { const int d = chamfer_fixed(dx, dy); best = d < best ? d : best; }
        }
    }
    return best;
}
__global__ void chamfer_dt_kernel(const uint8_t* bin, int T, int H, int W, int R, float* out) {
    const int64_t n = (int64_t)T * H * W;
    for (int64_t i = blockIdx.x * (int64_t)EB + threadIdx.x; i < n; i += (int64_t)gridDim.x * EB) {
        const int x = (int)(i % W); const int y = (int)((i / W) % H); const int64_t t = i / ((int64_t)W * H);
        const uint8_t* img = bin + t * H * W;
        int d = 0;
        if (img[(int64_t)y * W + x] > 0) { d = window_dist(img, H, W, x, y, R, false); if (d > R * C_HV) d = DIST_BIG; }   // beyond R the window minimum is not guaranteed global
        out[i] = (float)d * (1.0f / 65536.0f);
    }
}
// reference diffuerase.py:77-112
__global__ void feather_composite_kernel(const uint8_t* inp, const uint8_t* orig, const uint8_t* mask, int T, int H, int W, float feather,
                                         int R, uint8_t* out) {
    const int64_t n = (int64_t)T * H * W;
    for (int64_t i = blockIdx.x * (int64_t)EB + threadIdx.x; i < n; i += (int64_t)gridDim.x * EB) {
        const int x = (int)(i % W); const int y = (int)((i / W) % H); const int64_t t = i / ((int64_t)W * H);
        const uint8_t* img = mask + t * H * W;
        const bool inside = img[(int64_t)y * W + x] > 0;
        float alpha;
        if (feather > 0.f) {
            // d_in: distance of masked pixels to the nearest unmasked one; d_out: the converse (0 on the own side)
            const int d = window_dist(img, H, W, x, y, R, !inside);
            const float df = (float)d * (1.0f / 65536.0f);
            const float d_in = inside ? df : 0.f, d_out = inside ? 0.f : df;
            alpha = 0.5f + (d_in - d_out) / (2.0f * feather);
            alpha = fminf(fmaxf(alpha, 0.f), 1.f);
        } else alpha = inside ? 1.f : 0.f;
        const float om = 1.0f - alpha;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float a = alpha * (float)inp[i * 3 + c];
            const float b = om * (float)orig[i * 3 + c];
            float v = rintf(a + b);
            v = fminf(fmaxf(v, 0.f), 255.f);
            out[i * 3 + c] = (uint8_t)v;
        }
    }
}

// ---- 21x21 Gaussian soft-mask compose -----------------------------------------------------------
struct Taps { float k[21]; };
__device__ __forceinline__ int refl101(int i, int n) {
    i = i < 0 ? -i : i;
    if (i >= n) i = 2 * (n - 1) - i;
    return i < 0 ? 0 : (i > n - 1 ? n - 1 : i);
}
__global__ void blur_rows_kernel(const uint8_t* mask, int T, int H, int W, Taps tp, float* tmp) {
    const int64_t n = (int64_t)T * H * W;
    for (int64_t i = blockIdx.x * (int64_t)EB + threadIdx.x; i < n; i += (int64_t)gridDim.x * EB) {
        const int x = (int)(i % W);
        const uint8_t* row = mask + (i - x);
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < 21; ++j) {
            const float m = row[refl101(x + j - 10, W)] > 0 ? 1.f : 0.f;
            acc = acc + tp.k[j] * m;
        }
        tmp[i] = acc;
    }
}
__global__ void blur_cols_compose_kernel(const float* tmp, const float* pix, const uint8_t* orig, const uint8_t* mask, int T, int H, int W,
                                         Taps tp, uint8_t* out) {
    const int64_t n = (int64_t)T * H * W;
    for (int64_t i = blockIdx.x * (int64_t)EB + threadIdx.x; i < n; i += (int64_t)gridDim.x * EB) {
        const int x = (int)(i % W); const int y = (int)((i / W) % H); const int64_t t = i / ((int64_t)W * H);
        const float* plane = tmp + t * H * W;
        float mb = 0.f;
#pragma unroll
        for (int j = 0; j < 21; ++j) mb = mb + tp.k[j] * plane[(int64_t)refl101(y + j - 10, H) * W + x];
        const float m = mask[i] > 0 ? 1.f : 0.f;
        const float mp = 1.0f - (1.0f - m) * (1.0f - mb);
        const float om = 1.0f - mp;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float o = (float)orig[i * 3 + c] * (1.0f / 255.0f);
            const float a = pix[i * 3 + c] * mp;
            const float b = o * om;
            float v = rintf((a + b) * 255.0f);
            v = fminf(fmaxf(v, 0.f), 255.f);
            out[i * 3 + c] = (uint8_t)v;
        }
    }
}

}  // namespace

extern "C" int vv_mask_collapse_dilate(const uint8_t* masks, int T, int H, int W, int ch, int iters, uint8_t* out, uint8_t* tmp,
                                       int32_t* flags, void* stream) {
    if (!masks || !out || !tmp || !flags || T <= 0 || H <= 0 || W <= 0 || ch <= 0) VV_FAIL(VV_E_ARG, "vv_mask_collapse_dilate: bad args");
    hipStream_t st = (hipStream_t)stream;
    const int64_t npp = (int64_t)H * W, n = npp * T;
    if (hipMemsetAsync(flags, 0, sizeof(int32_t) * T, st) != hipSuccess) VV_FAIL(VV_E_LAUNCH, "vv_mask_collapse_dilate: memset failed");
    hipLaunchKernelGGL(collapse_kernel, grid_for(n), dim3(EB), 0, st, masks, npp, T, ch, out, flags);
    if (iters < 1) {
        // scipy semantics: iterate to convergence == the whole frame as soon as one pixel is set (4-connected grid)
        hipLaunchKernelGGL(fill_by_flag_kernel, grid_for(n), dim3(EB), 0, st, out, flags, npp, T);
    } else {
        uint8_t* a = out; uint8_t* b = tmp;
        for (int k = 0; k < iters; ++k) { hipLaunchKernelGGL(dilate_cross_kernel, grid_for(n), dim3(EB), 0, st, a, b, T, H, W); uint8_t* s = a; a = b; b = s; }
        if (a != out) hipLaunchKernelGGL(copy_u8_kernel, grid_for(n), dim3(EB), 0, st, a, out, n);
    }
    VV_CHECK_LAUNCH("vv_mask_collapse_dilate");
    return VV_OK;
}

extern "C" int vv_resize_bilinear_u8(const uint8_t* src, int T, int Hs, int Ws, int ch, uint8_t* dst, int Hd, int Wd, void* stream) {
    if (!src || !dst || T <= 0 || Hs <= 0 || Ws <= 0 || Hd <= 0 || Wd <= 0 || ch <= 0) VV_FAIL(VV_E_ARG, "vv_resize_bilinear_u8: bad args");
    hipLaunchKernelGGL(resize_bilinear_kernel, grid_for((int64_t)T * Hd * Wd), dim3(EB), 0, (hipStream_t)stream, src, T, Hs, Ws, ch, dst, Hd, Wd);
    VV_CHECK_LAUNCH("vv_resize_bilinear_u8");
    return VV_OK;
}

// ---- planar YCbCr (BT.601, limited or full range) -> RGB24: the GPU side of the frame loader.  Integer arithmetic, bit for bit
// the host routine vvio_ycbcr_to_rgb (vv_ffv1.c): chroma in 1/16 units (MPEG-2 4:2:0 siting: co-sited with even luma columns, midway between two
// luma rows; bilinear), 16.16 fixed-point matrix, round to nearest, clamp.  One thread per pixel, 3 output bytes per thread: HBM bound (4.5 B/pixel).
namespace {
__device__ __forceinline__ int ycc_chroma16(const uint8_t* __restrict__ c, int CW, int CH, int X, int Y, int hs, int vs) {
    int x0 = X >> hs, x1 = x0, wx0 = 4, wx1 = 0, y0 = Y >> vs, y1 = y0, wy0 = 4, wy1 = 0;
    if (hs == 1 && (X & 1)) { x1 = x0 + 1 < CW ? x0 + 1 : x0; wx0 = 2; wx1 = 2; }
    if (vs == 1) {
        if (Y & 1) { y1 = y0 + 1 < CH ? y0 + 1 : y0; wy0 = 3; wy1 = 1; }
        else { y1 = y0 > 0 ? y0 - 1 : y0; wy0 = 3; wy1 = 1; }
    }
    return wy0 * (wx0 * c[(int64_t)y0 * CW + x0] + wx1 * c[(int64_t)y0 * CW + x1]) + wy1 * (wx0 * c[(int64_t)y1 * CW + x0] + wx1 * c[(int64_t)y1 * CW + x1]);
}
__global__ __launch_bounds__(EB) void ycbcr_to_rgb_kernel(const uint8_t* __restrict__ y, const uint8_t* __restrict__ cb, const uint8_t* __restrict__ cr, int T, int H,
                                                           int W, int hs, int vs, int full, uint8_t* __restrict__ rgb) {
    const int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x;
    if (i >= (int64_t)T * H * W) return;
    const int X = (int)(i % W), Y = (int)((i / W) % H), t = (int)(i / ((int64_t)W * H));
    const int CW = (W + (1 << hs) - 1) >> hs, CH = (H + (1 << vs) - 1) >> vs;
    const int ky = full ? 65536 : 76309, yoff = full ? 0 : 16;
    const int krv = full ? 91881 : 104597, kgu = full ? 22554 : 25675, kgv = full ? 46802 : 53279, kbu = full ? 116130 : 132201;
    const int yy = 16 * ky * ((int)y[i] - yoff);
    const uint8_t* cbt = cb + (int64_t)t * CW * CH;
    const uint8_t* crt = cr + (int64_t)t * CW * CH;
    const int u = ycc_chroma16(cbt, CW, CH, X, Y, hs, vs) - 2048, v = ycc_chroma16(crt, CW, CH, X, Y, hs, vs) - 2048;
    const int r = (yy + krv * v + (1 << 19)) >> 20, g = (yy - kgu * u - kgv * v + (1 << 19)) >> 20, b = (yy + kbu * u + (1 << 19)) >> 20;
    uint8_t* o = rgb + i * 3;
    o[0] = (uint8_t)min(max(r, 0), 255); o[1] = (uint8_t)min(max(g, 0), 255); o[2] = (uint8_t)min(max(b, 0), 255);
}
}  // namespace

extern "C" int vv_ycbcr_to_rgb(const uint8_t* y, const uint8_t* cb, const uint8_t* cr, int T, int H, int W, int hshift, int vshift, int full_range, uint8_t* rgb,
                               void* stream) {
    if (!y || !cb || !cr || !rgb || T <= 0 || H <= 0 || W <= 0 || hshift < 0 || hshift > 2 || vshift < 0 || vshift > 2) VV_FAIL(VV_E_ARG, "vv_ycbcr_to_rgb: bad args");
    hipLaunchKernelGGL(ycbcr_to_rgb_kernel, grid_for((int64_t)T * H * W), dim3(EB), 0, (hipStream_t)stream, y, cb, cr, T, H, W, hshift, vshift, full_range, rgb);
    VV_CHECK_LAUNCH("vv_ycbcr_to_rgb");
    return VV_OK;
}

extern "C" int vv_resize_nearest_u8(const uint8_t* src, int T, int Hs, int Ws, int ch, uint8_t* dst, int Hd, int Wd, void* stream) {
    if (!src || !dst || T <= 0 || Hs <= 0 || Ws <= 0 || Hd <= 0 || Wd <= 0 || ch <= 0) VV_FAIL(VV_E_ARG, "vv_resize_nearest_u8: bad args");
    hipLaunchKernelGGL(resize_nearest_kernel, grid_for((int64_t)T * Hd * Wd), dim3(EB), 0, (hipStream_t)stream, src, T, Hs, Ws, ch, dst, Hd, Wd);
    VV_CHECK_LAUNCH("vv_resize_nearest_u8");
    return VV_OK;
}

extern "C" int vv_chamfer_dt(const uint8_t* bin, int T, int H, int W, int R, float* out, void* stream) {
    if (!bin || !out || T <= 0 || H <= 0 || W <= 0 || R < 1 || R > 64) VV_FAIL(VV_E_ARG, "vv_chamfer_dt: bad args");
    hipLaunchKernelGGL(chamfer_dt_kernel, grid_for((int64_t)T * H * W), dim3(EB), 0, (hipStream_t)stream, bin, T, H, W, R, out);
    VV_CHECK_LAUNCH("vv_chamfer_dt");
    return VV_OK;
}

extern "C" int vv_feather_composite(const uint8_t* inpainted, const uint8_t* orig, const uint8_t* mask2d, int T, int H, int W, float feather_px,
                                    uint8_t* out, void* stream) {
    if (!inpainted || !orig || !mask2d || !out || T <= 0 || H <= 0 || W <= 0) VV_FAIL(VV_E_ARG, "vv_feather_composite: bad args");
    if (feather_px > 64.f) VV_FAIL(VV_E_UNSUPPORTED, "vv_feather_composite: feather_px %.1f > 64", feather_px);
    const int R = feather_px > 0.f ? (int)ceilf(feather_px) : 0;
    hipLaunchKernelGGL(feather_composite_kernel, grid_for((int64_t)T * H * W), dim3(EB), 0, (hipStream_t)stream, inpainted, orig, mask2d, T, H, W,
                       feather_px, R, out);
    VV_CHECK_LAUNCH("vv_feather_composite");
    return VV_OK;
}

extern "C" int vv_blur_compose(const float* pix01, const uint8_t* orig, const uint8_t* mask2d, int T, int H, int W, const float* host_taps21, float* tmp,
                               uint8_t* out, void* stream) {
    if (!host_taps21) VV_FAIL(VV_E_ARG, "vv_blur_compose: null taps");
    if (!pix01 || !orig || !mask2d || !tmp || !out || T <= 0 || H <= 0 || W <= 0) VV_FAIL(VV_E_ARG, "vv_blur_compose: bad args");
    Taps tp;
    for (int j = 0; j < 21; ++j) tp.k[j] = host_taps21[j];
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = (int64_t)T * H * W;
    hipLaunchKernelGGL(blur_rows_kernel, grid_for(n), dim3(EB), 0, st, mask2d, T, H, W, tp, tmp);
    hipLaunchKernelGGL(blur_cols_compose_kernel, grid_for(n), dim3(EB), 0, st, tmp, pix01, orig, mask2d, T, H, W, tp, out);
    VV_CHECK_LAUNCH("vv_blur_compose");
    return VV_OK;
}
