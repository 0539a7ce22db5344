// K9/K10: RAFT correlation pyramid lookup, recurrent-update elementwise pieces, convex upsampling, bilinear warp,
// forward/backward consistency and flow-guided fill (reference call site diffuerase.py:52-57
// -> Propainter.forward).  HBM-bound gather / elementwise kernels; the dense parts of RAFT (encoders, all-pairs
// correlation, update-block convolutions) run on vv_conv_gemm.  FP contraction is off: the warp / consistency /
// fill arithmetic is specified operation by operation so that it matches the oracle bit for bit on equal inputs.
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
#define GRID_STRIDE(i, n) for (int64_t i = blockIdx.x * (int64_t)EB + threadIdx.x; i < (n); i += (int64_t)gridDim.x * EB)

__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + __expf(-x)); }

// ---- correlation pyramid ----------------------------------------------------------------------------------------
__global__ void avgpool2_kernel(const float* in, int64_t N, int h, int w, float* out) {
    const int ho = h / 2, wo = w / 2;
    const int64_t n = N * ho * wo;
    GRID_STRIDE(i, n) {
        const int x = (int)(i % wo); const int y = (int)((i / wo) % ho); const int64_t p = i / ((int64_t)wo * ho);
        const float* s = in + (p * h + 2 * y) * w + 2 * x;
        out[i] = ((s[0] + s[1]) + (s[w] + s[w + 1])) * 0.25f;
    }
}

struct Pyr { const float* lvl[4]; int h[4]; int w[4]; };

__device__ __forceinline__ float tap_zero(const float* plane, int Hp, int Wp, int y, int x) {
    return (x >= 0 && x < Wp && y >= 0 && y < Hp) ? plane[(int64_t)y * Wp + x] : 0.f;
}
__device__ __forceinline__ float bilinear_zero(const float* plane, int Hp, int Wp, float x, float y) {
    const float x0f = floorf(x), y0f = floorf(y);
    const float wx = x - x0f, wy = y - y0f;
    const int x0 = (int)x0f, y0 = (int)y0f;
    const float v00 = tap_zero(plane, Hp, Wp, y0, x0), v01 = tap_zero(plane, Hp, Wp, y0, x0 + 1);
    const float v10 = tap_zero(plane, Hp, Wp, y0 + 1, x0), v11 = tap_zero(plane, Hp, Wp, y0 + 1, x0 + 1);
    const float a = ((1.0f - wx) * (1.0f - wy)) * v00;
    const float b = (wx * (1.0f - wy)) * v01;
    const float c = ((1.0f - wx) * wy) * v10;
    const float d = (wx * wy) * v11;
    return ((a + b) + c) + d;
}

// out[n][l*81 + i*9 + j] = bilinear(corr_l[n], x/2^l + (i-4), y/2^l + (j-4)); channels >= 324 are zero padding
template <typename T>
__global__ void corr_lookup_kernel(const Pyr P, const float* coords, int64_t N, int cpad, unsigned short* out) {
    const int64_t n = N * cpad;
    GRID_STRIDE(i, n) {
        const int c = (int)(i % cpad); const int64_t p = i / cpad;
        float v = 0.f;
        if (c < 324) {
            const int l = c / 81, r = c - l * 81, ii = r / 9, jj = r - ii * 9;
            const float sc = 1.0f / (float)(1 << l);
            const float x = coords[p * 2] * sc + (float)(ii - 4), y = coords[p * 2 + 1] * sc + (float)(jj - 4);
            v = bilinear_zero(P.lvl[l] + p * P.h[l] * P.w[l], P.h[l], P.w[l], x, y);
        }
        out[i] = T::from_f32(v);
    }
}

// ---- recurrent update pieces -------------------------------------------------------------------------------------
// cn [M][256] fp32 -> net = tanh(cn[:, :128]) (fp32 [M][128] + h16 copy), inp = relu(cn[:, 128:]) -> xbuf[:, 0:128] (h16, ld 256)
template <typename T>
__global__ void ctx_split_kernel(const float* cn, int64_t M, float* net, unsigned short* net16, unsigned short* xbuf) {
    const int64_t n = M * 128;
    GRID_STRIDE(i, n) {
        const int c = (int)(i & 127); const int64_t m = i >> 7;
        const float t = tanhf(cn[m * 256 + c]);
        net[i] = t; net16[i] = T::from_f32(t);
        xbuf[m * 256 + c] = T::from_f32(fmaxf(cn[m * 256 + 128 + c], 0.f));
    }
}
// flow = coords1 - coords0 -> flow8 (h16 [M][8], 2 real channels) and xbuf[:, 254:256]
template <typename T>
__global__ void flow_prep_kernel(const float* coords1, int64_t M, int w, int h, unsigned short* flow8, unsigned short* xbuf) {
    GRID_STRIDE(m, M) {
        const float fx = coords1[m * 2] - (float)(m % w), fy = coords1[m * 2 + 1] - (float)((m / w) % h);      // rows of stacked h x w grids
        const unsigned pk = pack2<T>(fx, fy);
        *(uint4*)(flow8 + m * 8) = make_uint4(pk, 0, 0, 0);
        *(unsigned*)(xbuf + m * 256 + 254) = pk;
    }
}
// rh = sigmoid(zr[:, 128:256]) * h  (h16 [M][128])
template <typename T>
__global__ void gru_rh_kernel(const float* zr, const float* h, int64_t M, unsigned short* rh) {
    const int64_t n = M * 128;
    GRID_STRIDE(i, n) {
        const int c = (int)(i & 127); const int64_t m = i >> 7;
        rh[i] = T::from_f32(sigmoid_f(zr[m * 256 + 128 + c]) * h[i]);
    }
}
// h = (1 - z) h + z tanh(q), z = sigmoid(zr[:, 0:128]); writes fp32 (in place) + h16 copy
template <typename T>
__global__ void gru_update_kernel(const float* zr, const float* q, int64_t M, float* h, unsigned short* h16) {
    const int64_t n = M * 128;
    GRID_STRIDE(i, n) {
        const int c = (int)(i & 127); const int64_t m = i >> 7;
        const float z = sigmoid_f(zr[m * 256 + c]);
        const float v = (1.0f - z) * h[i] + z * tanhf(q[i]);
        h[i] = v; h16[i] = T::from_f32(v);
    }
}
__global__ void add_flow_kernel(float* coords1, const float* dflow, int ld, int64_t M) {
    GRID_STRIDE(i, M * 2) { const int64_t m = i >> 1; const int c = (int)(i & 1); coords1[i] = coords1[i] + dflow[m * ld + c]; }
}
__global__ void add_relu_kernel(const float* a, const float* b, float* out, int64_t n) {
    GRID_STRIDE(i, n) out[i] = fmaxf(a[i] + b[i], 0.f);
}
// convex 8x upsampling: coords1 [h][w][2] (absolute), mask [h][w][576] (already x0.25) -> flow [8h][8w][2]
__global__ void convex_upsample_kernel(const float* coords1, const float* mask, int F, int h, int w, float* out) {
    const int64_t n = (int64_t)F * h * w * 64;
    GRID_STRIDE(i, n) {
        const int sub = (int)(i & 63); const int64_t p = i >> 6;
        const int64_t f = p / ((int64_t)h * w), g0 = f * h * w;          // grid f of the stack
        const int x = (int)(p % w), y = (int)((p / w) % h), sy = sub >> 3, sx = sub & 7;
        const float* mk = mask + p * 576 + sub;           // mask[k*64 + sy*8 + sx], k = 0..8
        float mx = mk[0];
        for (int k = 1; k < 9; ++k) mx = fmaxf(mx, mk[k * 64]);
        float e[9], den = 0.f;
        for (int k = 0; k < 9; ++k) { e[k] = __expf(mk[k * 64] - mx); den += e[k]; }
        float fx = 0.f, fy = 0.f;
        for (int k = 0; k < 9; ++k) {
            const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
            float vx = 0.f, vy = 0.f;
            if (yy >= 0 && yy < h && xx >= 0 && xx < w) {
                const int64_t q = g0 + (int64_t)yy * w + xx;
                vx = 8.0f * (coords1[q * 2] - (float)xx); vy = 8.0f * (coords1[q * 2 + 1] - (float)yy);
            }
            const float wk = e[k] / den;
            fx += wk * vx; fy += wk * vy;
        }
        const int64_t o = ((f * h * 8 + (int64_t)(y * 8 + sy)) * (w * 8) + (x * 8 + sx)) * 2;
        out[o] = fx; out[o + 1] = fy;
    }
}

// ---- flow-guided propagation ---------------------------------------------------------------------------------
// planar-interleaved helpers: img [H][W][C] fp32, flow [H][W][2] fp32
__device__ __forceinline__ float bilinear_zero_c(const float* img, int H, int W, int C, int c, float x, float y) {
    const float x0f = floorf(x), y0f = floorf(y);
    const float wx = x - x0f, wy = y - y0f;
    const int x0 = (int)x0f, y0 = (int)y0f;
    auto tap = [&](int yy, int xx) { return (xx >= 0 && xx < W && yy >= 0 && yy < H) ? img[((int64_t)yy * W + xx) * C + c] : 0.f; };
    const float a = ((1.0f - wx) * (1.0f - wy)) * tap(y0, x0);
    const float b = (wx * (1.0f - wy)) * tap(y0, x0 + 1);
    const float cc = ((1.0f - wx) * wy) * tap(y0 + 1, x0);
    const float d = (wx * wy) * tap(y0 + 1, x0 + 1);
    return ((a + b) + cc) + d;
}
// valid(p) = |f_ab + warp(f_ba, f_ab)|^2 < 0.01 (|f_ab|^2 + |warp f_ba|^2) + 0.5
__global__ void fb_valid_kernel(const float* f_ab, const float* f_ba, int H, int W, uint8_t* valid) {
    const int64_t n = (int64_t)H * W;
    GRID_STRIDE(i, n) {
        const int x = (int)(i % W), y = (int)(i / W);
        const float ax = f_ab[i * 2], ay = f_ab[i * 2 + 1];
        const float sx = (float)x + ax, sy = (float)y + ay;
        const float bx = bilinear_zero_c(f_ba, H, W, 2, 0, sx, sy), by = bilinear_zero_c(f_ba, H, W, 2, 1, sx, sy);
        const float dx = ax + bx, dy = ay + by;
        const float lhs = dx * dx + dy * dy;
        const float rhs = 0.01f * ((ax * ax + ay * ay) + (bx * bx + by * by)) + 0.5f;
        valid[i] = lhs < rhs ? 1 : 0;
    }
}
// one sweep step: fill the unknown pixels of frame t from neighbour nb (both [H][W][3] fp32, in place on cur_t)
__global__ void prop_fill_kernel(float* cur_t, const float* cur_nb, uint8_t* known_t, const uint8_t* known_nb, const uint8_t* valid,
                                 const float* flow, int H, int W, uint8_t* filled_t) {
    const int64_t n = (int64_t)H * W;
    GRID_STRIDE(i, n) {
        const int x = (int)(i % W), y = (int)(i / W);
        const float fx = flow[i * 2], fy = flow[i * 2 + 1];
        const float sxf = (float)x + fx, syf = (float)y + fy;
        const int sx = (int)floorf(sxf + 0.5f), sy = (int)floorf(syf + 0.5f);
        const bool inb = sx >= 0 && sx < W && sy >= 0 && sy < H;
        const bool src_known = inb && known_nb[(int64_t)(inb ? sy : 0) * W + (inb ? sx : 0)] != 0;
        const bool take = known_t[i] == 0 && valid[i] != 0 && src_known;
        if (take) {
#pragma unroll
            for (int c = 0; c < 3; ++c) cur_t[i * 3 + c] = bilinear_zero_c(cur_nb, H, W, 3, c, sxf, syf);
        }
        filled_t[i] = take ? 1 : 0;
    }
    // NOTE: known_t is updated by prop_commit_kernel after this launch (readers of known_nb in the same launch see the old map)
}
__global__ void prop_commit_kernel(uint8_t* known_t, const uint8_t* filled_t, int64_t n) {
    GRID_STRIDE(i, n) if (filled_t[i]) known_t[i] = 1;
}
// combine the two sweeps, fill the rest with the mean colour, round to uint8
__global__ void prop_combine_kernel(const float* orig, const float* a, const float* b, const uint8_t* fa, const uint8_t* fb, const uint8_t* hole,
                                    const float* mean3, int64_t npix, uint8_t* out, uint8_t* filled) {
    GRID_STRIDE(i, npix) {
        const bool A = fa[i] != 0, B = fb[i] != 0, Hh = hole[i] != 0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = orig[i * 3 + c];
            if (A && B) v = (a[i * 3 + c] + b[i * 3 + c]) * 0.5f;
            else if (A) v = a[i * 3 + c];
            else if (B) v = b[i * 3 + c];
            else if (Hh) v = mean3[c];
            v = floorf(v + 0.5f);
            v = fminf(fmaxf(v, 0.f), 255.f);
            out[i * 3 + c] = (uint8_t)v;
        }
        if (filled) filled[i] = (A || B) ? 1 : 0;
    }
}
// per-frame mean colour of the non-hole pixels (double accumulation through integer sums: exact)
__global__ void masked_sum_kernel(const uint8_t* frame, const uint8_t* hole, int64_t npix, unsigned long long* sums /* [4]: r,g,b,count */) {
    unsigned long long s[4] = {0, 0, 0, 0};
    GRID_STRIDE(i, npix) {
        if (hole[i] == 0) { s[0] += frame[i * 3]; s[1] += frame[i * 3 + 1]; s[2] += frame[i * 3 + 2]; s[3] += 1; }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        unsigned long long v = s[k];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(&sums[k], v);
    }
}
__global__ void u8_is_zero_kernel(const uint8_t* in, uint8_t* out, int64_t n) { GRID_STRIDE(i, n) out[i] = in[i] == 0 ? 1 : 0; }
__global__ void u8_to_f32_kernel(const uint8_t* in, float* out, int64_t n) { GRID_STRIDE(i, n) out[i] = (float)in[i]; }
// image [T][H][W][3] u8 -> h16 [T][H][W][8] scaled to [-1,1] as 2*(x/255)-1 (RAFT input)
template <typename T>
__global__ void raft_prep_kernel(const uint8_t* img, int64_t npix, unsigned short* out) {
    GRID_STRIDE(i, npix) {
        float v[8];
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = 2.0f * ((float)img[i * 3 + c] / 255.0f) - 1.0f;
#pragma unroll
        for (int c = 3; c < 8; ++c) v[c] = 0.f;
        *(uint4*)(out + i * 8) = pack8<T>(v);
    }
}

}  // namespace

#define DT_DISPATCH(name, kern, grid, ...)                                                             \
    if (dtype == VV_BF16) hipLaunchKernelGGL(kern<BF16>, grid, dim3(EB), 0, (hipStream_t)stream, __VA_ARGS__);      \
    else if (dtype == VV_F16) hipLaunchKernelGGL(kern<F16>, grid, dim3(EB), 0, (hipStream_t)stream, __VA_ARGS__);   \
    else VV_FAIL(VV_E_ARG, name ": bad dtype");                                                        \
    VV_CHECK_LAUNCH(name);                                                                             \
    return VV_OK;

extern "C" int vv_avgpool2_f32(const float* in, int64_t N, int h, int w, float* out, void* stream) {
    if (!in || !out || N <= 0 || h < 2 || w < 2) VV_FAIL(VV_E_ARG, "vv_avgpool2_f32: bad args");
    hipLaunchKernelGGL(avgpool2_kernel, grid_for(N * (h / 2) * (w / 2)), dim3(EB), 0, (hipStream_t)stream, in, N, h, w, out);
    VV_CHECK_LAUNCH("vv_avgpool2_f32");
    return VV_OK;
}

extern "C" int vv_corr_lookup(const float* l0, const float* l1, const float* l2, const float* l3, int h, int w, const float* coords, int64_t N,
                              int cpad, void* out, int dtype, void* stream) {
    if (!l0 || !l1 || !l2 || !l3 || !coords || !out || N <= 0 || h < 8 || w < 8 || cpad < 324 || cpad % 8) VV_FAIL(VV_E_ARG, "vv_corr_lookup: bad args (feature map must be at least 8x8)");
    Pyr P;
    P.lvl[0] = l0; P.lvl[1] = l1; P.lvl[2] = l2; P.lvl[3] = l3;
    int hh = h, ww = w;
    for (int l = 0; l < 4; ++l) { P.h[l] = hh; P.w[l] = ww; hh /= 2; ww /= 2; }
    DT_DISPATCH("vv_corr_lookup", corr_lookup_kernel, grid_for(N * cpad), P, coords, N, cpad, (unsigned short*)out)
}

extern "C" int vv_raft_ctx_split(const float* cn, int64_t M, float* net, void* net16, void* xbuf, int dtype, void* stream) {
    if (!cn || !net || !net16 || !xbuf || M <= 0) VV_FAIL(VV_E_ARG, "vv_raft_ctx_split: bad args");
    DT_DISPATCH("vv_raft_ctx_split", ctx_split_kernel, grid_for(M * 128), cn, M, net, (unsigned short*)net16, (unsigned short*)xbuf)
}

extern "C" int vv_raft_flow_prep(const float* coords1, int64_t M, int w, int h, void* flow8, void* xbuf, int dtype, void* stream) {
    if (!coords1 || !flow8 || !xbuf || M <= 0 || w <= 0 || h <= 0 || M % ((int64_t)h * w)) VV_FAIL(VV_E_ARG, "vv_raft_flow_prep: bad args (M must be whole h x w grids)");
    DT_DISPATCH("vv_raft_flow_prep", flow_prep_kernel, grid_for(M), coords1, M, w, h, (unsigned short*)flow8, (unsigned short*)xbuf)
}

extern "C" int vv_gru_rh(const float* zr, const float* h, int64_t M, void* rh, int dtype, void* stream) {
    if (!zr || !h || !rh || M <= 0) VV_FAIL(VV_E_ARG, "vv_gru_rh: bad args");
    DT_DISPATCH("vv_gru_rh", gru_rh_kernel, grid_for(M * 128), zr, h, M, (unsigned short*)rh)
}

extern "C" int vv_gru_update(const float* zr, const float* q, int64_t M, float* h, void* h16, int dtype, void* stream) {
    if (!zr || !q || !h || !h16 || M <= 0) VV_FAIL(VV_E_ARG, "vv_gru_update: bad args");
    DT_DISPATCH("vv_gru_update", gru_update_kernel, grid_for(M * 128), zr, q, M, h, (unsigned short*)h16)
}

extern "C" int vv_add_flow(float* coords1, const float* dflow, int ld, int64_t M, void* stream) {
    if (!coords1 || !dflow || ld < 2 || M <= 0) VV_FAIL(VV_E_ARG, "vv_add_flow: bad args");
    hipLaunchKernelGGL(add_flow_kernel, grid_for(M * 2), dim3(EB), 0, (hipStream_t)stream, coords1, dflow, ld, M);
    VV_CHECK_LAUNCH("vv_add_flow");
    return VV_OK;
}

extern "C" int vv_add_relu_f32(const float* a, const float* b, float* out, int64_t n, void* stream) {
    if (!a || !b || !out || n <= 0) VV_FAIL(VV_E_ARG, "vv_add_relu_f32: bad args");
    hipLaunchKernelGGL(add_relu_kernel, grid_for(n), dim3(EB), 0, (hipStream_t)stream, a, b, out, n);
    VV_CHECK_LAUNCH("vv_add_relu_f32");
    return VV_OK;
}

extern "C" int vv_convex_upsample(const float* coords1, const float* mask, int F, int h, int w, float* out, void* stream) {
    if (!coords1 || !mask || !out || F <= 0 || h <= 0 || w <= 0) VV_FAIL(VV_E_ARG, "vv_convex_upsample: bad args");
    hipLaunchKernelGGL(convex_upsample_kernel, grid_for((int64_t)F * h * w * 64), dim3(EB), 0, (hipStream_t)stream, coords1, mask, F, h, w, out);
    VV_CHECK_LAUNCH("vv_convex_upsample");
    return VV_OK;
}

extern "C" int vv_fb_valid(const float* f_ab, const float* f_ba, int H, int W, uint8_t* valid, void* stream) {
    if (!f_ab || !f_ba || !valid || H <= 0 || W <= 0) VV_FAIL(VV_E_ARG, "vv_fb_valid: bad args");
    hipLaunchKernelGGL(fb_valid_kernel, grid_for((int64_t)H * W), dim3(EB), 0, (hipStream_t)stream, f_ab, f_ba, H, W, valid);
    VV_CHECK_LAUNCH("vv_fb_valid");
    return VV_OK;
}

extern "C" int vv_prop_fill(float* cur_t, const float* cur_nb, uint8_t* known_t, const uint8_t* known_nb, const uint8_t* valid, const float* flow,
                            int H, int W, uint8_t* filled_t, void* stream) {
    if (!cur_t || !cur_nb || !known_t || !known_nb || !valid || !flow || !filled_t || H <= 0 || W <= 0) VV_FAIL(VV_E_ARG, "vv_prop_fill: bad args");
    const int64_t n = (int64_t)H * W;
    hipLaunchKernelGGL(prop_fill_kernel, grid_for(n), dim3(EB), 0, (hipStream_t)stream, cur_t, cur_nb, known_t, known_nb, valid, flow, H, W, filled_t);
    hipLaunchKernelGGL(prop_commit_kernel, grid_for(n), dim3(EB), 0, (hipStream_t)stream, known_t, filled_t, n);
    VV_CHECK_LAUNCH("vv_prop_fill");
    return VV_OK;
}

extern "C" int vv_prop_combine(const float* orig, const float* a, const float* b, const uint8_t* fa, const uint8_t* fb, const uint8_t* hole,
                               int H, int W, const float* mean3, uint8_t* out, uint8_t* filled, void* stream) {
    if (!orig || !a || !b || !fa || !fb || !hole || !mean3 || !out || H <= 0 || W <= 0) VV_FAIL(VV_E_ARG, "vv_prop_combine: bad args");
    hipLaunchKernelGGL(prop_combine_kernel, grid_for((int64_t)H * W), dim3(EB), 0, (hipStream_t)stream, orig, a, b, fa, fb, hole, mean3, (int64_t)H * W, out, filled);
    VV_CHECK_LAUNCH("vv_prop_combine");
    return VV_OK;
}

extern "C" int vv_masked_sum_u8(const uint8_t* frame, const uint8_t* hole, int64_t npix, unsigned long long* sums4, void* stream) {
    if (!frame || !hole || !sums4 || npix <= 0) VV_FAIL(VV_E_ARG, "vv_masked_sum_u8: bad args");
    if (hipMemsetAsync(sums4, 0, 4 * sizeof(unsigned long long), (hipStream_t)stream) != hipSuccess) VV_FAIL(VV_E_LAUNCH, "vv_masked_sum_u8: memset failed");
    hipLaunchKernelGGL(masked_sum_kernel, grid_for(npix), dim3(EB), 0, (hipStream_t)stream, frame, hole, npix, sums4);
    VV_CHECK_LAUNCH("vv_masked_sum_u8");
    return VV_OK;
}

extern "C" int vv_u8_is_zero(const uint8_t* in, uint8_t* out, int64_t n, void* stream) {
    if (!in || !out || n <= 0) VV_FAIL(VV_E_ARG, "vv_u8_is_zero: bad args");
    hipLaunchKernelGGL(u8_is_zero_kernel, grid_for(n), dim3(EB), 0, (hipStream_t)stream, in, out, n);
    VV_CHECK_LAUNCH("vv_u8_is_zero");
    return VV_OK;
}

extern "C" int vv_u8_to_f32(const uint8_t* in, float* out, int64_t n, void* stream) {
    if (!in || !out || n <= 0) VV_FAIL(VV_E_ARG, "vv_u8_to_f32: bad args");
    hipLaunchKernelGGL(u8_to_f32_kernel, grid_for(n), dim3(EB), 0, (hipStream_t)stream, in, out, n);
    VV_CHECK_LAUNCH("vv_u8_to_f32");
    return VV_OK;
}

extern "C" int vv_raft_prep(const uint8_t* img, int64_t npix, void* out8, int dtype, void* stream) {
    if (!img || !out8 || npix <= 0) VV_FAIL(VV_E_ARG, "vv_raft_prep: bad args");
    DT_DISPATCH("vv_raft_prep", raft_prep_kernel, grid_for(npix), img, npix, (unsigned short*)out8)
}
