// SAM 2.1 video predictor: the kernels the masking path needs beyond the shared GEMM / attention / gather kernels.
// All HBM-bound elementwise / small-window work: rows are walked by consecutive lanes, one pass over the data wherever the
// arithmetic allows (LayerNorm re-reads its row from L1/L2).  Contracts: include/vvhip.h ("SAM 2" section).
#include "vv_common.h"

namespace {

__device__ __forceinline__ float wave_sum(float s) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    return s;
}

template <typename T> __device__ __forceinline__ void store_h16(void* p, int64_t i, float v) { ((unsigned short*)p)[i] = T::from_f32(v); }

__device__ __forceinline__ float act_f(float x, int act) {
    if (act == VV_ACT_GELU) return gelu_f(x);
    if (act == VV_ACT_RELU) return fmaxf(x, 0.f);
    if (act == VV_ACT_SIGMOID) return 1.0f / (1.0f + __expf(-x));
    if (act == VV_ACT_SILU) return silu_f(x);
    return x;
}

// ---- uint8 RGB -> normalised h16 [npix][cpad]; s2d > 1: space-to-depth, out [(H/s2d)*(W/s2d)][cpad], channel (dy*s2d + dx)*3 + c ----------------
template <typename T>
__global__ __launch_bounds__(256) void u8_normalize_kernel(const unsigned char* __restrict__ src, int H, int W, float m0, float m1, float m2,
                                                           float s0, float s1, float s2, unsigned short* __restrict__ out, int cpad, int s2d) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)H * W) return;
    const int y = (int)(i / W), x = (int)(i - (int64_t)y * W);
    const unsigned char* p = src + i * 3;
    const int Wb = W / s2d;
    unsigned short* o = out + ((int64_t)(y / s2d) * Wb + x / s2d) * cpad + ((y % s2d) * s2d + x % s2d) * 3;
    o[0] = T::from_f32(((float)p[0] / 255.0f - m0) * s0);
    o[1] = T::from_f32(((float)p[1] / 255.0f - m1) * s1);
    o[2] = T::from_f32(((float)p[2] / 255.0f - m2) * s2);
    if ((y % s2d) == s2d - 1 && (x % s2d) == s2d - 1)
        for (int c = 3 * s2d * s2d; c < cpad; ++c) (o - ((y % s2d) * s2d + x % s2d) * 3)[c] = 0;
}

// ---- LayerNorm over the last dimension, any C, fp32 in, optional activation, h16 or fp32 out: one wave per row -----------------
template <typename T>
__global__ __launch_bounds__(256) void layernorm_ex_kernel(const float* __restrict__ x, int64_t M, int C, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps, int act, void* __restrict__ out, int out_f32, int cpad) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + row * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += xr[c];
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
    for (int c = lane; c < C; c += 64) { const float d = xr[c] - mean; q += d * d; }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
    for (int c = lane; c < cpad; c += 64) {
        const float y = c < C ? act_f((xr[c] - mean) * rstd * gamma[c] + beta[c], act) : 0.f;       // columns C .. cpad-1: zero padding
        if (out_f32) ((float*)out)[row * cpad + c] = y;
        else store_h16<T>(out, row * cpad + c, y);
    }
}

// fast form: C % 4 == 0 and C <= 256 * NCH: the row lives in registers (float4 per lane and step), one read of x
template <typename T, int NCH>
__global__ __launch_bounds__(256) void layernorm_ex_vec_kernel(const float* __restrict__ x, int64_t M, int C, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float eps, int act, void* __restrict__ out, int out_f32, int cpad) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int C4 = C >> 2;
    const float4* xr = (const float4*)(x + row * C);
    float4 v[NCH];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = lane + 64 * i;
        v[i] = c < C4 ? xr[c] : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        if (lane + 64 * i < C4) {
            const float dx = v[i].x - mean, dy = v[i].y - mean, dz = v[i].z - mean, dw = v[i].w - mean;
            q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
    const int P4 = cpad >> 2;
#pragma unroll
    for (int i = 0; i < NCH + 1; ++i) {                              // one extra step covers the zero padding (cpad - C <= 252 columns)
        const int c = lane + 64 * i;
        if (c >= P4) continue;
        float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < C4 && i < NCH) {
            const float4 g = ((const float4*)gamma)[c], b = ((const float4*)beta)[c];
            y.x = act_f((v[i].x - mean) * rstd * g.x + b.x, act); y.y = act_f((v[i].y - mean) * rstd * g.y + b.y, act);
            y.z = act_f((v[i].z - mean) * rstd * g.z + b.z, act); y.w = act_f((v[i].w - mean) * rstd * g.w + b.w, act);
        }
        if (out_f32) ((float4*)out)[row * P4 + c] = y;
        else ((uint2*)out)[row * P4 + c] = make_uint2(pack2<T>(y.x, y.y), pack2<T>(y.z, y.w));
    }
}

// ---- 2x2 max pooling, NHWC ------------------------------------------------------------------------------------------------------
template <typename T, bool F32>
__global__ __launch_bounds__(256) void maxpool2_kernel(const void* __restrict__ x, int B, int H, int W, int C, int64_t in_bs, void* __restrict__ out) {
    const int Ho = H >> 1, Wo = W >> 1;
    const int64_t n = (int64_t)B * Ho * Wo * C;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i % C);
    int64_t r = i / C;
    const int xo = (int)(r % Wo); r /= Wo;
    const int yo = (int)(r % Ho);
    const int b = (int)(r / Ho);
    const int64_t base = (int64_t)b * in_bs + (((int64_t)2 * yo) * W + 2 * xo) * C + c;
    if (F32) {
        const float* p = (const float*)x;
        ((float*)out)[i] = fmaxf(fmaxf(p[base], p[base + C]), fmaxf(p[base + (int64_t)W * C], p[base + (int64_t)W * C + C]));
    } else {
        const unsigned short* p = (const unsigned short*)x;
        const float v = fmaxf(fmaxf(T::to_f32(p[base]), T::to_f32(p[base + C])),
                              fmaxf(T::to_f32(p[base + (int64_t)W * C]), T::to_f32(p[base + (int64_t)W * C + C])));
        ((unsigned short*)out)[i] = T::from_f32(v);       // the maximum of h16 values is one of them: exact
    }
}

// ---- rotary position encoding on an h16 matrix, in place: pairs (2i, 2i+1) of columns col0.. of rows < rows_rope --------------------
template <typename T>
__global__ __launch_bounds__(256) void rope_kernel(unsigned short* __restrict__ x, int64_t rows_rope, int ld, int col0, int D,
                                                   const float* __restrict__ cs, int n_table) {
    const int half = D >> 1;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows_rope * half) return;
    const int64_t r = i / half;
    const int pr = (int)(i - r * half);
    const float* t = cs + ((r % n_table) * half + pr) * 2;
    unsigned* p = (unsigned*)(x + r * ld + col0 + 2 * pr);
    const unsigned u = *p;
    const float re = T::to_f32(u & 0xffff), im = T::to_f32(u >> 16);
    *p = pack2<T>(re * t[0] - im * t[1], re * t[1] + im * t[0]);
}

// ---- depthwise k x k convolution, NHWC fp32, zero padding k/2 ----------------------------------------------------------------------
__global__ __launch_bounds__(256) void dwconv_kernel(const float* __restrict__ x, int H, int W, int C, const float* __restrict__ w,
                                                     const float* __restrict__ bias, int k, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)H * W * C) return;
    const int c = (int)(i % C);
    const int px = (int)((i / C) % W), py = (int)(i / ((int64_t)C * W));
    const int r = k >> 1;
    float acc = bias[c];
    for (int dy = 0; dy < k; ++dy) {
        const int yy = py + dy - r;
        if (yy < 0 || yy >= H) continue;
        for (int dx = 0; dx < k; ++dx) {
            const int xx = px + dx - r;
            if (xx < 0 || xx >= W) continue;
            acc = fmaf(x[((int64_t)yy * W + xx) * C + c], w[(c * k + dy) * k + dx], acc);
        }
    }
    out[i] = acc;
}

// ---- ConvTranspose2d(k = 2, stride 2) tail: y [h*w][4*C] (dy, dx, c) + bias (+ add) (act) -> [2h*2w][C] ---------------------------------
template <typename T>
__global__ __launch_bounds__(256) void pixel_shuffle2_kernel(const float* __restrict__ y, const float* __restrict__ bias, const float* __restrict__ add,
                                                             int h, int w, int C, int act, void* __restrict__ out, int out_f32) {
    const int64_t n = (int64_t)4 * h * w * C;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i % C);
    int64_t r = i / C;
    const int X = (int)(r % (2 * w)), Y = (int)(r / (2 * w));
    const int64_t src = (((int64_t)(Y >> 1) * w + (X >> 1)) * 4 + (Y & 1) * 2 + (X & 1)) * C + c;
    float v = y[src] + bias[c];
    if (add) v += add[i];
    v = act_f(v, act);
    if (out_f32) ((float*)out)[i] = v;
    else store_h16<T>(out, i, v);
}

// ---- bilinear resize, NHWC fp32, torch semantics (align_corners = False, no antialias) ------------------------------------------------
__global__ __launch_bounds__(256) void resize_bilinear_f32_kernel(const float* __restrict__ src, int Hs, int Ws, int C, float* __restrict__ dst,
                                                                  int Hd, int Wd, float sy, float sx) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)Hd * Wd * C) return;
    const int c = (int)(i % C);
    const int X = (int)((i / C) % Wd), Y = (int)(i / ((int64_t)C * Wd));
    float fy = ((float)Y + 0.5f) * sy - 0.5f, fx = ((float)X + 0.5f) * sx - 0.5f;
    fy = fy < 0.f ? 0.f : fy; fx = fx < 0.f ? 0.f : fx;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < Hs - 1 ? 1 : 0), x1 = x0 + (x0 < Ws - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
    const float* p = src + c;
    dst[i] = hy * (hx * p[((int64_t)y0 * Ws + x0) * C] + lx * p[((int64_t)y0 * Ws + x1) * C]) +
             ly * (hx * p[((int64_t)y1 * Ws + x0) * C] + lx * p[((int64_t)y1 * Ws + x1) * C]);
}

// ---- mask logits -> memory-encoder input: (binarise | sigmoid) * scale + bias, h16 [n][8] (channel 0) -------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void mask_mem_input_kernel(const float* __restrict__ logits, int64_t n, int binarize, float scale, float bias,
                                                             uint4* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float v = logits[i];
    const float m = (binarize ? (v > 0.f ? 1.f : 0.f) : 1.0f / (1.0f + __expf(-v))) * scale + bias;
    out[i] = make_uint4(pack2<T>(m, 0.f), 0u, 0u, 0u);
}

// ---- in-place activation ---------------------------------------------------------------------------------------------------------
template <typename T, bool F32>
__global__ __launch_bounds__(256) void act_kernel(void* __restrict__ x, int64_t n, int act) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (F32) ((float*)x)[i] = act_f(((float*)x)[i], act);
    else ((unsigned short*)x)[i] = T::from_f32(act_f(T::to_f32(((unsigned short*)x)[i]), act));
}

// ---- prompt encoder: random-Fourier position encoding of the click points + label embeddings -> [P][D] fp32 ---------------------------
__global__ __launch_bounds__(256) void prompt_points_kernel(const float* __restrict__ coords, const int* __restrict__ labels, int P, float inv_size,
                                                            const float* __restrict__ gauss, const float* __restrict__ table, int D, float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P * D) return;
    const int pt = i / D, c = i - pt * D, half = D >> 1;
    const int lab = labels[pt];
    float v = 0.f;
    if (lab >= 0) {
        const float cx = 2.f * ((coords[pt * 2] + 0.5f) * inv_size) - 1.f, cy = 2.f * ((coords[pt * 2 + 1] + 0.5f) * inv_size) - 1.f;
        const int j = c < half ? c : c - half;
        const float a = 6.283185307179586f * (cx * gauss[j] + cy * gauss[half + j]);
        v = c < half ? sinf(a) : cosf(a);
    }
    out[i] = v + table[(lab + 1) * D + c];       // table rows: not_a_point (label -1), point_embeddings 0..3
}

// ---- get_1d_sine_pe: pos [n] -> [n][dim] = [sin(pos / t_j) | cos(pos / t_j)], t_j = temperature^(2 (j / 2) / (dim / 2)) -----------------------
__global__ __launch_bounds__(256) void sine_pe_1d_kernel(const float* __restrict__ pos, int n, int dim, float temperature, float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n * dim) return;
    const int r = i / dim, c = i - r * dim, half = dim >> 1;
    const int j = c < half ? c : c - half;
    const float t = powf(temperature, (float)(2 * (j / 2)) / (float)half);
    const float a = pos[r] / t;
    out[i] = c < half ? sinf(a) : cosf(a);
}

// ---- mask decoder output selection (mask_decoder.py::forward + sam2_base.py::_forward_sam_heads), one block ---------------------------------
// sel[0] = index of the returned mask among the nm decoder masks, sel[1] = object appearing (score logit > 0), sel[2] = index of the
// output token the object pointer is projected from.
__global__ __launch_bounds__(256) void sam_select_kernel(const float* __restrict__ masks, int HW, int nm, const float* __restrict__ iou,
                                                         const float* __restrict__ obj_logit, int multimask, float delta, float thresh, int* __restrict__ sel) {
    __shared__ float red[2][4];
    float ai = 0.f, au = 0.f;
    for (int i = threadIdx.x; i < HW; i += 256) { const float v = masks[i]; ai += v > delta ? 1.f : 0.f; au += v > -delta ? 1.f : 0.f; }
    ai = wave_sum(ai); au = wave_sum(au);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = ai; red[1][threadIdx.x >> 6] = au; }
    __syncthreads();
    if (threadIdx.x == 0) {
        ai = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        au = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        int best = 1;
        for (int j = 2; j < nm; ++j) if (iou[j] > iou[best]) best = j;        // torch.argmax: the first maximum
        const bool stable = (au > 0.f ? ai / au : 1.f) >= thresh;
        sel[0] = multimask ? best : (stable ? 0 : best);
        sel[1] = obj_logit[0] > 0.f ? 1 : 0;
        sel[2] = multimask ? best : 0;
    }
}

__global__ __launch_bounds__(256) void sam_pick_kernel(const float* __restrict__ masks, int HW, const int* __restrict__ sel, float no_obj, float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= HW) return;
    out[i] = sel[1] ? masks[(int64_t)sel[0] * HW + i] : no_obj;
}

__global__ __launch_bounds__(256) void select_rows_kernel(const float* __restrict__ a, const float* __restrict__ b, const int* __restrict__ flag, int64_t n,
                                                          float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = flag[0] ? a[i] : b[i];
}

__global__ __launch_bounds__(256) void add_rowvec_unless_kernel(float* __restrict__ x, const float* __restrict__ vec, const float* __restrict__ score, int64_t M, int C) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < M * C && !(score[0] > 0.f)) x[i] += vec[i % C];
}

__global__ __launch_bounds__(256) void clamp_kernel(const float* __restrict__ x, int64_t n, float lo, float hi, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = fminf(fmaxf(x[i], lo), hi);
}

// ---- fill_holes_in_mask_scores: background (score <= 0) components, 8-connected, of area <= max_area become 0.1 ------------------------------
// A component of at most A pixels has diameter < A, so A rounds of min-label propagation converge on it; a label region that is small AND closed
// (no background neighbour carries another label) is exactly such a component -- larger components never qualify, converged or not.
__global__ __launch_bounds__(256) void holes_init_kernel(const float* __restrict__ m, int n, int* __restrict__ lab, int* __restrict__ area, int* __restrict__ open) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    lab[i] = m[i] <= 0.f ? i : -1;
    area[i] = 0; open[i] = 0;
}

__global__ __launch_bounds__(256) void holes_prop_kernel(int* __restrict__ lab, int H, int W) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= H * W) return;
    int l = lab[i];
    if (l < 0) return;
    const int y = i / W, x = i - y * W;
    for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
            const int yy = y + dy, xx = x + dx;
            if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
            const int o = lab[yy * W + xx];
            if (o >= 0 && o < l) l = o;
        }
    lab[i] = l;          // monotone: racing reads of neighbours only ever see larger-or-equal labels than their final value
}

__global__ __launch_bounds__(256) void holes_count_kernel(const int* __restrict__ lab, int H, int W, int* __restrict__ area, int* __restrict__ open) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= H * W) return;
    const int l = lab[i];
    if (l < 0) return;
    atomicAdd(&area[l], 1);
    const int y = i / W, x = i - y * W;
    bool op = false;
    for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
            const int yy = y + dy, xx = x + dx;
            if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
            const int o = lab[yy * W + xx];
            if (o >= 0 && o != l) { op = true; atomicOr(&open[o], 1); }
        }
    if (op) atomicOr(&open[l], 1);
}

__global__ __launch_bounds__(256) void holes_apply_kernel(float* __restrict__ m, int n, const int* __restrict__ lab, const int* __restrict__ area,
                                                          const int* __restrict__ open, int max_area) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int l = lab[i];
    if (l >= 0 && area[l] <= max_area && !open[l]) m[i] = 0.1f;
}

// ---- masks[j][p] = sum_c hyper[j][c] * up[p][c]   (mask_decoder.py: hyper_in @ upscaled_embedding), nm <= 8 ------------------------------------
__global__ __launch_bounds__(256) void hyper_masks_kernel(const float* __restrict__ hyper, const float* __restrict__ up, int HW, int C, int nm,
                                                          float* __restrict__ masks) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    const float* u = up + (int64_t)p * C;
    for (int c = 0; c < C; ++c) {
        const float v = u[c];
#pragma unroll
        for (int j = 0; j < 8; ++j) if (j < nm) acc[j] = fmaf(hyper[j * C + c], v, acc[j]);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) if (j < nm) masks[(int64_t)j * HW + p] = acc[j];
}

// ---- memory encoder, mask path, first two layers fused (memory_encoder.py::MaskDownSampler layers 0-1 and 3-4 after sam2_base.py::_encode_new_memory):
// layer 1: low-res logits [lo][lo] --bilinear (torch, align_corners = False)--> [S][S] --(binarise | sigmoid) * scale + bias--> conv 3x3 / stride 2 / pad 1
// (1 -> C1 channels) --> LayerNorm over channels --> GELU --> h16 [S/2 * S/2][cpad].  One thread per output pixel: the 1024^2 intermediate never exists.
template <typename T, int C1>
__global__ __launch_bounds__(256) void maskdown1_kernel(const float* __restrict__ logits, int lo, int S, int binarize, float scale, float bias,
                                                        const float* __restrict__ w, const float* __restrict__ b, const float* __restrict__ g,
                                                        const float* __restrict__ be, float eps, unsigned short* __restrict__ out, int cpad) {
    const int Ho = S >> 1;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Ho * Ho) return;
    const int oy = i / Ho, ox = i - oy * Ho;
    const float sc = (float)lo / (float)S;
    float acc[C1];
#pragma unroll
    for (int c = 0; c < C1; ++c) acc[c] = b[c];
    for (int ky = 0; ky < 3; ++ky) {
        const int Y = 2 * oy + ky - 1;
        if (Y < 0 || Y >= S) continue;
        float fy = ((float)Y + 0.5f) * sc - 0.5f; fy = fy < 0.f ? 0.f : fy;
        const int y0 = (int)fy, y1 = y0 + (y0 < lo - 1 ? 1 : 0);
        const float ly = fy - (float)y0, hy = 1.f - ly;
        for (int kx = 0; kx < 3; ++kx) {
            const int X = 2 * ox + kx - 1;
            if (X < 0 || X >= S) continue;
            float fx = ((float)X + 0.5f) * sc - 0.5f; fx = fx < 0.f ? 0.f : fx;
            const int x0 = (int)fx, x1 = x0 + (x0 < lo - 1 ? 1 : 0);
            const float lx = fx - (float)x0, hx = 1.f - lx;
            const float v = hy * (hx * logits[y0 * lo + x0] + lx * logits[y0 * lo + x1]) + ly * (hx * logits[y1 * lo + x0] + lx * logits[y1 * lo + x1]);
            const float m = (binarize ? (v > 0.f ? 1.f : 0.f) : 1.0f / (1.0f + __expf(-v))) * scale + bias;
#pragma unroll
            for (int c = 0; c < C1; ++c) acc[c] = fmaf(m, w[(c * 3 + ky) * 3 + kx], acc[c]);
        }
    }
    float mean = 0.f;
#pragma unroll
    for (int c = 0; c < C1; ++c) mean += acc[c];
    mean /= (float)C1;
    float var = 0.f;
#pragma unroll
    for (int c = 0; c < C1; ++c) { const float d = acc[c] - mean; var += d * d; }
    const float rstd = rsqrtf(var / (float)C1 + eps);
    unsigned short* o = out + (int64_t)i * cpad;
#pragma unroll
    for (int c = 0; c < C1; ++c) o[c] = T::from_f32(gelu_f((acc[c] - mean) * rstd * g[c] + be[c]));
    for (int c = C1; c < cpad; ++c) o[c] = 0;
}

// layers with a handful of channels: conv 3x3 / stride 2 / pad 1 (CIN -> COUT) + LayerNorm over channels + GELU, h16 [H*W][cin_pad] -> h16 [H/2*W/2][cpad]
template <typename T, int CIN, int COUT>
__global__ __launch_bounds__(256) void conv3s2_ln_gelu_kernel(const unsigned short* __restrict__ x, int H, int W, int cin_pad, const float* __restrict__ w,
                                                              const float* __restrict__ b, const float* __restrict__ g, const float* __restrict__ be, float eps,
                                                              unsigned short* __restrict__ out, int cpad) {
    const int Ho = H >> 1, Wo = W >> 1;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Ho * Wo) return;
    const int oy = i / Wo, ox = i - oy * Wo;
    float acc[COUT];
#pragma unroll
    for (int c = 0; c < COUT; ++c) acc[c] = b[c];
    for (int ky = 0; ky < 3; ++ky) {
        const int Y = 2 * oy + ky - 1;
        if (Y < 0 || Y >= H) continue;
        for (int kx = 0; kx < 3; ++kx) {
            const int X = 2 * ox + kx - 1;
            if (X < 0 || X >= W) continue;
            const unsigned short* px = x + ((int64_t)Y * W + X) * cin_pad;
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) {
                const float v = T::to_f32(px[ci]);
#pragma unroll
                for (int c = 0; c < COUT; ++c) acc[c] = fmaf(v, w[((c * CIN + ci) * 3 + ky) * 3 + kx], acc[c]);
            }
        }
    }
    float mean = 0.f;
#pragma unroll
    for (int c = 0; c < COUT; ++c) mean += acc[c];
    mean /= (float)COUT;
    float var = 0.f;
#pragma unroll
    for (int c = 0; c < COUT; ++c) { const float d = acc[c] - mean; var += d * d; }
    const float rstd = rsqrtf(var / (float)COUT + eps);
    unsigned short* o = out + (int64_t)i * cpad;
#pragma unroll
    for (int c = 0; c < COUT; ++c) o[c] = T::from_f32(gelu_f((acc[c] - mean) * rstd * g[c] + be[c]));
    for (int c = COUT; c < cpad; ++c) o[c] = 0;
}

inline dim3 grid1(int64_t n) { return dim3((unsigned)((n + 255) / 256)); }

}  // namespace

#define SAM2_DT(call_bf, call_f) do { if (dtype == VV_BF16) { call_bf; } else if (dtype == VV_F16) { call_f; } else VV_FAIL(VV_E_ARG, "bad dtype"); } while (0)

extern "C" int vv_u8_normalize(const uint8_t* src, int H, int W, const float* mean3, const float* istd3, void* out, int cpad, int s2d, int dtype, void* stream) {
    if (s2d <= 0) s2d = 1;
    if (!src || !mean3 || !istd3 || !out || H <= 0 || W <= 0 || cpad < 3 * s2d * s2d || H % s2d || W % s2d) VV_FAIL(VV_E_ARG, "vv_u8_normalize: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int64_t npix = (int64_t)H * W;
    SAM2_DT(hipLaunchKernelGGL(u8_normalize_kernel<BF16>, grid1(npix), dim3(256), 0, st, src, H, W, mean3[0], mean3[1], mean3[2], istd3[0], istd3[1], istd3[2], (unsigned short*)out, cpad, s2d),
            hipLaunchKernelGGL(u8_normalize_kernel<F16>, grid1(npix), dim3(256), 0, st, src, H, W, mean3[0], mean3[1], mean3[2], istd3[0], istd3[1], istd3[2], (unsigned short*)out, cpad, s2d));
    VV_CHECK_LAUNCH("vv_u8_normalize");
    return VV_OK;
}

extern "C" int vv_layernorm_ex(const float* x, int64_t M, int C, const float* gamma, const float* beta, float eps, int act, void* out, int out_dtype,
                               int cpad, int dtype, void* stream) {
    if (cpad <= 0) cpad = C;
    if (!x || !gamma || !beta || !out || M <= 0 || C <= 0 || cpad < C) VV_FAIL(VV_E_ARG, "vv_layernorm_ex: bad arguments");
    const int of32 = out_dtype == VV_F32;
    if (!of32 && out_dtype != dtype) VV_FAIL(VV_E_ARG, "vv_layernorm_ex: out_dtype must be VV_F32 or the h16 dtype");
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)((M + 3) / 4));
#define LNX_VEC(NCH) SAM2_DT(hipLaunchKernelGGL((layernorm_ex_vec_kernel<BF16, NCH>), grid, dim3(256), 0, st, x, M, C, gamma, beta, eps, act, out, of32, cpad), \
                            hipLaunchKernelGGL((layernorm_ex_vec_kernel<F16, NCH>), grid, dim3(256), 0, st, x, M, C, gamma, beta, eps, act, out, of32, cpad))
    if (C % 4 == 0 && cpad % 4 == 0 && C >= 64 && cpad - C <= 252) {
        if (C <= 256) { LNX_VEC(1); VV_CHECK_LAUNCH("vv_layernorm_ex"); return VV_OK; }
        if (C <= 768) { LNX_VEC(3); VV_CHECK_LAUNCH("vv_layernorm_ex"); return VV_OK; }
        if (C <= 1280) { LNX_VEC(5); VV_CHECK_LAUNCH("vv_layernorm_ex"); return VV_OK; }
        if (C <= 2304) { LNX_VEC(9); VV_CHECK_LAUNCH("vv_layernorm_ex"); return VV_OK; }
    }
#undef LNX_VEC
    SAM2_DT(hipLaunchKernelGGL(layernorm_ex_kernel<BF16>, grid, dim3(256), 0, st, x, M, C, gamma, beta, eps, act, out, of32, cpad),
            hipLaunchKernelGGL(layernorm_ex_kernel<F16>, grid, dim3(256), 0, st, x, M, C, gamma, beta, eps, act, out, of32, cpad));
    VV_CHECK_LAUNCH("vv_layernorm_ex");
    return VV_OK;
}

extern "C" int vv_maxpool2x2(const void* x, int x_dtype, int B, int H, int W, int C, int64_t in_bs, void* out, void* stream) {
    if (in_bs <= 0) in_bs = (int64_t)H * W * C;
    if (!x || !out || B <= 0 || H < 2 || W < 2 || (H & 1) || (W & 1) || C <= 0) VV_FAIL(VV_E_ARG, "vv_maxpool2x2: bad arguments (even H, W)");
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = (int64_t)B * (H / 2) * (W / 2) * C;
    if (x_dtype == VV_F32) hipLaunchKernelGGL((maxpool2_kernel<F16, true>), grid1(n), dim3(256), 0, st, x, B, H, W, C, in_bs, out);
    else if (x_dtype == VV_BF16) hipLaunchKernelGGL((maxpool2_kernel<BF16, false>), grid1(n), dim3(256), 0, st, x, B, H, W, C, in_bs, out);
    else if (x_dtype == VV_F16) hipLaunchKernelGGL((maxpool2_kernel<F16, false>), grid1(n), dim3(256), 0, st, x, B, H, W, C, in_bs, out);
    else VV_FAIL(VV_E_ARG, "vv_maxpool2x2: bad dtype");
    VV_CHECK_LAUNCH("vv_maxpool2x2");
    return VV_OK;
}

extern "C" int vv_rope_apply(void* x, int64_t rows_rope, int ld, int col0, int D, const float* cos_sin, int n_table, int dtype, void* stream) {
    if (!x || !cos_sin || rows_rope < 0 || D <= 0 || (D & 1) || (col0 & 1) || (ld & 1) || n_table <= 0) VV_FAIL(VV_E_ARG, "vv_rope_apply: bad arguments");
    if (rows_rope == 0) return VV_OK;
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = rows_rope * (D / 2);
    SAM2_DT(hipLaunchKernelGGL(rope_kernel<BF16>, grid1(n), dim3(256), 0, st, (unsigned short*)x, rows_rope, ld, col0, D, cos_sin, n_table),
            hipLaunchKernelGGL(rope_kernel<F16>, grid1(n), dim3(256), 0, st, (unsigned short*)x, rows_rope, ld, col0, D, cos_sin, n_table));
    VV_CHECK_LAUNCH("vv_rope_apply");
    return VV_OK;
}

extern "C" int vv_dwconv(const float* x, int H, int W, int C, const float* w, const float* bias, int k, float* out, void* stream) {
    if (!x || !w || !bias || !out || H <= 0 || W <= 0 || C <= 0 || k <= 0 || !(k & 1)) VV_FAIL(VV_E_ARG, "vv_dwconv: bad arguments (odd k)");
    hipLaunchKernelGGL(dwconv_kernel, grid1((int64_t)H * W * C), dim3(256), 0, (hipStream_t)stream, x, H, W, C, w, bias, k, out);
    VV_CHECK_LAUNCH("vv_dwconv");
    return VV_OK;
}

extern "C" int vv_pixel_shuffle2(const float* y, const float* bias, const float* add, int h, int w, int C, int act, void* out, int out_dtype, int dtype,
                                 void* stream) {
    if (!y || !bias || !out || h <= 0 || w <= 0 || C <= 0) VV_FAIL(VV_E_ARG, "vv_pixel_shuffle2: bad arguments");
    const int of32 = out_dtype == VV_F32;
    if (!of32 && out_dtype != dtype) VV_FAIL(VV_E_ARG, "vv_pixel_shuffle2: out_dtype must be VV_F32 or the h16 dtype");
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = (int64_t)4 * h * w * C;
    SAM2_DT(hipLaunchKernelGGL(pixel_shuffle2_kernel<BF16>, grid1(n), dim3(256), 0, st, y, bias, add, h, w, C, act, out, of32),
            hipLaunchKernelGGL(pixel_shuffle2_kernel<F16>, grid1(n), dim3(256), 0, st, y, bias, add, h, w, C, act, out, of32));
    VV_CHECK_LAUNCH("vv_pixel_shuffle2");
    return VV_OK;
}

extern "C" int vv_resize_bilinear_f32(const float* src, int Hs, int Ws, int C, float* dst, int Hd, int Wd, void* stream) {
    if (!src || !dst || Hs <= 0 || Ws <= 0 || Hd <= 0 || Wd <= 0 || C <= 0) VV_FAIL(VV_E_ARG, "vv_resize_bilinear_f32: bad arguments");
    hipLaunchKernelGGL(resize_bilinear_f32_kernel, grid1((int64_t)Hd * Wd * C), dim3(256), 0, (hipStream_t)stream, src, Hs, Ws, C, dst, Hd, Wd,
                       (float)Hs / (float)Hd, (float)Ws / (float)Wd);
    VV_CHECK_LAUNCH("vv_resize_bilinear_f32");
    return VV_OK;
}

extern "C" int vv_mask_mem_input(const float* logits, int64_t n, int binarize, float scale, float bias, void* out, int dtype, void* stream) {
    if (!logits || !out || n <= 0) VV_FAIL(VV_E_ARG, "vv_mask_mem_input: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    SAM2_DT(hipLaunchKernelGGL(mask_mem_input_kernel<BF16>, grid1(n), dim3(256), 0, st, logits, n, binarize, scale, bias, (uint4*)out),
            hipLaunchKernelGGL(mask_mem_input_kernel<F16>, grid1(n), dim3(256), 0, st, logits, n, binarize, scale, bias, (uint4*)out));
    VV_CHECK_LAUNCH("vv_mask_mem_input");
    return VV_OK;
}

namespace {
template <typename T>
__global__ __launch_bounds__(256) void act_vec8_kernel(uint4* __restrict__ x, int64_t n8, int act) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    float v[8];
    unpack8<T>(x[i], v);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = act_f(v[e], act);
    x[i] = pack8<T>(v);
}
}  // namespace

extern "C" int vv_act(void* x, int x_dtype, int64_t n, int act, void* stream) {
    if (!x || n <= 0) VV_FAIL(VV_E_ARG, "vv_act: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (x_dtype != VV_F32 && n % 8 == 0 && ((uintptr_t)x & 15) == 0) {       // h16, 16 bytes per thread
        if (x_dtype == VV_BF16) hipLaunchKernelGGL(act_vec8_kernel<BF16>, grid1(n / 8), dim3(256), 0, st, (uint4*)x, n / 8, act);
        else if (x_dtype == VV_F16) hipLaunchKernelGGL(act_vec8_kernel<F16>, grid1(n / 8), dim3(256), 0, st, (uint4*)x, n / 8, act);
        else VV_FAIL(VV_E_ARG, "vv_act: bad dtype");
        VV_CHECK_LAUNCH("vv_act");
        return VV_OK;
    }
    if (x_dtype == VV_F32) hipLaunchKernelGGL((act_kernel<F16, true>), grid1(n), dim3(256), 0, st, x, n, act);
    else if (x_dtype == VV_BF16) hipLaunchKernelGGL((act_kernel<BF16, false>), grid1(n), dim3(256), 0, st, x, n, act);
    else if (x_dtype == VV_F16) hipLaunchKernelGGL((act_kernel<F16, false>), grid1(n), dim3(256), 0, st, x, n, act);
    else VV_FAIL(VV_E_ARG, "vv_act: bad dtype");
    VV_CHECK_LAUNCH("vv_act");
    return VV_OK;
}

extern "C" int vv_prompt_points(const float* coords, const int32_t* labels, int P, float inv_size, const float* gauss, const float* table, int D, float* out,
                                void* stream) {
    if (!coords || !labels || !gauss || !table || !out || P <= 0 || D <= 0 || (D & 1)) VV_FAIL(VV_E_ARG, "vv_prompt_points: bad arguments");
    hipLaunchKernelGGL(prompt_points_kernel, grid1((int64_t)P * D), dim3(256), 0, (hipStream_t)stream, coords, labels, P, inv_size, gauss, table, D, out);
    VV_CHECK_LAUNCH("vv_prompt_points");
    return VV_OK;
}

extern "C" int vv_sine_pe_1d(const float* pos, int n, int dim, float temperature, float* out, void* stream) {
    if (!pos || !out || n <= 0 || dim <= 0 || (dim & 1)) VV_FAIL(VV_E_ARG, "vv_sine_pe_1d: bad arguments");
    hipLaunchKernelGGL(sine_pe_1d_kernel, grid1((int64_t)n * dim), dim3(256), 0, (hipStream_t)stream, pos, n, dim, temperature, out);
    VV_CHECK_LAUNCH("vv_sine_pe_1d");
    return VV_OK;
}

extern "C" int vv_sam_select(const float* masks, int HW, int nm, const float* iou, const float* obj_logit, int multimask, float delta, float thresh,
                             int32_t* sel, void* stream) {
    if (!masks || !iou || !obj_logit || !sel || HW <= 0 || nm < 2) VV_FAIL(VV_E_ARG, "vv_sam_select: bad arguments");
    hipLaunchKernelGGL(sam_select_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, masks, HW, nm, iou, obj_logit, multimask, delta, thresh, sel);
    VV_CHECK_LAUNCH("vv_sam_select");
    return VV_OK;
}

extern "C" int vv_sam_pick(const float* masks, int HW, const int32_t* sel, float no_obj_score, float* out, void* stream) {
    if (!masks || !sel || !out || HW <= 0) VV_FAIL(VV_E_ARG, "vv_sam_pick: bad arguments");
    hipLaunchKernelGGL(sam_pick_kernel, grid1(HW), dim3(256), 0, (hipStream_t)stream, masks, HW, sel, no_obj_score, out);
    VV_CHECK_LAUNCH("vv_sam_pick");
    return VV_OK;
}

extern "C" int vv_select_f32(const float* a, const float* b, const int32_t* flag, int64_t n, float* out, void* stream) {
    if (!a || !b || !flag || !out || n <= 0) VV_FAIL(VV_E_ARG, "vv_select_f32: bad arguments");
    hipLaunchKernelGGL(select_rows_kernel, grid1(n), dim3(256), 0, (hipStream_t)stream, a, b, flag, n, out);
    VV_CHECK_LAUNCH("vv_select_f32");
    return VV_OK;
}

extern "C" int vv_add_rowvec_unless(float* x, const float* vec, const float* score, int64_t M, int C, void* stream) {
    if (!x || !vec || !score || M <= 0 || C <= 0) VV_FAIL(VV_E_ARG, "vv_add_rowvec_unless: bad arguments");
    hipLaunchKernelGGL(add_rowvec_unless_kernel, grid1(M * C), dim3(256), 0, (hipStream_t)stream, x, vec, score, M, C);
    VV_CHECK_LAUNCH("vv_add_rowvec_unless");
    return VV_OK;
}

extern "C" int vv_clamp_f32(const float* x, int64_t n, float lo, float hi, float* out, void* stream) {
    if (!x || !out || n <= 0) VV_FAIL(VV_E_ARG, "vv_clamp_f32: bad arguments");
    hipLaunchKernelGGL(clamp_kernel, grid1(n), dim3(256), 0, (hipStream_t)stream, x, n, lo, hi, out);
    VV_CHECK_LAUNCH("vv_clamp_f32");
    return VV_OK;
}

extern "C" int vv_fill_holes(float* mask, int H, int W, int max_area, int32_t* ws, void* stream) {
    if (!mask || !ws || H <= 0 || W <= 0 || max_area <= 0 || max_area > 64) VV_FAIL(VV_E_ARG, "vv_fill_holes: bad arguments (0 < max_area <= 64)");
    hipStream_t st = (hipStream_t)stream;
    const int n = H * W;
    int *lab = ws, *area = ws + n, *open = ws + 2 * n;
    hipLaunchKernelGGL(holes_init_kernel, grid1(n), dim3(256), 0, st, mask, n, lab, area, open);
    for (int it = 0; it < max_area; ++it) hipLaunchKernelGGL(holes_prop_kernel, grid1(n), dim3(256), 0, st, lab, H, W);
    hipLaunchKernelGGL(holes_count_kernel, grid1(n), dim3(256), 0, st, lab, H, W, area, open);
    hipLaunchKernelGGL(holes_apply_kernel, grid1(n), dim3(256), 0, st, mask, n, lab, area, open, max_area);
    VV_CHECK_LAUNCH("vv_fill_holes");
    return VV_OK;
}

extern "C" int vv_hyper_masks(const float* hyper, const float* up, int HW, int C, int nm, float* masks, void* stream) {
    if (!hyper || !up || !masks || HW <= 0 || C <= 0 || nm <= 0 || nm > 8) VV_FAIL(VV_E_ARG, "vv_hyper_masks: bad arguments (nm <= 8)");
    hipLaunchKernelGGL(hyper_masks_kernel, grid1(HW), dim3(256), 0, (hipStream_t)stream, hyper, up, HW, C, nm, masks);
    VV_CHECK_LAUNCH("vv_hyper_masks");
    return VV_OK;
}

extern "C" int vv_sam2_maskdown(const float* logits, int lo, int S, int binarize, float scale, float bias, const float* w1, const float* b1, const float* g1,
                                const float* be1, const float* w2, const float* b2, const float* g2, const float* be2, float eps, void* mid, void* out, int dtype,
                                void* stream) {
    if (!logits || !w1 || !b1 || !g1 || !be1 || !w2 || !b2 || !g2 || !be2 || !mid || !out || lo <= 0 || S < 4 || (S & 3)) VV_FAIL(VV_E_ARG, "vv_sam2_maskdown: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int H1 = S / 2, H2 = S / 4;
    SAM2_DT(hipLaunchKernelGGL((maskdown1_kernel<BF16, 4>), grid1((int64_t)H1 * H1), dim3(256), 0, st, logits, lo, S, binarize, scale, bias, w1, b1, g1, be1, eps, (unsigned short*)mid, 8),
            hipLaunchKernelGGL((maskdown1_kernel<F16, 4>), grid1((int64_t)H1 * H1), dim3(256), 0, st, logits, lo, S, binarize, scale, bias, w1, b1, g1, be1, eps, (unsigned short*)mid, 8));
    SAM2_DT(hipLaunchKernelGGL((conv3s2_ln_gelu_kernel<BF16, 4, 16>), grid1((int64_t)H2 * H2), dim3(256), 0, st, (const unsigned short*)mid, H1, H1, 8, w2, b2, g2, be2, eps, (unsigned short*)out, 16),
            hipLaunchKernelGGL((conv3s2_ln_gelu_kernel<F16, 4, 16>), grid1((int64_t)H2 * H2), dim3(256), 0, st, (const unsigned short*)mid, H1, H1, 8, w2, b2, g2, be2, eps, (unsigned short*)out, 16));
    VV_CHECK_LAUNCH("vv_sam2_maskdown");
    return VV_OK;
}
