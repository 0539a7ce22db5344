/* ORACLE (test infrastructure, never linked into the product): plain-C restatement of the integer / byte image steps
 * on the hot path, fast enough for FULL-SIZE (720p / 1080p) bit-exact checks of the HIP kernels.
 *   dilate_cross_u8        reference diffuerase.py:30  (scipy.ndimage.binary_dilation, default 3x3 cross, k iterations;
 *                                                      k < 1: until convergence)
 *   distance_transform_l2_5 reference diffuerase.py:95-96 (cv2.distanceTransform(., DIST_L2, 5): two-pass 5x5 chamfer,
 *                                                      16.16 fixed point a=1 b=1.4 c=2.1969; restated from OpenCV's
 *                                                      published algorithm -- parity unpinned against a real cv2)
 *   feather_composite_u8   reference diffuerase.py:99-112 (alpha ramp + rint/clip composite, float32 arithmetic)
 *   resize_bilinear_u8     reference diffuerase.py:73  (cv2.resize INTER_LINEAR, 8-bit fixed-point path; unpinned)
 * The numpy versions in oracle/imageops_ref.py are the ones pinned against the reference fixtures; tests check C == numpy
 * on small inputs before using the C version at full size.
 * Build: make -C oracle   (gcc -O2 -fno-fast-math -ffp-contract=off)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define HV 65536
#define DIAG 91750
#define LONGD 143976
#define INIT_DIST0 (0x7fffffff >> 2)

void dilate_cross_u8(uint8_t* m, int H, int W, int iterations, uint8_t* tmp) {
    int it = 0;
    for (;;) {
        int changed = 0;
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                uint8_t v = m[y * W + x];
                if (y > 0) v |= m[(y - 1) * W + x];
                if (y < H - 1) v |= m[(y + 1) * W + x];
                if (x > 0) v |= m[y * W + x - 1];
                if (x < W - 1) v |= m[y * W + x + 1];
                tmp[y * W + x] = v;
                changed |= v != m[y * W + x];
            }
        memcpy(m, tmp, (size_t)H * W);
        ++it;
        if (iterations >= 1 && it >= iterations) break;
        if (iterations < 1 && !changed) break;
    }
}

void distance_transform_l2_5(const uint8_t* src, int H, int W, float* out) {
    const int B = 2, S = W + 2 * B;
    int64_t* t = (int64_t*)malloc(sizeof(int64_t) * (size_t)(H + 2 * B) * S);
    for (size_t i = 0; i < (size_t)(H + 2 * B) * S; ++i) t[i] = INIT_DIST0;
    static const int dy[8] = {-2, -2, -1, -1, -1, -1, -1, 0}, dx[8] = {-1, 1, -2, -1, 0, 1, 2, -1};
    static const int c[8] = {LONGD, LONGD, LONGD, DIAG, HV, DIAG, LONGD, HV};
    for (int i = 0; i < H; ++i)
        for (int j = 0; j < W; ++j) {
            int64_t* p = t + (size_t)(i + B) * S + j + B;
            if (src[i * W + j] == 0) { *p = 0; continue; }
            int64_t best = (int64_t)INIT_DIST0 * 2;
            for (int k = 0; k < 8; ++k) { int64_t v = p[dy[k] * S + dx[k]] + c[k]; if (v < best) best = v; }
            *p = best;
        }
    const float scale = 1.0f / 65536.0f;
    for (int i = H - 1; i >= 0; --i)
        for (int j = W - 1; j >= 0; --j) {
            int64_t* p = t + (size_t)(i + B) * S + j + B;
            int64_t t0 = *p;
            if (t0 > HV) {
                for (int k = 0; k < 8; ++k) { int64_t v = p[-dy[k] * S - dx[k]] + c[k]; if (v < t0) t0 = v; }
                *p = t0;
            }
            if (t0 > INIT_DIST0) t0 = INIT_DIST0;
            out[i * W + j] = (float)t0 * scale;
        }
    free(t);
}

void feather_composite_u8(const uint8_t* inp, const uint8_t* orig, const uint8_t* mask, int H, int W, float feather, uint8_t* out) {
    const size_t n = (size_t)H * W;
    uint8_t* bin = (uint8_t*)malloc(n);
    uint8_t* inv = (uint8_t*)malloc(n);
    float* din = (float*)malloc(n * sizeof(float));
    float* dout = (float*)malloc(n * sizeof(float));
    for (size_t i = 0; i < n; ++i) { bin[i] = mask[i] > 0 ? 255 : 0; inv[i] = (uint8_t)~bin[i]; }
    if (feather > 0.f) { distance_transform_l2_5(bin, H, W, din); distance_transform_l2_5(inv, H, W, dout); }
    for (size_t i = 0; i < n; ++i) {
        float a;
        if (feather > 0.f) {
            a = 0.5f + (din[i] - dout[i]) / (2.0f * feather);
            a = a < 0.f ? 0.f : (a > 1.f ? 1.f : a);
        } else a = bin[i] ? 1.f : 0.f;
        const float om = 1.0f - a;
        for (int ch = 0; ch < 3; ++ch) {
            const float x = a * (float)inp[i * 3 + ch], y = om * (float)orig[i * 3 + ch];
            float v = rintf(x + y);
            v = v < 0.f ? 0.f : (v > 255.f ? 255.f : v);
            out[i * 3 + ch] = (uint8_t)v;
        }
    }
    free(bin); free(inv); free(din); free(dout);
}

static void lin_coef(int d, int ssize, int dsize, int* s0, int* s1, int* a0, int* a1) {
    const double scale = (double)ssize / (double)dsize;
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
    *a0 = (int)rintf((1.0f - f) * 2048.0f);
    *a1 = (int)rintf(f * 2048.0f);
    *s0 = s; *s1 = s + 1 < ssize ? s + 1 : ssize - 1;
}

void resize_bilinear_u8(const uint8_t* src, int Hs, int Ws, int ch, uint8_t* dst, int Hd, int Wd) {
    for (int y = 0; y < Hd; ++y) {
        int y0, y1, b0, b1;
        lin_coef(y, Hs, Hd, &y0, &y1, &b0, &b1);
        for (int x = 0; x < Wd; ++x) {
            int x0, x1, a0, a1;
            lin_coef(x, Ws, Wd, &x0, &x1, &a0, &a1);
            for (int c = 0; c < ch; ++c) {
                const int h0 = src[((size_t)y0 * Ws + x0) * ch + c] * a0 + src[((size_t)y0 * Ws + x1) * ch + c] * a1;
                const int h1 = src[((size_t)y1 * Ws + x0) * ch + c] * a0 + src[((size_t)y1 * Ws + x1) * ch + c] * a1;
                int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
                dst[((size_t)y * Wd + x) * ch + c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
            }
        }
    }
}
