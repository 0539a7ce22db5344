// Shared device/host helpers for libvvhip (gfx950 only: wave64, MFMA 16x16x32, ds_read_b64_tr_b16).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/vvhip.h"

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;

struct BF16 {
    typedef __bf16 elem;
    static __device__ __forceinline__ unsigned short from_f32(float x) { return __builtin_bit_cast(unsigned short, (__bf16)x); }
    static __device__ __forceinline__ float to_f32(unsigned short u) { return __builtin_bit_cast(float, (unsigned)u << 16); }
    static __device__ __forceinline__ f32x4 mfma(uint4 a, uint4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
struct F16 {
    typedef _Float16 elem;
    static __device__ __forceinline__ unsigned short from_f32(float x) { return __builtin_bit_cast(unsigned short, (_Float16)x); }
    static __device__ __forceinline__ float to_f32(unsigned short u) { return (float)__builtin_bit_cast(_Float16, u); }
    static __device__ __forceinline__ f32x4 mfma(uint4 a, uint4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};

// two fp32 -> one dword of two h16 (round to nearest even).  A vector fptrunc lowers to ONE v_cvt_pk_{bf16,f16}_f32 on
// gfx950 and -- unlike inline asm -- stays visible to the compiler's hazard recognizer (an asm-written VGPR consumed by
// the next MFMA missed its wait states: NaNs in one fp16 instantiation).
typedef float vv_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 vv_bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 vv_f16x2 __attribute__((ext_vector_type(2)));
template <typename T> __device__ __forceinline__ unsigned pack2(float lo, float hi);
template <> __device__ __forceinline__ unsigned pack2<BF16>(float lo, float hi) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector((vv_f32x2){lo, hi}, vv_bf16x2));
}
template <> __device__ __forceinline__ unsigned pack2<F16>(float lo, float hi) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector((vv_f32x2){lo, hi}, vv_f16x2));
}
template <typename T> __device__ __forceinline__ uint4 pack8(const float* v) {
    return make_uint4(pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3]), pack2<T>(v[4], v[5]), pack2<T>(v[6], v[7]));
}
template <typename T> __device__ __forceinline__ void unpack8(uint4 u, float* v) {
    v[0] = T::to_f32(u.x & 0xffff); v[1] = T::to_f32(u.x >> 16);
    v[2] = T::to_f32(u.y & 0xffff); v[3] = T::to_f32(u.y >> 16);
    v[4] = T::to_f32(u.z & 0xffff); v[5] = T::to_f32(u.z >> 16);
    v[6] = T::to_f32(u.w & 0xffff); v[7] = T::to_f32(u.w >> 16);
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
// exact (erf) GELU, matching torch.nn.functional.gelu default.  erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below the h16 rounding
// that follows: one v_rcp + one v_exp + 6 FMAs instead of ocml's two-branch erff) -- the form the fused kernels (vv_motion.hip, vv_chain.hip) have
// used since round 2, the product default everywhere since round 4 (the GEGLU GEMMs gain 5-10 %: profiles/r3_gelu_as_ab.txt; parity re-validated
// against the 1e-3 asserts: profiles/r4_parity_gpu.txt).  -DVV_GELU_ERFF (lab) brings ocml's erff back.
#ifdef VV_GELU_ERFF
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
#else
__device__ __forceinline__ float gelu_f(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __frcp_rn(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f); p = fmaf(p, t, -0.284496736f); p = fmaf(p, t, 0.254829592f);
    const float erfz = 1.0f - p * t * __expf(-z * z);
    return 0.5f * x * (1.0f + copysignf(erfz, x));
}
#endif

// Two exact-erf GELUs at once WITHOUT transcendentals (round 6): gelu(x) = x/2 + |x|/2 * E(|x| / sqrt 2), E(z) = erf(z) on [0, 3.5] as z * Q(w), w = 2 z^2 / 3.5^2 - 1,
// Q = the degree-12 Chebyshev fit of erf(z) / z in z^2 (monomial coefficients in w, all <= 0.41 in magnitude: well conditioned in fp32), z clamped to 3.5
// (1 - erf(3.5) = 7.4e-7).  Evaluated in fp32: |error| <= 1.8e-6 over all x (the A&S form above: 4.7e-7; the h16 rounding that follows: 4.9e-4 relative) --
// tools/gelu_poly_fit.py generates the coefficients and prints the error.  ~20 packed fp32 instructions per PAIR (v_pk_mul / v_pk_fma: two lanes' worth per issue)
// against ~13 scalar instructions + v_rcp + v_exp per ELEMENT: the GEGLU epilogue of the short-K feed-forward GEMMs spends 30-45 % of its tile time in the
// activation (K = 640: 760 TFLOP/s against 985 at K = 1280, same kernel: DESIGN 5, round 6).
__device__ __forceinline__ vv_f32x2 gelu_poly2(vv_f32x2 x) {
    const vv_f32x2 ax = {fabsf(x.x), fabsf(x.y)};
    vv_f32x2 z = ax * 0.70710678118654752f;
    z = (vv_f32x2){fminf(z.x, 3.5f), fminf(z.y, 3.5f)};
    const vv_f32x2 w = __builtin_elementwise_fma(z * z, (vv_f32x2){0.16326530612244897f, 0.16326530612244897f}, (vv_f32x2){-1.0f, -1.0f});
    vv_f32x2 q = {0.0017835492035374045f, 0.0017835492035374045f};
#define VV_GP(c) q = __builtin_elementwise_fma(q, w, (vv_f32x2){c, c})
    VV_GP(-0.004138993564993143f); VV_GP(0.0036422861739993095f); VV_GP(-0.006848857272416353f); VV_GP(0.017900297418236732f);
    VV_GP(-0.030372897163033485f); VV_GP(0.04461858794093132f); VV_GP(-0.06463798880577087f); VV_GP(0.08848482370376587f);
    VV_GP(-0.1146334782242775f); VV_GP(0.1467439830303192f); VV_GP(-0.20070014894008636f); VV_GP(0.4038730561733246f);
#undef VV_GP
    return __builtin_elementwise_fma(ax * 0.5f, z * q, x * 0.5f);
}

// host-side error plumbing -------------------------------------------------------------------------------------
void vv_set_error(const char* fmt, ...);
#define VV_FAIL(code, ...) do { vv_set_error(__VA_ARGS__); return (code); } while (0)
#define VV_CHECK_LAUNCH(name) do { hipError_t e_ = hipGetLastError(); \
    if (e_ != hipSuccess) VV_FAIL(VV_E_LAUNCH, "%s: launch failed: %s", name, hipGetErrorString(e_)); } while (0)
