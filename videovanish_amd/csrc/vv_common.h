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

template <typename T> __device__ __forceinline__ unsigned pack2(float lo, float hi) {
    return (unsigned)T::from_f32(lo) | ((unsigned)T::from_f32(hi) << 16);
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
// exact (erf) GELU, matching torch.nn.functional.gelu default
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

// host-side error plumbing -------------------------------------------------------------------------------------
void vv_set_error(const char* fmt, ...);
#define VV_FAIL(code, ...) do { vv_set_error(__VA_ARGS__); return (code); } while (0)
#define VV_CHECK_LAUNCH(name) do { hipError_t e_ = hipGetLastError(); \
    if (e_ != hipSuccess) VV_FAIL(VV_E_LAUNCH, "%s: launch failed: %s", name, hipGetErrorString(e_)); } while (0)
