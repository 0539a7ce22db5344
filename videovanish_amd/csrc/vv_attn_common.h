// LDS / LDS-DMA helpers shared by the attention translation units (vv_attn.hip, vv_attn32.hip).
#pragma once
#include "vv_common.h"

namespace {

__device__ __forceinline__ uint2 ds_read_tr16(const unsigned char* lds_ptr) {
    typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_p;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)lds_ptr);
    return __builtin_bit_cast(uint2, v);
}

__device__ __forceinline__ void glds16(const void* gptr, void* lds_wave_base) {
    typedef const void __attribute__((address_space(1))) * gp_t;
    typedef void __attribute__((address_space(3))) * lp_t;
    __builtin_amdgcn_global_load_lds((gp_t)gptr, (lp_t)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ void glds16_asm(const void* gptr, void* lds_wave_base) {
    typedef void __attribute__((address_space(3))) * lp_t;
    const unsigned dst = (unsigned)(size_t)(lp_t)lds_wave_base;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gptr), "s"(dst) : "memory");
}

}  // namespace
