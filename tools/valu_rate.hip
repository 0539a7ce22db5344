// Micro-benchmark: issue cost (cycles per wave64 instruction) of the VALU ops the attention softmax is made of, measured
// relative to v_fma_f32 (4 cycles).  hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP16(X) X X X X X X X X X X X X X X X X
template <int OP>
__global__ void k(float* out, int iters) {
    float a[8]; f2 p[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; p[i] = f2{a[i], a[i] + 1.f}; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[i]));
                if (OP == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
                if (OP == 2) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p[i]));
                if (OP == 3) asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(a[i]));
                if (OP == 4) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(a[i]));
                if (OP == 5) asm volatile("v_mov_b64 %0, %0" : "+v"(p[i]));
                if (OP == 6) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(p[i]));
                if (OP == 7) asm volatile("v_exp_f16 %0, %0" : "+v"(a[i]));
                if (OP == 8) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(p[i]));
                if (OP == 9) asm volatile("v_lshl_add_u32 %0, %0, 3, %0" : "+v"(a[i]));
                if (OP == 10) asm volatile("v_pk_mul_f16 %0, %0, %0" : "+v"(a[i]));
                if (OP == 11) asm volatile("v_cvt_pk_f16_f32 %0, %0, %0" : "+v"(a[i]));
                if (OP == 12) asm volatile("v_ldexp_f32 %0, %0, %0" : "+v"(a[i]));
                if (OP == 13) asm volatile("v_max_f32 %0, %0, %0" : "+v"(a[i]));
                if (OP == 14) asm volatile("v_add_f32 %0, %0, %0" : "+v"(a[i]));
                if (OP == 15) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(a[i]));
                if (OP == 16) asm volatile("v_mov_b32 %0, %0" : "+v"(a[i]));
                if (OP == 17) asm volatile("v_sub_f32 %0, %0, %0" : "+v"(a[i]));
            }
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
typedef float f4 __attribute__((ext_vector_type(4)));
typedef short s8 __attribute__((ext_vector_type(8)));
// MODE 0: 8 independent MFMAs per iteration; 1: + 16 v_exp_f32; 2: + 16 v_max3_f32; 3: only the 16 exps; 4: + 32 v_exp
template <int MODE>
__global__ void km(float* out, int iters) {
    f4 acc[8]; float a[8];
    s8 x; for (int i = 0; i < 8; ++i) { x[i] = (short)(threadIdx.x + i); acc[i] = f4{0, 0, 0, 0}; a[i] = threadIdx.x * 0.001f + i; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE != 3) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %1, %0" : "+v"(acc[i]) : "v"(x));
            if (MODE == 1 || MODE == 3 || MODE == 4) { asm volatile("v_exp_f32 %0, %0" : "+v"(a[i])); asm volatile("v_exp_f32 %0, %0" : "+v"(a[(i + 4) & 7])); }
            if (MODE == 4) { asm volatile("v_exp_f32 %0, %0" : "+v"(a[(i + 2) & 7])); asm volatile("v_exp_f32 %0, %0" : "+v"(a[(i + 6) & 7])); }
            if (MODE == 2) { asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(a[i])); asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(a[(i + 4) & 7])); }
        }
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + acc[i][0] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void runm(float* d, const char* name, int wps) {
    const int iters = 2000, blocks = 256 * wps, thr = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    km<MODE><<<blocks, thr>>>(d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0); km<MODE><<<blocks, thr>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double it_per_simd = (double)blocks * (thr / 64) * iters / 1024.0;
    printf("%-44s %d waves/SIMD: %.2f ns per (8 MFMA [+VALU]) group per SIMD\n", name, wps, ms * 1e6 / it_per_simd);
}
template <int OP> double run(float* d, const char* name, double ref) {
    const int iters = 2000, blocks = 256 * 8, thr = 256;   // 8 waves per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, thr>>>(d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0); k<OP><<<blocks, thr>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)blocks * (thr / 64) * iters * 32 / 1024.0;
    const double ns = ms * 1e6 / instr_per_simd;
    printf("%-22s %.3f ns / wave-instr / SIMD  (= %.2f cycles if v_fma_f32 is 4)\n", name, ns, ref > 0 ? 4.0 * ns / ref : 4.0);
    return ns;
}
int main() {
    float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    double r = run<0>(d, "v_fma_f32", 0);
    run<1>(d, "v_exp_f32", r); run<7>(d, "v_exp_f16", r); run<2>(d, "v_pk_fma_f32", r); run<6>(d, "v_pk_mul_f32", r); run<8>(d, "v_pk_add_f32", r);
    run<3>(d, "v_max3_f32", r); run<4>(d, "v_cvt_pk_bf16_f32", r); run<11>(d, "v_cvt_pk_f16_f32", r); run<5>(d, "v_mov_b64", r);
    run<9>(d, "v_lshl_add_u32", r); run<10>(d, "v_pk_mul_f16", r); run<12>(d, "v_ldexp_f32", r);
    run<13>(d, "v_max_f32", r); run<14>(d, "v_add_f32", r); run<15>(d, "v_mul_f32", r); run<16>(d, "v_mov_b32", r); run<17>(d, "v_sub_f32", r);
    for (int wps : {1, 2, 4}) {
        runm<0>(d, "8 mfma 16x16x32 bf16", wps); runm<3>(d, "16 v_exp_f32", wps); runm<1>(d, "8 mfma + 16 v_exp_f32", wps);
        runm<4>(d, "8 mfma + 32 v_exp_f32", wps); runm<2>(d, "8 mfma + 16 v_max3_f32", wps);
    }
    return 0;
}
