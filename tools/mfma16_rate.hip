// Micro-benchmark: issue/pipe time of v_mfma_f32_16x16x16_f16 against v_mfma_f32_16x16x32_f16 on gfx950 (8 independent accumulators, 4 waves/SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef short s8 __attribute__((ext_vector_type(8)));
typedef short s4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void k(float* out, int iters) {
    f4 acc[8]; s8 x; s4 y;
    for (int i = 0; i < 8; ++i) { x[i] = (short)(threadIdx.x + i); acc[i] = f4{0, 0, 0, 0}; }
    for (int i = 0; i < 4; ++i) y[i] = (short)(threadIdx.x * 3 + i);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0 || MODE == 2) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %1, %0" : "+v"(acc[i]) : "v"(x));
            if (MODE == 1 || MODE == 2) asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %1, %0" : "+v"(acc[i]) : "v"(y));
            if (MODE == 3) asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %1, %0" : "+v"(acc[i]) : "v"(y));
        }
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(float* d, const char* name) {
    const int iters = 2000, blocks = 256 * 4, thr = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, thr>>>(d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0); k<MODE><<<blocks, thr>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s %.2f ns per group of 8 per SIMD\n", name, ms * 1e6 / ((double)blocks * 4 * iters / 1024.0));
}
int main() {
    float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>(d, "8 x 16x16x32 f16"); run<1>(d, "8 x 16x16x16 f16"); run<2>(d, "8 x (16x16x32 + 16x16x16) f16"); run<3>(d, "8 x 16x16x16 bf16");
    return 0;
}
