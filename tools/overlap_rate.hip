// Micro-benchmark: do the matrix pipe and the VALU of one SIMD overlap ACROSS waves?  Waves alternate an MFMA phase (16 MFMAs on 8
// accumulators) and a VALU phase (32 v_exp_f32 + 64 v_max3_f32) like the attention kernel's QK^T / softmax / PV phases.
//   MODE 0: MFMA phase only   1: VALU phase only   2: both, phase-alternating in every wave   3: role split (even waves of a SIMD MFMA only,
//   odd waves VALU only)      4: both, finely interleaved in program order (1 MFMA : 2 exp : 4 max3)
// hipcc --offload-arch=gfx950 -O3 tools/overlap_rate.hip -o /tmp/overlap_rate && /tmp/overlap_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef short s8 __attribute__((ext_vector_type(8)));
template <int MODE, bool AGPR>
__global__ __launch_bounds__(256) void kp(float* out, int iters) {
    f4 acc[8]; float a[8];
    s8 x; for (int i = 0; i < 8; ++i) { x[i] = (short)(threadIdx.x + i); acc[i] = f4{0, 0, 0, 0}; a[i] = threadIdx.x * 0.001f + i; }
    const int wave = threadIdx.x >> 6;            // 256 threads = 4 waves = one per SIMD; the role alternates with the block index
    const bool mrole = MODE != 3 || (blockIdx.x & 1) == 0, vrole = MODE != 3 || (blockIdx.x & 1) == 1;
    (void)wave;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 4) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (AGPR) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %1, %0" : "+a"(acc[i & 7]) : "v"(x));
                else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %1, %0" : "+v"(acc[i & 7]) : "v"(x));
                asm volatile("v_exp_f32 %0, %0" : "+v"(a[i & 7])); asm volatile("v_exp_f32 %0, %0" : "+v"(a[(i + 4) & 7]));
                asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(a[(i + 1) & 7])); asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(a[(i + 2) & 7]));
                asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(a[(i + 3) & 7])); asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(a[(i + 5) & 7]));
            }
            continue;
        }
        if ((MODE == 0 || MODE == 2 || MODE == 3) && mrole) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (AGPR) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %1, %0" : "+a"(acc[i & 7]) : "v"(x));
                else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %1, %0" : "+v"(acc[i & 7]) : "v"(x));
            }
        }
        if ((MODE == 1 || MODE == 2 || MODE == 3) && vrole) {
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                asm volatile("v_exp_f32 %0, %0" : "+v"(a[i & 7]));
                asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(a[(i + 1) & 7])); asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(a[(i + 2) & 7]));
            }
        }
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + acc[i][0] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE, bool AGPR> void run(float* d, const char* name, int wps) {
    const int iters = 2000, blocks = 256 * wps, thr = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kp<MODE, AGPR><<<blocks, thr>>>(d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0); kp<MODE, AGPR><<<blocks, thr>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: wps waves, each does `iters` rounds (role split: half the waves do each phase)
    const double rounds = (double)wps * iters * (MODE == 3 ? 0.5 : 1.0);
    printf("%-64s %s %d waves/SIMD: %8.1f ns per (16 MFMA | 32 exp + 64 max3) round per SIMD\n", name, AGPR ? "agpr" : "vgpr", wps, ms * 1e6 / rounds);
}
// Exact pairing: ONE 512-thread block per CU, waves 0-3 (one per SIMD) run only the MFMA phase, waves 4-7 (the second wave of each SIMD) only the
// VALU phase, same trip count -- the situation a "ping-pong" kernel with role-split wave groups creates.  MODE 0: both; 1: MFMA waves only (the others
// exit); 2: VALU waves only.
template <int MODE>
__global__ __launch_bounds__(512) void kpair(float* out, int iters) {
    f4 acc[8]; float a[8];
    s8 x; for (int i = 0; i < 8; ++i) { x[i] = (short)(threadIdx.x + i); acc[i] = f4{0, 0, 0, 0}; a[i] = threadIdx.x * 0.001f + i; }
    const bool mrole = threadIdx.x < 256;
    if ((MODE == 1 && !mrole) || (MODE == 2 && mrole)) return;
    for (int it = 0; it < iters; ++it) {
        if (mrole) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %1, %0" : "+v"(acc[i & 7]) : "v"(x));
        } else {
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                asm volatile("v_exp_f32 %0, %0" : "+v"(a[i & 7]));
                asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(a[(i + 1) & 7])); asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(a[(i + 2) & 7]));
            }
        }
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + acc[i][0] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void runpair(float* d, const char* name) {
    const int iters = 4000, blocks = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kpair<MODE><<<blocks, 512>>>(d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0); kpair<MODE><<<blocks, 512>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-72s %8.1f ns per round (16 MFMA on one wave | 32 exp + 64 max3 on the other wave of the SIMD)\n", name, ms * 1e6 / iters);
}
int main() {
    float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int wps : {2, 3, 4, 6}) {
        run<0, false>(d, "MFMA phase only (16 x 7.3 = 117 ns)", wps);
        run<1, false>(d, "VALU phase only (32 x 3.5 + 64 x 1.8 = 227 ns)", wps);
        run<2, false>(d, "phase-alternating waves", wps);
        run<2, true>(d, "phase-alternating waves", wps);
        run<3, false>(d, "role split across blocks (per MFMA+VALU pair of waves)", wps);
        run<4, false>(d, "interleaved in program order", wps);
        run<4, true>(d, "interleaved in program order", wps);
    }
    runpair<1>(d, "paired waves: MFMA wave alone");
    runpair<2>(d, "paired waves: VALU wave alone");
    runpair<0>(d, "paired waves: MFMA wave + VALU wave on every SIMD");
    return 0;
}
