// Micro-benchmark: the K/V tile stream of the d = 40 spatial attention kernel (64 keys x 80 B of K + 64 x 80 B of V per tile, L2 resident,
// 3 blocks of 256 threads per CU, barrier per tile, NO compute) in two lane mappings of the LDS-DMA:
//   MODE 0: today's padded LDS rows (K pitch 160 B, V pitch 96 B): 16 wave instructions per tile, pad lanes masked off
//   MODE 1: dense LDS rows (pitch 80 B): 10 wave instructions per tile, every lane active
// hipcc --offload-arch=gfx950 -O3 tools/attn_fill_rate.hip -o /tmp/afr && /tmp/afr
#include <hip/hip_runtime.h>
#include <stdio.h>
__device__ __forceinline__ void glds16(const void* g, void* l) {
    typedef const void __attribute__((address_space(1))) * gp_t;
    typedef void __attribute__((address_space(3))) * lp_t;
    __builtin_amdgcn_global_load_lds((gp_t)g, (lp_t)l, 16, 0, 0);
}
template <int MODE>
__global__ __launch_bounds__(256, 3) void fill(const unsigned char* K, const unsigned char* V, int ntiles, int nkv_sets, float* out) {
    __shared__ __attribute__((aligned(1024))) unsigned char sK[2][64 * 160];
    __shared__ __attribute__((aligned(1024))) unsigned char sV[2][64 * 96];
    const int t = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const size_t set = (size_t)(blockIdx.x & 7) % nkv_sets;          // blocks of one XCD share a (frame, head) pair
    const unsigned char* kp = K + set * (size_t)ntiles * 5120;
    const unsigned char* vp = V + set * (size_t)ntiles * 5120;
    float acc = 0.f;
    for (int it = 0; it < ntiles; ++it) {
        unsigned char* bK = sK[it & 1]; unsigned char* bV = sV[it & 1];
        const unsigned char* kt = kp + (size_t)it * 5120; const unsigned char* vt = vp + (size_t)it * 5120;
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int sidx = i * 256 + t, row = sidx / 10, ch = sidx % 10;
                if (i * 256 + wave * 64 < 640) { if (ch < 5) glds16(kt + row * 80 + ch * 16, bK + (i * 256 + wave * 64) * 16); }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int sidx = i * 256 + t, row = sidx / 6, ch = sidx % 6;
                if (i * 256 + wave * 64 < 384) { if (ch < 5) glds16(vt + row * 80 + ch * 16, bV + (i * 256 + wave * 64) * 16); }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int sidx = i * 256 + t;
                if (i * 256 + wave * 64 < 320) glds16(kt + sidx * 16, bK + (i * 256 + wave * 64) * 16);
                if (i * 256 + wave * 64 < 320) glds16(vt + sidx * 16, bV + (i * 256 + wave * 64) * 16);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        acc += *(float*)(bK + t * 4);
    }
    if (acc == 123.456f) out[0] = acc;
}
template <int MODE> void run(const unsigned char* K, const unsigned char* V, float* out, const char* name) {
    const int ntiles = 225, blocks = 256 * 3 * 8;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    fill<MODE><<<blocks, 256>>>(K, V, ntiles, 8, out);
    hipDeviceSynchronize();
    hipEventRecord(e0); fill<MODE><<<blocks, 256>>>(K, V, ntiles, 8, out); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double tiles = (double)blocks * ntiles;
    printf("%-28s %.3f ms  %.1f ns per tile per CU-slot(3/CU)  %.2f TB/s of K/V data into LDS\n", name, ms, ms * 1e6 / (tiles / (256.0 * 3)), tiles * 10240 / ms / 1e9);
}
int main() {
    unsigned char *K, *V; float* out;
    hipMalloc(&K, 8 * 225 * 5120 + 4096); hipMalloc(&V, 8 * 225 * 5120 + 4096); hipMalloc(&out, 16);
    hipMemset(K, 1, 8 * 225 * 5120); hipMemset(V, 1, 8 * 225 * 5120);
    for (int r = 0; r < 2; ++r) { run<0>(K, V, out, "padded rows, masked lanes"); run<1>(K, V, out, "dense rows"); }
    return 0;
}
