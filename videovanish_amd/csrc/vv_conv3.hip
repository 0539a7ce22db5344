// K1b (EXPERIMENT, opt-in with VV_CONV3_HALO256=1): 3x3 stride-1 convolution on MFMA, 256-pixel halo tile, 8 waves, software pipelined.
//
// vv_conv_gemm's im2col loader sits on the L2->LDS fill ceiling of its 128x160 tile (profiles/r1_gemm_ab.txt: 36 KB per k tile
// at ~15 TB/s chip-wide = ~1050 TF/s).  This kernel cuts the L2 bytes per FLOP by ~3x:
//   * M tile = 16 x 16 pixel patch of one frame; per 64-channel chunk its 18 x 18 halo (41 KB) is fetched ONCE and the nine
//     taps read their A operands out of it (the halo pixel of (row, col, tap) is a constant offset);
//   * the weight tile (160 x 64, 20 KB) of every (chunk, tap) is shared by 256 output rows instead of 128;
//   * everything is LDS-DMA (global_load_lds_dwordx4) into DISTINCT static buffers -- a ring of three weight tiles (fetched two
//     k tiles ahead) and two halo buffers (the next chunk's halo arrives in six slices spread over taps 0..5) -- so the
//     fetches of the next tiles fly during the MFMAs of the current one, with ONE barrier per k tile and explicit
//     s_waitcnt vmcnt(n) that only waits for the tile about to be consumed.
// 512 threads = 8 waves as 4 (patch rows) x 2 (80 channels); each wave 64 x 80 outputs = 4 x 5 mfma_16x16x32 tiles, swapped
// operands (lane owns 4 consecutive channels of one pixel), epilogue shared with vv_conv_gemm (vv_gemm_epilogue.h).
#include <stdlib.h>
#include <type_traits>
#include "vv_common.h"
#include "vv_gemm_epilogue.h"

namespace {

constexpr int HPX = 328;                 // 18 x 18 = 324 halo pixels, padded to whole 1 KB DMA blocks (328 * 8 slots = 41 waves)
constexpr int HSLOTS = HPX * 8;          // 16-byte slots per halo buffer
constexpr int HPASS = 6;                 // ceil(HSLOTS / 512)

__device__ __forceinline__ void glds16(const void* gptr, void* lds_wave_base) {
    typedef const void __attribute__((address_space(1))) * gp_t;
    typedef void __attribute__((address_space(3))) * lp_t;
    __builtin_amdgcn_global_load_lds((gp_t)gptr, (lp_t)lds_wave_base, 16, 0, 0);
}

template <int N> __device__ __forceinline__ void wait_vm() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else static_assert(N == 0, "unsupported count");
}

// PIPE: the MFMA operands are double buffered in registers at k-step granularity -- the ds_reads of step n+1 are issued before the
// MFMAs of step n, and the second half of tile k-1 runs AFTER the barrier of tile k (its operands were read before it), so the LDS
// burst that follows every barrier (all 8 waves read at once) overlaps matrix work instead of stalling the pipe.
template <typename T, int NT, bool PIPE>
__global__ __launch_bounds__(512, 1) void conv3_halo_kernel(const vv_conv_params p, const int tilesM, const int tilesN) {
    constexpr int MT = 4, BN = 2 * NT * 16;
    constexpr int BSLOTS = BN * 8;                       // 16-byte slots per weight tile (1280 for BN = 160, 1024 for 128)
    constexpr int BPASS = (BSLOTS + 511) / 512;          // 3 or 2 DMA instructions per thread and tile (uniform, see dma_b)
    __shared__ __attribute__((aligned(1024))) unsigned char hA0[HSLOTS * 16];
    __shared__ __attribute__((aligned(1024))) unsigned char hA1[HSLOTS * 16];
    __shared__ __attribute__((aligned(1024))) unsigned char sB0[BSLOTS * 16];
    __shared__ __attribute__((aligned(1024))) unsigned char sB1[BSLOTS * 16];
    __shared__ __attribute__((aligned(1024))) unsigned char sB2[BSLOTS * 16];

    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int lr = lane & 15, lq = lane >> 4;

    // XCD-aware block remap (blocks b and b+8 share an XCD): each XCD gets a contiguous range, column tiles of a patch adjacent
    const int nblk = tilesM * tilesN;
    int bid = blockIdx.x;
    {
        const int xcd = bid & 7, idx = bid >> 3, q = nblk >> 3, r = nblk & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_n = bid % tilesN, tile_m = bid / tilesN;
    const int n0 = tile_n * BN;
    const int PW = (p.Win + 15) >> 4, PH = (p.Hin + 15) >> 4;
    const int hf = tile_m / (PH * PW);
    const int pr = tile_m - hf * (PH * PW);
    const int hy0 = (pr / PW) * 16, hx0 = (pr % PW) * 16;
    const int Cin = p.C0 + p.C1;
    auto row_m = [&](int row, bool& ok) -> int {
        const int y = hy0 + (row >> 4), x = hx0 + (row & 15);
        ok = y < p.Hin && x < p.Win;
        return (hf * p.Hin + y) * p.Win + x;
    };

    // ---- halo slots: LDS slot (pass i, thread t) = halo pixel hp (0..327), physical chunk hc; holds logical chunk hc ^ (hp & 7).
    // Out-of-image pixels are zero-filled once in both buffers and their lanes masked in every DMA.  Pass 5 only has one real
    // wave (slots 2560..2623); the other waves repeat their pass-4 slot so that every wave issues the same number of DMAs.
    int hpix[HPASS]; bool hok[HPASS]; int hslot[HPASS];
#pragma unroll
    for (int i = 0; i < HPASS; ++i) {
        int idx = i * 512 + t, wbase_slot = i * 512 + wave * 64;
        if (idx < HSLOTS) { *(uint4*)(hA0 + idx * 16) = make_uint4(0, 0, 0, 0); *(uint4*)(hA1 + idx * 16) = make_uint4(0, 0, 0, 0); }
        if (wbase_slot >= HSLOTS) { idx -= 512; wbase_slot -= 512; }          // duplicate of the previous pass (wave-uniform)
        const int hp = idx >> 3;
        const int y = hy0 - 1 + hp / 18, x = hx0 - 1 + hp % 18;
        hok[i] = hp < 324 && y >= 0 && y < p.Hin && x >= 0 && x < p.Win;
        hpix[i] = (hf * p.Hin + y) * p.Win + x;
        hslot[i] = wbase_slot * 16;                              // wave base of the slot block (lane-linear inside)
    }
    const int hsw = ((t & 7) ^ ((t >> 3) & 7)) << 3;             // logical channel offset of this thread's slots
    // ---- weight slots: slot (pass i, thread t) = tile row (i*512 + t) >> 3, physical chunk t & 7 holding logical (t & 7) ^ (row & 7).
    // BN = 160: pass 2 covers rows 128..159 with waves 0..3; waves 4..7 repeat their pass-1 slot (same data, same place).
    const unsigned short* wbase = (const unsigned short*)p.weight;
    int brow[BPASS], bslot[BPASS];
#pragma unroll
    for (int i = 0; i < BPASS; ++i) {
        int idx = i * 512 + t, wbase_slot = i * 512 + wave * 64;
        if (wbase_slot >= BSLOTS) { idx -= 512; wbase_slot -= 512; }
        brow[i] = idx >> 3;
        bslot[i] = wbase_slot * 16;
    }
    const int bsw = ((t & 7) ^ ((t >> 3) & 7)) << 3;
    __syncthreads();      // zero fill complete before the first DMA lands

    auto dma_halo_pass = [&](int c, int i, unsigned char* buf) {
        int cc = c * 64;
        const unsigned char* src = (const unsigned char*)p.in0;
        int Cs = p.C0;
        if (cc >= p.C0) { src = (const unsigned char*)p.in1; cc -= p.C0; Cs = p.C1; }
        if (hok[i]) glds16(src + ((int64_t)hpix[i] * Cs + cc + hsw) * 2, buf + hslot[i]);
    };
    auto dma_b = [&](int kofs, unsigned char* buf) {
#pragma unroll
        for (int i = 0; i < BPASS; ++i)
            glds16(wbase + (int64_t)(n0 + brow[i]) * p.Kpad + kofs + bsw, buf + bslot[i]);
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto mma = [&](const unsigned char* hA, const unsigned char* sB, int tapofs) {
        // keep the (cheap) tap-dependent address arithmetic inside the k tile: without this the 18 unrolled taps' operand
        // addresses are all hoisted out of the chunk loop and spill
        asm volatile("" : "+v"(tapofs));
        const unsigned char* b = sB + (wc * NT * 16) * 128;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            uint4 af[MT], bf[NT];
            const int ch = s * 4 + lq;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int hp = (wr * MT + i) * 18 + lr + tapofs;
                af[i] = *(const uint4*)(hA + hp * 128 + ((ch ^ (hp & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int row = j * 16 + lr;
                bf[j] = *(const uint4*)(b + row * 128 + ((ch ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = T::mfma(bf[j], af[i], acc[i][j]);
        }
    };

    const int nchunks = Cin >> 6;
    const int nkt = nchunks * 9;
    auto kofs_of = [&](int kk) { const int c = kk / 9, tap = kk - c * 9; return tap * Cin + c * 64; };
    // ---- prologue: halo of chunk 0 (all six slices), weight tiles 0 and 1
#pragma unroll
    for (int i = 0; i < HPASS; ++i) dma_halo_pass(0, i, hA0);
    dma_b(kofs_of(0), sB0);
    if (nkt > 1) dma_b(kofs_of(1), sB1);
    // ---- one channel chunk = nine k tiles; PAR = chunk parity selects the halo buffer; weight ring index = tap % 3
    struct Frag { uint4 a[MT]; uint4 b[NT]; };
    auto read_step = [&](Frag& r, const unsigned char* hA, const unsigned char* sB, int tapofs, const int s) {
        asm volatile("" : "+v"(tapofs));
        const unsigned char* b = sB + (wc * NT * 16) * 128;
        const int ch = s * 4 + lq;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int hp = (wr * MT + i) * 18 + lr + tapofs;
            r.a[i] = *(const uint4*)(hA + hp * 128 + ((ch ^ (hp & 7)) << 4));
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int row = j * 16 + lr;
            r.b[j] = *(const uint4*)(b + row * 128 + ((ch ^ (row & 7)) << 4));
        }
    };
    auto mma_step = [&](const Frag& r) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = T::mfma(r.b[j], r.a[i], acc[i][j]);
    };
    Frag f0, f1;
    auto chunk = [&](const int c, auto par_tag) {
        constexpr int PAR = decltype(par_tag)::value;
        unsigned char* hcur = PAR ? hA1 : hA0;
        unsigned char* hnext = PAR ? hA0 : hA1;
        const bool more_chunks = c + 1 < nchunks;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int kk = c * 9 + tap;
            unsigned char* bcur = tap % 3 == 0 ? sB0 : tap % 3 == 1 ? sB1 : sB2;
            unsigned char* bnext2 = (tap + 2) % 3 == 0 ? sB0 : (tap + 2) % 3 == 1 ? sB1 : sB2;
            // in-order VM queue of this wave, oldest first: ... B(kk) [halo slice] B(kk+1) [halo slice].  Everything up to B(kk) must
            // have landed; leaving the BPASS newest in flight never under-waits (a halo slice among them only makes the wait
            // cover one DMA of B(kk+1) as well).  A wave whose halo lanes are all out of the image skips that slice entirely.
            if (kk + 1 < nkt) wait_vm<BPASS>(); else wait_vm<0>();
            if (PIPE) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // my operand reads of tile kk-1 are done before its buffer is refilled
            __builtin_amdgcn_s_barrier();       // (not __syncthreads(): its fence would drain vmcnt(0))
            if (kk + 2 < nkt) dma_b(kofs_of(kk + 2), bnext2);
            const int tapofs = (tap / 3) * 18 + tap % 3;
            if constexpr (PIPE) {
                read_step(f0, hcur, bcur, tapofs, 0);
                if (kk > 0) mma_step(f1);                 // second k step of tile kk-1 (operands read before the barrier)
                read_step(f1, hcur, bcur, tapofs, 1);
                mma_step(f0);
            } else {
                mma(hcur, bcur, tapofs);
            }
            if (tap < HPASS && more_chunks) dma_halo_pass(c + 1, tap, hnext);
        }
    };
    for (int c = 0; c < nchunks; c += 2) {
        chunk(c, std::integral_constant<int, 0>{});
        if (c + 1 < nchunks) chunk(c + 1, std::integral_constant<int, 1>{});
    }
    if constexpr (PIPE) mma_step(f1);                    // second k step of the last tile
    const int HWo = p.Hin * p.Win;
    gemm_epilogue<T, MT, NT>(p, acc, wr * MT * 16, n0 + wc * NT * 16, lr, lq, HWo, row_m);
}

template <typename T, int NT>
int launch(const vv_conv_params& p, hipStream_t st) {
    const int tilesM = p.F * ((p.Hin + 15) / 16) * ((p.Win + 15) / 16), tilesN = p.Npad / (2 * NT * 16);
    const char* e = getenv("VV_CONV3_PIPE");
    if (e && e[0] == '0') hipLaunchKernelGGL((conv3_halo_kernel<T, NT, false>), dim3(tilesM * tilesN), dim3(512), 0, st, p, tilesM, tilesN);
    else hipLaunchKernelGGL((conv3_halo_kernel<T, NT, true>), dim3(tilesM * tilesN), dim3(512), 0, st, p, tilesM, tilesN);
    VV_CHECK_LAUNCH("vv_conv3_halo");
    return VV_OK;
}

}  // namespace

// Internal entry (called by vv_conv_gemm's dispatcher, not exported in include/vvhip.h): returns VV_OK after launching, or
// a negative value < -1000 when the shape is not eligible (caller falls through to the generic kernels).
extern "C" int vv_conv3_halo_try(const vv_conv_params* pp, int dtype, void* stream) {
    const vv_conv_params& p = *pp;
    // measured slower than the 128-pixel halo tile of vv_conv_gemm at 3 blocks per CU (profiles/r1_gemm_ab.txt, seventh A/B): the
    // eight waves of the single resident block run in lockstep behind the barrier, LDS reads and MFMAs do not overlap -> opt-in
    const char* e = getenv("VV_CONV3_HALO256");
    const bool off = !(e && e[0] == '1');
    const int kw = p.ksize_w > 0 ? p.ksize_w : p.ksize;
    if (off || p.in_dtype == VV_F32 || p.ksize != 3 || kw != 3 || p.stride != 1 || p.pad_t != 1 || p.pad_l != 1 || p.Hv != p.Hin || p.Wv != p.Win ||
        p.Hout != p.Hin || p.Wout != p.Win || p.epilogue == VV_EPI_GEGLU || p.C0 % 64 || p.C1 % 64 || p.Kpad != p.K) return -2000;
    if (p.Npad % 160 != 0 && p.Npad % 128 != 0) return -2000;
    const int64_t cover = (int64_t)((p.Hin + 15) / 16) * 16 * ((p.Win + 15) / 16) * 16;
    if (cover * 10 > (int64_t)p.Hin * p.Win * 11) return -2000;                      // patch grid wastes > 10 %
    if ((int64_t)p.F * p.Hin * p.Win > 0x7fffffff) return -2000;
    hipStream_t st = (hipStream_t)stream;
    if (p.Npad % 160 == 0) return dtype == VV_BF16 ? launch<BF16, 5>(p, st) : launch<F16, 5>(p, st);
    return dtype == VV_BF16 ? launch<BF16, 4>(p, st) : launch<F16, 4>(p, st);
}
