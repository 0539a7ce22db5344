// K1/K6 (large-tile form): implicit-GEMM convolution / linear layer on 256 x {320,256} x 64 tiles for the compute-bound
// shapes of the UNet / BrushNet / VAE (same contract as vv_conv_gemm; dispatched from vv_gemm.hip, see include/vvhip.h).
//
// Why a second tiling: the 128 x 160 kernel (vv_gemm.hip) moves 36 KB of L2->LDS bytes per 5.2 MFLOP k tile and the LDS-DMA
// stream tops out at ~27 B/clk/CU (tools/fill_rate.hip), which caps it near half of the MFMA peak however many blocks share a
// CU.  A 256 x 320 tile moves 72 KB per 21 MFLOP: twice the reuse per byte.  One 512-thread block (8 waves, 2 per SIMD, <= 256
// VGPRs) owns a CU, so nothing overlaps between blocks any more: the k loop is software pipelined inside the block instead --
// two LDS stages in DISTINCT __shared__ arrays (so hipcc does not order the ds_reads of stage k behind the LDS-DMA of stage
// k+1), the DMA of tile k+1 issued before the MFMAs of tile k, one vmcnt(0)+barrier per k tile (cdna_hip_programming.md 5,
// "minimum 2-phase" form of the 256^2 template).
//
// Waves: 2 (M) x 4 (N); a wave owns 128 rows x NT*16 columns = MT(8) x NT accumulator tiles of v_mfma_f32_16x16x32 with
// swapped operands (lane owns 4 consecutive output channels of one row -> shared 16-byte epilogue, vv_gemm_epilogue.h).
// LDS image: [row][64 k] h16, 128-byte rows, 16-byte chunk c of row r stored at chunk c ^ (r & 7); filled by
// global_load_lds_dwordx4 with the swizzle applied on the per-lane SOURCE address (the LDS side of the DMA is lane-linear).
//   LIN  : plain [M][K] h16 matrix (linear layers, 1x1 stride-1 convs)
//   CONV : im2col gather, <= 9 taps, stride 1/2, zero padding, two-source channel concat, no fused resize; per row only the
//          pixel index of tap (0,0) and a tap-validity bit mask are kept (out-of-image taps read a zero page)
#include "vv_common.h"
#include "vv_gemm_epilogue.h"

namespace {

enum { G256_LIN = 0, G256_CONV = 1 };

__device__ __attribute__((aligned(64))) const unsigned int g256_zero_page[16] = {0};

__device__ __forceinline__ void glds16(const void* gptr, void* lds_wave_base) {
    typedef const void __attribute__((address_space(1))) * gp_t;
    typedef void __attribute__((address_space(3))) * lp_t;
    __builtin_amdgcn_global_load_lds((gp_t)gptr, (lp_t)lds_wave_base, 16, 0, 0);
}

template <typename T, int NT, int MODE>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(const vv_conv_params p, const int M, const int tilesM, const int tilesN) {
    constexpr int BM = 256, BN = 4 * NT * 16, MT = 8;
    constexpr int AP = BM / 64;        // 1 KB pieces (8 rows x 128 B) of the A tile per wave: 4
    constexpr int BP = BN / 64;        // ... of the B tile per wave: 5 (BN = 320) or 4 (BN = 256)
    __shared__ __attribute__((aligned(1024))) unsigned char sA0[BM * 128];
    __shared__ __attribute__((aligned(1024))) unsigned char sA1[BM * 128];
    __shared__ __attribute__((aligned(1024))) unsigned char sB0[BN * 128];
    __shared__ __attribute__((aligned(1024))) unsigned char sB1[BN * 128];

    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int lr = lane & 15, lq = lane >> 4;

    // XCD-aware bijective remap: blocks b and b+8 share an XCD -> every XCD walks a contiguous range of tiles (column tiles of
    // one row panel fastest), so a row panel of A and the weight matrix stay in ONE L2
    const int nblk = tilesM * tilesN;
    int bid = blockIdx.x;
    {
        const int xcd = bid & 7, idx = bid >> 3, q = nblk >> 3, r = nblk & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_n = bid % tilesN, tile_m = bid / tilesN;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- per-thread DMA state.  Piece i of this wave = tile rows (wave + 8 i) * 8 .. + 7; lane -> row (lane >> 3), physical
    // chunk (lane & 7) which holds logical chunk (lane & 7) ^ (row & 7) = (lane & 7) ^ (lane >> 3)
    const int prow = lane >> 3;
    const int lchunk = (lane & 7) ^ prow;
    const int HWo = p.Hout * p.Wout;
    const int Cin = p.C0 + p.C1;
    const int KW = p.ksize_w > 0 ? p.ksize_w : p.ksize;
    const unsigned char* aptr[MODE == G256_LIN ? AP : 1];
    int pix9[MODE == G256_CONV ? AP : 1];
    unsigned okm[MODE == G256_CONV ? (AP + 1) / 2 : 1];      // bit (i & 1) * 16 + tap: tap of piece-row i lies inside the image
    if constexpr (MODE == G256_LIN) {
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            int m = m0 + (wave + 8 * i) * 8 + prow;
            m = m < M ? m : M - 1;                            // rows past M are clamped (never stored by the epilogue)
            aptr[i] = (const unsigned char*)p.in0 + ((int64_t)m * p.C0 + lchunk * 8) * 2;
        }
    } else {
#pragma unroll
        for (int i = 0; i < (AP + 1) / 2; ++i) okm[i] = 0u;
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const int m = m0 + (wave + 8 * i) * 8 + prow;
            const bool v = m < M;
            const int mm = v ? m : 0;
            const int f = mm / HWo, rem = mm - f * HWo;
            const int y = rem / p.Wout, x = rem - y * p.Wout;
            const int yb = y * p.stride - p.pad_t, xb = x * p.stride - p.pad_l;
            pix9[i] = (f * p.Hin + yb) * p.Win + xb;
            unsigned bits = 0u;
            for (int tap = 0; tap < p.ksize * KW; ++tap) {
                const int yv = yb + tap / KW, xv = xb + tap % KW;
                if (v && yv >= 0 && yv < p.Hin && xv >= 0 && xv < p.Win) bits |= 1u << tap;
            }
            okm[i >> 1] |= bits << ((i & 1) * 16);
        }
    }
    const unsigned char* wptr = (const unsigned char*)p.weight + ((int64_t)(n0 + wave * 8 + prow) * p.Kpad + lchunk * 8) * 2;
    const int64_t wstep = (int64_t)64 * p.Kpad * 2;           // 8 pieces = 64 weight rows further

    auto dma_tile = [&](const int kt, unsigned char* bufA, unsigned char* bufB) {
        if constexpr (MODE == G256_LIN) {
#pragma unroll
            for (int i = 0; i < AP; ++i) glds16(aptr[i] + kt * 128, bufA + (wave + 8 * i) * 1024);
        } else {
            const int k0 = kt * 64;
            const int tap = k0 / Cin;
            int cc = k0 - tap * Cin;
            const int ky = tap / KW, kx = tap - ky * KW;
            const unsigned char* src = (const unsigned char*)p.in0;
            int Cs = p.C0;
            if (cc >= p.C0) { src = (const unsigned char*)p.in1; cc -= p.C0; Cs = p.C1; }
            const int dpix = ky * p.Win + kx;
            const int csrc = cc + lchunk * 8;
#pragma unroll
            for (int i = 0; i < AP; ++i) {
                const bool ok = (okm[i >> 1] >> ((i & 1) * 16 + tap)) & 1u;
                const void* g = ok ? (const void*)(src + ((int64_t)(pix9[i] + dpix) * Cs + csrc) * 2) : (const void*)g256_zero_page;
                glds16(g, bufA + (wave + 8 * i) * 1024);
            }
        }
        const unsigned char* w = wptr + kt * 128;
#pragma unroll
        for (int i = 0; i < BP; ++i) glds16(w + i * wstep, bufB + (wave + 8 * i) * 1024);
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](const unsigned char* cA, const unsigned char* cB) {
        const unsigned char* a = cA + (wr * 128 + lr) * 128;
        const unsigned char* b = cB + (wc * NT * 16 + lr) * 128;
        const int sw = lr & 7;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int off = ((s * 4 + lq) ^ sw) << 4;
            uint4 bf[NT], af[MT];
#pragma unroll
            for (int j = 0; j < NT; ++j) bf[j] = *(const uint4*)(b + j * 2048 + off);
#pragma unroll
            for (int i = 0; i < MT; ++i) af[i] = *(const uint4*)(a + i * 2048 + off);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = T::mfma(bf[j], af[i], acc[i][j]);
        }
    };

    const int nk = p.Kpad / 64;
    dma_tile(0, sA0, sB0);
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {
        if (kt + 1 < nk) dma_tile(kt + 1, sA1, sB1);
        compute(sA0, sB0);
        __syncthreads();          // vmcnt(0): tile kt+1 has landed; barrier: everybody is done reading stage 0
        if (kt + 1 < nk) {
            if (kt + 2 < nk) dma_tile(kt + 2, sA0, sB0);
            compute(sA1, sB1);
            __syncthreads();
        }
    }

    auto row_m = [&](int row, bool& ok) -> int { ok = m0 + row < M; return m0 + row; };
    // LEAN: one block per CU, nothing else covers the epilogue; the operand stages are dead (the k loop ended with a barrier): a wave-private strip tile in sB0 / sB1
    gemm_epilogue<T, MT, NT, true, true>(p, acc, wr * 128, n0 + wc * NT * 16, lr, lq, HWo, row_m, nullptr, (float*)(wr ? sB1 : sB0) + wc * (16 * (NT * 16 + 4)));
}


// ---------------------------------------------------------------------------------------------------------------------------
// 8-phase form (256 x 256 x 64): the two waves of every SIMD run one barrier apart, so that while waves 0-3 issue their 16
// MFMAs of a phase, waves 4-7 read the operands of theirs from LDS and issue the LDS-DMA of a later half tile -- matrix beside
// memory on every SIMD in every barrier interval (cdna_hip_programming.md 5, "The 256^2 8-phase template").
//   * a k tile = 4 half tiles of 16 KB: A0 / A1 = the first / second 64 rows of each wave-row group, B0 / B1 = the first /
//     second 32 columns of each wave-column group.  A wave's 128 x 64 output is four 64 x 32 quadrants, one per phase, in the
//     order (A0,B0) (A0,B1) (A1,B1) (A1,B0): phase 0 reads 12 operand fragments from LDS, phase 1 four, phase 2 eight, phase 3 none.
//   * half tiles are DMA'd in the order they are first read: S = A0(0) B0(0) B1(0) A1(0) A0(1) ...; phase g issues S[g+5] and then
//     waits until all but the 3 newest have landed (vmcnt(6): 2 DMA instructions per half tile and wave), so S[g+1], S[g+2] -- what
//     phase g+1 reads -- are retired one barrier before they are read, and 3 half tiles stay in flight across the barriers.
//   * 8 LDS slots (2 k tiles x 4 half tiles) in distinct __shared__ arrays, k loop unrolled by two so every slot is static.
// LDS-DMA issued from inline asm: hipcc does not see the LDS write, so it inserts no conservative vmcnt(0) before the ds_reads at
// the loop head (it cannot count DMA instructions across the back edge); every wait is the hand-placed counted one.
// M0 = wave-uniform LDS destination, written in the same statement that reads it (cdna_hip_programming.md 5.7).
__device__ __forceinline__ void glds16_asm(const void* gptr, void* lds_wave_base) {
    typedef void __attribute__((address_space(3))) * lp_t;
    const unsigned dst = (unsigned)(size_t)(lp_t)lds_wave_base;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gptr), "s"(dst) : "memory");
}

template <typename T, int MODE, bool ASMDMA>
__global__ __launch_bounds__(512, 2) void gemm256p_kernel(const vv_conv_params p, const int M, const int tilesM, const int tilesN) {
    constexpr int MT = 8, NT = 4;
    __shared__ __attribute__((aligned(1024))) unsigned char sl0[16384], sl1[16384], sl2[16384], sl3[16384];   // k tile parity 0: A0 B0 B1 A1
    __shared__ __attribute__((aligned(1024))) unsigned char sl4[16384], sl5[16384], sl6[16384], sl7[16384];   // parity 1

    auto dma16 = [](const void* g, void* l) { if constexpr (ASMDMA) glds16_asm(g, l); else glds16(g, l); };
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int lr = lane & 15, lq = lane >> 4;
    const int nblk = tilesM * tilesN;
    int bid = blockIdx.x;
    {
        const int xcd = bid & 7, idx = bid >> 3, q = nblk >> 3, r = nblk & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_n = bid % tilesN, tile_m = bid / tilesN;
    const int m0 = tile_m * 256, n0 = tile_n * 256;

    // ---- DMA state.  A half tile = 128 slot rows = 16 pieces of 8 rows; this wave fills pieces wave and wave + 8.
    // slot row sr of A-half h <-> tile row (sr >> 6) * 128 + h * 64 + (sr & 63); of B-half h <-> tile column (sr >> 5) * 64 + h * 32 + (sr & 31)
    const int prow = lane >> 3;
    const int lchunk = (lane & 7) ^ prow;
    const int HWo = p.Hout * p.Wout;
    const int Cin = p.C0 + p.C1;
    const int KW = p.ksize_w > 0 ? p.ksize_w : p.ksize;
    const unsigned char* aptr[MODE == G256_LIN ? 4 : 1];       // [half * 2 + piece]
    int pix9[MODE == G256_CONV ? 4 : 1];
    unsigned okm[MODE == G256_CONV ? 2 : 1];
    if constexpr (MODE == G256_CONV) { okm[0] = 0u; okm[1] = 0u; }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int sr = (wave + 8 * i) * 8 + prow;
            const int m = m0 + (sr >> 6) * 128 + h * 64 + (sr & 63);
            if constexpr (MODE == G256_LIN) {
                const int mc = m < M ? m : M - 1;
                aptr[h * 2 + i] = (const unsigned char*)p.in0 + ((int64_t)mc * p.C0 + lchunk * 8) * 2;
            } else {
                const bool v = m < M;
                const int mm = v ? m : 0;
                const int f = mm / HWo, rem = mm - f * HWo;
                const int y = rem / p.Wout, x = rem - y * p.Wout;
                const int yb = y * p.stride - p.pad_t, xb = x * p.stride - p.pad_l;
                pix9[h * 2 + i] = (f * p.Hin + yb) * p.Win + xb;
                unsigned bits = 0u;
                for (int tap = 0; tap < p.ksize * KW; ++tap) {
                    const int yv = yb + tap / KW, xv = xb + tap % KW;
                    if (v && yv >= 0 && yv < p.Hin && xv >= 0 && xv < p.Win) bits |= 1u << tap;
                }
                okm[h] |= bits << (i * 16);
            }
        }
    const unsigned char* wptr[4];                              // [half * 2 + piece]
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int sr = (wave + 8 * i) * 8 + prow;
            const int n = n0 + (sr >> 5) * 64 + h * 32 + (sr & 31);
            wptr[h * 2 + i] = (const unsigned char*)p.weight + ((int64_t)n * p.Kpad + lchunk * 8) * 2;
        }

    auto dma_a = [&](const int kt, const int h, unsigned char* slot) {
        if constexpr (MODE == G256_LIN) {
            dma16(aptr[h * 2] + kt * 128, slot + wave * 1024);
            dma16(aptr[h * 2 + 1] + kt * 128, slot + (wave + 8) * 1024);
        } else {
            const int k0 = kt * 64;
            const int tap = k0 / Cin;
            int cc = k0 - tap * Cin;
            const int ky = tap / KW, kx = tap - ky * KW;
            const unsigned char* src = (const unsigned char*)p.in0;
            int Cs = p.C0;
            if (cc >= p.C0) { src = (const unsigned char*)p.in1; cc -= p.C0; Cs = p.C1; }
            const int dpix = ky * p.Win + kx;
            const int csrc = cc + lchunk * 8;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const bool ok = (okm[h] >> (i * 16 + tap)) & 1u;
                const void* g = ok ? (const void*)(src + ((int64_t)(pix9[h * 2 + i] + dpix) * Cs + csrc) * 2) : (const void*)g256_zero_page;
                dma16(g, slot + (wave + 8 * i) * 1024);
            }
        }
    };
    auto dma_b = [&](const int kt, const int h, unsigned char* slot) {
        dma16(wptr[h * 2] + kt * 128, slot + wave * 1024);
        dma16(wptr[h * 2 + 1] + kt * 128, slot + (wave + 8) * 1024);
    };
    const int nk = p.Kpad / 64;
    // issue half tile number i of the sequence S (i = 4 * kt + {0: A0, 1: B0, 2: B1, 3: A1}) into its slot; parity = kt & 1
    auto issue = [&](const int kt, const int which, unsigned char* slot) {
        if (kt < nk) {
            if (which == 0) dma_a(kt, 0, slot);
            else if (which == 1) dma_b(kt, 0, slot);
            else if (which == 2) dma_b(kt, 1, slot);
            else dma_a(kt, 1, slot);
        }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    uint4 fa[4][2], fb0[2][2], fb1[2][2];       // [tile][k step]
    const int sw = lr & 7;
    const int off0 = ((0 * 4 + lq) ^ sw) << 4, off1 = ((1 * 4 + lq) ^ sw) << 4;
    // slot row of this wave's A sub-tile i (0..3): wr * 64 + i * 16 + lr ; of B sub-tile j (0..1): wc * 32 + j * 16 + lr
    const int arow = (wr * 64 + lr) * 128, brow = (wc * 32 + lr) * 128;
    auto read_a = [&](const unsigned char* slot) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { fa[i][0] = *(const uint4*)(slot + arow + i * 2048 + off0); fa[i][1] = *(const uint4*)(slot + arow + i * 2048 + off1); }
    };
    auto read_b = [&](const unsigned char* slot, uint4 (&fb)[2][2]) {
#pragma unroll
        for (int j = 0; j < 2; ++j) { fb[j][0] = *(const uint4*)(slot + brow + j * 2048 + off0); fb[j][1] = *(const uint4*)(slot + brow + j * 2048 + off1); }
    };
    auto mma = [&](const int ah, const int bh, uint4 (&fb)[2][2]) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[ah * 4 + i][bh * 2 + j] = T::mfma(fb[j][s], fa[i][s], acc[ah * 4 + i][bh * 2 + j]);
        __builtin_amdgcn_s_setprio(0);
    };
    auto bar = [&]() { __builtin_amdgcn_s_barrier(); };
    // wait until all but the 3 newest half tiles (6 DMA instructions of this wave) have landed; near the end of the k loop,
    // where nothing new is issued any more, everything
    auto wait_dma = [&](const bool tail) {
        if (tail) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    };
    auto wait_lds = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };

    // one k tile = 4 phases.  (cA0, cB0, cB1, cA1) = this tile's slots, (nA0 ..) = the other parity's.
    auto ktile = [&](const int kt, unsigned char* cA0, unsigned char* cB0, unsigned char* cB1, unsigned char* cA1,
                     unsigned char* nA0, unsigned char* nB0, unsigned char* nB1, unsigned char* nA1) {
        const bool tail = kt + 2 >= nk;          // S[g+5] runs out during the last two k tiles: drain instead of counting
        // phase 0: quadrant (A0, B0); issue B0(kt+1)
        read_a(cA0); read_b(cB0, fb0);
        issue(kt + 1, 1, nB0); wait_dma(tail);
        bar(); wait_lds(); mma(0, 0, fb0); bar();
        // phase 1: quadrant (A0, B1); issue B1(kt+1)
        read_b(cB1, fb1);
        issue(kt + 1, 2, nB1); wait_dma(tail);
        bar(); wait_lds(); mma(0, 1, fb1); bar();
        // phase 2: quadrant (A1, B1); issue A1(kt+1)
        read_a(cA1);
        issue(kt + 1, 3, nA1); wait_dma(tail);
        bar(); wait_lds(); mma(1, 1, fb1); bar();
        // phase 3: quadrant (A1, B0); issue A0(kt+2) into THIS parity's A0 slot (last read in phase 0)
        issue(kt + 2, 0, cA0); wait_dma(tail);
        bar(); mma(1, 0, fb0); bar();
    };

    // prologue: S[0..4] = the whole k tile 0 + A0(1); retire S[0], S[1]
    issue(0, 0, sl0); issue(0, 1, sl1); issue(0, 2, sl2); issue(0, 3, sl3); issue(1, 0, sl4);
    wait_dma(nk < 2);
    bar();
    if (wr == 1) bar();                          // stagger: waves 4-7 run one barrier behind waves 0-3
    for (int kt = 0; kt < nk; kt += 2) {
        ktile(kt, sl0, sl1, sl2, sl3, sl4, sl5, sl6, sl7);
        if (kt + 1 < nk) ktile(kt + 1, sl4, sl5, sl6, sl7, sl0, sl1, sl2, sl3);
    }
    if (wr == 0) bar();                          // re-align the two groups

    auto row_m = [&](int row, bool& ok) -> int { ok = m0 + row < M; return m0 + row; };
    gemm_epilogue<T, MT, NT, true>(p, acc, wr * 128, n0 + wc * 64, lr, lq, HWo, row_m);
}

template <typename T, int NT, int MODE>
int launch256(const vv_conv_params& p, int M, hipStream_t st) {
    constexpr int BN = 4 * NT * 16;
    const int tilesM = (M + 255) / 256, tilesN = p.Npad / BN;
    hipLaunchKernelGGL((gemm256_kernel<T, NT, MODE>), dim3(tilesM * tilesN), dim3(512), 0, st, p, M, tilesM, tilesN);
    VV_CHECK_LAUNCH("vv_conv_gemm(256-row tile)");
    return VV_OK;
}

template <typename T, int MODE, bool ASMDMA>
int launch256p(const vv_conv_params& p, int M, hipStream_t st) {
    const int tilesM = (M + 255) / 256, tilesN = p.Npad / 256;
    hipLaunchKernelGGL((gemm256p_kernel<T, MODE, ASMDMA>), dim3(tilesM * tilesN), dim3(512), 0, st, p, M, tilesM, tilesN);
    VV_CHECK_LAUNCH("vv_conv_gemm(256-row tile, 8-phase)");
    return VV_OK;
}

template <typename T>
int launch256_t(const vv_conv_params& p, int M, bool lin, int form, hipStream_t st) {
    if (form == 2) return lin ? launch256p<T, G256_LIN, false>(p, M, st) : launch256p<T, G256_CONV, false>(p, M, st);
    if (form == 3) return lin ? launch256p<T, G256_LIN, true>(p, M, st) : launch256p<T, G256_CONV, true>(p, M, st);
    const bool n5 = p.Npad % 320 == 0 && p.epilogue != VV_EPI_GEGLU;     // GEGLU pairs value/gate tiles: needs an even NT
    if (n5) return lin ? launch256<T, 5, G256_LIN>(p, M, st) : launch256<T, 5, G256_CONV>(p, M, st);
    return lin ? launch256<T, 4, G256_LIN>(p, M, st) : launch256<T, 4, G256_CONV>(p, M, st);
}

}  // namespace

// Eligibility + launch.  Returns VV_OK / an error after launching, or -1000 when the shape is not eligible (caller falls back
// to the 128-row kernels).  `force`: 0 = only where the heuristic expects a win; 1 = the 2-phase kernel whenever the shape is
// ELIGIBLE; 2 = the 8-phase kernel whenever eligible (needs Npad % 256 == 0).
// Build split (build.sh): this source is compiled twice, -DVV_DT_ONLY=0 holds the BF16 instantiations behind vv_gemm256_launch_bf16, -DVV_DT_ONLY=1 the
// F16 ones plus the entry point below (two translation units of ~50 s instead of one of ~100 s: the longest pole of a clean build).  Without the
// macro everything lives in one unit.
#if defined(VV_DT_ONLY) && VV_DT_ONLY == 0
extern "C" int vv_gemm256_launch_bf16(const vv_conv_params* pp, int M, int lin, int form, void* stream) {
    return launch256_t<BF16>(*pp, M, lin != 0, form, (hipStream_t)stream);
}
#else
#if defined(VV_DT_ONLY)
extern "C" int vv_gemm256_launch_bf16(const vv_conv_params* pp, int M, int lin, int form, void* stream);
#endif
extern "C" int vv_gemm256_try(const vv_conv_params* pp, int dtype, int force, void* stream) {
    const vv_conv_params& p = *pp;
    const int kw = p.ksize_w > 0 ? p.ksize_w : p.ksize;
    if (p.in_dtype == VV_F32 || p.Kpad != p.K || (p.C0 & 63) || (p.C1 & 63) || p.sc_oh > 0) return -1000;      // (the scattered store lives in the 128-row kernels)
    if (p.Hv != p.Hin || p.Wv != p.Win || p.ksize * kw > 9) return -1000;
    if (!(p.Npad % 320 == 0 && p.epilogue != VV_EPI_GEGLU) && p.Npad % 256 != 0) return -1000;
    if (force >= 2 && p.Npad % 256 != 0) return -1000;
    const int64_t M64 = (int64_t)p.F * p.Hout * p.Wout;
    const int M = (int)M64;
    const bool lin = p.ksize == 1 && kw == 1 && p.stride == 1 && p.pad_t == 0 && p.pad_l == 0 && p.C1 == 0 && p.Hout == p.Hin && p.Wout == p.Win;
    int form = force >= 2 ? force : 1;
    if (!force) {
        // One block per CU: nothing overlaps the (fp32 residual) epilogue, and the grid is quantised to whole rounds of 256 blocks.
        // Measured A/B against the 128-row kernels on the shapes of a 720p step (tools/bench_gemm256.py, profiles/r2_gemm256_ab.txt):
        // the 8-phase form wins on long-k / wide-N linears (GEGLU projections with K >= 640: x1.15-1.34, the level-2 QKV and FF
        // output projections: x1.10-1.16; a plain 8192^3 GEMM: x1.45 = 1.2 PFLOP/s), the 2-phase form on 3x3 convs with K >= 5760
        // (x1.04-1.28; its im2col address arithmetic sits badly in the 8-phase load segments); short-K layers stay on the
        // 4-blocks-per-CU kernels.
        const bool n256 = p.Npad % 256 == 0;
        const int BN = (p.Npad % 320 == 0 && p.epilogue != VV_EPI_GEGLU) ? 320 : 256;
        bool win = false;
        // Round 4: re-measured in back-to-back loops of 0.4 s per form (the board is power capped; the round-2 table came from 6-launch loops that
        // ran while the clock was still ramping: profiles/r4_gemm256_steady.txt): the 3x3 convolutions of levels 0 / 1 (M > 65536) are faster on the
        // 128-row tiles whatever their K (x1.06-1.18), level 2 stays here (x1.12-1.34); the level-2 FF output projection (K = 5120, N = 1280) takes
        // the 2-phase 256x320 form (x1.14 over the 128-row tile, x1.09 over the 8-phase form; the level-1 one, K = 2560, gained x1.05 standalone and lost
        // 2 % in the pipeline: stays on the 128-row tile).
        // Round 5 (second session): with the lean epilogue (vv_gemm_epilogue.h: the 256-row kernels no longer pay 40 serialised bias-load / store round
        // trips per wave) the 2-phase form also wins on the level-1 / level-2 linears with K >= 640 the 128-row tiles used to keep -- QKV K 640 N 1920
        // x1.12, FF output K 2560 N 640 x1.10, attention / block output projections K 640 N 640 x1.04 and K 1280 N 1280 x1.05 -- and on the level-1
        // GEGLU projection (K 640 N 5120) against the 8-phase form (x1.04); in the pipeline, two chunks in flight, all five together: 19.24 -> 19.06 s per
        // chunk, bit-identical output (tools/ab_tiles.py, profiles/r5_epilogue_ab.txt).  K = 320 (level 0) stays on the 128-row tiles.
        if (lin) {
            if (p.K >= 5120 && p.Npad % 320 == 0 && p.Npad < 3840 && p.epilogue != VV_EPI_GEGLU) { win = true; form = 1; }
            else if (n256 && ((p.epilogue == VV_EPI_GEGLU && p.K >= 1280) || (p.K >= 1280 && p.Npad >= 3840) || p.K >= 5120)) { win = true; form = 3; }
            else if (p.K >= 640) { win = true; form = 1; }
        } else if (p.ksize == 3 && p.stride == 1 && p.K >= 5760 && M64 <= 65536 && BN == 320) { win = true; form = 1; }      // (the 256x256 form lost on every VAE shape)
        const int64_t tiles = ((M64 + 255) / 256) * (p.Npad / (form == 3 ? 256 : BN));
        if (!win || tiles < 400) return -1000;
    }
    hipStream_t st = (hipStream_t)stream;
#if defined(VV_DT_ONLY)
    return dtype == VV_BF16 ? vv_gemm256_launch_bf16(&p, M, lin ? 1 : 0, form, stream) : launch256_t<F16>(p, M, lin, form, st);
#else
    return dtype == VV_BF16 ? launch256_t<BF16>(p, M, lin, form, st) : launch256_t<F16>(p, M, lin, form, st);
#endif
}
#endif
