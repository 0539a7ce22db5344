// K1/K6: implicit-GEMM convolution / linear on MFMA (gfx950).  See include/vvhip.h (vv_conv_gemm).
//
// Tile: BM x BN x 64, 256 threads = 4 waves in a WR x WC grid, each wave MT x NT tiles of 16x16 (mfma 16x16x32).
// LDS holds double-buffered XOR-swizzled [row][64] h16 images of the A (activation, im2col-gathered from NHWC) and
// B (weight, [N][K] K-contiguous) tiles; MFMA operands are read with ds_read_b128.
//   FAST path (h16 activations, channel counts multiples of 64): both tiles are filled by LDS-DMA
//     (global_load_lds_dwordx4, per-lane source address = the im2col gather, swizzle applied on the SOURCE address,
//     out-of-image taps read a zero page), no staging VGPRs, no ds_write.  Three buffering variants were measured A/B in
//     one process (VV_GEMM_SPLIT, profiles/r1_gemm_ab.txt): 2 = SINGLE buffer (fill -> barrier -> MFMAs -> barrier; 36 KB
//     of LDS, so 3-4 co-resident blocks per CU hide each other's fills) is the fastest and the default; 0 = one array
//     double buffered (hipcc drains vmcnt(0) before the ds_reads); 1 = double buffered in DISTINCT __shared__ arrays with
//     the k loop unrolled by two (the DMA of tile k+1 really flies during the MFMAs of tile k) -- the slowest of the three.
//   generic / fp32-activation paths: register staged (issue-early, convert, write-late).
// The MFMA is issued with swapped operands (D = W_tile * A_tile^T) so every lane owns 4 CONSECUTIVE output channels
// of one output row: the epilogue (bias, time-embedding vector, residuals, GEGLU, cast) is 16-byte vectorised.
// The block index is remapped so that the column tiles of one row panel run on the same XCD (shared L2).
#include <stdlib.h>
#include <type_traits>
#include "vv_common.h"
#include "vv_gemm_epilogue.h"

extern "C" int vv_gemm256_try(const vv_conv_params* pp, int dtype, int force, void* stream);

// A/B switches of the lab build (-DVV_AB: environment variables read once per process).  The product build takes the measured
// defaults (profiles/r1_gemm_ab.txt) with no getenv in any launch path.
#ifdef VV_AB
#define VV_AB_ENV(name) (getenv(name) != nullptr)
#define VV_AB_INT(name, dflt) (getenv(name) ? atoi(getenv(name)) : (dflt))
extern "C" int vv_conv3_halo_try(const vv_conv_params* pp, int dtype, void* stream);
#else
#define VV_AB_ENV(name) false
#define VV_AB_INT(name, dflt) (dflt)
#endif

namespace {

[[maybe_unused]] constexpr int BK = 64;
enum { MODE_H16 = 0, MODE_F32 = 1, MODE_FAST = 2, MODE_FAST32 = 3, MODE_HALO = 4, MODE_LIN = 5, MODE_FAST9 = 6 };
constexpr int HALO_PX = 184;   // (8+2) x (16+2) = 180 halo pixels of an 8x16 output patch, padded to whole 1 KB DMA blocks

__device__ __attribute__((aligned(64))) const unsigned int g_zero_page[16] = {0};

__device__ __forceinline__ void glds16(const void* gptr, void* lds_wave_base) {
    typedef const void __attribute__((address_space(1))) * gp_t;
    typedef void __attribute__((address_space(3))) * lp_t;
    __builtin_amdgcn_global_load_lds((gp_t)gptr, (lp_t)lds_wave_base, 16, 0, 0);
}

template <typename T, int WR, int WC, int MT, int NT, int MODE, int SPLIT, int BKT, int OCCW = 2, bool GN = false>
__global__ __launch_bounds__(256, OCCW) void conv_gemm_kernel(const vv_conv_params p, const int M, const int tilesM, const int tilesN) {
    constexpr int BM = WR * MT * 16, BN = WC * NT * 16;
    constexpr int CH = BKT / 8;                 // 16-byte chunks per tile row (8 for a 64-wide k tile, 4 for 32)
    constexpr int SH = CH == 8 ? 3 : 2;         // log2(CH)
    constexpr int RP = BKT * 2;                 // LDS row pitch in bytes
    constexpr int RPB = 256 >> SH;              // tile rows filled by one 256-thread pass (32 or 64)
    constexpr int AR = BM / RPB;                // A chunks staged per thread
    constexpr int BCH = (BN + RPB - 1) / RPB;   // B chunks staged per thread
    constexpr int KS = BKT / 32;                // MFMA k steps per tile
    // HALO (3x3, stride 1, h16): the M tile is an 8 x 16 pixel patch of one frame; per 64-channel chunk its 10 x 18 halo is
    // DMA'd ONCE and the 9 taps read their A operands out of it (L2->LDS bytes per 9 k tiles: 23 + 9*20 KB instead of 9*36)
    // LIN (plain linear layer / 1x1 stride-1 conv, one h16 source): rows of A are consecutive, no gather state at all -> the kernel
    // fits 128 VGPRs and a 4th block shares the CU
    constexpr bool AF32 = MODE == MODE_F32, A32 = MODE == MODE_FAST32, HALO = MODE == MODE_HALO, LIN = MODE == MODE_LIN;
    // FAST9 (h16 im2col, at most 9 taps, no fused resize): per row only the pixel index of tap (0,0) and a 9-bit tap-validity mask are
    // kept (two rows per VGPR) instead of coordinates + frame + flags -> the 128x160 kernel fits 128 VGPRs = 4 blocks per CU
    constexpr bool F9 = MODE == MODE_FAST9;
    constexpr bool FAST = MODE == MODE_FAST || A32 || HALO || LIN || F9;
    static_assert(!HALO || (WR * MT == 8 && SPLIT == 2 && BKT == 64), "HALO: 128-row tile, single buffer");
    constexpr int HP = HALO ? (HALO_PX * 8 + 255) / 256 : 1;    // halo DMA passes
    // FAST32: the fp32 A tile is DMA'd as fp32 (256-byte rows, 16 chunks) and rounded to h16 when the operand is read
    constexpr int RPA = A32 ? BKT * 4 : RP;     // A row pitch
    constexpr int SHA = A32 ? 4 : SH;           // log2(A chunks per row)
    constexpr int RPBA = 256 >> SHA;            // A rows filled per pass
    constexpr int ARA = BM / RPBA;              // A chunks staged per thread
    // XOR swizzle of the chunk index that makes the ds_read_b128 operand reads conflict free (tools/lds_bank_model.py)
    auto SWZ = [](int row) { return CH == 8 ? (row & 7) : ((row >> 1) & 3); };
    __shared__ __attribute__((aligned(16))) unsigned char sA0[SPLIT == 0 ? 16 : (HALO ? HALO_PX * 128 : BM * RPA)];
    __shared__ __attribute__((aligned(16))) unsigned char sA1[SPLIT == 1 ? BM * RP : 16];
    __shared__ __attribute__((aligned(16))) unsigned char sB0[SPLIT == 0 ? 16 : BN * RP];
    __shared__ __attribute__((aligned(16))) unsigned char sB1[SPLIT == 1 ? BN * RP : 16];
    __shared__ __attribute__((aligned(16))) unsigned char sAB[SPLIT == 0 ? 2 * (BM + BN) * RP : 16];

    __shared__ __attribute__((aligned(16))) float sBias[(MODE == MODE_HALO || MODE == MODE_FAST32) ? 4 : BN];      // the block's bias columns: the non-LEAN epilogue reads them from here (vv_gemm_epilogue.h)

    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = wave / WC, wc = wave % WC;

    // ---- XCD-aware block remap (bijective): blocks b and b+8 share an XCD -> give each XCD a contiguous range
    const int nblk = tilesM * tilesN;
    int bid = blockIdx.x;
    {
        const int xcd = bid & 7, idx = bid >> 3, q = nblk >> 3, r = nblk & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_n = bid % tilesN, tile_m = bid / tilesN;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    // (not for the fp32-operand loader: with this alone its HBM-bound zero convolutions LOSE 9-14 % -- profiles/r5_epilogue_ab.txt; at 128 x 160 it runs the LEAN + STAGED
    //  form instead, like the halo-tile kernels, which need no staged copy)
    const bool stage_bias = MODE != MODE_HALO && MODE != MODE_FAST32 && (p.N & 3) == 0 && p.epilogue != VV_EPI_GEGLU;       // (made visible by the k loop's barriers)
    if (stage_bias && t < BN / 4) {
        const int n = n0 + 4 * t;
        *(float4*)(sBias + 4 * t) = (p.bias && n < p.N) ? *(const float4*)(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // HALO: tile_m -> (frame, patch row, patch column)
    int hf = 0, hy0 = 0, hx0 = 0;
    if (HALO) {
        const int PW = (p.Win + 15) >> 4, PH = (p.Hin + 7) >> 3;
        hf = tile_m / (PH * PW);
        const int r = tile_m - hf * (PH * PW);
        hy0 = (r / PW) * 8; hx0 = (r % PW) * 16;
    }
    // tile row -> output row m (and whether it exists)
    auto row_m = [&](int row, bool& ok) -> int {
        if (HALO) {
            const int y = hy0 + (row >> 4), x = hx0 + (row & 15);
            ok = y < p.Hin && x < p.Win;
            if (p.sc_oh > 0) return (hf * p.sc_oh + y * p.sc_sy + p.sc_oy) * p.sc_ow + x * p.sc_sx + p.sc_ox;      // output scatter (nn.UpConv2x parity launches)
            return (hf * p.Hin + y) * p.Win + x;
        }
        ok = m0 + row < M;
        if (p.sc_oh > 0 && ok) {      // output scatter (ABI 9): row (f, y, x) of this launch -> row of the [F][sc_oh][sc_ow] grid; residuals and the store follow it
            const int m = m0 + row, HWs = p.Hout * p.Wout;
            const int f = m / HWs, r = m - f * HWs, y = r / p.Wout, x = r - y * p.Wout;
            return (f * p.sc_oh + y * p.sc_sy + p.sc_oy) * p.sc_ow + x * p.sc_sx + p.sc_ox;
        }
        return m0 + row;
    };

    const int Cin = p.C0 + p.C1;
    const int KW = p.ksize_w > 0 ? p.ksize_w : p.ksize;
    const int HWo = p.Hout * p.Wout;
    const int c8 = t & (CH - 1);               // this thread's 16-byte slot inside the k tile row
    const int rsw = SWZ(t >> SH);              // swizzle key of every row this thread stages (rows differ by RPB, key unchanged)
    // per-row gather state
    int rpix[ARA], ryb[ARA], rxb[ARA], rfr[ARA];
    bool rv[ARA];
    // LIN: byte pointer of (row m0 + (t >> 3), chunk swizzled); rows past M are clamped to M-1 (never stored by the epilogue)
    const unsigned char* linA[LIN ? ARA : 1];
    if constexpr (LIN) {
#pragma unroll
        for (int i = 0; i < ARA; ++i) {
            int m = m0 + (t >> SHA) + RPBA * i;
            m = m < M ? m : M - 1;
            linA[i] = (const unsigned char*)p.in0 + ((int64_t)m * p.C0 + ((c8 ^ rsw) << 3)) * 2;
        }
    }
    unsigned okm[F9 ? (ARA + 1) / 2 : 1];       // FAST9: bit (i & 1) * 16 + tap of okm[i >> 1] = tap of row i is inside the image
    int pix9[F9 ? ARA : 1];
    if constexpr (F9) {
#pragma unroll
        for (int i = 0; i < (ARA + 1) / 2; ++i) okm[i] = 0u;
#pragma unroll
        for (int i = 0; i < ARA; ++i) {
            const int m = m0 + (t >> SHA) + RPBA * i;
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
#pragma unroll
    for (int i = 0; i < ((LIN || F9) ? 0 : ARA); ++i) {
        const int m = m0 + (t >> SHA) + RPBA * i;
        rv[i] = m < M;
        const int mm = rv[i] ? m : 0;
        const int f = mm / HWo, rem = mm - f * HWo;
        const int y = rem / p.Wout, x = rem - y * p.Wout;
        ryb[i] = y * p.stride - p.pad_t; rxb[i] = x * p.stride - p.pad_l;
        rpix[i] = (f * p.Hin + ryb[i]) * p.Win + rxb[i];     // pixel index of tap (0,0)
        rfr[i] = f;
    }
    // HALO slots: LDS slot (pass i, thread t) = halo pixel hp, physical chunk hc; holds logical chunk hc ^ (hp & 7)
    int hpix[HP]; bool hok[HP];
    if (HALO) {
#pragma unroll
        for (int i = 0; i < HP; ++i) {
            const int idx = i * 256 + t, hp = idx >> 3;
            const int y = hy0 - p.pad_t + hp / 18, x = hx0 - p.pad_l + hp % 18;      // (3x3: pad 1; the 2x2 parity convolutions of nn.UpConv2x: pad 0 or 1 per axis)
            hok[i] = hp < 180 && y >= 0 && y < p.Hin && x >= 0 && x < p.Win;
            hpix[i] = (hf * p.Hin + y) * p.Win + x;
            if (idx < HALO_PX * 8) *(uint4*)(sA0 + idx * 16) = make_uint4(0, 0, 0, 0);     // out-of-image halo stays zero (those lanes are masked in the DMA)
        }
        __syncthreads();
    }
    const bool resize = (p.Hv != p.Hin) || (p.Wv != p.Win);
    const unsigned short* wbase = (const unsigned short*)p.weight;

    uint4 ra[FAST ? 1 : AR];
    uint4 ra2[AF32 ? AR : 1];
    uint4 rb[FAST ? 1 : BCH];

    // ---- FAST: LDS-DMA fill of k tile kt (lies inside one tap and one source) into the given buffers
    auto dma_tile = [&](int kt, unsigned char* bufA, unsigned char* bufB) {
        if constexpr (LIN) {
            unsigned char* a = bufA + wave * 1024;
#pragma unroll
            for (int i = 0; i < ARA; ++i) glds16(linA[i] + kt * (BKT * 2), a + i * RPBA * RPA);
            unsigned char* b = bufB + wave * 1024;
            const unsigned short* wrow = wbase + (int64_t)(n0 + (t >> SH)) * p.Kpad + kt * BKT + ((c8 ^ rsw) << 3);
#pragma unroll
            for (int i = 0; i < BCH; ++i) {
                if (BN % RPB == 0 || (t >> SH) + RPB * i < BN) glds16(wrow + (int64_t)(RPB * i) * p.Kpad, b + i * RPB * RP);
            }
            return;
        }
        const int k0 = kt * BKT;
        const int tap = k0 / Cin;
        int cc = k0 - tap * Cin;
        const int ky = tap / KW, kx = tap - ky * KW;
        const unsigned char* src = (const unsigned char*)p.in0;
        int Cs = p.C0;
        if (cc >= p.C0) { src = (const unsigned char*)p.in1; cc -= p.C0; Cs = p.C1; }
        const int dpix = ky * p.Win + kx;
        // swizzle on the SOURCE address, LDS image stays lane-linear.  FAST32: physical 16-byte chunk pc of row r holds the
        // logical 8-float group c = (pc>>1) ^ (r&7), half h = (pc&1) ^ (c&1)  (both ds_read_b128 of an operand conflict free)
        const int c32 = ((t & 15) >> 1) ^ ((t >> 4) & 7);
        const int csrc = A32 ? cc + c32 * 8 + (((t & 1) ^ (c32 & 1)) << 2) : cc + ((c8 ^ rsw) << 3);
        constexpr int ES = A32 ? 4 : 2;
        unsigned char* a = bufA + wave * 1024;
        if constexpr (F9) {
#pragma unroll
            for (int i = 0; i < ARA; ++i) {
                const bool ok = (okm[i >> 1] >> ((i & 1) * 16 + tap)) & 1u;
                const void* g = ok ? (const void*)(src + ((int64_t)(pix9[i] + dpix) * Cs + csrc) * 2) : (const void*)g_zero_page;
                glds16(g, a + i * RPBA * RPA);
            }
        } else {
#pragma unroll
        for (int i = 0; i < ARA; ++i) {
            const int yv = ryb[i] + ky, xv = rxb[i] + kx;
            const bool ok = rv[i] && yv >= 0 && yv < p.Hv && xv >= 0 && xv < p.Wv;
            int pix = rpix[i] + dpix;
            if (resize) pix = (rfr[i] * p.Hin + (yv * p.Hin) / p.Hv) * p.Win + (xv * p.Win) / p.Wv;   // fused nearest upsample
            const void* g = ok ? (const void*)(src + ((int64_t)pix * Cs + csrc) * ES) : (const void*)g_zero_page;
            glds16(g, a + i * RPBA * RPA);
        }
        }
        unsigned char* b = bufB + wave * 1024;
        const unsigned short* wrow = wbase + (int64_t)(n0 + (t >> SH)) * p.Kpad + k0 + ((c8 ^ rsw) << 3);
#pragma unroll
        for (int i = 0; i < BCH; ++i) {
            if (BN % RPB == 0 || (t >> SH) + RPB * i < BN) glds16(wrow + (int64_t)(RPB * i) * p.Kpad, b + i * RPB * RP);
        }
    };
    // ---- HALO: halo of channel chunk c (64 channels of one source); B tile of (chunk, tap)
    auto dma_halo = [&](int c) {
        int cc = c * 64;
        const unsigned char* src = (const unsigned char*)p.in0;
        int Cs = p.C0;
        if (cc >= p.C0) { src = (const unsigned char*)p.in1; cc -= p.C0; Cs = p.C1; }
        const int cs = cc + ((((t & 7) ^ ((t >> 3) & 7))) << 3);      // (pass*256 + t) >> 3 has the same low 3 bits as t >> 3
#pragma unroll
        for (int i = 0; i < HP; ++i) {
            if (i * 256 + wave * 64 < HALO_PX * 8) {
                if (hok[i]) glds16(src + ((int64_t)hpix[i] * Cs + cs) * 2, sA0 + (i * 256 + wave * 64) * 16);
            }
        }
    };
    auto dma_b = [&](int kofs, unsigned char* bufB) {
        unsigned char* b = bufB + wave * 1024;
        const unsigned short* wrow = wbase + (int64_t)(n0 + (t >> SH)) * p.Kpad + kofs + ((c8 ^ rsw) << 3);
#pragma unroll
        for (int i = 0; i < BCH; ++i) {
            if (BN % RPB == 0 || (t >> SH) + RPB * i < BN) glds16(wrow + (int64_t)(RPB * i) * p.Kpad, b + i * RPB * RP);
        }
    };
    // ---- generic: register staged
    auto load_tile = [&](int kt) {
        const int k = kt * BKT + c8 * 8;
        const bool kvalid = k < p.K;
        const int tap = kvalid ? k / Cin : 0;
        int cc = k - tap * Cin;
        const int ky = tap / KW, kx = tap - ky * KW;
        const unsigned char* src = (const unsigned char*)p.in0;
        int Cs = p.C0;
        if (cc >= p.C0) { src = (const unsigned char*)p.in1; cc -= p.C0; Cs = p.C1; }
        const int dpix = ky * p.Win + kx;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const int yv = ryb[i] + ky, xv = rxb[i] + kx;
            const bool ok = kvalid && rv[i] && yv >= 0 && yv < p.Hv && xv >= 0 && xv < p.Wv;
            int pix = rpix[i] + dpix;
            if (resize) pix = (rfr[i] * p.Hin + (yv * p.Hin) / p.Hv) * p.Win + (xv * p.Win) / p.Wv;
            const int64_t off = (int64_t)pix * Cs + cc;
            if (AF32) {
                const float4* g = (const float4*)(src + off * 4);
                if (ok) { ra[i] = *(const uint4*)g; ra2[i] = *(const uint4*)(g + 1); }
                else { ra[i] = make_uint4(0, 0, 0, 0); ra2[i] = make_uint4(0, 0, 0, 0); }
            } else {
                ra[i] = ok ? *(const uint4*)(src + off * 2) : make_uint4(0, 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < BCH; ++i) {
            const int row = (t >> SH) + RPB * i;
            if (BN % RPB == 0 || row < BN) rb[i] = *(const uint4*)(wbase + (int64_t)(n0 + row) * p.Kpad + kt * BKT + c8 * 8);
        }
    };
    auto store_tile = [&](unsigned char* a, unsigned char* b) {
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const int row = (t >> SH) + RPB * i;
            uint4 v;
            if (AF32) {
                float f[8];
                *(uint4*)&f[0] = ra[i]; *(uint4*)&f[4] = ra2[i];
                v = pack8<T>(f);
            } else v = ra[i];
            *(uint4*)(a + row * RP + ((c8 ^ SWZ(row)) << 4)) = v;
        }
#pragma unroll
        for (int i = 0; i < BCH; ++i) {
            const int row = (t >> SH) + RPB * i;
            if (BN % RPB == 0 || row < BN) *(uint4*)(b + row * RP + ((c8 ^ SWZ(row)) << 4)) = rb[i];
        }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.Kpad / BKT;
    const int lr = lane & 15, lq = lane >> 4;
    // one k tile: prefetch the next tile into (nA,nB), run the MFMAs on (cA,cB), then barrier
    auto k_step = [&](int kt, const unsigned char* cA, const unsigned char* cB, unsigned char* nA, unsigned char* nB, int tapofs = 0) {
        const bool more = kt + 1 < nk;
        if (more) { if (FAST) dma_tile(kt + 1, nA, nB); else load_tile(kt + 1); }
        const unsigned char* a = cA + (wr * MT * 16) * RPA;
        const unsigned char* b = cB + (wc * NT * 16) * RP;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            uint4 af[MT], bf[NT];
            const int ch = s * 4 + lq;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int row = i * 16 + lr;
                if constexpr (HALO) {
                    const int hp = (wr * MT + i) * 18 + lr + tapofs;       // halo pixel of (patch row, column lr) shifted by the tap
                    af[i] = *(const uint4*)(cA + hp * 128 + ((ch ^ (hp & 7)) << 4));
                } else if constexpr (A32) {
                    const int q = (ch ^ (row & 7)) << 1, hb = ch & 1;
                    float f[8];
                    *(uint4*)&f[0] = *(const uint4*)(a + row * RPA + ((q | hb) << 4));
                    *(uint4*)&f[4] = *(const uint4*)(a + row * RPA + ((q | (hb ^ 1)) << 4));
                    af[i] = pack8<T>(f);
                } else af[i] = *(const uint4*)(a + row * RP + ((ch ^ SWZ(row)) << 4));
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int row = j * 16 + lr;
                bf[j] = *(const uint4*)(b + row * RP + ((ch ^ SWZ(row)) << 4));
            }
            // swapped operands: D[n-in-tile][m-in-tile] -> lane (lr,lq) owns row m = ..+lr, channels n = ..+4*lq+{0..3}
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = T::mfma(bf[j], af[i], acc[i][j]);
        }
        if (!FAST && more) store_tile(nA, nB);
        __syncthreads();     // with LDS-DMA in flight hipcc drains vmcnt(0) here: the prefetch overlapped the MFMAs
    };

    if constexpr (HALO) {
        const int nchunks = Cin >> 6;
        for (int c = 0; c < nchunks; ++c) {
#pragma unroll 1
            for (int tap = 0; tap < p.ksize * KW; ++tap) {
                if (tap == 0) dma_halo(c);
                dma_b(tap * Cin + c * 64, sB0);
                __syncthreads();
                k_step(nk, sA0, sB0, sA0, sB0, (tap / KW) * 18 + tap % KW);
            }
        }
    } else if (SPLIT == 2) {
        // single buffer: fill -> barrier -> MFMAs -> barrier; half the LDS, so twice the co-resident blocks hide the fill
        for (int kt = 0; kt < nk; ++kt) {
            if (FAST) dma_tile(kt, sA0, sB0);
            else { load_tile(kt); store_tile(sA0, sB0); }
            __syncthreads();
            k_step(nk, sA0, sB0, sA0, sB0);      // kt argument = nk: no prefetch inside, ends with a barrier
        }
    } else if (SPLIT == 1) {
        if (FAST) dma_tile(0, sA0, sB0);
        else { load_tile(0); store_tile(sA0, sB0); }
        __syncthreads();
        for (int kt = 0; kt < nk; kt += 2) {
            k_step(kt, sA0, sB0, sA1, sB1);
            if (kt + 1 < nk) k_step(kt + 1, sA1, sB1, sA0, sB0);
        }
    } else {
        unsigned char* a2 = sAB; unsigned char* b2 = sAB + 2 * BM * RP;
        if (FAST) dma_tile(0, a2, b2);
        else { load_tile(0); store_tile(a2, b2); }
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            k_step(kt, a2 + cur * BM * RP, b2 + cur * BN * RP, a2 + (cur ^ 1) * BM * RP, b2 + (cur ^ 1) * BN * RP);
        }
    }

    // LEAN epilogue (vv_gemm_epilogue.h) in the halo-tile 3x3 kernels: +1.5-3.4 % on the ResBlock / VAE convolutions with a residual, +0.3-1.1 % on the
    // others.  NOT in the loaders held to 128 VGPRs for a fourth block per CU (LIN, FAST9: it spills there, -30 %).  The fp32-operand loader (FAST32: zero
    // convolutions, conv_shortcuts -- HBM-bound fp32-in / fp32-out layers) loses 8-10 % with the lean form alone and GAINS 9-15 % with lean + STAGED (row-major
    // residual reads / stores through a wave-private LDS tile): profiles/r5_epilogue_ab.txt
    gemm_epilogue<T, MT, NT, MODE == MODE_HALO || (MODE == MODE_FAST32 && NT == 5), (MODE == MODE_HALO || MODE == MODE_FAST32) && NT == 5, GN>(p, acc, wr * MT * 16, n0 + wc * NT * 16, lr, lq, HWo, row_m, stage_bias ? sBias + wc * NT * 16 : nullptr,
                                                ((HALO || A32) && NT == 5) ? (float*)sA0 + wave * (16 * (NT * 16 + 4)) : nullptr,      // (halo tiles: a strip = 16 consecutive pixels of one image row; the halo buffer is dead after the k loop's last barrier)
                                                // GroupNorm partials of the output (round 6): row block = (patch of the frame, wave row): [frame][PH * PW * WR][N][2]
                                                GN ? p.gn_partials + (((int64_t)tile_m * WR + wr) * p.N + n0 + wc * NT * 16) * 2 : nullptr);
}

template <typename T, int WR, int WC, int MT, int NT, int MODE>
int launch_cfg(const vv_conv_params& p, int M, hipStream_t st) {
    constexpr int BM = WR * MT * 16, BN = WC * NT * 16;
    if (p.Npad % BN != 0) VV_FAIL(VV_E_ARG, "vv_conv_gemm: Npad %d not a multiple of tile N %d", p.Npad, BN);
    int tilesM = (M + BM - 1) / BM;
    const int tilesN = p.Npad / BN;
    if constexpr (MODE == MODE_HALO) tilesM = p.F * ((p.Hin + 7) / 8) * ((p.Win + 15) / 16);
    static const int split = VV_AB_INT("VV_GEMM_SPLIT", 2);
    // 128x128 tiles: 130 VGPRs uncapped; capping at 128 (4 spilled) lets a 4th block share the CU (LDS 4 x 32-40 KB)
    static const bool occ4 = !VV_AB_ENV("VV_GEMM_NO_OCC4");
    constexpr bool CAN4 = WR * WC == 4 && (MODE == MODE_LIN || MODE == MODE_FAST9 || (NT == 4 && (MODE == MODE_FAST || MODE == MODE_HALO)));
    if (CAN4 && occ4 && (MODE == MODE_HALO || split == 2)) { if constexpr (CAN4) hipLaunchKernelGGL((conv_gemm_kernel<T, WR, WC, MT, NT, MODE, 2, 64, 4>), dim3(tilesM * tilesN), dim3(256), 0, st, p, M, tilesM, tilesN); }
    else if constexpr (MODE == MODE_FAST32 || MODE == MODE_HALO) hipLaunchKernelGGL((conv_gemm_kernel<T, WR, WC, MT, NT, MODE, 2, 64>), dim3(tilesM * tilesN), dim3(256), 0, st, p, M, tilesM, tilesN);
#ifdef VV_AB
    else if (split == 1) hipLaunchKernelGGL((conv_gemm_kernel<T, WR, WC, MT, NT, MODE, 1, 64>), dim3(tilesM * tilesN), dim3(256), 0, st, p, M, tilesM, tilesN);
    else if (split == 3 && MODE == MODE_FAST) hipLaunchKernelGGL((conv_gemm_kernel<T, WR, WC, MT, NT, MODE, 1, 32>), dim3(tilesM * tilesN), dim3(256), 0, st, p, M, tilesM, tilesN);
    else if (split == 0) hipLaunchKernelGGL((conv_gemm_kernel<T, WR, WC, MT, NT, MODE, 0, 64>), dim3(tilesM * tilesN), dim3(256), 0, st, p, M, tilesM, tilesN);
#endif
    else hipLaunchKernelGGL((conv_gemm_kernel<T, WR, WC, MT, NT, MODE, 2, 64>), dim3(tilesM * tilesN), dim3(256), 0, st, p, M, tilesM, tilesN);
    VV_CHECK_LAUNCH("vv_conv_gemm");
    return VV_OK;
}

template <typename T, int MODE>
int launch_t(const vv_conv_params& p, int M, hipStream_t st) {
    // tile choice: GEGLU needs an even number of N tiles per wave; N % 160 == 0 -> 128x160; tiny N -> 128x16
#ifdef VV_AB      // lab: 128 x 320 tile (wave tile 64 x 160: 14 operand fragments per 40 MFMAs instead of 9 per 20; 2 blocks per CU), profiles/r3_gemm_n320_ab.txt
    static const bool n320 = VV_AB_ENV("VV_GEMM_N320");
    if constexpr (MODE != MODE_H16 && MODE != MODE_F32) {
        if (n320 && p.Npad % 320 == 0 && M >= 4096) return launch_cfg<T, 2, 2, 4, 10, MODE>(p, M, st);
    }
#endif
    if (p.epilogue == VV_EPI_GEGLU) return launch_cfg<T, 2, 2, 4, 4, MODE>(p, M, st);
    // N a multiple of both: the 128x128 tile runs 4 blocks per CU (128 VGPRs) against 3 for 128x160 -> +2..8 % on the LDS-DMA
    // loaders when there are enough row tiles (profiles/r1_gemm_ab.txt, eighth A/B)
    // (opt-in: in the pipeline the 3x3 convs lose 2-3 % with it, and the linear layers now run 128x160 at 4 blocks through LIN)
    static const bool pref128 = VV_AB_ENV("VV_GEMM_PREF128") && !VV_AB_ENV("VV_GEMM_NO_OCC4");
    if (pref128 && p.Npad % 128 == 0 && (MODE == MODE_FAST || MODE == MODE_HALO) && M >= 16384) return launch_cfg<T, 2, 2, 4, 4, MODE>(p, M, st);
    if (p.Npad % 160 == 0) return launch_cfg<T, 2, 2, 4, 5, MODE>(p, M, st);   // (a 256x160 4-wave tile measured the same: profiles/r1_gemm_ab.txt)
    if (p.Npad % 128 == 0) return launch_cfg<T, 2, 2, 4, 4, MODE>(p, M, st);
    if (p.Npad % 16 == 0 && p.Npad <= 64) return launch_cfg<T, 4, 1, 2, 1, MODE>(p, M, st);
    VV_FAIL(VV_E_ARG, "vv_conv_gemm: unsupported Npad %d (need %%160, %%128 or 16..64 %%16)", p.Npad);
}

template <typename T>
int launch_mode(const vv_conv_params& p, int M, hipStream_t st) {
    const bool fast = (p.C0 % 64 == 0) && (p.C1 % 64 == 0) && p.Kpad == p.K;
    static const bool no32 = VV_AB_ENV("VV_GEMM_NO_FAST32");
    constexpr int dt = std::is_same<T, BF16>::value ? VV_BF16 : VV_F16;
    if (p.gn_partials) {      // GroupNorm partials out of the epilogue: only the 128 x 160 halo-tile kernel with the staged fp32 epilogue writes them (vvhip.h)
        const bool ok = fast && p.in_dtype != VV_F32 && p.ksize == 3 && (p.ksize_w == 0 || p.ksize_w == 3) && p.pad_t == 1 && p.pad_l == 1 && p.stride == 1 && p.sc_oh == 0 &&
                        p.Hv == p.Hin && p.Wv == p.Win && p.Hout == p.Hin && p.Wout == p.Win && p.epilogue != VV_EPI_GEGLU && p.Npad % 160 == 0 && (p.N & 3) == 0 && (p.ldo & 3) == 0 &&
                        p.out_dtype == VV_F32 && p.split_heads <= 0 && !p.rowvec && !p.res1 && p.act == VV_ACT_NONE && (!p.res0 || p.res_dtype == VV_F32);
        if (!ok) VV_FAIL(VV_E_UNSUPPORTED, "vv_conv_gemm: gn_partials needs the 128 x 160 halo-tile 3x3 kernel with the staged fp32 epilogue (see vvhip.h)");
        const int tilesM = p.F * ((p.Hin + 7) / 8) * ((p.Win + 15) / 16), tilesN = p.Npad / 160;
        hipLaunchKernelGGL((conv_gemm_kernel<T, 2, 2, 4, 5, MODE_HALO, 2, 64, 2, true>), dim3(tilesM * tilesN), dim3(256), 0, st, p, M, tilesM, tilesN);      // its own instantiation: the
        VV_CHECK_LAUNCH("vv_conv_gemm");                                                                                                                          // plain kernel keeps its registers
        return VV_OK;
    }
    if (p.tile_hint != 1) {   // compute-bound shapes: the 256-row tile kernel (vv_gemm256.hip)
        const int r = vv_gemm256_try(&p, dt, p.tile_hint >= 2 ? p.tile_hint - 1 : 0, (void*)st);
        if (r > -1000) return r;
    }
#ifdef VV_AB
    if (p.sc_oh == 0) {   // opt-in 256-pixel halo kernel (vv_conv3.hip; measured slower than the 128-row halo tile); it has no output scatter
        const int r = vv_conv3_halo_try(&p, dt, (void*)st);
        if (r > -1000) return r;
    }
#endif
    static const bool nohalo = VV_AB_ENV("VV_GEMM_NO_HALO");
    // 3x3 / pad 1, and (round 5) the 2x2 / pad 0 or 1 parity convolutions of nn.UpConv2x with their scattered store: the 10 x 18 halo of an 8 x 16 patch starts
    // at (y0 - pad_t, x0 - pad_l) and holds every tap of both kernel sizes
    const bool halo3 = p.ksize == 3 && (p.ksize_w == 0 || p.ksize_w == 3) && p.pad_t == 1 && p.pad_l == 1 && p.sc_oh == 0;
    const bool halo2 = p.ksize == 2 && (p.ksize_w == 0 || p.ksize_w == 2) && p.pad_t >= 0 && p.pad_t <= 1 && p.pad_l >= 0 && p.pad_l <= 1;
    if (fast && !nohalo && p.in_dtype != VV_F32 && (halo3 || halo2) && p.stride == 1 &&
        p.Hv == p.Hin && p.Wv == p.Win && p.Hout == p.Hin && p.Wout == p.Win && p.epilogue != VV_EPI_GEGLU && (p.Npad % 160 == 0 || p.Npad % 128 == 0 || (p.Npad % 16 == 0 && p.Npad <= 64))) {      // (the narrow tile: conv_out layers, 4 MFMAs per k tile -- all data movement, the halo saves 3/4 of it)
        // patch grid waste <= 15 % (the halo tile is worth 17-25 %)
        const int64_t cover = (int64_t)((p.Hin + 7) / 8) * 8 * ((p.Win + 15) / 16) * 16;
        if (cover * 100 <= (int64_t)p.Hin * p.Win * 115) return launch_t<T, MODE_HALO>(p, M, st);
    }
    static const bool nolin = VV_AB_ENV("VV_GEMM_NO_LIN");
    if (fast && !nolin && p.in_dtype != VV_F32 && p.ksize == 1 && p.ksize_w <= 1 && p.stride == 1 && p.pad_t == 0 && p.pad_l == 0 && p.C1 == 0 &&
        p.Hv == p.Hin && p.Wv == p.Win && p.Hout == p.Hin && p.Wout == p.Win) return launch_t<T, MODE_LIN>(p, M, st);
    if (p.in_dtype == VV_F32) return (fast && !no32) ? launch_t<T, MODE_FAST32>(p, M, st) : launch_t<T, MODE_F32>(p, M, st);
    static const bool no9 = VV_AB_ENV("VV_GEMM_NO_FAST9");
    if (fast && !no9 && p.Hv == p.Hin && p.Wv == p.Win && p.ksize * (p.ksize_w > 0 ? p.ksize_w : p.ksize) <= 9) return launch_t<T, MODE_FAST9>(p, M, st);
    return fast ? launch_t<T, MODE_FAST>(p, M, st) : launch_t<T, MODE_H16>(p, M, st);
}

}  // namespace

// Build split (build.sh): compiled twice, -DVV_DT_ONLY=0 = the BF16 instantiations behind vv_conv_gemm_launch_bf16, -DVV_DT_ONLY=1 = the F16 ones plus
// the entry point (see vv_gemm256.hip).
#if defined(VV_DT_ONLY) && VV_DT_ONLY == 0
extern "C" int vv_conv_gemm_launch_bf16(const vv_conv_params* pp, int M, void* stream) { return launch_mode<BF16>(*pp, M, (hipStream_t)stream); }
#else
#if defined(VV_DT_ONLY)
extern "C" int vv_conv_gemm_launch_bf16(const vv_conv_params* pp, int M, void* stream);
#endif
extern "C" int vv_conv_gn_partial_blocks(int Hout, int Wout) { return ((Hout + 7) / 8) * ((Wout + 15) / 16) * 2; }      // 8 x 16 patches x 2 wave rows

extern "C" int vv_conv_gemm(const vv_conv_params* pp, int dtype, void* stream) {
    if (!pp) VV_FAIL(VV_E_ARG, "vv_conv_gemm: null params");
    const vv_conv_params& p = *pp;
    if (dtype != VV_BF16 && dtype != VV_F16) VV_FAIL(VV_E_ARG, "vv_conv_gemm: dtype must be VV_BF16 or VV_F16");
    if (!p.in0 || !p.weight || !p.out) VV_FAIL(VV_E_ARG, "vv_conv_gemm: null tensor pointer");
    if (p.C0 <= 0 || p.C0 % 8 || p.C1 < 0 || p.C1 % 8 || (p.C1 > 0 && !p.in1)) VV_FAIL(VV_E_ARG, "vv_conv_gemm: C0=%d C1=%d must be multiples of 8", p.C0, p.C1);
    const int kw_ = p.ksize_w > 0 ? p.ksize_w : p.ksize;
    if (p.ksize < 1 || p.ksize > 7 || p.ksize == 4 || p.ksize == 6 || kw_ < 1 || kw_ > 7 || kw_ == 4 || kw_ == 6) VV_FAIL      // 2x2: ConvTranspose-style / space-to-depth layers of SAM 2
       (VV_E_ARG, "vv_conv_gemm: kernel %dx%d", p.ksize, kw_);
    if (p.act != VV_ACT_NONE && p.act != VV_ACT_RELU && p.act != VV_ACT_LRELU) VV_FAIL(VV_E_ARG, "vv_conv_gemm: act %d", p.act);
    if (p.stride != 1 && p.stride != 2) VV_FAIL(VV_E_ARG, "vv_conv_gemm: stride %d", p.stride);
    if (p.K != p.ksize * kw_ * (p.C0 + p.C1)) VV_FAIL(VV_E_ARG, "vv_conv_gemm: K=%d != kh*kw*Cin", p.K);
    if (p.Kpad % BK || p.Kpad < p.K) VV_FAIL(VV_E_ARG, "vv_conv_gemm: Kpad=%d must be a multiple of 64 >= K", p.Kpad);
    if (p.in_dtype != VV_F32 && p.in_dtype != dtype) VV_FAIL(VV_E_ARG, "vv_conv_gemm: in_dtype mismatch");
    if (p.out_dtype != VV_F32 && p.out_dtype != dtype) VV_FAIL(VV_E_ARG, "vv_conv_gemm: out_dtype mismatch");
    if ((p.res0 || p.res1) && p.res_dtype != VV_F32 && p.res_dtype != dtype) VV_FAIL(VV_E_ARG, "vv_conv_gemm: res_dtype mismatch");
    if (p.epilogue == VV_EPI_GEGLU && (p.N % 32 || p.rowvec || p.res0 || p.res1 || (p.ldo & 3))) VV_FAIL(VV_E_ARG, "vv_conv_gemm: GEGLU needs N%%32==0, ldo%%4==0 and no residual/rowvec");
    const int64_t stok_ = p.split_tokens < 0 ? -(int64_t)p.split_tokens : p.split_tokens;
    if (p.split_heads > 0 && (p.split_dim <= 0 || (p.split_dim & 3) || stok_ <= 0 || p.N != 3 * p.split_heads * p.split_dim || p.out_dtype == VV_F32 ||
                              p.res0 || p.res1 || p.epilogue == VV_EPI_GEGLU || ((int64_t)p.F * p.Hout * p.Wout) % stok_ ||
                              (int64_t)p.N * stok_ > 0x7fffffff))
        VV_FAIL(VV_E_ARG, "vv_conv_gemm: split_heads needs N = 3*heads*dim, dim %% 4 == 0, h16 output, no residual / GEGLU, M %% split_tokens == 0");
    if (p.F <= 0 || p.Hout <= 0 || p.Wout <= 0 || p.Hin <= 0 || p.Win <= 0 || p.Hv <= 0 || p.Wv <= 0) VV_FAIL(VV_E_ARG, "vv_conv_gemm: bad geometry");
    const int64_t M64 = (int64_t)p.F * p.Hout * p.Wout;
    if (p.sc_oh != 0) {
        if (p.sc_oh < 0 || p.sc_ow <= 0 || p.sc_sy <= 0 || p.sc_sx <= 0 || p.sc_oy < 0 || p.sc_ox < 0 || (int64_t)(p.Hout - 1) * p.sc_sy + p.sc_oy >= p.sc_oh ||
            (int64_t)(p.Wout - 1) * p.sc_sx + p.sc_ox >= p.sc_ow)
            VV_FAIL(VV_E_ARG, "vv_conv_gemm: output scatter (%d x %d, step %d x %d, origin %d, %d) does not hold the %d x %d grid", p.sc_oh, p.sc_ow, p.sc_sy, p.sc_sx,
                    p.sc_oy, p.sc_ox, p.Hout, p.Wout);
        if (p.epilogue == VV_EPI_GEGLU || p.split_heads > 0 || p.rowvec) VV_FAIL(VV_E_ARG, "vv_conv_gemm: output scatter with GEGLU / split_heads / rowvec");
        if ((int64_t)p.F * p.sc_oh * p.sc_ow > 0x7fffffff) VV_FAIL(VV_E_ARG, "vv_conv_gemm: more than 2^31 pixels");
    }
    if (M64 > 0x7fffffff || (int64_t)p.F * p.Hin * p.Win > 0x7fffffff) VV_FAIL(VV_E_ARG, "vv_conv_gemm: more than 2^31 pixels");
    const int M = (int)M64;
    hipStream_t st = (hipStream_t)stream;
#if defined(VV_DT_ONLY)
    return dtype == VV_BF16 ? vv_conv_gemm_launch_bf16(&p, M, stream) : launch_mode<F16>(p, M, st);
#else
    return dtype == VV_BF16 ? launch_mode<BF16>(p, M, st) : launch_mode<F16>(p, M, st);
#endif
}
#endif
