// K1/K6: implicit-GEMM convolution / linear on MFMA (gfx950).  See include/vvhip.h (vv_conv_gemm).
//
// Tile: BM x BN x 64, 256 threads = 4 waves in a WR x WC grid, each wave MT x NT tiles of 16x16 (mfma 16x16x32).
// A (activations) is gathered im2col-style straight from NHWC global memory into registers (16 B = 8 channels per
// lane, one k-tile ahead of the MFMAs: issue-early / write-late), then written to an XOR-swizzled LDS image
// [row][64] so the MFMA operand reads are ds_read_b128.  B (weights, [N][K] K-contiguous) is staged the same way.
// The block index is remapped so that the column tiles of one row panel run on the same XCD (shared L2).
#include "vv_common.h"

namespace {

constexpr int BK = 64;

template <typename T, int WR, int WC, int MT, int NT, bool AF32>
__global__ __launch_bounds__(256) void conv_gemm_kernel(const vv_conv_params p, const int M, const int tilesM, const int tilesN) {
    constexpr int BM = WR * MT * 16, BN = WC * NT * 16;
    constexpr int AR = BM / 32;                 // A rows staged per thread
    constexpr int BCH = (BN * 8 + 255) / 256;   // B chunks staged per thread
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sA = smem;                          // [2][BM][128 B]
    unsigned char* sB = smem + 2 * BM * 128;           // [2][BN][128 B]

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
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

    const int Cin = p.C0 + p.C1;
    const int HWo = p.Hout * p.Wout;
    const int c8 = t & 7;                      // this thread's 16-byte chunk column inside the 64-wide k tile
    // per-row gather state
    int rf[AR], ryb[AR], rxb[AR];
    bool rv[AR];
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        const int m = m0 + (t >> 3) + 32 * i;
        rv[i] = m < M;
        const int mm = rv[i] ? m : 0;
        const int f = mm / HWo, rem = mm - f * HWo;
        const int y = rem / p.Wout, x = rem - y * p.Wout;
        rf[i] = f; ryb[i] = y * p.stride - p.pad_t; rxb[i] = x * p.stride - p.pad_l;
    }
    const bool resize = (p.Hv != p.Hin) || (p.Wv != p.Win);
    const unsigned short* wbase = (const unsigned short*)p.weight;

    uint4 ra[AR];          // staged A chunks (h16)  -- or first half of fp32
    uint4 ra2[AF32 ? AR : 1];
    uint4 rb[BCH];

    auto load_tile = [&](int kt) {
        const int k = kt * BK + c8 * 8;
        const bool kvalid = k < p.K;
        const int tap = kvalid ? k / Cin : 0;
        int cc = k - tap * Cin;
        const int ky = tap / p.ksize, kx = tap - ky * p.ksize;
        const unsigned char* src = (const unsigned char*)p.in0;
        int Cs = p.C0;
        if (cc >= p.C0) { src = (const unsigned char*)p.in1; cc -= p.C0; Cs = p.C1; }
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            int yv = ryb[i] + ky, xv = rxb[i] + kx;
            bool ok = kvalid && rv[i] && yv >= 0 && yv < p.Hv && xv >= 0 && xv < p.Wv;
            if (resize) { yv = (yv * p.Hin) / p.Hv; xv = (xv * p.Win) / p.Wv; }
            const int64_t pix = ((int64_t)rf[i] * p.Hin + yv) * p.Win + xv;
            const int64_t off = pix * Cs + cc;
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
            const int ch = t + 256 * i;
            if (BN * 8 % 256 == 0 || ch < BN * 8) {
                const int row = ch >> 3, c = ch & 7;
                rb[i] = *(const uint4*)(wbase + (int64_t)(n0 + row) * p.Kpad + kt * BK + c * 8);
            }
        }
    };
    auto store_tile = [&](int buf) {
        unsigned char* a = sA + buf * BM * 128;
        unsigned char* b = sB + buf * BN * 128;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const int row = (t >> 3) + 32 * i;
            uint4 v;
            if (AF32) {
                float f[8];
                *(uint4*)&f[0] = ra[i]; *(uint4*)&f[4] = ra2[i];
                v = pack8<T>(f);
            } else v = ra[i];
            *(uint4*)(a + row * 128 + ((c8 ^ (row & 7)) << 4)) = v;
        }
#pragma unroll
        for (int i = 0; i < BCH; ++i) {
            const int ch = t + 256 * i;
            if (BN * 8 % 256 == 0 || ch < BN * 8) {
                const int row = ch >> 3, c = ch & 7;
                *(uint4*)(b + row * 128 + ((c ^ (row & 7)) << 4)) = rb[i];
            }
        }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.Kpad / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    const int lr = lane & 15, lq = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1);
        const unsigned char* a = sA + cur * BM * 128 + (wr * MT * 16) * 128;
        const unsigned char* b = sB + cur * BN * 128 + (wc * NT * 16) * 128;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            uint4 af[MT], bf[NT];
            const int ch = s * 4 + lq;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int row = i * 16 + lr;
                af[i] = *(const uint4*)(a + row * 128 + ((ch ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int row = j * 16 + lr;
                bf[j] = *(const uint4*)(b + row * 128 + ((ch ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = T::mfma(af[i], bf[j], acc[i][j]);
        }
        if (kt + 1 < nk) store_tile(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: C/D layout of mfma 16x16: col = lane&15, row = (lane>>4)*4 + reg
    const bool geglu = p.epilogue == VV_EPI_GEGLU;
    const int N = p.N;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + wr * MT * 16 + i * 16 + lq * 4 + r;
            if (m >= M) continue;
            const float* rowv = p.rowvec ? p.rowvec + (int64_t)(m / HWo) * N : nullptr;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int n = n0 + wc * NT * 16 + j * 16 + lr;
                if (geglu) {
                    if (NT % 2 == 0 && (j & 1) == 0 && n < N) {
                        const float val = acc[i][j][r] + (p.bias ? p.bias[n] : 0.f);
                        const float gate = acc[i][j + (NT % 2 == 0 ? 1 : 0)][r] + (p.bias ? p.bias[n + 16] : 0.f);
                        const float o = val * gelu_f(gate);
                        const int64_t oc = (int64_t)m * p.ldo + ((n0 + wc * NT * 16 + j * 16) >> 1) + lr;
                        if (p.out_dtype == VV_F32) ((float*)p.out)[oc] = o;
                        else ((unsigned short*)p.out)[oc] = T::from_f32(o);
                    }
                    continue;
                }
                if (n >= N) continue;
                float v = acc[i][j][r];
                if (p.bias) v += p.bias[n];
                v *= p.out_scale;
                if (rowv) v += rowv[n];
                const int64_t ri = (int64_t)m * N + n;
                if (p.res0) v += (p.res_dtype == VV_F32) ? ((const float*)p.res0)[ri] : T::to_f32(((const unsigned short*)p.res0)[ri]);
                if (p.res1) v += (p.res_dtype == VV_F32) ? ((const float*)p.res1)[ri] : T::to_f32(((const unsigned short*)p.res1)[ri]);
                const int64_t oc = (int64_t)m * p.ldo + n;
                if (p.out_dtype == VV_F32) ((float*)p.out)[oc] = v;
                else ((unsigned short*)p.out)[oc] = T::from_f32(v);
            }
        }
    }
}

template <typename T, int WR, int WC, int MT, int NT, bool AF32>
int launch_cfg(const vv_conv_params& p, int M, hipStream_t st) {
    constexpr int BM = WR * MT * 16, BN = WC * NT * 16;
    if (p.Npad % BN != 0) VV_FAIL(VV_E_ARG, "vv_conv_gemm: Npad %d not a multiple of tile N %d", p.Npad, BN);
    const int tilesM = (M + BM - 1) / BM, tilesN = p.Npad / BN;
    const size_t lds = 2 * (BM + BN) * 128;
    auto kern = conv_gemm_kernel<T, WR, WC, MT, NT, AF32>;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            VV_FAIL(VV_E_LAUNCH, "vv_conv_gemm: cannot set dynamic LDS size %zu", lds);
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(tilesM * tilesN), dim3(256), lds, st, p, M, tilesM, tilesN);
    VV_CHECK_LAUNCH("vv_conv_gemm");
    return VV_OK;
}

template <typename T, bool AF32>
int launch_t(const vv_conv_params& p, int M, hipStream_t st) {
    // tile choice: GEGLU needs an even number of N tiles per wave; N % 160 == 0 -> 128x160; tiny N -> 128x16
    if (p.epilogue == VV_EPI_GEGLU) return launch_cfg<T, 2, 2, 4, 4, AF32>(p, M, st);
    if (p.Npad % 160 == 0) return launch_cfg<T, 2, 2, 4, 5, AF32>(p, M, st);
    if (p.Npad % 128 == 0) return launch_cfg<T, 2, 2, 4, 4, AF32>(p, M, st);
    if (p.Npad % 16 == 0 && p.Npad <= 64) return launch_cfg<T, 4, 1, 2, 1, AF32>(p, M, st);
    VV_FAIL(VV_E_ARG, "vv_conv_gemm: unsupported Npad %d (need %%160, %%128 or 16..64 %%16)", p.Npad);
}

}  // namespace

extern "C" int vv_conv_gemm(const vv_conv_params* pp, int dtype, void* stream) {
    if (!pp) VV_FAIL(VV_E_ARG, "vv_conv_gemm: null params");
    const vv_conv_params& p = *pp;
    if (dtype != VV_BF16 && dtype != VV_F16) VV_FAIL(VV_E_ARG, "vv_conv_gemm: dtype must be VV_BF16 or VV_F16");
    if (!p.in0 || !p.weight || !p.out) VV_FAIL(VV_E_ARG, "vv_conv_gemm: null tensor pointer");
    if (p.C0 <= 0 || p.C0 % 8 || p.C1 < 0 || p.C1 % 8 || (p.C1 > 0 && !p.in1)) VV_FAIL(VV_E_ARG, "vv_conv_gemm: C0=%d C1=%d must be multiples of 8", p.C0, p.C1);
    if (p.ksize != 1 && p.ksize != 3) VV_FAIL(VV_E_ARG, "vv_conv_gemm: ksize %d", p.ksize);
    if (p.stride != 1 && p.stride != 2) VV_FAIL(VV_E_ARG, "vv_conv_gemm: stride %d", p.stride);
    if (p.K != p.ksize * p.ksize * (p.C0 + p.C1)) VV_FAIL(VV_E_ARG, "vv_conv_gemm: K=%d != ks^2*Cin", p.K);
    if (p.Kpad % BK || p.Kpad < p.K) VV_FAIL(VV_E_ARG, "vv_conv_gemm: Kpad=%d must be a multiple of 64 >= K", p.Kpad);
    if (p.in_dtype != VV_F32 && p.in_dtype != dtype) VV_FAIL(VV_E_ARG, "vv_conv_gemm: in_dtype mismatch");
    if (p.out_dtype != VV_F32 && p.out_dtype != dtype) VV_FAIL(VV_E_ARG, "vv_conv_gemm: out_dtype mismatch");
    if ((p.res0 || p.res1) && p.res_dtype != VV_F32 && p.res_dtype != dtype) VV_FAIL(VV_E_ARG, "vv_conv_gemm: res_dtype mismatch");
    if (p.epilogue == VV_EPI_GEGLU && (p.N % 32 || p.rowvec || p.res0 || p.res1)) VV_FAIL(VV_E_ARG, "vv_conv_gemm: GEGLU needs N%%32==0 and no residual/rowvec");
    if (p.F <= 0 || p.Hout <= 0 || p.Wout <= 0 || p.Hin <= 0 || p.Win <= 0 || p.Hv <= 0 || p.Wv <= 0) VV_FAIL(VV_E_ARG, "vv_conv_gemm: bad geometry");
    const int64_t M64 = (int64_t)p.F * p.Hout * p.Wout;
    if (M64 > 0x7fffffff) VV_FAIL(VV_E_ARG, "vv_conv_gemm: M too large");
    const int M = (int)M64;
    hipStream_t st = (hipStream_t)stream;
    const bool af32 = p.in_dtype == VV_F32;
    if (dtype == VV_BF16) return af32 ? launch_t<BF16, true>(p, M, st) : launch_t<BF16, false>(p, M, st);
    return af32 ? launch_t<F16, true>(p, M, st) : launch_t<F16, false>(p, M, st);
}
