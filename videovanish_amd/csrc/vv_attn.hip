// K3/K4/K5: flash attention on MFMA for gfx950 (spatial self-attention, 77-token cross-attention, temporal
// attention over the frame axis, VAE mid-block attention).  See include/vvhip.h (vv_attention).
//
// Per wave: QT tiles of 16 queries.  S^T = K Q^T (mfma 16x16x32: A = K rows from LDS, B = Q held in registers),
// so every lane owns ONE query column: the online-softmax max/sum/rescale are lane-local plus two cross-lane
// exchanges, and the accumulator registers of S^T are -- after exp2 and packing -- directly the B operand of
// O^T = V^T P^T.  V^T operands come from the row-major V tile through ds_read_b64_tr_b16 (hardware transpose).
// Head dims that are not MFMA multiples are zero-padded in LDS only (K-dim to 32, V-dim to 16).
#include <stdlib.h>
#include <type_traits>
#include "vv_attn_common.h"
#ifndef VV_ATTN_PART
#define VV_ATTN_PART 0
#endif

namespace {

// DMA = true: K/V tiles go global -> LDS by LDS-DMA into two static buffers (tile k+1 in flight during the MFMAs and the
// softmax of tile k, ONE barrier per tile, no staging registers); DMA = false: register staged through dynamic LDS.
// KIND only names the instantiation (0 spatial self-attention, 1 cross-attention to the text tokens; temporal attention has its own
// tile shape): profilers then report the launches of each use separately (profiles/*_kernel_stats.csv).
template <typename T, int D, int QT, int KVT, int NW, bool PREFETCH, int OCC = 1, bool DMA = false, int KIND = 0>
__global__ __launch_bounds__(NW * 64, OCC) void attn_kernel(const vv_attn_params p, const int nqt) {
    constexpr int DK = (D + 31) / 32 * 32, DKC = DK / 8, KS = DK / 32;
    constexpr int DV = (D + 15) / 16 * 16, DVC = DV / 8, NDT = DV / 16;
    constexpr int KT = KVT / 16, US = KVT / 32;
    constexpr int PK = DK * 2 + 32, PV = DV * 2 + ((DV * 2) % 64 == 0 ? 32 : 0);      // conflict-free ds_read_b128 / ds_read_b64_tr_b16 (bank model: tools/lds_bank_model.py)
    constexpr int NT = NW * 64;
    constexpr int KCH = (KVT * DKC + NT - 1) / NT, VCH = (KVT * DVC + NT - 1) / NT;
    constexpr int BQ = NW * QT * 16;
    // spare zero-padded V column (d = 40 -> 48): put 1.0 there, then row D of O^T accumulates sum_k P = the softmax
    // denominator on the MATRIX pipe instead of 16 packed adds per tile on the (issue-bound) VALU
    constexpr bool ONES = DV > D;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sK = smem;
    unsigned char* sV = smem + KVT * PK;
    constexpr int NSK = KVT * PK / 16, NSV = KVT * PV / 16;            // 16-byte LDS slots per K / V tile (multiples of 64)
    constexpr int KP = (NSK + NT - 1) / NT, VP = (NSV + NT - 1) / NT;  // DMA passes
    static_assert(!DMA || (NSK % 64 == 0 && NSV % 64 == 0), "K/V tile must be whole 1 KB wave blocks");
    __shared__ __attribute__((aligned(1024))) unsigned char dK0[DMA ? KVT * PK : 16];
    __shared__ __attribute__((aligned(1024))) unsigned char dK1[DMA ? KVT * PK : 16];
    __shared__ __attribute__((aligned(1024))) unsigned char dV0[DMA ? KVT * PV : 16];
    __shared__ __attribute__((aligned(1024))) unsigned char dV1[DMA ? KVT * PV : 16];

    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int li = lane & 15, lg = lane >> 4;
    // XCD-aware decode: blocks i and i+8 share an XCD (and its L2).  Give every XCD its own (batch, head) pairs and walk
    // that pair's query tiles on it, so the pair's K/V (re-read by every query tile) stays resident in ONE 4 MiB L2.
    int qt, h, b;
    {
        const int nbh = p.B * p.heads;
        const int full = (nbh / 8) * 8;                       // pairs handled in XCD-striped rounds of 8
        const int bid = blockIdx.x;
        int bh;
        if (bid < full * nqt) { const int xcd = bid & 7, idx = bid >> 3; bh = (idx / nqt) * 8 + xcd; qt = idx % nqt; }
        else { const int r = bid - full * nqt; bh = full + r / nqt; qt = r % nqt; }
        h = bh % p.heads; b = bh / p.heads;
    }

    const unsigned short* Q = (const unsigned short*)p.q + (int64_t)b * p.q_bs + (int64_t)h * (p.q_hs ? p.q_hs : D);
    const unsigned short* Kp = (const unsigned short*)p.k + (int64_t)b * p.k_bs + (int64_t)h * (p.k_hs ? p.k_hs : D);
    const unsigned short* Vp = (const unsigned short*)p.v + (int64_t)b * p.v_bs + (int64_t)h * (p.v_hs ? p.v_hs : D);
    unsigned short* O = (unsigned short*)p.o + (int64_t)b * p.o_bs + (int64_t)h * (p.o_hs ? p.o_hs : D);

    const int q0 = qt * BQ + wave * QT * 16;
    // ---- Q fragments (B operand of S^T): lane holds Q[q0 + j*16 + li][s*32 + lg*8 .. +7]
    uint4 qf[QT][KS];
#pragma unroll
    for (int j = 0; j < QT; ++j)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int q = q0 + j * 16 + li, d0 = s * 32 + lg * 8;
            qf[j][s] = (q < p.Nq && d0 < D) ? *(const uint4*)(Q + (int64_t)q * p.q_rs + d0) : make_uint4(0, 0, 0, 0);
        }

    f32x4 oacc[NDT][QT];
#pragma unroll
    for (int d = 0; d < NDT; ++d)
#pragma unroll
        for (int j = 0; j < QT; ++j) oacc[d][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float mrun[QT], lrun[QT];
#pragma unroll
    for (int j = 0; j < QT; ++j) { mrun[j] = -1e30f; lrun[j] = 0.f; }
    const float c = p.q_prescaled ? 1.0f : p.scale * 1.4426950408889634f;

    uint4 rk[KCH], rv[VCH];
    auto load_kv = [&](int kv0) {
#pragma unroll
        for (int i = 0; i < KCH; ++i) {
            const int idx = t + NT * i;
            const int row = idx / DKC, ch = idx - row * DKC;
            const int key = kv0 + row;
            rk[i] = (idx < KVT * DKC && key < p.Nkv && ch * 8 < D) ? *(const uint4*)(Kp + (int64_t)key * p.k_rs + ch * 8) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < VCH; ++i) {
            const int idx = t + NT * i;
            const int row = idx / DVC, ch = idx - row * DVC;
            const int key = kv0 + row;
            rv[i] = (idx < KVT * DVC && key < p.Nkv && ch * 8 < D) ? *(const uint4*)(Vp + (int64_t)key * p.v_rs + ch * 8) : make_uint4(0, 0, 0, 0);
            if (ONES && ch * 8 == D) rv[i].x = (unsigned)T::from_f32(1.0f);        // V[key][D] = 1 (low half-word), rest of the pad stays 0
        }
    };
    auto store_kv = [&]() {
#pragma unroll
        for (int i = 0; i < KCH; ++i) {
            const int idx = t + NT * i;
            const int row = idx / DKC, ch = idx - row * DKC;
            if (KVT * DKC % NT == 0 || idx < KVT * DKC) *(uint4*)(sK + row * PK + ch * 16) = rk[i];
        }
#pragma unroll
        for (int i = 0; i < VCH; ++i) {
            const int idx = t + NT * i;
            const int row = idx / DVC, ch = idx - row * DVC;
            if (KVT * DVC % NT == 0 || idx < KVT * DVC) *(uint4*)(sV + row * PV + ch * 16) = rv[i];
        }
    };

    const int ntiles = (p.Nkv + KVT - 1) / KVT;
    // one KV tile out of (sK, sV); MASK is a compile-time flag so that only the LAST (ragged) tile pays for the key mask
    auto compute = [&](const int it, const unsigned char* sK, const unsigned char* sV, auto mask_tag) {
        constexpr bool MASK = decltype(mask_tag)::value;
        const int kv0 = it * KVT;

        // ---- S^T = K Q^T
        f32x4 sacc[KT][QT];
        auto qk = [&]() {
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int j = 0; j < QT; ++j) sacc[kt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KS; ++s) {
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) {
                    const uint4 kf = *(const uint4*)(sK + (kt * 16 + li) * PK + (s * 4 + lg) * 16);
#pragma unroll
                    for (int j = 0; j < QT; ++j) sacc[kt][j] = T::mfma(kf, qf[j][s], sacc[kt][j]);
                }
            }
            if (MASK) {
#pragma unroll
                for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (kv0 + kt * 16 + lg * 4 + r >= p.Nkv) {
#pragma unroll
                            for (int j = 0; j < QT; ++j) sacc[kt][j][r] = -1e30f;
                        }
            }
        };
        uint4 pb[US][QT];
        // ---- O^T += V^T P^T   (A = V^T via transposed LDS reads; k-slot order matches the packing of pb)
        auto pv = [&]() {
#pragma unroll
            for (int u = 0; u < US; ++u) {
#pragma unroll
                for (int d = 0; d < NDT; ++d) {
                    const unsigned char* a0 = sV + (u * 32 + 4 * lg + (li >> 2)) * PV + (d * 16 + 4 * (li & 3)) * 2;
                    const uint2 lo = ds_read_tr16(a0);
                    const uint2 hi = ds_read_tr16(a0 + 16 * PV);
                    const uint4 vf = make_uint4(lo.x, lo.y, hi.x, hi.y);
#pragma unroll
                    for (int j = 0; j < QT; ++j) oacc[d][j] = T::mfma(vf, pb[u][j], oacc[d][j]);
                }
            }
        };
        qk();
        // ---- online softmax (per query column = per lane, replicated over the 4 lane groups)
#pragma unroll
        for (int j = 0; j < QT; ++j) {
            float mx = sacc[0][j][0];
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sacc[kt][j][r]);
            mx = fmaxf(mx, __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, mx), (16 << 10) | 0x1f)));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float mnew = fmaxf(mrun[j], mx);
            // the kernel is VALU-issue bound (every wave64 VALU op costs 4 cycles, v_exp 8): use packed fp32 ops for the
            // exponent arguments, and skip the O^T / l rescale when no query row of this wave raised its maximum
            const bool grew = __builtin_amdgcn_ballot_w64(mnew > mrun[j]) != 0;     // wave-uniform
            const float mc = mnew * c;
            const vv_f32x2 c2 = {c, c}, nmc2 = {-mc, -mc};
            vv_f32x2 ps2 = {0.f, 0.f};
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    const vv_f32x2 a2 = __builtin_elementwise_fma((vv_f32x2){sacc[kt][j][r], sacc[kt][j][r + 1]}, c2, nmc2);   // v_pk_fma_f32
                    const float e0 = __builtin_amdgcn_exp2f(a2.x), e1 = __builtin_amdgcn_exp2f(a2.y);
                    sacc[kt][j][r] = e0; sacc[kt][j][r + 1] = e1;
                    if (!ONES) ps2 += (vv_f32x2){e0, e1};                      // v_pk_add_f32
                }
            const float ps = ps2.x + ps2.y;
            if (grew) {
                const float alpha = __builtin_amdgcn_exp2f((mrun[j] - mnew) * c);
                if (!ONES) lrun[j] = lrun[j] * alpha + ps;
#pragma unroll
                for (int d = 0; d < NDT; ++d) oacc[d][j] *= alpha;
            } else {
                if (!ONES) lrun[j] += ps;
            }
            mrun[j] = mnew;
#pragma unroll
            for (int u = 0; u < US; ++u)
                pb[u][j] = make_uint4(pack2<T>(sacc[2 * u][j][0], sacc[2 * u][j][1]), pack2<T>(sacc[2 * u][j][2], sacc[2 * u][j][3]),
                                      pack2<T>(sacc[2 * u + 1][j][0], sacc[2 * u + 1][j][1]), pack2<T>(sacc[2 * u + 1][j][2], sacc[2 * u + 1][j][3]));
        }
        pv();
    };
    const bool ragged = (p.Nkv % KVT) != 0;
    if constexpr (DMA) {
        // ---- per-thread DMA slots: LDS slot (pass i, thread t) <-> (tile row, 16-byte chunk).  Slots that hold no tensor
        // data (head-dim / pitch padding, the ONES column) are written ONCE here (with the rest of both buffers) and never
        // touched by the DMA: the lanes that own them are masked off (a masked lane's LDS slot = M0 + 16 * lane is not written).
        unsigned koff[KP], voff[VP];      // byte offsets from Kp / Vp of the slot's source in the NEXT tile to issue
        bool kdata[KP], vdata[VP];
#pragma unroll
        for (int i = 0; i < KP; ++i) {
            const int sidx = i * NT + t, row = sidx / (PK / 16), ch = sidx - row * (PK / 16);
            kdata[i] = ch * 8 < D && sidx < NSK;
            koff[i] = (unsigned)(row * (int)p.k_rs + ch * 8) * 2u;
            if (sidx < NSK) {
                const uint4 fill = make_uint4(0, 0, 0, 0);
                *(uint4*)(dK0 + sidx * 16) = fill; *(uint4*)(dK1 + sidx * 16) = fill;
            }
        }
#pragma unroll
        for (int i = 0; i < VP; ++i) {
            const int sidx = i * NT + t, row = sidx / (PV / 16), ch = sidx - row * (PV / 16);
            vdata[i] = ch * 8 < D && sidx < NSV;
            voff[i] = (unsigned)(row * (int)p.v_rs + ch * 8) * 2u;
            if (sidx < NSV) {
                const uint4 fill = make_uint4((ONES && ch * 8 == D) ? (unsigned)T::from_f32(1.0f) : 0u, 0, 0, 0);
                *(uint4*)(dV0 + sidx * 16) = fill; *(uint4*)(dV1 + sidx * 16) = fill;
            }
        }
        __syncthreads();    // the fill is complete before the first DMA lands (rows past Nkv of a ragged tile stay finite)
        const unsigned kstep = (unsigned)(KVT * (int)p.k_rs * 2), vstep = (unsigned)(KVT * (int)p.v_rs * 2);
        int issued = 0;     // tiles issued so far (= index of the tile the offsets address)
        auto dma_issue = [&](unsigned char* bK, unsigned char* bV, auto check_tag) {
            constexpr bool CHECK = decltype(check_tag)::value;      // ragged last tile: rows past Nkv keep the stale (finite) tile
            const int kv0 = issued * KVT;
            // tile base = wave-uniform pointer (SGPR pair), per-lane part = a constant 32-bit offset: the DMA takes the saddr + voffset
            // form and the loop carries no per-lane address arithmetic
            const unsigned char* kt = (const unsigned char*)Kp + (size_t)issued * kstep;
            const unsigned char* vt = (const unsigned char*)Vp + (size_t)issued * vstep;
#pragma unroll
            for (int i = 0; i < KP; ++i) {
                if (NSK % NT == 0 || i * NT + wave * 64 < NSK) {
                    if (kdata[i] && (!CHECK || kv0 + (i * NT + t) / (PK / 16) < p.Nkv)) glds16(kt + koff[i], bK + (i * NT + wave * 64) * 16);
                }
            }
#pragma unroll
            for (int i = 0; i < VP; ++i) {
                if (NSV % NT == 0 || i * NT + wave * 64 < NSV) {
                    if (vdata[i] && (!CHECK || kv0 + (i * NT + t) / (PV / 16) < p.Nkv)) glds16(vt + voff[i], bV + (i * NT + wave * 64) * 16);
                }
            }
            ++issued;
        };
        auto step = [&](const int it, unsigned char* cK, unsigned char* cV, unsigned char* nK, unsigned char* nV) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's share of tile `it` has landed
            __syncthreads();                                      // ... everybody's has, and everybody is done with tile it-1
            if (it + 1 < ntiles) {
                if (ragged && it + 2 == ntiles) dma_issue(nK, nV, std::true_type{});
                else dma_issue(nK, nV, std::false_type{});
            }
            if (ragged && it + 1 == ntiles) compute(it, cK, cV, std::true_type{});
            else compute(it, cK, cV, std::false_type{});
        };
        if (ragged && ntiles == 1) dma_issue(dK0, dV0, std::true_type{});
        else dma_issue(dK0, dV0, std::false_type{});
        for (int it = 0; it < ntiles; it += 2) {
            step(it, dK0, dV0, dK1, dV1);
            if (it + 1 < ntiles) step(it + 1, dK1, dV1, dK0, dV0);
        }
    } else {
        load_kv(0);
        store_kv();
        __syncthreads();
        auto tile_step = [&](const int it, auto mask_tag) {
            if (PREFETCH && it + 1 < ntiles) load_kv((it + 1) * KVT);
            compute(it, sK, sV, mask_tag);
            __syncthreads();
            if (it + 1 < ntiles) {
                if (!PREFETCH) load_kv((it + 1) * KVT);
                store_kv();
                __syncthreads();
            }
        };
        const int nfull = ragged ? ntiles - 1 : ntiles;
        for (int it = 0; it < nfull; ++it) tile_step(it, std::false_type{});
        if (ragged) tile_step(ntiles - 1, std::true_type{});
    }
    // ---- finalize: O[q][d] = O^T[d][q] / l
#pragma unroll
    for (int j = 0; j < QT; ++j) {
        float l;
        if (ONES) {
            // denominator = O^T[D][q]: register (D % 16) % 4 of d-tile D/16 on lane group (D % 16) / 4
            l = __shfl(oacc[D / 16][j][(D % 16) % 4], ((D % 16) / 4) * 16 + li);
        } else {
            l = lrun[j];
            l += __shfl_xor(l, 16);
            l += __shfl_xor(l, 32);
        }
        const float inv = 1.0f / l;
        const int q = q0 + j * 16 + li;
        if (q < p.Nq) {
            if (p.lse && lg == 0) p.lse[((int64_t)b * p.heads + h) * p.Nq + q] = mrun[j] * c + __log2f(l);      // log-sum-exp of this call's keys (log2 domain) for vv_attention_merge
#pragma unroll
            for (int d = 0; d < NDT; ++d) {
                const int dd = d * 16 + lg * 4;
                if (dd < D) {
                    const uint2 o2 = make_uint2(pack2<T>(oacc[d][j][0] * inv, oacc[d][j][1] * inv), pack2<T>(oacc[d][j][2] * inv, oacc[d][j][3] * inv));
                    *(uint2*)(O + (int64_t)q * p.o_rs + dd) = o2;
                }
            }
        }
    }
}


template <typename T, int D, int QT, int KVT, int NW, bool PREFETCH, int OCC = 1, bool DMA = false, int KIND = 0>
int attn_launch(const vv_attn_params& p, hipStream_t st) {
    constexpr int DK = (D + 31) / 32 * 32, DV = (D + 15) / 16 * 16;
    constexpr int PK = DK * 2 + 32, PV = DV * 2 + ((DV * 2) % 64 == 0 ? 32 : 0);   // conflict-free ds_read_b128 / ds_read_b64_tr_b16 (bank model: tools/lds_bank_model.py)
    constexpr int BQ = NW * QT * 16;
    const size_t lds = DMA ? 0 : (size_t)KVT * (PK + PV);
    const int nqt = (p.Nq + BQ - 1) / BQ;
    const int64_t nblk = (int64_t)p.B * p.heads * nqt;
    if (nblk > 0x7fffffff) VV_FAIL(VV_E_ARG, "vv_attention: grid too large");
    auto kern = attn_kernel<T, D, QT, KVT, NW, PREFETCH, OCC, DMA, KIND>;
    static bool attr_done = false;
    if (!attr_done && lds > 48 * 1024) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            VV_FAIL(VV_E_LAUNCH, "vv_attention: cannot set dynamic LDS size %zu", lds);
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(NW * 64), lds, st, p, nqt);
    VV_CHECK_LAUNCH("vv_attention");
    return VV_OK;
}

template <typename T, int D>
int attn_dispatch(const vv_attn_params& p, hipStream_t st) {
    if constexpr (D >= 512) {
        // (round 6: eight waves of 16 queries instead of four -- the waves share the block's 32-key K / V tiles: the VAE mid attention, 4 frames x 14400 tokens, 4.32 -> 3.58 ms;
        //  profiles/r6_attn160_ab.txt)
        if (p.Nq >= 256) return attn_launch<T, D, 1, 32, 8, false>(p, st);
        return attn_launch<T, D, 1, 32, 4, false>(p, st);
    } else {
        // short sequences (temporal attention over <=32 frames, tiny test shapes): one wave per block, 32-key tiles
        // (round 6: at d = 160 two waves of 16 queries share the one 32-key tile of a temporal attention: 0.0865 -> 0.072 ms at 32 frames x 920 pixels; d = 80: +2 %, unchanged;
        //  profiles/r6_attn160_ab.txt)
        if (D >= 128 && p.Nq <= 32 && p.Nkv <= 32 && p.Nq > 16) return attn_launch<T, D, 1, 32, 2, true>(p, st);
        if (p.Nq <= 32 && p.Nkv <= 32) return attn_launch<T, D, 2, 32, 1, true>(p, st);
        if constexpr (D <= 80) {
            // default for d <= 64: K/V by LDS-DMA, double buffered, 3 waves/SIMD (d = 80 would spill: stays register staged)
            const bool cross = p.Nkv < 128 && p.Nq != p.Nkv;
            if (D <= 64) return cross ? attn_launch<T, D, 2, 64, 4, false, 3, true, 1>(p, st) : attn_launch<T, D, 2, 64, 4, false, 3, true, 0>(p, st);
            return cross ? attn_launch<T, D, 2, 64, 4, true, 1, false, 1>(p, st) : attn_launch<T, D, 2, 64, 4, true, 1, false, 0>(p, st);
        }
        // one long head (SAM 2 memory attention: d = 256, 4096 queries x up to 28736 keys) is 32 blocks whatever the block shape below 8 waves:
        // eight waves of 16 queries keep the block's K/V tile loads covered (1.80 -> 1.05 ms; blocks of fewer than four waves are 7x slower:
        // the register-staged loader wants 256 threads), profiles/r3_sam2_attn256_ab.txt
        if constexpr (D == 256) {
            if ((int64_t)p.B * p.heads * ((p.Nq + 127) / 128) <= 128) return attn_launch<T, D, 1, 64, 8, true>(p, st);
        }
        // d = 160 spatial self-attention (level 2: 32 frames x 8 heads x 920 tokens at 720p): eight waves of 16 queries share a block's K / V tile loads --
        // 0.372 -> 0.325 ms (+14.5 %) against four waves of 32 queries; four waves of 16: 0.329; 32-key tiles: 0.509 (round 6, profiles/r6_attn160_ab.txt)
        // (... and the level-2 cross attention to the 77 text tokens: 0.089 -> 0.076 ms; d = 80 cross attention: no gain, unchanged)
        if constexpr (D == 160) {
            if (p.Nq >= 256) return attn_launch<T, D, 1, 64, 8, true>(p, st);
        }
        return attn_launch<T, D, 2, 64, 4, true>(p, st);
    }
}

template <typename T>
int attn_by_d(const vv_attn_params& p, hipStream_t st) {
    switch (p.D) {
#if VV_ATTN_PART == 0      // small head dims: built with -mllvm -amdgpu-mfma-vgpr-form (accumulators in arch VGPRs)
        case 32: return attn_dispatch<T, 32>(p, st);
        case 40: return attn_dispatch<T, 40>(p, st);
        case 64: return attn_dispatch<T, 64>(p, st);
        case 80: return attn_dispatch<T, 80>(p, st);
#else                      // large head dims need the AGPR half of the register file for O^T
        case 128: return attn_dispatch<T, 128>(p, st);
        case 160: return attn_dispatch<T, 160>(p, st);
        case 256: return attn_dispatch<T, 256>(p, st);
        case 512: return attn_dispatch<T, 512>(p, st);
#endif
        default: VV_FAIL(VV_E_UNSUPPORTED, "vv_attention: head dim %d not built (32,40,64,80,128,160,256,512)", p.D);
    }
}

}  // namespace

#if VV_ATTN_PART == 0
namespace {
template <typename T>
__global__ __launch_bounds__(256) void attn_merge_kernel(const unsigned short* __restrict__ parts, const float* __restrict__ lse, int S, int heads, int Nq, int D,
                                                         int ld, unsigned short* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;        // one thread per (query, head, pair of channels)
    const int half = D >> 1;
    if (i >= (int64_t)Nq * heads * half) return;
    const int c2 = (int)(i % half);
    const int h = (int)((i / half) % heads);
    const int q = (int)(i / ((int64_t)half * heads));
    float mx = -3.0e38f;
    for (int s = 0; s < S; ++s) mx = fmaxf(mx, lse[((int64_t)s * heads + h) * Nq + q]);
    float a0 = 0.f, a1 = 0.f, den = 0.f;
    for (int s = 0; s < S; ++s) {
        const float w = __builtin_amdgcn_exp2f(lse[((int64_t)s * heads + h) * Nq + q] - mx);
        const unsigned u = *(const unsigned*)(parts + ((int64_t)s * Nq + q) * ld + h * D + 2 * c2);
        a0 = fmaf(w, T::to_f32(u & 0xffff), a0); a1 = fmaf(w, T::to_f32(u >> 16), a1); den += w;
    }
    const float inv = 1.0f / den;
    *(unsigned*)(out + (int64_t)q * ld + h * D + 2 * c2) = pack2<T>(a0 * inv, a1 * inv);
}
}  // namespace

extern "C" int vv_attention_merge(const void* o_parts, const float* lse, int S, int heads, int Nq, int D, int ld, void* out, int dtype, void* stream) {
    if (!o_parts || !lse || !out || S <= 0 || heads <= 0 || Nq <= 0 || D <= 0 || (D & 1) || (ld & 1) || ld < heads * D) VV_FAIL(VV_E_ARG, "vv_attention_merge: bad arguments");
    const int64_t n = (int64_t)Nq * heads * (D / 2);
    const dim3 grid((unsigned)((n + 255) / 256));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == VV_BF16) hipLaunchKernelGGL(attn_merge_kernel<BF16>, grid, dim3(256), 0, st, (const unsigned short*)o_parts, lse, S, heads, Nq, D, ld, (unsigned short*)out);
    else if (dtype == VV_F16) hipLaunchKernelGGL(attn_merge_kernel<F16>, grid, dim3(256), 0, st, (const unsigned short*)o_parts, lse, S, heads, Nq, D, ld, (unsigned short*)out);
    else VV_FAIL(VV_E_ARG, "vv_attention_merge: bad dtype");
    VV_CHECK_LAUNCH("vv_attention_merge");
    return VV_OK;
}

extern "C" int vv_attention_large_d(const vv_attn_params* pp, int dtype, void* stream);
extern "C" int vv_attention_mfma32(const vv_attn_params* pp, int dtype, void* stream);
#ifdef VV_AB      // lab build: VV_ATTN_VARIANT selects an experimental variant / timing probe of vv_attn_lab.hip
extern "C" int vv_attention_lab(const vv_attn_params* pp, int dtype, void* stream);
#endif

extern "C" int vv_attention(const vv_attn_params* pp, int dtype, void* stream) {
    if (!pp) VV_FAIL(VV_E_ARG, "vv_attention: null params");
    const vv_attn_params& p = *pp;
    if (!p.q || !p.k || !p.v || !p.o) VV_FAIL(VV_E_ARG, "vv_attention: null pointer");
    if (p.B <= 0 || p.heads <= 0 || p.Nq <= 0 || p.Nkv <= 0) VV_FAIL(VV_E_ARG, "vv_attention: empty problem");
    if ((p.q_rs | p.k_rs | p.v_rs | p.o_rs | p.q_bs | p.k_bs | p.v_bs | p.o_bs) & 3) VV_FAIL(VV_E_ARG, "vv_attention: strides must be multiples of 4 elements (q/k/v: 8)");
    if ((p.q_rs | p.k_rs | p.v_rs | p.q_bs | p.k_bs | p.v_bs | p.q_hs | p.k_hs | p.v_hs) & 7) VV_FAIL(VV_E_ARG, "vv_attention: q/k/v strides must be multiples of 8 elements");
    if (p.o_hs & 3) VV_FAIL(VV_E_ARG, "vv_attention: o_hs must be a multiple of 4 elements");
    if (dtype != VV_BF16 && dtype != VV_F16) VV_FAIL(VV_E_ARG, "vv_attention: bad dtype");
    if (p.lse && p.D == 40) VV_FAIL(VV_E_UNSUPPORTED, "vv_attention: lse output is not available at D = 40");
#ifdef VV_AB
    if (getenv("VV_ATTN_VARIANT")) return vv_attention_lab(pp, dtype, stream);
#endif
    if (p.D == 40 || p.D == 80) {      // vv_attn32.hip: the 32x32x16 kernels with an optimistic softmax reference take the spatial self-attention shapes
        const int r = vv_attention_mfma32(pp, dtype, stream);
        if (r != -1000) return r;
    }
    if (p.D > 80) return vv_attention_large_d(pp, dtype, stream);
    return dtype == VV_BF16 ? attn_by_d<BF16>(p, (hipStream_t)stream) : attn_by_d<F16>(p, (hipStream_t)stream);
}
#else
extern "C" int vv_attention_large_d(const vv_attn_params* pp, int dtype, void* stream) {
    const vv_attn_params& p = *pp;
    return dtype == VV_BF16 ? attn_by_d<BF16>(p, (hipStream_t)stream) : attn_by_d<F16>(p, (hipStream_t)stream);
}
#endif
