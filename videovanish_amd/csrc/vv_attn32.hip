// Spatial self-attention on the FULL-SIZE matrix instruction (v_mfma_f32_32x32x16) with an optimistic softmax reference: head dims 40 (UNet level 0: the
// dominant kernel of the denoise step) and 80 (level 1; SAM 2's 72 -> 80 padded heads).  Split off vv_attn.hip (which keeps the 16x16x32 flash kernels
// of every other shape and calls vv_attention_mfma32 first for D = 40 / 80); built with -mllvm -amdgpu-mfma-vgpr-form like the small-head part there.
#include <type_traits>
#include "vv_attn_common.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------------------
// d = 40 spatial self-attention on the FULL-SIZE matrix instruction (v_mfma_f32_32x32x16): the dominant kernel of the denoise step.
//
// Why another kernel: on the 16x16x32 form an MFMA occupies the SIMD's vector-issue port for 8 of its 16 cycles, and the 28 MFMAs of a
// 32-query x 64-key tile then take as much issue time as the 32 v_exp_f32 the tile also needs -- the kernel above is issue bound at ~48 % matrix
// pipe.  A 32x32x16 MFMA does twice the work per issue (8 of 32 cycles), and its K step of 16 pads d = 40 + the two lazy-maximum slots to 48
// instead of 64: QK^T is 6 MFMAs per tile instead of 16, PV 8 (M = d padded to 64) instead of 12, with 6 + 16 LDS fragment reads instead of 8 + 12.
//
// Per wave: 32 queries (lane & 31 = the query, like the kernel above a lane owns its query's softmax row; lane >> 5 = h splits the keys).
//   S^T[32 keys][32 q] = K Q^T:  A = K rows (ds_read_b128, 16 B = k slots 16s + 8h .. +7), B = Q in registers (3 k steps: 40 data + slots 40,41 =
//   -m hi/lo against 1.0 in K + 6 zeros).  Accumulator register i of a lane holds key (i & 3) + 8 (i >> 2) + 4 h.
//   P = exp2(S^T) packed pairwise IS the B operand of O^T += V^T P^T (k order of step s': key 16 s' + 8 (j >> 2) + 4 h + (j & 3), j = 0..7):
//   no cross-lane movement; the A operand V^T comes through ds_read_b64_tr_b16 with exactly that key order (two reads of 4 consecutive keys).
//   O^T rows 0..39 = the output, row 40 = sum_k P (ONES column of V), rows 41..63 padding.
// LDS images (both 96-byte rows, filled by LDS-DMA, every wave instruction a whole KB):
//   K: row = key, 16-byte chunk c stored at position c ^ ((row >> 3) & 1)  -> ds_read_b128 of 16 rows x one chunk is conflict free;
//   V: key 8 g + 4 b + q stored at row 8 g + 2 q + b                      -> the 4 rows of a transposed read are 2 apart: conflict free.
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <typename T> struct Mfma32;
template <> struct Mfma32<BF16> {
    static __device__ __forceinline__ f32x16 run(uint4 a, uint4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct Mfma32<F16> {
    static __device__ __forceinline__ f32x16 run(uint4 a, uint4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};

// self-attention (Nq == Nkv): the score of query q against ITS OWN key, from the lane's Q fragments (lane half h holds the 16 s + 8 h .. +7 slots of
// the three k steps: 24 + 16 of the 40 products); -1e30 when there is no such key
template <typename T>
__device__ __forceinline__ float attn40_diag_score(const vv_attn_params& p, const unsigned char* Kp, const uint4 (&qx)[3], const int q, const int h) {
    float sum = 0.f;
    const bool on = p.Nq == p.Nkv && q < p.Nkv;
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) {
        const int d0 = 16 * s3 + 8 * h;
        if (on && d0 < 40) {
            float qv[8], kv[8];
            unpack8<T>(qx[s3], qv);
            unpack8<T>(*(const uint4*)(Kp + ((int64_t)q * p.k_rs + d0) * 2), kv);
#pragma unroll
            for (int e = 0; e < 8; ++e) sum += qv[e] * kv[e];
        }
    }
    sum += __shfl_xor(sum, 32);
    return on ? sum : -1e30f;
}

// OPTIMISTIC reference (this kernel) instead of the lazy running one: the softmax reference m of a query is fixed ONCE, before the key loop, from
// the exact maximum of its scores against a 64-key sample spread over the whole sequence -- plus, for self-attention (Nq == Nkv), the query's OWN
// key: trained self-attention maps are often diagonal dominant, and a dominant key outside the sample is what forces the slow repeat below -- (+ MARGIN),
// and rides in the pad slots as before.  MARGIN = -4 since round 4 (was +4): the sample maximum maps to P = 2^4, so a later score may beat it by 12
// binary orders before fp16 overflows (the repeat below makes any input correct), while scores down to 18 orders BELOW the sample maximum keep the full
// 11-bit precision and 28 orders stay representable -- with +4 a bulk of keys 12..20 orders below one sampled outlier went subnormal / to zero although
// thousands of them still carry part of the softmax mass (tests/test_kernels_gpu.py::test_attention_d40_heavy_tail).  The
// loop body then has no maximum, no rescale, no overflow test and NO BRANCH: scores -> exp2 -> pack -> PV, software pipelined one tile deep
// (the QK^T MFMAs of tile t+1 are independent of the exponentials of tile t, so the matrix pipe and the VALU work side by side inside the wave's own
// instruction stream -- the only place where they overlap well on this chip, profiles/r2_attn_lazy_ab.txt).  P = 2^(s - m) may exceed 1: h16 keeps its
// relative precision up to 2^16 (fp16) and the sums are fp32, so a later score may beat the sample maximum by up to 16 + MARGIN binary orders before
// anything is lost.  Beyond that P overflows to inf, the denominator (row 40 of O^T) comes out non-finite, and the BLOCK repeats its keys once with the
// exact maximum (a QK^T-only sweep first): correct for any input, slow only for the blocks that hit it.  bf16 cannot overflow at all.
template <typename T, int NW, int OCC, bool RAGGED>
__global__ __launch_bounds__(NW * 64, OCC) void attn40_kernel(const vv_attn_params p, const int nqt) {
    // LDS: dense 80-byte rows (5 chunks of 8 h16) -- a K or V tile is exactly 5 KB = 5 LDS-DMA wave instructions with EVERY lane active (no
    // exec masking, no pad slots), 10 per 64-key tile.  The constant operand slots come from a region of 1.0 instead of from the rows:
    //   K slots 40..47 (Q carries -m hi, -m lo, 0 x 6 there)  and  V columns 40..43 (O^T rows 40.. = sum_k P, the softmax denominator).
    // Row orders: K natural (80-byte pitch: 16 consecutive rows x one chunk hit 16 different 16-byte bank slots);
    //             V key 16 g + 4 j + q at row 16 g + 4 q + j (the 4 rows of one transposed read are 4 apart: conflict free at 80 bytes).
    constexpr int D = 40, KVT = 64, PR = 80, NCH = 5;
    constexpr int NT = NW * 64, BQ = NW * 32;
    constexpr int TILE = KVT * PR;                            // 5120
    constexpr int NPC = 2 * TILE / 1024;                      // 10 DMA pieces (1 KB each) per tile: 0..4 = K, 5..9 = V
    constexpr int PPW = (NPC + NW - 1) / NW;                  // pieces per wave (piece j -> wave j % NW)
    constexpr int KONES = 32 * PR + 64, VONES = 4096 + 64;   // bytes of 1.0 behind each tile buffer (reached with the key-block / k-step immediates)
    constexpr float MARGIN = -4.0f;     // P = 2^4 at the sample maximum: see the comment above attn40_kernel (round 4)
    // FOUR arrays, not one: hipcc drains vmcnt(0) in front of a ds_read that may alias an LDS-DMA in flight, and tells buffers apart only as
    // distinct __shared__ objects (the DMA of step `it` targets the buffers the step does not read)
    __shared__ __attribute__((aligned(1024))) unsigned char dK0[TILE + KONES];
    __shared__ __attribute__((aligned(1024))) unsigned char dK1[TILE + KONES];
    __shared__ __attribute__((aligned(1024))) unsigned char dV0[TILE + VONES];
    __shared__ __attribute__((aligned(1024))) unsigned char dV1[TILE + VONES];

    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int r = lane & 31, h = lane >> 5;
    int qt, hd, b;
    {
        const int nbh = p.B * p.heads;
        const int full = (nbh / 8) * 8;
        const int bid = blockIdx.x;
        int bh;
        if (bid < full * nqt) { const int xcd = bid & 7, idx = bid >> 3; bh = (idx / nqt) * 8 + xcd; qt = idx % nqt; }
        else { const int rr = bid - full * nqt; bh = full + rr / nqt; qt = rr % nqt; }
        hd = bh % p.heads; b = bh / p.heads;
    }
    const unsigned short* Q = (const unsigned short*)p.q + (int64_t)b * p.q_bs + (int64_t)hd * (p.q_hs ? p.q_hs : D);
    const unsigned char* Kp = (const unsigned char*)((const unsigned short*)p.k + (int64_t)b * p.k_bs + (int64_t)hd * (p.k_hs ? p.k_hs : D));
    const unsigned char* Vp = (const unsigned char*)((const unsigned short*)p.v + (int64_t)b * p.v_bs + (int64_t)hd * (p.v_hs ? p.v_hs : D));
    unsigned short* O = (unsigned short*)p.o + (int64_t)b * p.o_bs + (int64_t)hd * (p.o_hs ? p.o_hs : D);

    // ---- Q fragments: lane (r, h) holds Q[q0 + r][16 s + 8 h .. +7]; chunk 5 (s = 2, h = 1) is the pad chunk: slots 40, 41 = -m (hi, lo)
    const int q0 = qt * BQ + wave * 32;
    uint4 qf[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int q = q0 + r, d0 = 16 * s + 8 * h;
        qf[s] = (q < p.Nq && d0 < D) ? *(const uint4*)(Q + (int64_t)q * p.q_rs + d0) : make_uint4(0, 0, 0, 0);
        if (!p.q_prescaled) {
            float qv[8];
            unpack8<T>(qf[s], qv);
#pragma unroll
            for (int e = 0; e < 8; ++e) qv[e] *= p.scale * 1.4426950408889634f;
            qf[s] = pack8<T>(qv);
        }
    }
    const float dg = attn40_diag_score<T>(p, Kp, qf, q0 + r, h);      // self-attention: score against the query's own key (part of the reference sample)
    // ---- constant regions (written once): 1.0 everywhere; the tile buffers start as zeros (rows of a ragged tile that are never loaded stay finite)
    {
        const unsigned one2 = (unsigned)T::from_f32(1.0f) * 0x10001u;
        const uint4 ones = make_uint4(one2, one2, one2, one2), zero = make_uint4(0, 0, 0, 0);
        // (two loops per region, not `i < TILE / 16 ? zero : ones`: hipcc turned that select into a 32-byte SCRATCH array indexed by the condition -- every thread of the
        //  launch wrote and re-read it: 0.12 GB of HBM writes per level-0 launch, 29 % of the kernel's WRITE_SIZE; found in round 6 through private_segment_fixed_size = 48)
        for (int i = t; i < TILE / 16; i += NT) { *(uint4*)(dK0 + i * 16) = zero; *(uint4*)(dK1 + i * 16) = zero; *(uint4*)(dV0 + i * 16) = zero; *(uint4*)(dV1 + i * 16) = zero; }
        for (int i = TILE / 16 + t; i < (TILE + KONES) / 16; i += NT) { *(uint4*)(dK0 + i * 16) = ones; *(uint4*)(dK1 + i * 16) = ones; }
        for (int i = TILE / 16 + t; i < (TILE + VONES) / 16; i += NT) { *(uint4*)(dV0 + i * 16) = ones; *(uint4*)(dV1 + i * 16) = ones; }
    }
    __syncthreads();

    // ---- this wave's DMA pieces: piece j = wave + NW * i; slot = (j % 5) * 64 + lane = row * 5 + chunk of the K (j < 5) or V tile
    unsigned doff[PPW];          // byte offset of the slot's source inside a tile (K: key = row; V: key = 16 g + 4 (row & 3) + ((row >> 2) & 3))
    int dkey[PPW];               // key index inside the tile (ragged last tile)
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int j = wave + NW * i, slot = (j % 5) * 64 + lane, row = slot / NCH, ch = slot - row * NCH;
        const bool isv = j >= 5;
        const int key = isv ? (row & ~15) + 4 * (row & 3) + ((row >> 2) & 3) : row;
        dkey[i] = key;
        doff[i] = (unsigned)(key * (int)(isv ? p.v_rs : p.k_rs) + ch * 8) * 2u;
    }
    const int ntiles = (p.Nkv + KVT - 1) / KVT;
    const int nlast = p.Nkv - (ntiles - 1) * KVT;             // keys in the last tile (KVT unless RAGGED)
    const unsigned kstep = (unsigned)(KVT * (int)p.k_rs * 2), vstep = (unsigned)(KVT * (int)p.v_rs * 2);
    // K part (kind 0), V part (kind 1) of tile `it` -> LDS buffer at byte offset `dst`; stride > 1: the 64-key SAMPLE (keys 0, stride, 2 stride, ..)
    auto dma = [&](const int kind, const int it, unsigned char* dst, const unsigned stride) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int j = wave + NW * i;                          // wave-uniform
            if (j < NPC && (j >= 5) == (kind == 1)) {
                const unsigned char* src = kind ? Vp + (size_t)it * vstep : Kp + (size_t)it * kstep;
                unsigned off = doff[i];
                if (stride != 1) off += (unsigned)(dkey[i] * (int)p.k_rs * 2) * (stride - 1);
                if (RAGGED && it + 1 == ntiles && stride == 1) { if (dkey[i] < nlast) glds16(src + off, dst + (j % 5) * 1024); }
                else glds16(src + off, dst + (j % 5) * 1024);
            }
        }
    };

    // ---- lane-constant LDS read addresses (byte offsets into a tile buffer; the key block / k step are instruction immediates)
    const int ka0 = r * PR + 16 * h;                             // K chunks h (s = 0) and 2 + h (s = 1: + 32)
    const int ka2 = h ? TILE : r * PR + 64;                     // s = 2: chunk 4 for h = 0, the constant chunk (1.0: slots 40..47) for h = 1
    // transposed V read: 16-lane group g = lane >> 4 (g & 1 = cb: which 16 columns, g >> 1 = h); lane 4 q + pp of the group addresses row q, columns 4 pp ..
    const int vq = (lane >> 2) & 3, vpp = lane & 3, vcb = (lane >> 4) & 1;
    const int va0 = (4 * vq + h) * PR + (16 * vcb + 4 * vpp) * 2;             // d block 0; + (16 (2 kb + s2) + 2 j4) rows
    // d block 1 = columns 32..47 on the 16x16x32 form (k group g = lane >> 4 holds keys 32 kb + 16 (g & 1) + 4 (g >> 1) + {0..3, 8..11}): columns 32..39
    // are data (pp = 0, 1), 40..43 the constant (pp = 2: 1.0 -> O^T rows 40..43 = sum_k P), pp = 3 feeds rows that are never read (repeats pp = 1)
    const int va1 = vpp == 2 ? TILE : (16 * vcb + 4 * vq + h) * PR + 64 + 8 * (vpp & 1);

    // S^T = K Q^T for the 64 keys of the K tile at byte offset kb0 (2 key blocks x 3 k steps)
    auto qk = [&](const unsigned char* sK, f32x16 (&sacc)[2]) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int i = 0; i < 16; ++i) sacc[kb][i] = 0.f;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const uint4 kf = s < 2 ? *(const uint4*)(sK + kb * 32 * PR + ka0 + 32 * s) : *(const uint4*)(sK + kb * 32 * PR + ka2);
                sacc[kb] = Mfma32<T>::run(kf, qf[s], sacc[kb]);
            }
        }
    };
    auto mask_last = [&](f32x16 (&sacc)[2]) {                     // keys past Nkv of the ragged last tile: -inf scores (P = 0)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (kb * 32 + (i & 3) + 8 * (i >> 2) + 4 * h >= nlast) sacc[kb][i] = -1e30f;
    };
    auto row_max = [&](const f32x16 (&sacc)[2]) {
        float mx = sacc[0][0];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) mx = fmaxf(mx, sacc[kb][i]);
        return fmaxf(mx, __shfl_xor(mx, 32));
    };
    auto set_reference = [&](const float target) {                // pad slots 40, 41 of Q <- -target as the nearest h16 hi + lo pair
        const unsigned short hi = T::from_f32(-target);
        const unsigned short lo = T::from_f32(-target - T::to_f32(hi));
        qf[2].x = h == 1 ? ((unsigned)hi | ((unsigned)lo << 16)) : qf[2].x;
    };

    f32x16 oacc;              // O^T rows 0..31 (32x32x16 layout: lane = query, registers + h = d)
    f32x4 o2[2];              // O^T rows 32..47 for the queries 0..15 / 16..31 of the wave (16x16x32 layout: lane & 15 = query, 4 (lane >> 4) + reg = d - 32)
    float la = 0.f, lb = 0.f;
#pragma nounroll
    for (int attempt = 0; attempt < 2; ++attempt) {
        // ---- reference: attempt 0 = maximum over the 64-key sample + MARGIN; attempt 1 (after an overflow) = the exact maximum over all keys
        f32x16 sA[2], sB[2];
        if (attempt == 0) {
            dma(0, 0, dK1, (unsigned)(p.Nkv / KVT));
            dma(0, 0, dK0, 1u);
            dma(1, 0, dV0, 1u);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            qk(dK1, sA);
            set_reference(fmaxf(row_max(sA), dg) + MARGIN);
        } else {
            qf[2].x = h == 1 ? 0u : qf[2].x;                      // plain scores again
            float mx = -1e30f;
            __syncthreads();
            dma(0, 0, dK0, 1u);
            for (int it = 0; it < ntiles; ++it) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (it + 1 < ntiles) { if (it & 1) dma(0, it + 1, dK0, 1u); else dma(0, it + 1, dK1, 1u); }
                if (it & 1) qk(dK1, sA); else qk(dK0, sA);
                if (RAGGED && it + 1 == ntiles) mask_last(sA);
                mx = fmaxf(mx, row_max(sA));
            }
            set_reference(mx);
            __syncthreads();
            dma(0, 0, dK0, 1u);
            dma(1, 0, dV0, 1u);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();                                          // everybody is done with the sample (the last sweep tile) in K buffer 1
        if (ntiles > 1) dma(0, 1, dK1, 1u);
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[i] = 0.f;
        o2[0] = f32x4{0.f, 0.f, 0.f, 0.f}; o2[1] = f32x4{0.f, 0.f, 0.f, 0.f};
        qk(dK0, sA);
        if (RAGGED && ntiles == 1) mask_last(sA);

        // one pipeline step: S_next = K(it+1) Q^T  |  P = exp2(S_cur), packed  |  O^T += V(it)^T P^T     -- ONE basic block, no branch
        auto body = [&](const unsigned char* nK, const unsigned char* cV, f32x16 (&sc)[2], f32x16 (&sn)[2]) {
            qk(nK, sn);
            uint4 pb[2][2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                for (int i = 0; i < 16; ++i) sc[kb][i] = __builtin_amdgcn_exp2f(sc[kb][i]);
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
                    pb[kb][s2] = make_uint4(pack2<T>(sc[kb][8 * s2 + 0], sc[kb][8 * s2 + 1]), pack2<T>(sc[kb][8 * s2 + 2], sc[kb][8 * s2 + 3]),
                                            pack2<T>(sc[kb][8 * s2 + 4], sc[kb][8 * s2 + 5]), pack2<T>(sc[kb][8 * s2 + 6], sc[kb][8 * s2 + 7]));
            }
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    // rows 0..31 of O^T: keys 32 kb + 16 s2 + 8 j4 + 4 h + q, j4 = 0 (elements 0..3), 1 (elements 4..7): V rows 16 (2 kb + s2) + 4 q + 2 j4 + h
                    const int g0 = 16 * (2 * kb + s2) * PR;
                    const uint2 lo = ds_read_tr16(cV + g0 + va0), hi = ds_read_tr16(cV + g0 + 2 * PR + va0);
                    const uint4 vf = make_uint4(lo.x, lo.y, hi.x, hi.y);
                    oacc = Mfma32<T>::run(vf, pb[kb][s2], oacc);
                }
                // rows 32..47 on the 16x16x32 form (M = 16 instead of a second, three-quarters empty 32-row block): v_permlane16_swap turns the
                // (s2 = 0, s2 = 1) dword pairs of P -- rows {q 0..15 | q 16..31} x {h = 0 | h = 1} -- into the B operands of the two query tiles:
                // every 16-lane row then holds queries 0..15 (resp. 16..31) and k group g = (h, s2) = keys 32 kb + 16 s2 + 4 h + {0..3, 8..11}
                uint4 pa, pq;
                {
                    auto s0 = __builtin_amdgcn_permlane16_swap(pb[kb][0].x, pb[kb][1].x, false, false);
                    auto s1 = __builtin_amdgcn_permlane16_swap(pb[kb][0].y, pb[kb][1].y, false, false);
                    auto s2_ = __builtin_amdgcn_permlane16_swap(pb[kb][0].z, pb[kb][1].z, false, false);
                    auto s3 = __builtin_amdgcn_permlane16_swap(pb[kb][0].w, pb[kb][1].w, false, false);
                    pa = make_uint4(s0[0], s1[0], s2_[0], s3[0]);
                    pq = make_uint4(s0[1], s1[1], s2_[1], s3[1]);
                }
                const uint2 lo = ds_read_tr16(cV + kb * 32 * PR + va1), hi = ds_read_tr16(cV + kb * 32 * PR + 2 * PR + va1);
                const uint4 vf = make_uint4(lo.x, lo.y, hi.x, hi.y);
                o2[0] = T::mfma(vf, pa, o2[0]); o2[1] = T::mfma(vf, pq, o2[1]);
            }
        };
        // step `it`: K(it+1) and V(it) have landed (issued one step ago); issue K(it+2) over K(it) and V(it+1) over V(it-1)
        auto step = [&](const int it, unsigned char* kA, unsigned char* kB, unsigned char* vA, unsigned char* vB, f32x16 (&sc)[2], f32x16 (&sn)[2]) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (it + 2 < ntiles) dma(0, it + 2, kA, 1u);
            if (it + 1 < ntiles) dma(1, it + 1, vB, 1u);
            body(kB, vA, sc, sn);                                 // (the last step computes scores of a tile that does not exist: stale K, never used)
            if (RAGGED && it + 2 == ntiles) mask_last(sn);
        };
        for (int it = 0; it < ntiles; it += 2) {
            step(it, dK0, dK1, dV0, dV1, sA, sB);
            if (it + 1 < ntiles) step(it + 1, dK1, dK0, dV1, dV0, sB, sA);
        }
        // ---- denominator = O^T row 40 = register 0 of the row-32.. tiles on the lanes 32..47 (k group 2).  Non-finite (some P overflowed) or zero:
        // the block repeats its keys with the exact maximum
        la = __shfl(o2[0][0], 32 + (lane & 15));
        lb = __shfl(o2[1][0], 32 + (lane & 15));
        const bool bad = !(la > 0.f && la < 3.0e38f && lb > 0.f && lb < 3.0e38f);
        if (attempt == 1 || !__syncthreads_or(bad ? 1 : 0)) break;
    }
    // ---- finalize: O[q][d] = O^T[d][q] / l.  Rows 0..31: register i on lane (r, h) is d = (i & 3) + 8 (i >> 2) + 4 h of query r
    {
        const float inv = 1.0f / (r < 16 ? la : lb);
        const int q = q0 + r;
        if (q < p.Nq) {
            unsigned short* orow = O + (int64_t)q * p.o_rs;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *(uint2*)(orow + 8 * g + 4 * h) = make_uint2(pack2<T>(oacc[4 * g] * inv, oacc[4 * g + 1] * inv), pack2<T>(oacc[4 * g + 2] * inv, oacc[4 * g + 3] * inv));
        }
    }
    // rows 32..39: query tile qt, lane (c = lane & 15, g = lane >> 4 < 2): d = 32 + 4 g + reg of query 16 qt + c
#pragma unroll
    for (int qt2 = 0; qt2 < 2; ++qt2) {
        const float inv = 1.0f / (qt2 ? lb : la);
        const int q = q0 + 16 * qt2 + (lane & 15);
        if (q < p.Nq && (lane >> 4) < 2)
            *(uint2*)(O + (int64_t)q * p.o_rs + 32 + 4 * (lane >> 4)) = make_uint2(pack2<T>(o2[qt2][0] * inv, o2[qt2][1] * inv), pack2<T>(o2[qt2][2] * inv, o2[qt2][3] * inv));
    }
}

template <typename T, int NW, int OCC, bool RAGGED>
__global__ __launch_bounds__(NW * 64, OCC) void attn40q2_kernel(const vv_attn_params p, const int nqt) {
    // TWO 32-query blocks per wave (a, b): every K / V fragment read from LDS and every LDS-DMA piece serves 64 queries -- half the LDS and L2 -> LDS
    // bytes per FLOP of attn40_kernel (the chip is power limited on this kernel: fewer bytes moved = a higher clock).  Not pipelined across
    // tiles; the two blocks overlap each other instead: QK_a, QK_b | exp_a, PV_a | exp_b, PV_b in ONE basic block per tile.
    // LDS images, constant regions and row orders: as attn40_kernel above.
    constexpr int D = 40, KVT = 64, PR = 80, NCH = 5;
    constexpr int NT = NW * 64, QB = 2, BQ = NW * 32 * QB;
    constexpr int TILE = KVT * PR;                            // 5120
    constexpr int NPC = 2 * TILE / 1024;                      // 10 DMA pieces (1 KB each) per tile: 0..4 = K, 5..9 = V
    constexpr int PPW = (NPC + NW - 1) / NW;                  // pieces per wave (piece j -> wave j % NW)
    constexpr int KONES = 32 * PR + 64, VONES = 4096 + 64;   // bytes of 1.0 behind each tile buffer (reached with the key-block / k-step immediates)
    constexpr float MARGIN = -4.0f;     // P = 2^4 at the sample maximum: see the comment above attn40_kernel (round 4)
    // FOUR arrays, not one: hipcc drains vmcnt(0) in front of a ds_read that may alias an LDS-DMA in flight, and tells buffers apart only as
    // distinct __shared__ objects (the DMA of step `it` targets the buffers the step does not read)
    __shared__ __attribute__((aligned(1024))) unsigned char dK0[TILE + KONES];
    __shared__ __attribute__((aligned(1024))) unsigned char dK1[TILE + KONES];
    __shared__ __attribute__((aligned(1024))) unsigned char dV0[TILE + VONES];
    __shared__ __attribute__((aligned(1024))) unsigned char dV1[TILE + VONES];

    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int r = lane & 31, h = lane >> 5;
    int qt, hd, b;
    {
        const int nbh = p.B * p.heads;
        const int full = (nbh / 8) * 8;
        const int bid = blockIdx.x;
        int bh;
        if (bid < full * nqt) { const int xcd = bid & 7, idx = bid >> 3; bh = (idx / nqt) * 8 + xcd; qt = idx % nqt; }
        else { const int rr = bid - full * nqt; bh = full + rr / nqt; qt = rr % nqt; }
        hd = bh % p.heads; b = bh / p.heads;
    }
    const unsigned short* Q = (const unsigned short*)p.q + (int64_t)b * p.q_bs + (int64_t)hd * (p.q_hs ? p.q_hs : D);
    const unsigned char* Kp = (const unsigned char*)((const unsigned short*)p.k + (int64_t)b * p.k_bs + (int64_t)hd * (p.k_hs ? p.k_hs : D));
    const unsigned char* Vp = (const unsigned char*)((const unsigned short*)p.v + (int64_t)b * p.v_bs + (int64_t)hd * (p.v_hs ? p.v_hs : D));
    unsigned short* O = (unsigned short*)p.o + (int64_t)b * p.o_bs + (int64_t)hd * (p.o_hs ? p.o_hs : D);

    // ---- Q fragments: lane (r, h) holds Q[q0 + r][16 s + 8 h .. +7]; chunk 5 (s = 2, h = 1) is the pad chunk: slots 40, 41 = -m (hi, lo)
    const int q0 = qt * BQ + wave * 32 * QB;
    const bool active = q0 < p.Nq;       // wave-uniform: a wave past Nq (N = 14400: three of the four waves of the last query block) feeds the DMA ring only
    uint4 qf[QB][3];
#pragma unroll
    for (int x = 0; x < QB; ++x)
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int q = q0 + 32 * x + r, d0 = 16 * s + 8 * h;
        qf[x][s] = (q < p.Nq && d0 < D) ? *(const uint4*)(Q + (int64_t)q * p.q_rs + d0) : make_uint4(0, 0, 0, 0);
        if (!p.q_prescaled) {
            float qv[8];
            unpack8<T>(qf[x][s], qv);
#pragma unroll
            for (int e = 0; e < 8; ++e) qv[e] *= p.scale * 1.4426950408889634f;
            qf[x][s] = pack8<T>(qv);
        }
    }
    float dg[QB];                                              // self-attention: score against the query's own key (part of the reference sample)
#pragma unroll
    for (int x = 0; x < QB; ++x) dg[x] = attn40_diag_score<T>(p, Kp, qf[x], q0 + 32 * x + r, h);
    // ---- constant regions (written once): 1.0 everywhere; the tile buffers start as zeros (rows of a ragged tile that are never loaded stay finite)
    {
        const unsigned one2 = (unsigned)T::from_f32(1.0f) * 0x10001u;
        const uint4 ones = make_uint4(one2, one2, one2, one2), zero = make_uint4(0, 0, 0, 0);
        // (two loops per region, not `i < TILE / 16 ? zero : ones`: hipcc turned that select into a 32-byte SCRATCH array indexed by the condition -- every thread of the
        //  launch wrote and re-read it: 0.12 GB of HBM writes per level-0 launch, 29 % of the kernel's WRITE_SIZE; found in round 6 through private_segment_fixed_size = 48)
        for (int i = t; i < TILE / 16; i += NT) { *(uint4*)(dK0 + i * 16) = zero; *(uint4*)(dK1 + i * 16) = zero; *(uint4*)(dV0 + i * 16) = zero; *(uint4*)(dV1 + i * 16) = zero; }
        for (int i = TILE / 16 + t; i < (TILE + KONES) / 16; i += NT) { *(uint4*)(dK0 + i * 16) = ones; *(uint4*)(dK1 + i * 16) = ones; }
        for (int i = TILE / 16 + t; i < (TILE + VONES) / 16; i += NT) { *(uint4*)(dV0 + i * 16) = ones; *(uint4*)(dV1 + i * 16) = ones; }
    }
    __syncthreads();

    // ---- this wave's DMA pieces: piece j = wave + NW * i; slot = (j % 5) * 64 + lane = row * 5 + chunk of the K (j < 5) or V tile
    unsigned doff[PPW];          // byte offset of the slot's source inside a tile (K: key = row; V: key = 16 g + 4 (row & 3) + ((row >> 2) & 3))
    int dkey[PPW];               // key index inside the tile (ragged last tile)
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int j = wave + NW * i, slot = (j % 5) * 64 + lane, row = slot / NCH, ch = slot - row * NCH;
        const bool isv = j >= 5;
        const int key = isv ? (row & ~15) + 4 * (row & 3) + ((row >> 2) & 3) : row;
        dkey[i] = key;
        doff[i] = (unsigned)(key * (int)(isv ? p.v_rs : p.k_rs) + ch * 8) * 2u;
    }
    const int ntiles = (p.Nkv + KVT - 1) / KVT;
    const int nlast = p.Nkv - (ntiles - 1) * KVT;             // keys in the last tile (KVT unless RAGGED)
    const unsigned kstep = (unsigned)(KVT * (int)p.k_rs * 2), vstep = (unsigned)(KVT * (int)p.v_rs * 2);
    // K part (kind 0), V part (kind 1) of tile `it` -> LDS buffer at byte offset `dst`; stride > 1: the 64-key SAMPLE (keys 0, stride, 2 stride, ..)
    auto dma = [&](const int kind, const int it, unsigned char* dst, const unsigned stride) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int j = wave + NW * i;                          // wave-uniform
            if (j < NPC && (j >= 5) == (kind == 1)) {
                const unsigned char* src = kind ? Vp + (size_t)it * vstep : Kp + (size_t)it * kstep;
                unsigned off = doff[i];
                if (stride != 1) off += (unsigned)(dkey[i] * (int)p.k_rs * 2) * (stride - 1);
                if (RAGGED && it + 1 == ntiles && stride == 1) { if (dkey[i] < nlast) glds16(src + off, dst + (j % 5) * 1024); }
                else glds16(src + off, dst + (j % 5) * 1024);
            }
        }
    };

    // ---- lane-constant LDS read addresses (byte offsets into a tile buffer; the key block / k step are instruction immediates)
    const int ka0 = r * PR + 16 * h;                             // K chunks h (s = 0) and 2 + h (s = 1: + 32)
    const int ka2 = h ? TILE : r * PR + 64;                     // s = 2: chunk 4 for h = 0, the constant chunk (1.0: slots 40..47) for h = 1
    // transposed V read: 16-lane group g = lane >> 4 (g & 1 = cb: which 16 columns, g >> 1 = h); lane 4 q + pp of the group addresses row q, columns 4 pp ..
    const int vq = (lane >> 2) & 3, vpp = lane & 3, vcb = (lane >> 4) & 1;
    const int va0 = (4 * vq + h) * PR + (16 * vcb + 4 * vpp) * 2;             // d block 0; + (16 (2 kb + s2) + 2 j4) rows
    // d block 1 = columns 32..47 on the 16x16x32 form (k group g = lane >> 4 holds keys 32 kb + 16 (g & 1) + 4 (g >> 1) + {0..3, 8..11}): columns 32..39
    // are data (pp = 0, 1), 40..43 the constant (pp = 2: 1.0 -> O^T rows 40..43 = sum_k P), pp = 3 feeds rows that are never read (repeats pp = 1)
    const int va1 = vpp == 2 ? TILE : (16 * vcb + 4 * vq + h) * PR + 64 + 8 * (vpp & 1);

    // S^T = K Q^T for the 64 keys of the K tile at byte offset kb0 (2 key blocks x 3 k steps)
    auto qk = [&](const unsigned char* sK, const int x, f32x16 (&sacc)[2]) {      // (the second block's fragment reads are the first block's: CSE'd)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int i = 0; i < 16; ++i) sacc[kb][i] = 0.f;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const uint4 kf = s < 2 ? *(const uint4*)(sK + kb * 32 * PR + ka0 + 32 * s) : *(const uint4*)(sK + kb * 32 * PR + ka2);
                sacc[kb] = Mfma32<T>::run(kf, qf[x][s], sacc[kb]);
            }
        }
    };
    auto mask_last = [&](f32x16 (&sacc)[2]) {                     // keys past Nkv of the ragged last tile: -inf scores (P = 0)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (kb * 32 + (i & 3) + 8 * (i >> 2) + 4 * h >= nlast) sacc[kb][i] = -1e30f;
    };
    auto row_max = [&](const f32x16 (&sacc)[2]) {
        float mx = sacc[0][0];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) mx = fmaxf(mx, sacc[kb][i]);
        return fmaxf(mx, __shfl_xor(mx, 32));
    };
    auto set_reference = [&](const int x, const float target) {   // pad slots 40, 41 of Q <- -target as the nearest h16 hi + lo pair
        const unsigned short hi = T::from_f32(-target);
        const unsigned short lo = T::from_f32(-target - T::to_f32(hi));
        qf[x][2].x = h == 1 ? ((unsigned)hi | ((unsigned)lo << 16)) : qf[x][2].x;
    };

    f32x16 oacc[QB];          // O^T rows 0..31 per query block (32x32x16 layout: lane = query, registers + h = d)
    f32x4 o2[QB][2];          // O^T rows 32..47 for the queries 0..15 / 16..31 of each block (16x16x32 layout)
    float la[QB], lb[QB];
#pragma nounroll
    for (int attempt = 0; attempt < 2; ++attempt) {
        // ---- reference: attempt 0 = maximum over the 64-key sample + MARGIN; attempt 1 (after an overflow) = the exact maximum over all keys
        f32x16 sc[QB][2];
        if (attempt == 0) {
            dma(0, 0, dK1, (unsigned)(p.Nkv / KVT));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
#pragma unroll
            for (int x = 0; x < QB; ++x) { qk(dK1, x, sc[x]); set_reference(x, fmaxf(row_max(sc[x]), dg[x]) + MARGIN); }
        } else {
            float mx[QB];
#pragma unroll
            for (int x = 0; x < QB; ++x) { qf[x][2].x = h == 1 ? 0u : qf[x][2].x; mx[x] = -1e30f; }      // plain scores again
            __syncthreads();
            dma(0, 0, dK0, 1u);
            for (int it = 0; it < ntiles; ++it) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (it + 1 < ntiles) { if (it & 1) dma(0, it + 1, dK0, 1u); else dma(0, it + 1, dK1, 1u); }
#pragma unroll
                for (int x = 0; x < QB; ++x) {
                    if (it & 1) qk(dK1, x, sc[x]); else qk(dK0, x, sc[x]);
                    if (RAGGED && it + 1 == ntiles) mask_last(sc[x]);
                    mx[x] = fmaxf(mx[x], row_max(sc[x]));
                }
            }
#pragma unroll
            for (int x = 0; x < QB; ++x) set_reference(x, mx[x]);
        }
        __syncthreads();                                          // everybody is done with the sample (the last sweep tile)
        dma(0, 0, dK0, 1u);
        dma(1, 0, dV0, 1u);
#pragma unroll
        for (int x = 0; x < QB; ++x) {
#pragma unroll
            for (int i = 0; i < 16; ++i) oacc[x][i] = 0.f;
            o2[x][0] = f32x4{0.f, 0.f, 0.f, 0.f}; o2[x][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        // one tile: scores of both blocks, then per block  P = exp2(S) packed, O^T += V^T P^T   -- ONE basic block, no branch
        auto body = [&](const unsigned char* cK, const unsigned char* cV, const bool last) {
#pragma unroll
            for (int x = 0; x < QB; ++x) { qk(cK, x, sc[x]); if (RAGGED && last) mask_last(sc[x]); }
#pragma unroll
            for (int x = 0; x < QB; ++x) {
                uint4 pb[2][2];
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) sc[x][kb][i] = __builtin_amdgcn_exp2f(sc[x][kb][i]);
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2)
                        pb[kb][s2] = make_uint4(pack2<T>(sc[x][kb][8 * s2 + 0], sc[x][kb][8 * s2 + 1]), pack2<T>(sc[x][kb][8 * s2 + 2], sc[x][kb][8 * s2 + 3]),
                                                pack2<T>(sc[x][kb][8 * s2 + 4], sc[x][kb][8 * s2 + 5]), pack2<T>(sc[x][kb][8 * s2 + 6], sc[x][kb][8 * s2 + 7]));
                }
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        const int g0 = 16 * (2 * kb + s2) * PR;
                        const uint2 lo = ds_read_tr16(cV + g0 + va0), hi = ds_read_tr16(cV + g0 + 2 * PR + va0);
                        oacc[x] = Mfma32<T>::run(make_uint4(lo.x, lo.y, hi.x, hi.y), pb[kb][s2], oacc[x]);
                    }
                    uint4 pa, pq;
                    {
                        auto s0 = __builtin_amdgcn_permlane16_swap(pb[kb][0].x, pb[kb][1].x, false, false);
                        auto s1 = __builtin_amdgcn_permlane16_swap(pb[kb][0].y, pb[kb][1].y, false, false);
                        auto s2_ = __builtin_amdgcn_permlane16_swap(pb[kb][0].z, pb[kb][1].z, false, false);
                        auto s3 = __builtin_amdgcn_permlane16_swap(pb[kb][0].w, pb[kb][1].w, false, false);
                        pa = make_uint4(s0[0], s1[0], s2_[0], s3[0]);
                        pq = make_uint4(s0[1], s1[1], s2_[1], s3[1]);
                    }
                    const uint2 lo = ds_read_tr16(cV + kb * 32 * PR + va1), hi = ds_read_tr16(cV + kb * 32 * PR + 2 * PR + va1);
                    const uint4 vf = make_uint4(lo.x, lo.y, hi.x, hi.y);
                    o2[x][0] = T::mfma(vf, pa, o2[x][0]); o2[x][1] = T::mfma(vf, pq, o2[x][1]);
                }
            }
        };
        // step `it`: tile `it` (K and V) has landed; issue tile it+1 into the other buffers
        auto step = [&](const int it, unsigned char* cK, unsigned char* cV, unsigned char* nK, unsigned char* nV) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (it + 1 < ntiles) { dma(0, it + 1, nK, 1u); dma(1, it + 1, nV, 1u); }
            if (active) body(cK, cV, it + 1 == ntiles);
        };
        for (int it = 0; it < ntiles; it += 2) {
            step(it, dK0, dV0, dK1, dV1);
            if (it + 1 < ntiles) step(it + 1, dK1, dV1, dK0, dV0);
        }
        // ---- denominators (O^T row 40 = register 0 of the row-32.. tiles on the lanes 32..47); non-finite or zero: repeat with the exact maximum
        bool bad = false;
#pragma unroll
        for (int x = 0; x < QB; ++x) {
            la[x] = __shfl(o2[x][0][0], 32 + (lane & 15));
            lb[x] = __shfl(o2[x][1][0], 32 + (lane & 15));
            bad = bad || !(la[x] > 0.f && la[x] < 3.0e38f && lb[x] > 0.f && lb[x] < 3.0e38f);
        }
        if (attempt == 1 || !__syncthreads_or(active && bad ? 1 : 0)) break;
    }
    __syncthreads();      // (the second attempt leaves the loop without a barrier: every wave is done reading the tile buffers before one of them reuses its own below)
    if (!active) return;
    // ---- finalize: O[q][d] = O^T[d][q] / l
    // Head-major output (o_rs == D: the 64 queries of this wave are 64 * 80 contiguous bytes) and a full tile: the O tile goes through LDS -- this wave's own tile
    // buffer, dead since the loop's last barrier -- and leaves as five fully contiguous 1 KB stores (16 bytes per lane) instead of 8-byte pieces at an 80-byte
    // stride per instruction (round 6: the pieces left L2 as partial 64-byte bursts; WRITE_SIZE 1.40x the algorithmic bytes even in the head-major layout).
    if (p.o_rs == D && q0 + 64 <= p.Nq) {
        unsigned char* mine = wave == 0 ? dK0 : (wave == 1 ? dK1 : (wave == 2 ? dV0 : dV1));
        if (NW <= 4) {
#pragma unroll
            for (int x = 0; x < QB; ++x) {
                const float inv = 1.0f / (r < 16 ? la[x] : lb[x]);
                unsigned char* row = mine + (32 * x + r) * (D * 2);
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *(uint2*)(row + (8 * g + 4 * h) * 2) = make_uint2(pack2<T>(oacc[x][4 * g] * inv, oacc[x][4 * g + 1] * inv), pack2<T>(oacc[x][4 * g + 2] * inv, oacc[x][4 * g + 3] * inv));
#pragma unroll
                for (int qt2 = 0; qt2 < 2; ++qt2) {
                    const float inv2 = 1.0f / (qt2 ? lb[x] : la[x]);
                    if ((lane >> 4) < 2)
                        *(uint2*)(mine + (32 * x + 16 * qt2 + (lane & 15)) * (D * 2) + (32 + 4 * (lane >> 4)) * 2) =
                            make_uint2(pack2<T>(o2[x][qt2][0] * inv2, o2[x][qt2][1] * inv2), pack2<T>(o2[x][qt2][2] * inv2, o2[x][qt2][3] * inv2));
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            unsigned char* dst = (unsigned char*)(O + (int64_t)q0 * D);
#pragma unroll
            for (int it = 0; it < (64 * D * 2) / 1024; ++it) *(uint4*)(dst + (it * 64 + lane) * 16) = *(const uint4*)(mine + (it * 64 + lane) * 16);
            return;
        }
    }
#pragma unroll
    for (int x = 0; x < QB; ++x) {
        {
            const float inv = 1.0f / (r < 16 ? la[x] : lb[x]);
            const int q = q0 + 32 * x + r;
            if (q < p.Nq) {
                unsigned short* orow = O + (int64_t)q * p.o_rs;
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *(uint2*)(orow + 8 * g + 4 * h) = make_uint2(pack2<T>(oacc[x][4 * g] * inv, oacc[x][4 * g + 1] * inv), pack2<T>(oacc[x][4 * g + 2] * inv, oacc[x][4 * g + 3] * inv));
            }
        }
#pragma unroll
        for (int qt2 = 0; qt2 < 2; ++qt2) {
            const float inv = 1.0f / (qt2 ? lb[x] : la[x]);
            const int q = q0 + 32 * x + 16 * qt2 + (lane & 15);
            if (q < p.Nq && (lane >> 4) < 2)
                *(uint2*)(O + (int64_t)q * p.o_rs + 32 + 4 * (lane >> 4)) = make_uint2(pack2<T>(o2[x][qt2][0] * inv, o2[x][qt2][1] * inv), pack2<T>(o2[x][qt2][2] * inv, o2[x][qt2][3] * inv));
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// d = 80 (UNet level 1: 3600 tokens per frame at 720p; SAM 2's 72 -> 80 padded heads) in the form of attn40q2_kernel (round 4): S^T = K Q^T and
// O^T rows 0..63 on v_mfma_f32_32x32x16, rows 64..79 on 16x16x32 through v_permlane16_swap, 64 queries per wave, optimistic softmax reference,
// branch-free tile body.  What d = 80 changes:
//   * 80 = 5 k steps of 16 exactly: there is no pad slot for the reference to ride in.  It enters as the C OPERAND of the first MFMA of every score
//     block instead: a per-query-block register tile negm (all 16 registers of a lane = -m of the lane's query, fp32: no hi / lo split), read by the
//     out-of-place MFMA -- no instruction, no extra k step (a sixth k step would be +10 % matrix-pipe time).
//   * all 16 rows of the 16x16x32 block are data, so the softmax denominator does not come off the matrix pipe: l += dot2(P pair, (1, 1)) on the
//     packed, ROUNDED probabilities (v_dot2_f32_f16: 16 instructions per 32 queries x 64 keys) -- numerator and denominator see the same P.
//   * LDS images, 160-byte rows (10 chunks of 8 h16), all 20 LDS-DMA pieces of a tile with every lane active:
//       K: row = key, chunk c stored at position c ^ ((row >> 3) & 1)  -> the ds_read_b128 of 16 rows x one chunk is conflict free (tools/lds_bank_model.py);
//       V: key 8 A + 4 b + q stored at row 8 A + 2 q + b               -> the 4 rows of a transposed read are 2 apart: conflict free at 160 bytes
//          (the 16x16x32 block's reads are 2-way conflicted: 4 of 24 reads per tile).
// Waves whose queries all lie past Nq (the last query tile of N = 3600 holds 16 queries) keep feeding the DMA ring but skip the tile body.
template <typename T> __device__ __forceinline__ float psum2(unsigned u, float acc);
template <> __device__ __forceinline__ float psum2<F16>(unsigned u, float acc) {
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(vv_f16x2, u), (vv_f16x2){(_Float16)1.0f, (_Float16)1.0f}, acc, false);
}
template <> __device__ __forceinline__ float psum2<BF16>(unsigned u, float acc) {
    return acc + __builtin_bit_cast(float, u << 16) + __builtin_bit_cast(float, u & 0xffff0000u);
}

template <typename T, int NW, int OCC, bool RAGGED>
__global__ __launch_bounds__(NW * 64, OCC) void attn80_kernel(const vv_attn_params p, const int nqt) {
    constexpr int D = 80, KVT = 64, PR = 160, NCH = 10, QB = 2, NS = 5;
    constexpr int NT = NW * 64, BQ = NW * 32 * QB;
    constexpr int TILE = KVT * PR;                            // 10240
    constexpr int NPK = TILE / 1024;                          // 10 DMA pieces (1 KB each) per K tile, 10 per V tile
    constexpr int PPW = 2 * NPK / NW;                         // 5 pieces per wave and tile (piece j = wave + NW i: j < 10 = K, else V)
    static_assert(NW == 4, "piece distribution assumes four waves");
    constexpr float MARGIN = -4.0f;                           // P = 2^4 at the sample maximum (see attn40_kernel)
    __shared__ __attribute__((aligned(1024))) unsigned char dK0[TILE];
    __shared__ __attribute__((aligned(1024))) unsigned char dK1[TILE];
    __shared__ __attribute__((aligned(1024))) unsigned char dV0[TILE];
    __shared__ __attribute__((aligned(1024))) unsigned char dV1[TILE];

    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int r = lane & 31, h = lane >> 5;
    int qt, hd, b;
    {
        const int nbh = p.B * p.heads;
        const int full = (nbh / 8) * 8;
        const int bid = blockIdx.x;
        int bh;
        if (bid < full * nqt) { const int xcd = bid & 7, idx = bid >> 3; bh = (idx / nqt) * 8 + xcd; qt = idx % nqt; }
        else { const int rr = bid - full * nqt; bh = full + rr / nqt; qt = rr % nqt; }
        hd = bh % p.heads; b = bh / p.heads;
    }
    const unsigned short* Q = (const unsigned short*)p.q + (int64_t)b * p.q_bs + (int64_t)hd * (p.q_hs ? p.q_hs : D);
    const unsigned char* Kp = (const unsigned char*)((const unsigned short*)p.k + (int64_t)b * p.k_bs + (int64_t)hd * (p.k_hs ? p.k_hs : D));
    const unsigned char* Vp = (const unsigned char*)((const unsigned short*)p.v + (int64_t)b * p.v_bs + (int64_t)hd * (p.v_hs ? p.v_hs : D));
    unsigned short* O = (unsigned short*)p.o + (int64_t)b * p.o_bs + (int64_t)hd * (p.o_hs ? p.o_hs : D);

    // ---- Q fragments: lane (r, h) holds Q[q0 + 32 x + r][16 s + 8 h .. +7], s = 0..4
    const int q0 = qt * BQ + wave * 32 * QB;
    const bool active = q0 < p.Nq;                             // wave-uniform
    uint4 qf[QB][NS];
    float dg[QB];                                              // self-attention: score against the query's own key (part of the reference sample)
#pragma unroll
    for (int x = 0; x < QB; ++x) {
        const int q = q0 + 32 * x + r;
        float sum = 0.f;
        const bool on = p.Nq == p.Nkv && q < p.Nkv;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int d0 = 16 * s + 8 * h;
            qf[x][s] = q < p.Nq ? *(const uint4*)(Q + (int64_t)q * p.q_rs + d0) : make_uint4(0, 0, 0, 0);
            float qv[8];
            unpack8<T>(qf[x][s], qv);
            if (!p.q_prescaled) {
#pragma unroll
                for (int e = 0; e < 8; ++e) qv[e] *= p.scale * 1.4426950408889634f;
                qf[x][s] = pack8<T>(qv);
                unpack8<T>(qf[x][s], qv);
            }
            if (on) {
                float kv[8];
                unpack8<T>(*(const uint4*)(Kp + ((int64_t)q * p.k_rs + d0) * 2), kv);
#pragma unroll
                for (int e = 0; e < 8; ++e) sum += qv[e] * kv[e];
            }
        }
        sum += __shfl_xor(sum, 32);
        dg[x] = on ? sum : -1e30f;
    }
    // the tile buffers start as zeros (rows of a ragged tile that are never loaded stay finite)
    for (int i = t; i < TILE / 16; i += NT) {
        const uint4 zero = make_uint4(0, 0, 0, 0);
        *(uint4*)(dK0 + i * 16) = zero; *(uint4*)(dK1 + i * 16) = zero; *(uint4*)(dV0 + i * 16) = zero; *(uint4*)(dV1 + i * 16) = zero;
    }
    __syncthreads();

    // ---- this wave's DMA pieces: piece j = wave + NW i; slot = (j % 10) * 64 + lane = row * 10 + position of the K (j < 10) or V tile
    unsigned doff[PPW];          // byte offset of the slot's source inside a tile = key * row stride + chunk * 16 (chunk * 16 < row stride: the key is doff / row stride)
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int j = wave + NW * i, slot = (j % NPK) * 64 + lane, row = slot / NCH, pos = slot - row * NCH;
        const bool isv = j >= NPK;
        const int key = isv ? (row & ~7) + 4 * (row & 1) + ((row >> 1) & 3) : row;
        const int ch = isv ? pos : pos ^ ((row >> 3) & 1);
        doff[i] = (unsigned)(key * (int)(isv ? p.v_rs : p.k_rs) + ch * 8) * 2u;
    }
    const int ntiles = (p.Nkv + KVT - 1) / KVT;
    const int nlast = p.Nkv - (ntiles - 1) * KVT;             // keys in the last tile (KVT unless RAGGED)
    const unsigned kstep = (unsigned)(KVT * (int)p.k_rs * 2), vstep = (unsigned)(KVT * (int)p.v_rs * 2);
    // K part (want & 1) / V part (want & 2) of tile `it` -> the buffers bK / bV; stride > 1: the 64-key SAMPLE (keys 0, stride, 2 stride, ..), K only
    auto dma = [&](const int want, const int it, unsigned char* bK, unsigned char* bV, const unsigned stride) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int j = wave + NW * i;                          // wave-uniform
            const bool isv = j >= NPK;
            if (isv ? (want & 2) != 0 : (want & 1) != 0) {
                const unsigned char* src = isv ? Vp + (size_t)it * vstep : Kp + (size_t)it * kstep;
                unsigned char* dst = (isv ? bV : bK) + (j % NPK) * 1024;
                unsigned off = doff[i];
                const unsigned rs2 = (unsigned)(isv ? p.v_rs : p.k_rs) * 2u;
                if (stride != 1) off += (off / rs2) * rs2 * (stride - 1);                            // (once per block)
                if (RAGGED && it + 1 == ntiles && stride == 1) { if (off < (unsigned)nlast * rs2) glds16(src + off, dst); }
                else glds16(src + off, dst);
            }
        }
    };

    // ---- lane-constant LDS read addresses (byte offsets into a tile buffer; key block / k step / d block are instruction immediates)
    const int ka = r * PR + 16 * (h ^ ((r >> 3) & 1));            // K chunk 2 s + h of row r sits at position 2 s + (h ^ bit 3 of r): + 32 s
    const int vq = (lane >> 2) & 3, vpp = lane & 3, vcb = (lane >> 4) & 1;
    const int va0 = (2 * vq + h) * PR + (16 * vcb + 4 * vpp) * 2;              // rows 0..31 of O^T; + 64: rows 32..63; + 16 (2 kb + s2) rows; hi: + 8 rows
    const int va1 = (16 * vcb + 2 * vq + h) * PR + 128 + 8 * vpp;             // rows 64..79 (16x16x32: k group (s2 = vcb, h)); + 32 kb rows; hi: + 8 rows

    // S^T block kb (32 keys) of query block x = K Q^T + C
    auto qk = [&](const unsigned char* sK, const int kb, const int x, const f32x16& c0) {
        f32x16 acc = Mfma32<T>::run(*(const uint4*)(sK + kb * 32 * PR + ka), qf[x][0], c0);
#pragma unroll
        for (int s = 1; s < NS; ++s) acc = Mfma32<T>::run(*(const uint4*)(sK + kb * 32 * PR + ka + 32 * s), qf[x][s], acc);
        return acc;
    };
    auto mask_last = [&](f32x16& sacc, const int kb) {            // keys past Nkv of the ragged last tile: -inf scores (P = 0)
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (kb * 32 + (i & 3) + 8 * (i >> 2) + 4 * h >= nlast) sacc[i] = -1e30f;
    };
    auto row_max = [&](const f32x16& sacc, float mx) {
#pragma unroll
        for (int i = 0; i < 16; ++i) mx = fmaxf(mx, sacc[i]);
        return mx;
    };
    auto splat = [&](const float v) { f32x16 c; 
#pragma unroll
        for (int i = 0; i < 16; ++i) c[i] = v;
        return c; };

    f32x16 negm[QB];          // -reference of the lane's query in every register: the C operand of the first score MFMA
    f32x16 oacc[QB][2];       // O^T rows 0..31, 32..63 per query block (32x32x16 layout: lane = query, registers + h = d)
    f32x4 o2[QB][2];          // O^T rows 64..79 for the queries 0..15 / 16..31 of each block (16x16x32 layout)
    float l[QB];              // softmax denominators (this lane's half of the keys until the final exchange)
    float ref[QB];
    const f32x16 zero16 = splat(0.f);
    // ---- reference: attempt 0 = maximum over the 64-key sample (and the own key) + MARGIN; attempt 1 (after an overflow) = the exact maximum
    auto reference = [&](const int attempt) {
        if (attempt == 0) {
            dma(1, 0, dK1, dV1, (unsigned)(p.Nkv / KVT));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
#pragma unroll
            for (int x = 0; x < QB; ++x) {
                float mx = dg[x];
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) mx = row_max(qk(dK1, kb, x, zero16), mx);
                ref[x] = fmaxf(mx, __shfl_xor(mx, 32)) + MARGIN;
            }
        } else {
            float mx[QB];
#pragma unroll
            for (int x = 0; x < QB; ++x) mx[x] = -1e30f;
            __syncthreads();
            dma(1, 0, dK0, dV0, 1u);
            for (int it = 0; it < ntiles; ++it) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (it + 1 < ntiles) { if (it & 1) dma(1, it + 1, dK0, dV0, 1u); else dma(1, it + 1, dK1, dV1, 1u); }
#pragma unroll
                for (int x = 0; x < QB; ++x)
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) {
                        f32x16 sc = (it & 1) ? qk(dK1, kb, x, zero16) : qk(dK0, kb, x, zero16);
                        if (RAGGED && it + 1 == ntiles) mask_last(sc, kb);
                        mx[x] = row_max(sc, mx[x]);
                    }
            }
#pragma unroll
            for (int x = 0; x < QB; ++x) ref[x] = fmaxf(mx[x], __shfl_xor(mx[x], 32));
        }
#pragma unroll
        for (int x = 0; x < QB; ++x) negm[x] = splat(-ref[x]);
    };
    // all key tiles with the reference in place; true = some denominator of the BLOCK is non-finite or zero
    auto tiles = [&]() {
        __syncthreads();                                          // everybody is done with the sample (the last sweep tile)
        dma(3, 0, dK0, dV0, 1u);
#pragma unroll
        for (int x = 0; x < QB; ++x) {
            oacc[x][0] = zero16; oacc[x][1] = zero16;
            o2[x][0] = f32x4{0.f, 0.f, 0.f, 0.f}; o2[x][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            l[x] = 0.f;
        }
        // one tile, per 32-key block: scores of both query blocks, then per query block P = exp2(S) packed, l += sum P, O^T += V^T P^T -- ONE basic block
        auto body = [&](const unsigned char* cK, const unsigned char* cV, const bool last) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                f32x16 sc[QB];
#pragma unroll
                for (int x = 0; x < QB; ++x) {
                    sc[x] = qk(cK, kb, x, negm[x]);
                    if (RAGGED && last) mask_last(sc[x], kb);
                }
                uint4 pb[QB][2];
#pragma unroll
                for (int x = 0; x < QB; ++x) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) sc[x][i] = __builtin_amdgcn_exp2f(sc[x][i]);
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        pb[x][s2] = make_uint4(pack2<T>(sc[x][8 * s2 + 0], sc[x][8 * s2 + 1]), pack2<T>(sc[x][8 * s2 + 2], sc[x][8 * s2 + 3]),
                                               pack2<T>(sc[x][8 * s2 + 4], sc[x][8 * s2 + 5]), pack2<T>(sc[x][8 * s2 + 6], sc[x][8 * s2 + 7]));
                        l[x] = psum2<T>(pb[x][s2].x, l[x]); l[x] = psum2<T>(pb[x][s2].y, l[x]);      // (a second chain costs the ragged instantiation two spills in the loop)
                        l[x] = psum2<T>(pb[x][s2].z, l[x]); l[x] = psum2<T>(pb[x][s2].w, l[x]);
                    }
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    // rows 0..63 of O^T: keys 32 kb + 16 s2 + 8 j4 + 4 h + q, j4 = 0 (elements 0..3), 1 (elements 4..7): V rows 16 (2 kb + s2) + 8 j4 + 2 q + h
                    const int g0 = 16 * (2 * kb + s2) * PR;
#pragma unroll
                    for (int db = 0; db < 2; ++db) {
                        const uint2 lo = ds_read_tr16(cV + g0 + va0 + 64 * db), hi = ds_read_tr16(cV + g0 + 8 * PR + va0 + 64 * db);
                        const uint4 vf = make_uint4(lo.x, lo.y, hi.x, hi.y);
#pragma unroll
                        for (int x = 0; x < QB; ++x) oacc[x][db] = Mfma32<T>::run(vf, pb[x][s2], oacc[x][db]);
                    }
                }
                // rows 64..79 on the 16x16x32 form: v_permlane16_swap turns the (s2 = 0, s2 = 1) dword pairs of P into the B operands of the two
                // 16-query tiles (k group g = lane >> 4 = (s2 = g & 1, h = g >> 1): keys 32 kb + 16 s2 + 4 h + {0..3, 8..11}), as in attn40_kernel
                const uint2 lo = ds_read_tr16(cV + kb * 32 * PR + va1), hi = ds_read_tr16(cV + kb * 32 * PR + 8 * PR + va1);
                const uint4 vf = make_uint4(lo.x, lo.y, hi.x, hi.y);
#pragma unroll
                for (int x = 0; x < QB; ++x) {
                    auto s0 = __builtin_amdgcn_permlane16_swap(pb[x][0].x, pb[x][1].x, false, false);
                    auto s1 = __builtin_amdgcn_permlane16_swap(pb[x][0].y, pb[x][1].y, false, false);
                    auto s2_ = __builtin_amdgcn_permlane16_swap(pb[x][0].z, pb[x][1].z, false, false);
                    auto s3 = __builtin_amdgcn_permlane16_swap(pb[x][0].w, pb[x][1].w, false, false);
                    const uint4 pa = make_uint4(s0[0], s1[0], s2_[0], s3[0]);
                    const uint4 pq = make_uint4(s0[1], s1[1], s2_[1], s3[1]);
                    o2[x][0] = T::mfma(vf, pa, o2[x][0]); o2[x][1] = T::mfma(vf, pq, o2[x][1]);
                }
            }
        };
        // step `it`: tile `it` (K and V) has landed; issue tile it+1 into the other buffers.  `last` (the ragged tile: masked scores) is a compile-time
        // property of the call site, so every instance of the tile body stays ONE basic block
        auto step = [&](const int it, unsigned char* cK, unsigned char* cV, unsigned char* nK, unsigned char* nV, auto lastc) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (it + 1 < ntiles) dma(3, it + 1, nK, nV, 1u);
            if (active) body(cK, cV, decltype(lastc)::value);
        };
        const int nplain = RAGGED ? ntiles - 1 : ntiles;
        int it = 0;
        for (; it + 1 < nplain; it += 2) {
            step(it, dK0, dV0, dK1, dV1, std::false_type{});
            step(it + 1, dK1, dV1, dK0, dV0, std::false_type{});
        }
        if (it < nplain) { step(it, dK0, dV0, dK1, dV1, std::false_type{}); ++it; }
        if (RAGGED) {
            if (it & 1) step(it, dK1, dV1, dK0, dV0, std::true_type{});
            else step(it, dK0, dV0, dK1, dV1, std::true_type{});
        }
        // ---- denominators: both key halves; non-finite (some P overflowed) or zero: the block repeats its keys with the exact maximum
        bool bad = false;
#pragma unroll
        for (int x = 0; x < QB; ++x) {
            l[x] += __shfl_xor(l[x], 32);
            bad = bad || !(l[x] > 0.f && l[x] < 3.0e38f);
        }
        return __syncthreads_or(active && bad ? 1 : 0) != 0;
    };
    reference(0);
    if (tiles()) { reference(1); tiles(); }
    if (!active) return;
    // ---- finalize: O[q][d] = O^T[d][q] / l
#pragma unroll
    for (int x = 0; x < QB; ++x) {
        {
            const float inv = 1.0f / l[x];
            const int q = q0 + 32 * x + r;
            if (q < p.Nq) {
                unsigned short* orow = O + (int64_t)q * p.o_rs;
#pragma unroll
                for (int db = 0; db < 2; ++db)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        *(uint2*)(orow + 32 * db + 8 * g + 4 * h) = make_uint2(pack2<T>(oacc[x][db][4 * g] * inv, oacc[x][db][4 * g + 1] * inv),
                                                                              pack2<T>(oacc[x][db][4 * g + 2] * inv, oacc[x][db][4 * g + 3] * inv));
                if (p.lse && h == 0) p.lse[((int64_t)b * p.heads + hd) * p.Nq + q] = ref[x] + __log2f(l[x]);
            }
        }
        // rows 64..79: 16-query tile qt2, lane (c = lane & 15, g = lane >> 4): d = 64 + 4 g + reg of query 16 qt2 + c
#pragma unroll
        for (int qt2 = 0; qt2 < 2; ++qt2) {
            const float inv = 1.0f / __shfl(l[x], 16 * qt2 + (lane & 15));
            const int q = q0 + 32 * x + 16 * qt2 + (lane & 15);
            if (q < p.Nq)
                *(uint2*)(O + (int64_t)q * p.o_rs + 64 + 4 * (lane >> 4)) = make_uint2(pack2<T>(o2[x][qt2][0] * inv, o2[x][qt2][1] * inv), pack2<T>(o2[x][qt2][2] * inv, o2[x][qt2][3] * inv));
        }
    }
}

template <typename T, int NW, int OCC>
int attn80_launch(const vv_attn_params& p, hipStream_t st) {
    constexpr int BQ = NW * 64;
    const int nqt = (p.Nq + BQ - 1) / BQ;
    const int64_t nblk = (int64_t)p.B * p.heads * nqt;
    if (nblk > 0x7fffffff) VV_FAIL(VV_E_ARG, "vv_attention: grid too large");
    if (p.Nkv % 64) hipLaunchKernelGGL((attn80_kernel<T, NW, OCC, true>), dim3((unsigned)nblk), dim3(NW * 64), 0, st, p, nqt);
    else hipLaunchKernelGGL((attn80_kernel<T, NW, OCC, false>), dim3((unsigned)nblk), dim3(NW * 64), 0, st, p, nqt);
    VV_CHECK_LAUNCH("vv_attention(d80, 64 queries per wave)");
    return VV_OK;
}

template <typename T, int NW, int OCC>
int attn40q2_launch(const vv_attn_params& p, hipStream_t st) {
    constexpr int BQ = NW * 64;
    const int nqt = (p.Nq + BQ - 1) / BQ;
    const int64_t nblk = (int64_t)p.B * p.heads * nqt;
    if (nblk > 0x7fffffff) VV_FAIL(VV_E_ARG, "vv_attention: grid too large");
    if (p.Nkv % 64) hipLaunchKernelGGL((attn40q2_kernel<T, NW, OCC, true>), dim3((unsigned)nblk), dim3(NW * 64), 0, st, p, nqt);
    else hipLaunchKernelGGL((attn40q2_kernel<T, NW, OCC, false>), dim3((unsigned)nblk), dim3(NW * 64), 0, st, p, nqt);
    VV_CHECK_LAUNCH("vv_attention(d40, 64 queries per wave)");
    return VV_OK;
}

template <typename T, int NW, int OCC>
int attn40_launch(const vv_attn_params& p, hipStream_t st) {
    constexpr int BQ = NW * 32;
    const int nqt = (p.Nq + BQ - 1) / BQ;
    const int64_t nblk = (int64_t)p.B * p.heads * nqt;
    if (nblk > 0x7fffffff) VV_FAIL(VV_E_ARG, "vv_attention: grid too large");
    if (p.Nkv % 64) hipLaunchKernelGGL((attn40_kernel<T, NW, OCC, true>), dim3((unsigned)nblk), dim3(NW * 64), 0, st, p, nqt);
    else hipLaunchKernelGGL((attn40_kernel<T, NW, OCC, false>), dim3((unsigned)nblk), dim3(NW * 64), 0, st, p, nqt);
    VV_CHECK_LAUNCH("vv_attention(d40, 32x32x16)");
    return VV_OK;
}

}  // namespace

// the shapes these kernels take (anything else: -1000, the caller falls back to the 16x16x32 kernels): self-attention (or any Nq / Nkv that is not the
// 77-key cross-attention shape) over at least 64 keys, more than 32 queries; d = 40: 64 queries per wave (2 waves/SIMD) on long sequences, 32 (3 waves/SIMD)
// below 1024; d = 80: from 512 queries
extern "C" int vv_attention_mfma32(const vv_attn_params* pp, int dtype, void* stream) {
    const vv_attn_params& p = *pp;
    hipStream_t st = (hipStream_t)stream;
    const bool cross = p.Nkv < 128 && p.Nq != p.Nkv;
    if ((p.Nq <= 32 && p.Nkv <= 32) || cross || p.Nkv < 64) return -1000;
    if (p.D == 40) {
        if (dtype == VV_BF16) return p.Nq >= 1024 ? attn40q2_launch<BF16, 4, 2>(p, st) : attn40_launch<BF16, 4, 3>(p, st);
        return p.Nq >= 1024 ? attn40q2_launch<F16, 4, 2>(p, st) : attn40_launch<F16, 4, 3>(p, st);
    }
    if (p.D == 80 && p.Nq >= 512) return dtype == VV_BF16 ? attn80_launch<BF16, 4, 2>(p, st) : attn80_launch<F16, 4, 2>(p, st);
    return -1000;
}
