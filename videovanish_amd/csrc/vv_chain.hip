// Fused tail of a spatial Transformer2D block at C = 320 (the level-0 blocks of the UNet and of BrushNet: 14 per denoise step): everything after
// the self-attention core is per TOKEN, so it runs as ONE kernel with the token's row in registers -- the design of vv_motion.hip:
//   t   = t_in + Wo1 o + bo1                              (attn1 output projection + residual; o = the attention core's h16 output)
//   t  += Wo2 CrossAttn(LN2(t), text K/V) + bo2           (attn2: 77 constant text keys per head; K_h and V_h^T are part of the weight stream)
//   t  += W2 GEGLU(W1 LN3(t) + b1) + b2                   (feed-forward)
//   out = x + [res1 +] Wout t + bout                      (proj_out + the block's residual)
// The unfused path (nn.SpatialTransformer) runs these as 9 launches around 8 fp32 / h16 [M, 320 .. 2560] intermediates in HBM: at K = 320 every one of
// those GEMMs is bandwidth bound (AI 53-64 FLOP/B with the fp32 trunk read and written around each layer).
// One wave owns 32 consecutive tokens end to end (fp32 trunk in 160 registers, h16 activations in 80 as MFMA B fragments); a block = 4 waves = 128
// tokens shares the weight stream: 462 pre-swizzled [64 x 64] h16 slabs (packing.pack_chain_stream) through the 10-slot LDS ring by LDS-DMA.
// Fragment conventions, PERM32 and the ring protocol: vv_motion.hip.
#include <type_traits>
#include "vv_common.h"

namespace {

constexpr int CC = 320, CH = 8, CD = 40, NKEY = 77;      // CD: head dim
constexpr int NSLOT = 10, AHEAD = 6, SLAB = 8192;
// fp32 parameter block (floats): offsets
constexpr int Q_BO1 = 0, Q_LN2G = 320, Q_LN2B = 640, Q_BO2 = 960, Q_LN3G = 1280, Q_LN3B = 1600, Q_B1 = 1920, Q_B2 = 4480, Q_BOUT = 4800, Q_TOTAL = 5120;
constexpr int N_SLABS = 25 + CH * (5 + 2 + 2 + 5) + 20 * 15 + 25;      // 462

__device__ __forceinline__ void glds16_asm(const void* gptr, void* lds_wave_base) {
    typedef void __attribute__((address_space(3))) * lp_t;
    const unsigned dst = (unsigned)(size_t)(lp_t)lds_wave_base;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gptr), "s"(dst) : "memory");
}

// two GELUs at once on packed fp32 math (exact erf GELU through Abramowitz-Stegun 7.1.26; see vv_motion.hip)
__device__ __forceinline__ vv_f32x2 gelu2(vv_f32x2 x) {
#ifdef VV_CHAIN_PROBE_NOGELU      // timing probe only (wrong results): what the activation's VALU work costs
    return x;
#endif
    const vv_f32x2 ax = {fabsf(x.x), fabsf(x.y)};
    const vv_f32x2 z = ax * 0.70710678118654752f;
    const vv_f32x2 d = __builtin_elementwise_fma(z, (vv_f32x2){0.3275911f, 0.3275911f}, (vv_f32x2){1.0f, 1.0f});
    const vv_f32x2 t = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    vv_f32x2 q = __builtin_elementwise_fma(t, (vv_f32x2){1.061405429f, 1.061405429f}, (vv_f32x2){-1.453152027f, -1.453152027f});
    q = __builtin_elementwise_fma(q, t, (vv_f32x2){1.421413741f, 1.421413741f});
    q = __builtin_elementwise_fma(q, t, (vv_f32x2){-0.284496736f, -0.284496736f});
    q = __builtin_elementwise_fma(q, t, (vv_f32x2){0.254829592f, 0.254829592f});
    q = q * t;
    const vv_f32x2 ez = z * z * -1.4426950408889634f;
    const vv_f32x2 e = {__builtin_amdgcn_exp2f(ez.x), __builtin_amdgcn_exp2f(ez.y)};
    const vv_f32x2 erfc = q * e;
    return __builtin_elementwise_fma(ax * 0.5f, (vv_f32x2){1.0f, 1.0f} - erfc, x * 0.5f);
}

// TT = token tiles (of 16) per wave.  TT = 2: the round-3 form, 4 waves x 32 tokens, one wave per SIMD with the whole 512-entry register file
// (trunk in AGPRs: every VALU touch of it pays v_accvgpr_read / write, and nothing runs on the SIMD while the wave's GELU / LayerNorm / softmax do).
// TT = 1 (round 5, default): 8 waves x 16 tokens, two waves per SIMD at <= 256 registers each (all architectural: no AGPR copies); waves 4..7 --
// the SIMD partners of 0..3 -- run LAG slab PAIRS behind the others through the same ring, so one partner's VALU phase (GELU of an FF chunk, a
// LayerNorm, the cross-attention softmax) falls under the other's MFMAs instead of both stalling the matrix pipe together (the waves of a block
// meet at one barrier per slab pair: without the lag the two partners run in lockstep).  Ring: NS slots; a slot is re-filled AHEAD slabs ahead of the
// leaders, and the laggards may still have fragment reads of pair b - LAG - 1 in flight when pair b is being synchronised: NS >= AHEAD + 4 + 2 LAG.
template <typename T, int TT, int LAG, int AH>
__global__ __launch_bounds__(128 / (16 * TT) * 64, TT == 2 ? 1 : 2) void chain_c320_kernel(const vv_chain_params p) {
    constexpr int NW = 128 / (16 * TT), NS = AH + 4 + 2 * LAG, NPIECE = 8 / NW;       // waves, ring slots, 1 KB LDS-DMA pieces per wave and slab
    __shared__ __attribute__((aligned(1024))) unsigned char ring[NS * SLAB];
    __shared__ __attribute__((aligned(16))) float prm[Q_TOTAL];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int64_t row0 = (int64_t)blockIdx.x * 128 + wave * (16 * TT);

    for (int i = tid * 4; i < Q_TOTAL; i += NW * 64 * 4) *(float4*)(prm + i) = *(const float4*)(p.params + i);
    // ---- weight stream: slab s at p.stream + s * SLAB; each wave copies 8 / NW KB of every slab
    const unsigned char* sbase = (const unsigned char*)p.stream + (wave * NPIECE) * 1024 + lane * 16;
    int issued = 0, consumed = 0;
    auto issue = [&]() {
        unsigned char* dst = ring + (issued % NS) * SLAB + (wave * NPIECE) * 1024;
        const unsigned char* src = sbase + (int64_t)issued * SLAB;
#ifndef VV_PROBE_NODMA
        glds16_asm(src, dst);
        if constexpr (NPIECE == 2) glds16_asm(src + 1024, dst + 1024);
#else
        asm volatile("" :: "v"(src), "v"(dst));
#endif
        ++issued;
    };
    auto wait_landed = [&]() {      // all but the newest AHEAD slabs of this wave's share have landed
        static_assert(AH * NPIECE == 6 || AH * NPIECE == 8 || AH * NPIECE == 10 || AH * NPIECE == 12, "vmcnt literal");
        if constexpr (AH * NPIECE == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if constexpr (AH * NPIECE == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else if constexpr (AH * NPIECE == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    };
    // next slab of the stream.  Slabs are synchronised in PAIRS (vv_motion.hip): the EVEN slab of a pair issues two more slabs, waits until all but
    // the newest AHEAD have landed (this wave's share) and joins the barrier; the odd one just advances.  The parity of every slab's stream index is
    // a compile-time property of the call site (`even_tag`): a run-time test would cut the instruction stream into one basic block per slab
    // and the fragment reads of slab i+1 could not be scheduled under the MFMAs of slab i.
    // `tail_tag`: only the last group of the stream (proj_out) can run out of slabs to issue; everywhere else the issue is unconditional (no branch).
    auto next_slab = [&](auto even_tag, auto tail_tag) -> const unsigned char* {
        if constexpr (decltype(even_tag)::value) {
            if (!decltype(tail_tag)::value || issued < N_SLABS) { issue(); issue(); wait_landed(); }
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifndef VV_PROBE_NOBARRIER
            __builtin_amdgcn_s_barrier();
#endif
        }
        const unsigned char* s = ring + (consumed % NS) * SLAB;
        ++consumed;
        return s;
    };

    // ---- inputs: a = o (h16, fragments in PERM32 k order), t = t_in (fp32 trunk).  Rows past M repeat row M - 1 (never stored)
    uint4 a[10][TT];
    f32x4 t[20][TT];
    {
        __syncthreads();       // parameter block visible
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
            int64_t row = row0 + tt * 16 + li;
            if (row >= p.M) row = p.M - 1;
            const unsigned short* orow = (const unsigned short*)p.o + row * CC;
            const float* trow = p.t_in + row * CC;
#pragma unroll
            for (int s = 0; s < 10; ++s) {
                const uint2 lo = *(const uint2*)(orow + 32 * s + 4 * lg), hi = *(const uint2*)(orow + 32 * s + 16 + 4 * lg);
                a[s][tt] = make_uint4(lo.x, lo.y, hi.x, hi.y);
            }
#pragma unroll
            for (int j = 0; j < 20; ++j) {
                const float4 v = *(const float4*)(trow + 16 * j + 4 * lg);
                t[j][tt] = f32x4{v.x, v.y, v.z, v.w};
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll 1
        for (int i = 0; i < AH; ++i) issue();
        if (LAG > 0 && wave >= NW / 2) {      // the laggards pass LAG synchronisation steps without consuming: from here on they run 2 LAG slabs behind
#pragma unroll 1
            for (int i = 0; i < LAG; ++i) { issue(); issue(); wait_landed(); __builtin_amdgcn_s_barrier(); }
        }
    }

    // D += W_slab * X^T for RT row tiles and KK k steps of one slab, split into the fragment reads (LDS -> registers) and the MFMAs so that a group
    // of slabs runs software pipelined: the reads of slab i+1 are in flight under the MFMAs of slab i.  (With ONE wave per SIMD and the four
    // waves of a block released by the same barrier, reads that are waited for right before their MFMAs leave the matrix pipe idle for the whole
    // LDS round trip -- 8 KB per wave, all four waves at once -- on every slab.)
    struct WF { uint4 w[2][4]; };
    auto slab_load = [&](const unsigned char* s, auto rt_tag, auto kk_tag, WF& f) {
        constexpr int RT = decltype(rt_tag)::value, KK = decltype(kk_tag)::value;
        const int sw = li & 7;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            const int off = ((kk * 4 + lg) ^ sw) << 4;
#pragma unroll
#ifndef VV_PROBE_NOLDS
            for (int rt = 0; rt < RT; ++rt) f.w[kk][rt] = *(const uint4*)(s + (rt * 16 + li) * 128 + off);
#else
            for (int rt = 0; rt < RT; ++rt) { f.w[kk][rt] = make_uint4(off + rt, (unsigned)(size_t)s, kk, rt); asm volatile("" : "+v"(f.w[kk][rt].x)); }
#endif
        }
    };
    auto slab_fma = [&](const WF& f, auto rt_tag, auto kk_tag, f32x4* acc /* [RT][TT] */, const uint4 (&x0)[TT], const uint4 (&x1)[TT]) {
        constexpr int RT = decltype(rt_tag)::value, KK = decltype(kk_tag)::value;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int tt = 0; tt < TT; ++tt) acc[rt * TT + tt] = T::mfma(f.w[kk][rt], kk ? x1[tt] : x0[tt], acc[rt * TT + tt]);
    };
    using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    using I4 = std::integral_constant<int, 4>;
    using EVEN = std::true_type; using ODD = std::false_type;
    // a group of N slabs of the same shape whose first slab has stream-index parity P0 (0 = even): acc_of(i) / x0_of(i) / x1_of(i) name the
    // accumulator tile block and the operand k steps of slab i
    auto slab_group = [&](auto p0_tag, auto n_tag, auto rt_tag, auto kk_tag, auto&& acc_of, auto&& x0_of, auto&& x1_of, auto tail) {
        constexpr int P0 = decltype(p0_tag)::value, N = decltype(n_tag)::value, RT = decltype(rt_tag)::value, KK = decltype(kk_tag)::value;
        WF f[2];
        slab_load(next_slab(std::bool_constant<P0 == 0>{}, tail), rt_tag, kk_tag, f[0]);
#pragma unroll
        for (int i = 0; i < N; ++i) {
            if (i + 1 < N) {
                if (((P0 + i + 1) & 1) == 0) slab_load(next_slab(EVEN{}, tail), rt_tag, kk_tag, f[(i + 1) & 1]);
                else slab_load(next_slab(ODD{}, tail), rt_tag, kk_tag, f[(i + 1) & 1]);
            }
            slab_fma(f[i & 1], rt_tag, kk_tag, acc_of(i), x0_of(i), x1_of(i));
            if (i + 1 < N) __builtin_amdgcn_sched_group_barrier(0x100, KK * RT, 0);      // next slab's reads first ...
            __builtin_amdgcn_sched_group_barrier(0x008, TT * KK * RT, 0);                 // ... then this slab's MFMAs
#ifdef VV_CHAIN_PIN
            // ... and nothing crosses into the next slab's region: without this fence hipcc fills the MFMA group with the MFMAs of the slab whose reads it
            // has just issued (the group barriers order instruction TYPES, not instances), folds f[0] / f[1] into one register set and every slab waits
            // out its own LDS round trip (round 5: the ISA of rounds 3-4 was never software pipelined)
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
    };
    using P0E = std::integral_constant<int, 0>; using P0O = std::integral_constant<int, 1>;
    using N5 = std::integral_constant<int, 5>; using N10 = std::integral_constant<int, 10>; using N25 = std::integral_constant<int, 25>;
    using BODY = std::false_type; using TAIL = std::true_type;
    auto dense320 = [&](auto p0_tag, f32x4 (&acc)[20][TT], auto tail) {      // 5 row blocks x 5 k tiles
        slab_group(p0_tag, N25{}, I4{}, I2{}, [&](int i) { return &acc[(i / 5) * 4][0]; }, [&](int i) -> const uint4 (&)[TT] { return a[2 * (i % 5)]; },
                   [&](int i) -> const uint4 (&)[TT] { return a[2 * (i % 5) + 1]; }, tail);
    };
    auto frag = [&](const f32x4& lo, const f32x4& hi) -> uint4 {
        return make_uint4(pack2<T>(lo[0], lo[1]), pack2<T>(lo[2], lo[3]), pack2<T>(hi[0], hi[1]), pack2<T>(hi[2], hi[3]));
    };
    auto add_bias = [&](const int off) {
#pragma unroll
        for (int j = 0; j < 20; ++j) {
            const float4 b = *(const float4*)(prm + off + 16 * j + 4 * lg);
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) { t[j][tt][0] += b.x; t[j][tt][1] += b.y; t[j][tt][2] += b.z; t[j][tt][3] += b.w; }
        }
    };
    auto layer_norm = [&](const int goff, const int boff) {
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 20; ++j) s += (t[j][tt][0] + t[j][tt][1]) + (t[j][tt][2] + t[j][tt][3]);
            s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);
            const float mean = s * (1.0f / CC);
            float q = 0.f;
#pragma unroll
            for (int j = 0; j < 20; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float d = t[j][tt][r] - mean; q += d * d; }
            q += __shfl_xor(q, 16); q += __shfl_xor(q, 32);
            const float rstd = rsqrtf(q * (1.0f / CC) + 1e-5f);
#pragma unroll
            for (int s2 = 0; s2 < 10; ++s2) {
                f32x4 y[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int j = 2 * s2 + h, c = 16 * j + 4 * lg;
                    const float4 g = *(const float4*)(prm + goff + c), b = *(const float4*)(prm + boff + c);
                    y[h][0] = (t[j][tt][0] - mean) * rstd * g.x + b.x; y[h][1] = (t[j][tt][1] - mean) * rstd * g.y + b.y;
                    y[h][2] = (t[j][tt][2] - mean) * rstd * g.z + b.z; y[h][3] = (t[j][tt][3] - mean) * rstd * g.w + b.w;
                }
                a[s2][tt] = frag(y[0], y[1]);
            }
        }
    };

    // ---- attn1 output projection: t = t_in + Wo1 o + bo1        (stream slabs 0..24)
    dense320(P0E{}, t, BODY{});
    add_bias(Q_BO1);

    // ---- attn2: cross-attention to the 77 text keys.  Per head: q (5 slabs) | S^T = K_h q^T (2 slabs: key rows 0..63, 64..79) | softmax |
    //      O^T = V_h^T P^T (2 slabs: keys 0..63, 64..95) | t += Wo2[:, head] O (5 slabs)
    layer_norm(Q_LN2G, Q_LN2B);
    const float sc = 0.15811388300841897f * 1.4426950408889634f;      // 40^-1/2 * log2(e)
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int h = 0; h < CH; ++h) {
        f32x4 qa[3][TT];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) qa[i][tt] = z4;
        // (a head is 14 slabs and starts at an odd stream index: 25 + 14 h)
        slab_group(P0O{}, N5{}, I3{}, I2{}, [&](int) { return &qa[0][0]; }, [&](int i) -> const uint4 (&)[TT] { return a[2 * i]; },
                   [&](int i) -> const uint4 (&)[TT] { return a[2 * i + 1]; }, BODY{});
        uint4 q0[TT], q1[TT];                          // [token tile]: k steps 0 (d = PERM32) and 1 (d = 32 + 4 lg + e, e < 4)
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) { q0[tt] = frag(qa[0][tt], qa[1][tt]); q1[tt] = frag(qa[2][tt], z4); }
        f32x4 sT[5][TT];                              // [key tile][token tile]: lane = token li, registers = keys 16 kt + 4 lg + r
#pragma unroll
        for (int kt = 0; kt < 5; ++kt)
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) sT[kt][tt] = z4;
        {      // K_h: key rows 0..63 (4 tiles), then 64..79 (1 tile)
            WF f0, f1;
            slab_load(next_slab(EVEN{}, BODY{}), I4{}, I2{}, f0);
            slab_load(next_slab(ODD{}, BODY{}), I1{}, I2{}, f1);
            slab_fma(f0, I4{}, I2{}, &sT[0][0], q0, q1);
            slab_fma(f1, I1{}, I2{}, &sT[4][0], q0, q1);
            __builtin_amdgcn_sched_group_barrier(0x100, 10, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 10 * TT, 0);
        }
        uint4 pf[3][TT];                              // [k step of 32 keys][token tile]
        float inv[TT];
#pragma unroll
        for (int qt = 0; qt < TT; ++qt) {
            if (lg == 3) { sT[4][qt][1] = -1e30f; sT[4][qt][2] = -1e30f; sT[4][qt][3] = -1e30f; }      // keys 77, 78, 79 do not exist
            float m = sT[0][qt][0];
#pragma unroll
            for (int kt = 0; kt < 5; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) m = fmaxf(m, sT[kt][qt][r]);
            m = fmaxf(m, __shfl_xor(m, 16)); m = fmaxf(m, __shfl_xor(m, 32));
            const float mc = m * sc;
            float l = 0.f;
#pragma unroll
            for (int kt = 0; kt < 5; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float e = __builtin_amdgcn_exp2f(sT[kt][qt][r] * sc - mc); sT[kt][qt][r] = e; l += e; }
            l += __shfl_xor(l, 16); l += __shfl_xor(l, 32);
            inv[qt] = 1.0f / l;
            pf[0][qt] = frag(sT[0][qt], sT[1][qt]); pf[1][qt] = frag(sT[2][qt], sT[3][qt]); pf[2][qt] = frag(sT[4][qt], z4);
        }
        f32x4 oT[3][TT];                              // [d tile][token tile]
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) oT[i][tt] = z4;
        {      // V_h^T: keys 0..63 (2 k steps), then 64..95 (1 k step)
            WF f0, f1;
            slab_load(next_slab(EVEN{}, BODY{}), I3{}, I2{}, f0);
            slab_load(next_slab(ODD{}, BODY{}), I3{}, I1{}, f1);
            slab_fma(f0, I3{}, I2{}, &oT[0][0], pf[0], pf[1]);
            slab_fma(f1, I3{}, I1{}, &oT[0][0], pf[2], pf[2]);
            __builtin_amdgcn_sched_group_barrier(0x100, 9, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 9 * TT, 0);
        }
        uint4 o0[TT], o1[TT];
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
#pragma unroll
            for (int i = 0; i < 3; ++i) oT[i][tt] *= inv[tt];
            o0[tt] = frag(oT[0][tt], oT[1][tt]); o1[tt] = frag(oT[2][tt], z4);
        }
        slab_group(P0E{}, N5{}, I4{}, I2{}, [&](int i) { return &t[i * 4][0]; }, [&](int) -> const uint4 (&)[TT] { return o0; },
                   [&](int) -> const uint4 (&)[TT] { return o1; }, BODY{});
    }
    add_bias(Q_BO2);

    // ---- GEGLU feed-forward, 20 chunks of 64 hidden units: 10 slabs of W1 (value / gate rows interleaved per 16), 5 slabs of W2
    layer_norm(Q_LN3G, Q_LN3B);
    // (chunk c is 15 slabs and starts at stream index 137 + 15 c: odd for even c, even for odd c -> two chunks per loop iteration)
    auto ff_chunk = [&](const int c, auto p0_tag) {
        f32x4 g[8][TT];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) g[i][tt] = z4;
        slab_group(p0_tag, N10{}, I4{}, I2{}, [&](int i) { return &g[(i / 5) * 4][0]; }, [&](int i) -> const uint4 (&)[TT] { return a[2 * (i % 5)]; },
                   [&](int i) -> const uint4 (&)[TT] { return a[2 * (i % 5) + 1]; }, BODY{});
        uint4 hf0[TT], hf1[TT];
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
            f32x4 hv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 bv = *(const float4*)(prm + Q_B1 + c * 128 + (2 * i) * 16 + 4 * lg), bg = *(const float4*)(prm + Q_B1 + c * 128 + (2 * i + 1) * 16 + 4 * lg);
                const vv_f32x2 g01 = gelu2((vv_f32x2){g[2 * i + 1][tt][0] + bg.x, g[2 * i + 1][tt][1] + bg.y});
                const vv_f32x2 g23 = gelu2((vv_f32x2){g[2 * i + 1][tt][2] + bg.z, g[2 * i + 1][tt][3] + bg.w});
                hv[i][0] = (g[2 * i][tt][0] + bv.x) * g01.x; hv[i][1] = (g[2 * i][tt][1] + bv.y) * g01.y;
                hv[i][2] = (g[2 * i][tt][2] + bv.z) * g23.x; hv[i][3] = (g[2 * i][tt][3] + bv.w) * g23.y;
            }
            hf0[tt] = frag(hv[0], hv[1]); hf1[tt] = frag(hv[2], hv[3]);
        }
        slab_group(p0_tag, N5{}, I4{}, I2{}, [&](int i) { return &t[i * 4][0]; }, [&](int) -> const uint4 (&)[TT] { return hf0; },
                   [&](int) -> const uint4 (&)[TT] { return hf1; }, BODY{});
    };
#pragma unroll 1
    for (int c = 0; c < 20; c += 2) { ff_chunk(c, P0O{}); ff_chunk(c + 1, P0E{}); }
    add_bias(Q_B2);

    // ---- proj_out (+ bias + x [+ res1])
#pragma unroll
    for (int tt = 0; tt < TT; ++tt)
#pragma unroll
        for (int s2 = 0; s2 < 10; ++s2) a[s2][tt] = frag(t[2 * s2][tt], t[2 * s2 + 1][tt]);
#pragma unroll
    for (int j = 0; j < 20; ++j)
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) t[j][tt] = z4;
    dense320(P0O{}, t, TAIL{});          // stream slabs 437..461
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
        const int64_t r = row0 + tt * 16 + li;
        if (r < p.M) {
            const int64_t row = r * CC;
#pragma unroll
            for (int j = 0; j < 20; ++j) {
                const int c = 16 * j + 4 * lg;
                const float4 b = *(const float4*)(prm + Q_BOUT + c);
                const float4 xr = *(const float4*)(p.x + row + c);
                float v0 = t[j][tt][0] + b.x + xr.x, v1 = t[j][tt][1] + b.y + xr.y, v2 = t[j][tt][2] + b.z + xr.z, v3 = t[j][tt][3] + b.w + xr.w;
                if (p.res1) { const float4 r4 = *(const float4*)(p.res1 + row + c); v0 += r4.x; v1 += r4.y; v2 += r4.z; v3 += r4.w; }
                if (p.out_dtype == VV_F32) *(float4*)((float*)p.out + row + c) = make_float4(v0, v1, v2, v3);
                else *(uint2*)((unsigned short*)p.out + row + c) = make_uint2(pack2<T>(v0, v1), pack2<T>(v2, v3));
            }
        }
    }
}



// ---------------------------------------------------------------------------------------------------------------------------
// ROW-SPLIT form of the same tail (round 5).  What the counters said about the forms above (profiles/r5_chain_forms.txt): the kernel is not at the
// board's power limit (2.39 GHz), its LDS pipe is 16 % busy in the 4-wave form -- the matrix pipe idles because ONE wave per SIMD cannot overlap its
// own VALU phases (GELU, LayerNorm, softmax, the AGPR copies of a 512-register kernel) and waits (barrier, LDS round trips) with MFMAs; and the
// 8 x 16-token form needs one ds_read_b128 per 16-cycle MFMA per wave = exactly the LDS pipe's 256 B/clk when the matrix pipe is full.
// Here a block is still 128 tokens behind one weight ring, but its 8 waves are 4 token groups x 2 ROW HALVES: the two waves of a pair own the same
// 32 tokens and split every [64 x 64] slab by rows (wave hf reads row tiles 2 hf, 2 hf + 1: 4 fragment reads feed 8 MFMAs -- half the LDS bytes
// per MFMA of the 16-token form at two waves per SIMD and <= 256 architectural registers each).  A wave therefore holds the fp32 trunk of
// 160 channels x 32 tokens (80 registers; channel = 64 rb + 32 hf + 16 rt + 4 lg + r) and the FULL h16 activation row of its tokens as B fragments
// (80 registers), of which it produces only the k steps s = 2 kt + hf itself: whenever a layer ends (LayerNorm, GEGLU, the trunk before proj_out) the
// partners swap their halves through LDS, lane for lane (the fragment a lane needs from its partner is the one the same lane of the partner holds).
// LayerNorm statistics: each wave's (mean, M2) over its 160 channels, swapped and merged with Chan's formula.
// Cross-attention (77 text keys): per head the q projection, S^T, softmax and V^T P^T run on ONE token tile per wave (tile hf: the 16-token form, 9 of a
// head's 14 slabs), the head's O goes through LDS and its output projection is row-split again; stream order per head pair: q K V^T | q K V^T | Wo | Wo.
// Synchronisation: explicit steps (LDS-DMA issue of the slabs about to be consumed + counted vmcnt + one barrier) in front of every slab pair of a group.
constexpr int RS_NS = 10, RS_AH = 6, RS_XBUF = 40960;
#ifdef VV_PROBE_NOVMWAIT
#define VV_WAIT6 do {} while (0)
#else
#define VV_WAIT6 asm volatile("s_waitcnt vmcnt(6)" ::: "memory")
#endif

template <typename T>
__global__ __launch_bounds__(512, 2) void chain_rs_c320_kernel(const vv_chain_params p) {
    __shared__ __attribute__((aligned(1024))) unsigned char ring[RS_NS * SLAB];
    __shared__ __attribute__((aligned(16))) unsigned char xbuf[RS_XBUF];
    __shared__ __attribute__((aligned(16))) float sbuf[8 * 2 * 64 * 2];
    __shared__ __attribute__((aligned(16))) float prm[Q_TOTAL];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int grp = wave & 3, hf = wave >> 2, pw = wave ^ 4;
    const bool hi = hf != 0;
    const int64_t row0 = (int64_t)blockIdx.x * 128 + grp * 32;

    for (int i = tid * 4; i < Q_TOTAL; i += 512 * 4) *(float4*)(prm + i) = *(const float4*)(p.params + i);
    // ---- weight stream: every wave copies 1 KB of every slab
    const unsigned char* sbase = (const unsigned char*)p.stream + wave * 1024 + lane * 16;
    int issued = 0, islot = 0, cslot = 0;
    auto issue = [&]() {
#if defined(VV_PROBE_REGLOAD)       // timing probe (wrong results): the same VMEM instruction count and bytes as the DMA, into a sink register set instead of LDS
        asm volatile("global_load_dwordx4 a[0:3], %0, off" :: "v"(sbase + (int64_t)issued * SLAB) : "memory", "a0", "a1", "a2", "a3");
#elif defined(VV_PROBE_SAMESLAB)      // timing probe (wrong results): every DMA reads slab (issued & 7) -- the issue cost without the stream's memory side
        glds16_asm(sbase + (int64_t)(issued & 7) * SLAB, ring + islot * SLAB + wave * 1024);
#elif !defined(VV_PROBE_NODMA)
        glds16_asm(sbase + (int64_t)issued * SLAB, ring + islot * SLAB + wave * 1024);
#endif
        ++issued;
        islot = islot + 1 == RS_NS ? 0 : islot + 1;
    };
    using BODY = std::false_type; using TAIL = std::true_type;
    using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>; using I4 = std::integral_constant<int, 4>;
    // one synchronisation step in front of NI slabs: issue NI more, wait until all but the newest RS_AH have landed (this wave's share), meet.
    // XCH: the step also publishes LDS writes of this wave (an exchange): they have to be complete before the barrier.
    auto sync = [&](auto ni_tag, auto tail_tag, auto xch_tag) {
        constexpr int NI = decltype(ni_tag)::value;
        if constexpr (decltype(tail_tag)::value) {
            if (issued + NI <= N_SLABS) {
#pragma unroll
                for (int i = 0; i < NI; ++i) issue();
                VV_WAIT6;
            } else {
                while (issued < N_SLABS) issue();
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        } else {
#pragma unroll
            for (int i = 0; i < NI; ++i) issue();
            VV_WAIT6;
        }
        if constexpr (decltype(xch_tag)::value) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef VV_PROBE_NOBARRIER
        if constexpr (decltype(xch_tag)::value)
#endif
        __builtin_amdgcn_s_barrier();
    };
    using NOX = std::false_type; using XCH = std::true_type;
    auto slab = [&]() -> const unsigned char* {
        const unsigned char* s = ring + cslot * SLAB;
        cslot = cslot + 1 == RS_NS ? 0 : cslot + 1;
        return s;
    };
    auto meet = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); };      // exchange-only barrier (no slab)

    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    auto frag = [&](const f32x4& lo, const f32x4& hi_) -> uint4 {
        return make_uint4(pack2<T>(lo[0], lo[1]), pack2<T>(lo[2], lo[3]), pack2<T>(hi_[0], hi_[1]), pack2<T>(hi_[2], hi_[3]));
    };
    auto sel = [&](const uint4& a_, const uint4& b_) -> uint4 { return hi ? b_ : a_; };      // wave-uniform select

    // ---- row-split slab groups: this wave's two row tiles (2 hf, 2 hf + 1) of N [64 x 64] slabs
    struct WF2 { uint4 w[2][2]; };      // [kk][rt]
    const int rs_off = hf * 4096 + li * 128, sw = li & 7;
    auto load_rs = [&](const unsigned char* s, WF2& f) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int off = ((kk * 4 + lg) ^ sw) << 4;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) f.w[kk][rt] = *(const uint4*)(s + rs_off + rt * 2048 + off);
        }
    };
    auto fma_rs = [&](const WF2& f, f32x4* acc /* [2][2] = [rt][tt] */, const uint4 (&x0)[2], const uint4 (&x1)[2]) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) acc[rt * 2 + tt] = T::mfma(f.w[kk][rt], kk ? x1[tt] : x0[tt], acc[rt * 2 + tt]);
    };
    // N slabs, a step in front of every pair (PRE: the caller has already made the first one)
    auto group_rs = [&](auto n_tag, auto pre_tag, auto&& acc_of, auto&& x0_of, auto&& x1_of, auto tail) {
        constexpr int N = decltype(n_tag)::value;
        WF2 f[2];
        if constexpr (!decltype(pre_tag)::value) { if constexpr (N >= 2) sync(I2{}, tail, NOX{}); else sync(I1{}, tail, NOX{}); }
        load_rs(slab(), f[0]);
#pragma unroll
        for (int i = 0; i < N; ++i) {
            if (i + 1 < N) {
                if (((i + 1) & 1) == 0) { if (i + 2 < N) sync(I2{}, tail, NOX{}); else sync(I1{}, tail, NOX{}); }
                load_rs(slab(), f[(i + 1) & 1]);
            }
            fma_rs(f[i & 1], acc_of(i), x0_of(i), x1_of(i));
            if (i + 1 < N) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
#ifdef VV_CHAIN_PIN
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
    };
    using N5 = std::integral_constant<int, 5>; using N10 = std::integral_constant<int, 10>; using N25 = std::integral_constant<int, 25>;
    using NOPRE = std::false_type; using PRE = std::true_type;

    // ---- state: trunk t[2 rb + rt][tt] (own channels), activations a0[kt][tt] / a1[kt][tt] = k steps 2 kt / 2 kt + 1 of the full row
    f32x4 t[10][2];
    uint4 a0[5][2], a1[5][2];
    auto chan = [&](const int j) -> int { return 64 * (j >> 1) + 32 * hf + 16 * (j & 1) + 4 * lg; };      // first of the 4 channels of t[j][.][0..3]
    {
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            int64_t row = row0 + tt * 16 + li;
            if (row >= p.M) row = p.M - 1;
            const unsigned short* orow = (const unsigned short*)p.o + row * CC;
            const float* trow = p.t_in + row * CC;
#pragma unroll
            for (int kt = 0; kt < 5; ++kt) {
                const uint2 l0 = *(const uint2*)(orow + 64 * kt + 4 * lg), h0 = *(const uint2*)(orow + 64 * kt + 16 + 4 * lg);
                const uint2 l1 = *(const uint2*)(orow + 64 * kt + 32 + 4 * lg), h1 = *(const uint2*)(orow + 64 * kt + 48 + 4 * lg);
                a0[kt][tt] = make_uint4(l0.x, l0.y, h0.x, h0.y);
                a1[kt][tt] = make_uint4(l1.x, l1.y, h1.x, h1.y);
            }
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                const float4 v = *(const float4*)(trow + chan(j));
                t[j][tt] = f32x4{v.x, v.y, v.z, v.w};
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();       // parameter block visible
#pragma unroll 1
        for (int i = 0; i < RS_AH; ++i) issue();
    }
    auto add_bias = [&](const int off) {
#pragma unroll
        for (int j = 0; j < 10; ++j) {
            const float4 b = *(const float4*)(prm + off + chan(j));
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) { t[j][tt][0] += b.x; t[j][tt][1] += b.y; t[j][tt][2] += b.z; t[j][tt][3] += b.w; }
        }
    };
    auto dense320 = [&](f32x4 (&acc)[10][2], auto tail) {      // 5 row blocks x 5 k tiles
        group_rs(N25{}, NOPRE{}, [&](int i) { return &acc[(i / 5) * 2][0]; }, [&](int i) -> const uint4 (&)[2] { return a0[i % 5]; },
                 [&](int i) -> const uint4 (&)[2] { return a1[i % 5]; }, tail);
    };
    // own[rb][tt] = h16(LN(t) g + b) of this wave's channels = k step 2 rb + hf of the row; statistics merged with the partner's
    auto layer_norm = [&](const int goff, const int boff, uint4 (&own)[5][2]) {
        float mloc[2], m2loc[2];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 10; ++j) s += (t[j][tt][0] + t[j][tt][1]) + (t[j][tt][2] + t[j][tt][3]);
            s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);
            mloc[tt] = s * (1.0f / 160);
            float q = 0.f;
#pragma unroll
            for (int j = 0; j < 10; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float d = t[j][tt][r] - mloc[tt]; q += d * d; }
            q += __shfl_xor(q, 16); q += __shfl_xor(q, 32);
            m2loc[tt] = q;
        }
        *(float4*)(sbuf + (wave * 64 + lane) * 4) = make_float4(mloc[0], m2loc[0], mloc[1], m2loc[1]);
        meet();
        const float4 o4 = *(const float4*)(sbuf + (pw * 64 + lane) * 4);
        const float om[2] = {o4.x, o4.z}, oq[2] = {o4.y, o4.w};
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const float mean = 0.5f * (mloc[tt] + om[tt]), dm = mloc[tt] - om[tt];
            const float rstd = rsqrtf((m2loc[tt] + oq[tt] + 80.0f * dm * dm) * (1.0f / CC) + 1e-5f);
#pragma unroll
            for (int rb = 0; rb < 5; ++rb) {
                f32x4 y[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int j = 2 * rb + h, c = chan(j);
                    const float4 g = *(const float4*)(prm + goff + c), b = *(const float4*)(prm + boff + c);
                    y[h][0] = (t[j][tt][0] - mean) * rstd * g.x + b.x; y[h][1] = (t[j][tt][1] - mean) * rstd * g.y + b.y;
                    y[h][2] = (t[j][tt][2] - mean) * rstd * g.z + b.z; y[h][3] = (t[j][tt][3] - mean) * rstd * g.w + b.w;
                }
                own[rb][tt] = frag(y[0], y[1]);
            }
        }
    };
    unsigned char* const xmine = xbuf + wave * 5120 + lane * 16;
    const unsigned char* const xpart = xbuf + pw * 5120 + lane * 16;
    // full swap of the partners' halves: a0 / a1 <- (own, partner's) for both token tiles (two rounds through the 40 KB buffer)
    auto swap_full = [&](const uint4 (&own)[5][2]) {
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            if (tt) meet();      // the partner has read round 0
#pragma unroll
            for (int rb = 0; rb < 5; ++rb) *(uint4*)(xmine + rb * 1024) = own[rb][tt];
            meet();
#pragma unroll
            for (int rb = 0; rb < 5; ++rb) {
                const uint4 o = *(const uint4*)(xpart + rb * 1024);
                a0[rb][tt] = sel(own[rb][tt], o);
                a1[rb][tt] = sel(o, own[rb][tt]);
            }
        }
    };

    // ---- attn1 output projection: t = t_in + Wo1 o + bo1        (stream slabs 0..24)
    dense320(t, BODY{});
    add_bias(Q_BO1);

    // ---- attn2: cross-attention to the 77 text keys, on token tile hf of the pair (x0[kt] / x1[kt]: the tile's full row)
    {
        uint4 own[5][2];
        layer_norm(Q_LN2G, Q_LN2B, own);
        uint4 x0[5], x1[5];
        // the partner needs my k steps of ITS tile (1 - hf); I need its k steps of mine
#pragma unroll
        for (int rb = 0; rb < 5; ++rb) *(uint4*)(xmine + rb * 1024) = sel(own[rb][1], own[rb][0]);
        meet();
#pragma unroll
        for (int rb = 0; rb < 5; ++rb) {
            const uint4 o = *(const uint4*)(xpart + rb * 1024), m = sel(own[rb][0], own[rb][1]);
            x0[rb] = sel(m, o);
            x1[rb] = sel(o, m);
        }
        meet();      // everybody has read: the buffer is free for the heads' O
        const float sc = 0.15811388300841897f * 1.4426950408889634f;      // 40^-1/2 * log2(e)
        struct WF { uint4 w[2][4]; };
        auto load_full = [&](const unsigned char* s, auto rt_tag, auto kk_tag, WF& f) {
            constexpr int RT = decltype(rt_tag)::value, KK = decltype(kk_tag)::value;
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                const int off = ((kk * 4 + lg) ^ sw) << 4;
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) f.w[kk][rt] = *(const uint4*)(s + (rt * 16 + li) * 128 + off);
            }
        };
        auto fma_full = [&](const WF& f, auto rt_tag, auto kk_tag, f32x4* acc /* [RT] */, const uint4& y0, const uint4& y1) {
            constexpr int RT = decltype(rt_tag)::value, KK = decltype(kk_tag)::value;
#pragma unroll
            for (int kk = 0; kk < KK; ++kk)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) acc[rt] = T::mfma(f.w[kk][rt], kk ? y1 : y0, acc[rt]);
        };
        auto head_core = [&](const int hs) {      // q K V^T of one head for token tile hf -> O fragments into xbuf slot hs
            f32x4 qa[3] = {z4, z4, z4};
            {
                WF f[2];
                sync(I2{}, BODY{}, NOX{});
                load_full(slab(), I3{}, I2{}, f[0]);
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    if (i + 1 < 5) {
                        if (((i + 1) & 1) == 0) { if (i + 2 < 5) sync(I2{}, BODY{}, NOX{}); else sync(I1{}, BODY{}, NOX{}); }
                        load_full(slab(), I3{}, I2{}, f[(i + 1) & 1]);
                    }
                    fma_full(f[i & 1], I3{}, I2{}, qa, x0[i], x1[i]);
                    if (i + 1 < 5) __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
                }
            }
            const uint4 q0 = frag(qa[0], qa[1]), q1 = frag(qa[2], z4);
            f32x4 sT[5] = {z4, z4, z4, z4, z4};      // [key tile]: lane = token li, registers = keys 16 kt + 4 lg + r
            {
                WF f0, f1;
                sync(I2{}, BODY{}, NOX{});
                load_full(slab(), I4{}, I2{}, f0);
                load_full(slab(), I1{}, I2{}, f1);
                fma_full(f0, I4{}, I2{}, &sT[0], q0, q1);
                fma_full(f1, I1{}, I2{}, &sT[4], q0, q1);
            }
            if (lg == 3) { sT[4][1] = -1e30f; sT[4][2] = -1e30f; sT[4][3] = -1e30f; }      // keys 77, 78, 79 do not exist
            float m = sT[0][0];
#pragma unroll
            for (int kt = 0; kt < 5; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) m = fmaxf(m, sT[kt][r]);
            m = fmaxf(m, __shfl_xor(m, 16)); m = fmaxf(m, __shfl_xor(m, 32));
            const float mc = m * sc;
            float l = 0.f;
#pragma unroll
            for (int kt = 0; kt < 5; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float e = __builtin_amdgcn_exp2f(sT[kt][r] * sc - mc); sT[kt][r] = e; l += e; }
            l += __shfl_xor(l, 16); l += __shfl_xor(l, 32);
            const float inv = 1.0f / l;
            const uint4 pf0 = frag(sT[0], sT[1]), pf1 = frag(sT[2], sT[3]), pf2 = frag(sT[4], z4);
            f32x4 oT[3] = {z4, z4, z4};
            {
                WF f0, f1;
                sync(I2{}, BODY{}, NOX{});
                load_full(slab(), I3{}, I2{}, f0);
                load_full(slab(), I3{}, I1{}, f1);
                fma_full(f0, I3{}, I2{}, oT, pf0, pf1);
                fma_full(f1, I3{}, I1{}, oT, pf2, pf2);
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) oT[i] *= inv;
            unsigned char* dst = xbuf + (hs * 8 + wave) * 2048 + lane * 16;
            *(uint4*)dst = frag(oT[0], oT[1]);
            *(uint4*)(dst + 1024) = frag(oT[2], z4);
        };
        auto head_out = [&](const int hs, auto xch) {      // t += Wo2[:, head] O for both token tiles of the group (row-split)
            sync(I2{}, BODY{}, xch);
            uint4 o0[2], o1[2];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const unsigned char* src = xbuf + (hs * 8 + grp + 4 * tt) * 2048 + lane * 16;
                o0[tt] = *(const uint4*)src;
                o1[tt] = *(const uint4*)(src + 1024);
            }
            group_rs(N5{}, PRE{}, [&](int i) { return &t[i * 2][0]; }, [&](int) -> const uint4 (&)[2] { return o0; },
                     [&](int) -> const uint4 (&)[2] { return o1; }, BODY{});
        };
#pragma unroll 1
        for (int hp = 0; hp < CH / 2; ++hp) {
            head_core(0);
            head_core(1);
            head_out(0, XCH{});
            head_out(1, NOX{});
        }
    }
    add_bias(Q_BO2);

    // ---- GEGLU feed-forward, 20 chunks of 64 hidden units: 10 slabs of W1, 5 slabs of W2.  W1 rows per slab: [value | gate] of 16 hidden units for
    //      each half (packing.pack_chain_stream): this wave gets value and gate of hidden 0..15 (+ 32 hf) from the first five slabs and of 16..31 (+ 32 hf)
    //      from the next five = exactly k step hf of W2's 64-wide k; the other k step comes from the partner
    {
        uint4 own[5][2];
        layer_norm(Q_LN3G, Q_LN3B, own);
        swap_full(own);
    }
#pragma unroll 1
    for (int c = 0; c < 20; ++c) {
        f32x4 g[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) g[i][tt] = z4;
        group_rs(N10{}, NOPRE{}, [&](int i) { return &g[(i / 5) * 2][0]; }, [&](int i) -> const uint4 (&)[2] { return a0[i % 5]; },
                 [&](int i) -> const uint4 (&)[2] { return a1[i % 5]; }, BODY{});
        uint4 hown[2];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            f32x4 hv[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float* bp = prm + Q_B1 + c * 128 + i * 64 + hf * 32 + 4 * lg;
                const float4 bv = *(const float4*)bp, bg = *(const float4*)(bp + 16);
                const vv_f32x2 g01 = gelu2((vv_f32x2){g[2 * i + 1][tt][0] + bg.x, g[2 * i + 1][tt][1] + bg.y});
                const vv_f32x2 g23 = gelu2((vv_f32x2){g[2 * i + 1][tt][2] + bg.z, g[2 * i + 1][tt][3] + bg.w});
                hv[i][0] = (g[2 * i][tt][0] + bv.x) * g01.x; hv[i][1] = (g[2 * i][tt][1] + bv.y) * g01.y;
                hv[i][2] = (g[2 * i][tt][2] + bv.z) * g23.x; hv[i][3] = (g[2 * i][tt][3] + bv.w) * g23.y;
            }
            hown[tt] = frag(hv[0], hv[1]);
        }
        unsigned char* dst = xbuf + ((c & 1) * 8 + wave) * 2048 + lane * 16;
        *(uint4*)dst = hown[0];
        *(uint4*)(dst + 1024) = hown[1];
        sync(I2{}, BODY{}, XCH{});
        const unsigned char* src = xbuf + ((c & 1) * 8 + pw) * 2048 + lane * 16;
        const uint4 hp0 = *(const uint4*)src, hp1 = *(const uint4*)(src + 1024);
        const uint4 h0[2] = {sel(hown[0], hp0), sel(hown[1], hp1)}, h1[2] = {sel(hp0, hown[0]), sel(hp1, hown[1])};
        group_rs(N5{}, PRE{}, [&](int i) { return &t[i * 2][0]; }, [&](int) -> const uint4 (&)[2] { return h0; },
                 [&](int) -> const uint4 (&)[2] { return h1; }, BODY{});
    }
    add_bias(Q_B2);

    // ---- proj_out (+ bias + x [+ res1])
    {
        uint4 own[5][2];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int rb = 0; rb < 5; ++rb) own[rb][tt] = frag(t[2 * rb][tt], t[2 * rb + 1][tt]);
        meet();      // the last FF exchange buffer has been read by everybody
        swap_full(own);
    }
#pragma unroll
    for (int j = 0; j < 10; ++j)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) t[j][tt] = z4;
    dense320(t, TAIL{});          // stream slabs 437..461
#ifdef VV_PROBE_REGLOAD
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        const int64_t r = row0 + tt * 16 + li;
        if (r < p.M) {
            const int64_t row = r * CC;
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                const int c = chan(j);
                const float4 b = *(const float4*)(prm + Q_BOUT + c);
                const float4 xr = *(const float4*)(p.x + row + c);
                float v0 = t[j][tt][0] + b.x + xr.x, v1 = t[j][tt][1] + b.y + xr.y, v2 = t[j][tt][2] + b.z + xr.z, v3 = t[j][tt][3] + b.w + xr.w;
                if (p.res1) { const float4 r4 = *(const float4*)(p.res1 + row + c); v0 += r4.x; v1 += r4.y; v2 += r4.z; v3 += r4.w; }
                if (p.out_dtype == VV_F32) *(float4*)((float*)p.out + row + c) = make_float4(v0, v1, v2, v3);
                else *(uint2*)((unsigned short*)p.out + row + c) = make_uint2(pack2<T>(v0, v1), pack2<T>(v2, v3));
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------------
// COLUMN-SPLIT form of the tail (round 5; LAB form, -DVV_CHAIN_FORM=2: correct, 2.16 ms against 2.07 ms of the row-split form -- one wave per SIMD pays its
// GELU / LayerNorm / AGPR-copy VALU time and every LDS round trip in full; kept because it has no weight ring at all, see profiles/r5_chain_forms.txt).  What the probes of the ring forms said (profiles/r5_chain_forms.txt): with the weight
// stream's LDS-DMA removed the row-split kernel runs 2.08 -> 1.31 ms, with the same bytes loaded into registers instead 1.57 ms -- staging weights
// that every wave reads anyway through LDS costs more than the MFMAs they feed.  So here NO weight touches LDS: wave w of a 4-wave block owns output
// channels 80 w .. 80 w + 79 of every layer for all 128 tokens, its weights are a PRIVATE stream of ready-made A-operand fragments (1 KB each:
// packing.pack_chain_stream_columns) read with plain global_load_dwordx4 through a 10-fragment register ring that runs 2 k steps ahead across
// layer boundaries, and what the waves share -- the layer's INPUT activations, h16 [128 tokens][320] -- sits in LDS (80 KB, 16-byte chunks XOR-swizzled
// by the token so that the B-operand reads are conflict free) and is read 8 fragments per 40 MFMAs.  The fp32 trunk [80 channels x 128 tokens] stays in
// 160 accumulator registers; a layer's output is written back to the activation buffer (own columns) behind a barrier: ~35 barriers per block instead
// of 231 ring steps.  LayerNorm: per-wave (mean, M2) over its 80 channels through LDS, merged by Chan's formula.  Cross-attention: wave w does heads
// 2 w, 2 w + 1 (q projection padded to 48 rows, K_h / V_h^T fragments from the stream, scores / softmax / PV per token-tile pair as in the forms
// above); the four heads of a phase leave O in a 40 KB buffer and the matching half of Wo2 follows.  GEGLU: 10 chunks of 128 hidden units (32 per wave).
constexpr int CS_NF = 10, CS_FRAGS = 870;
constexpr int P_BO1 = 0, P_LN2G = 320, P_LN2B = 640, P_BO2 = 960, P_LN3G = 1280, P_LN3B = 1600, P_B1V = 1920, P_B1G = 3200, P_B2 = 4480, P_BOUT = 4800;

template <typename T>
__global__ __launch_bounds__(256, 1) void chain_cs_c320_kernel(const vv_chain_params p) {
    __shared__ __attribute__((aligned(1024))) unsigned char act[128 * 640];
    __shared__ __attribute__((aligned(1024))) unsigned char hbuf[40960];
    __shared__ __attribute__((aligned(16))) float stats[4 * 128 * 2];
    __shared__ __attribute__((aligned(16))) float prm[Q_TOTAL];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int64_t row0 = (int64_t)blockIdx.x * 128;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};

    for (int i = tid * 4; i < Q_TOTAL; i += 256 * 4) *(float4*)(prm + i) = *(const float4*)(p.params + i);
    // block barrier for LDS hand-offs: raw s_barrier behind an LDS-only wait (__syncthreads() would also drain vmcnt: the weight ring's loads in flight)
    auto bar = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); };

    // ---- the wave's weight stream and its register ring
    const unsigned char* wst = (const unsigned char*)p.stream + (int64_t)wave * CS_FRAGS * 1024 + lane * 16;
    uint4 wr[CS_NF];
#pragma unroll
    for (int i = 0; i < CS_NF; ++i) { wr[i] = *(const uint4*)wst; wst += 1024; }
    // take fragment i of the ring and refill the slot with the stream's next fragment
#ifdef VV_PROBE_NOWLOAD      // timing probe (wrong results): the ring is never refilled
    auto take = [&](const int i) -> uint4 { uint4 v = wr[i]; asm volatile("" : "+v"(v.x)); return v; };
#else
    auto take = [&](const int i) -> uint4 { const uint4 v = wr[i]; wr[i] = *(const uint4*)wst; wst += 1024; return v; };
#endif

    // ---- activation buffers.  act: [128][320] h16, chunk c (16 B) of token row n at chunk c ^ ((n >> 1) & 7); hbuf as 2 x [128][64] (GEGLU chunks): chunk c
    //      at c ^ ((n >> 1) & 7); as [128][160] (O of four heads): chunk c at c ^ ((n >> 2) & 3)
    const int sA = li >> 1, sO = li >> 2;
    auto act_rd = [&](const int ks, const int tt) -> uint4 { return *(const uint4*)(act + (16 * tt + li) * 640 + (((4 * ks + lg) ^ sA) << 4)); };
    auto act_wr = [&](const int ch /* multiple of 4 */, const int tt, const uint2 v) {
        *(uint2*)(act + (16 * tt + li) * 640 + ((((ch >> 3)) ^ sA) << 4) + ((ch & 4) << 1)) = v;
    };
    auto ob_rd = [&](const int ks, const int tt) -> uint4 { return *(const uint4*)(hbuf + (16 * tt + li) * 320 + (((4 * ks + lg) ^ sO) << 4)); };
    auto ob_wr = [&](const int ch, const int tt, const uint2 v) { *(uint2*)(hbuf + (16 * tt + li) * 320 + (((ch >> 3) ^ sO) << 4) + ((ch & 4) << 1)) = v; };
    auto pk4 = [&](const f32x4& v) -> uint2 { return make_uint2(pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3])); };
    auto frag = [&](const f32x4& lo, const f32x4& hi) -> uint4 {
        return make_uint4(pack2<T>(lo[0], lo[1]), pack2<T>(lo[2], lo[3]), pack2<T>(hi[0], hi[1]), pack2<T>(hi[2], hi[3]));
    };

    // ---- inputs: o (h16) -> act; t_in (own channels) -> trunk.  Rows past M repeat row M - 1 (never stored)
    f32x4 t[5][8];
    {
        // o: 128 rows x 40 chunks = 5120 chunks, 20 per thread: chunk q of the block = (row q / 40, chunk q % 40)
#pragma unroll 4
        for (int q = tid; q < 128 * 40; q += 256) {
            const int n = q / 40, c = q - n * 40;
            int64_t row = row0 + n;
            if (row >= p.M) row = p.M - 1;
            const uint4 v = *(const uint4*)((const unsigned short*)p.o + row * CC + c * 8);
            *(uint4*)(act + n * 640 + ((c ^ ((n >> 1) & 7)) << 4)) = v;
        }
#pragma unroll
        for (int tt = 0; tt < 8; ++tt) {
            int64_t row = row0 + tt * 16 + li;
            if (row >= p.M) row = p.M - 1;
            const float* trow = p.t_in + row * CC + 80 * wave + 4 * lg;
#pragma unroll
            for (int rt = 0; rt < 5; ++rt) { const float4 v = *(const float4*)(trow + 16 * rt); t[rt][tt] = f32x4{v.x, v.y, v.z, v.w}; }
        }
        __syncthreads();
    }

    // acc[RT][8] += W (RT row tiles of the stream, KS k steps) x buffer: per k step 8 B fragments (double buffered: the reads of step ks + 1 are issued
    // before the MFMAs of step ks) and RT ring fragments starting at ring position (R0 + RT ks) % 10
    auto layer_cb = [&](auto rt_tag, auto ks_tag, auto r0_tag, f32x4* acc /* [RT][8] */, auto&& rd, auto&& cb) {
        constexpr int RT = decltype(rt_tag)::value, KS = decltype(ks_tag)::value, R0 = decltype(r0_tag)::value;
        uint4 bb[2][8];
#pragma unroll
        for (int tt = 0; tt < 8; ++tt) bb[0][tt] = rd(0, tt);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + 1 < KS) {
#pragma unroll
                for (int tt = 0; tt < 8; ++tt) bb[(ks + 1) & 1][tt] = rd(ks + 1, tt);
            }
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const uint4 wa = take((R0 + RT * ks + rt) % CS_NF);
#pragma unroll
                for (int tt = 0; tt < 8; ++tt) acc[rt * 8 + tt] = T::mfma(wa, bb[ks & 1][tt], acc[rt * 8 + tt]);
            }
            cb(ks);      // independent VALU work of the caller (GEGLU of the previous chunk): scheduled among this step's MFMAs
#ifndef VV_CS_NO_PIN
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
    };
    auto layer = [&](auto rt_tag, auto ks_tag, auto r0_tag, f32x4* acc, auto&& rd) { layer_cb(rt_tag, ks_tag, r0_tag, acc, rd, [](int) {}); };
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>; using I5 = std::integral_constant<int, 5>;
    using I10 = std::integral_constant<int, 10>; using I0 = std::integral_constant<int, 0>;
    auto rdA = [&](const int ks, const int tt) -> uint4 { return act_rd(ks, tt); };
    auto rdO = [&](const int ks, const int tt) -> uint4 { return ob_rd(ks, tt); };
    auto add_bias = [&](const int off) {
#pragma unroll
        for (int rt = 0; rt < 5; ++rt) {
            const float4 b = *(const float4*)(prm + off + 80 * wave + 16 * rt + 4 * lg);
#pragma unroll
            for (int tt = 0; tt < 8; ++tt) { t[rt][tt][0] += b.x; t[rt][tt][1] += b.y; t[rt][tt][2] += b.z; t[rt][tt][3] += b.w; }
        }
    };
    // act <- h16(LN(t) g + b) (own columns); the caller's next barrier publishes it
    auto layer_norm = [&](const int goff, const int boff) {
        float ml[8], m2[8];
#pragma unroll
        for (int tt = 0; tt < 8; ++tt) {
            float s = 0.f;
#pragma unroll
            for (int rt = 0; rt < 5; ++rt) s += (t[rt][tt][0] + t[rt][tt][1]) + (t[rt][tt][2] + t[rt][tt][3]);
            s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);
            ml[tt] = s * (1.0f / 80);
            float q = 0.f;
#pragma unroll
            for (int rt = 0; rt < 5; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float d = t[rt][tt][r] - ml[tt]; q += d * d; }
            q += __shfl_xor(q, 16); q += __shfl_xor(q, 32);
            m2[tt] = q;
            if (lg == 0) *(float2*)(stats + ((wave * 128) + 16 * tt + li) * 2) = make_float2(ml[tt], q);
        }
        bar();                // statistics of all four waves; everybody is also done reading the activation buffer
#pragma unroll
        for (int tt = 0; tt < 8; ++tt) {
            float mw[4], qw[4];
#pragma unroll
            for (int w4 = 0; w4 < 4; ++w4) { const float2 v = *(const float2*)(stats + (w4 * 128 + 16 * tt + li) * 2); mw[w4] = v.x; qw[w4] = v.y; }
            const float mean = 0.25f * ((mw[0] + mw[1]) + (mw[2] + mw[3]));
            float M2 = (qw[0] + qw[1]) + (qw[2] + qw[3]);
#pragma unroll
            for (int w4 = 0; w4 < 4; ++w4) { const float d = mw[w4] - mean; M2 += 80.0f * d * d; }
            const float rstd = rsqrtf(M2 * (1.0f / CC) + 1e-5f);
#pragma unroll
            for (int rt = 0; rt < 5; ++rt) {
                const int c = 80 * wave + 16 * rt + 4 * lg;
                const float4 g = *(const float4*)(prm + goff + c), b = *(const float4*)(prm + boff + c);
                const f32x4 y = {(t[rt][tt][0] - mean) * rstd * g.x + b.x, (t[rt][tt][1] - mean) * rstd * g.y + b.y,
                                 (t[rt][tt][2] - mean) * rstd * g.z + b.z, (t[rt][tt][3] - mean) * rstd * g.w + b.w};
                act_wr(c, tt, pk4(y));
            }
        }
    };

    // ---- attn1 output projection: t = t_in + Wo1 o + bo1
    layer(I5{}, I10{}, I0{}, &t[0][0], rdA);
    add_bias(P_BO1);

    // ---- attn2: cross-attention to the 77 text keys; this wave's heads 2 w (phase 0) and 2 w + 1 (phase 1)
    layer_norm(P_LN2G, P_LN2B);
    bar();
    const float sc = 0.15811388300841897f * 1.4426950408889634f;      // 40^-1/2 * log2(e)
#pragma unroll 1
    for (int ph = 0; ph < 2; ++ph) {
        uint4 qf0[8], qf1[8];      // q of the head as B fragments: k step 0 = d in PERM32 order, k step 1 = d 32..39 + zeros
        {
            f32x4 qa[3][8];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int tt = 0; tt < 8; ++tt) qa[i][tt] = z4;
            layer(I3{}, I10{}, I0{}, &qa[0][0], rdA);
#pragma unroll
            for (int tt = 0; tt < 8; ++tt) { qf0[tt] = frag(qa[0][tt], qa[1][tt]); qf1[tt] = frag(qa[2][tt], z4); }
        }
        uint4 kw[10], vw[9];
#pragma unroll
        for (int i = 0; i < 10; ++i) kw[i] = take(i);
#pragma unroll
        for (int i = 0; i < 10; ++i) { const uint4 v = take(i); if (i < 9) vw[i] = v; }
        if (ph) bar();                // the O buffer: phase 0's half of Wo2 has been read by everybody
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {      // token tile pairs
            f32x4 sT[5][2];                    // [key tile][tile of the pair]: lane = token li, registers = keys 16 kt + 4 lg + r
#pragma unroll
            for (int kt = 0; kt < 5; ++kt)
#pragma unroll
                for (int u = 0; u < 2; ++u) sT[kt][u] = z4;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int kt = 0; kt < 5; ++kt)
#pragma unroll
                    for (int u = 0; u < 2; ++u) sT[kt][u] = T::mfma(kw[kk * 5 + kt], kk ? qf1[2 * pp + u] : qf0[2 * pp + u], sT[kt][u]);
            uint4 pf[3][2];
            float inv[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (lg == 3) { sT[4][u][1] = -1e30f; sT[4][u][2] = -1e30f; sT[4][u][3] = -1e30f; }      // keys 77, 78, 79 do not exist
                float m = sT[0][u][0];
#pragma unroll
                for (int kt = 0; kt < 5; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) m = fmaxf(m, sT[kt][u][r]);
                m = fmaxf(m, __shfl_xor(m, 16)); m = fmaxf(m, __shfl_xor(m, 32));
                const float mc = m * sc;
                float l = 0.f;
#pragma unroll
                for (int kt = 0; kt < 5; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { const float e = __builtin_amdgcn_exp2f(sT[kt][u][r] * sc - mc); sT[kt][u][r] = e; l += e; }
                l += __shfl_xor(l, 16); l += __shfl_xor(l, 32);
                inv[u] = 1.0f / l;
                pf[0][u] = frag(sT[0][u], sT[1][u]); pf[1][u] = frag(sT[2][u], sT[3][u]); pf[2][u] = frag(sT[4][u], z4);
            }
            f32x4 oT[3][2];                    // [d tile][tile of the pair]
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int u = 0; u < 2; ++u) oT[i][u] = z4;
#pragma unroll
            for (int kk = 0; kk < 3; ++kk)
#pragma unroll
                for (int dt = 0; dt < 3; ++dt)
#pragma unroll
                    for (int u = 0; u < 2; ++u) oT[dt][u] = T::mfma(vw[kk * 3 + dt], pf[kk][u], oT[dt][u]);
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int dt = 0; dt < 3; ++dt)
                    if (16 * dt + 4 * lg < CD) ob_wr(CD * wave + 16 * dt + 4 * lg, 2 * pp + u, pk4(oT[dt][u] * inv[u]));
        }
        bar();                // O of the phase's four heads complete
        // t += Wo2[:, the phase's heads] O     (25 fragments + 5 of padding)
        layer(I5{}, I5{}, I0{}, &t[0][0], rdO);
#pragma unroll
        for (int i = 5; i < 10; ++i) (void)take(i);
    }
    add_bias(P_BO2);

    // ---- GEGLU feed-forward, 20 chunks of 64 hidden units (16 per wave): g = W1 LN3(t) (20 fragments), t += W2 GEGLU(g) (10).  Software pipelined over the
    //      chunks (stream order W1 (0) | W1 (1), W2 (0) | W1 (2), W2 (1) | ...): the GEGLU of chunk c -- ~2.6 k cycles of VALU on a wave that has its SIMD to
    //      itself -- runs tile by tile INSIDE the k steps of W1 (c + 1), under that layer's MFMAs, from a second accumulator set; hidden activations
    //      alternate between the two halves of hbuf: one barrier per chunk
    layer_norm(P_LN3G, P_LN3B);
    bar();
    auto zero_g = [&](f32x4 (&g)[2][8]) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int tt = 0; tt < 8; ++tt) g[i][tt] = z4;
    };
    auto geglu_tile = [&](const f32x4 (&g)[2][8], const int tt, const float4 bv, const float4 bg, unsigned char* hb) {
        const vv_f32x2 g01 = gelu2((vv_f32x2){g[1][tt][0] + bg.x, g[1][tt][1] + bg.y});
        const vv_f32x2 g23 = gelu2((vv_f32x2){g[1][tt][2] + bg.z, g[1][tt][3] + bg.w});
        const f32x4 hv = {(g[0][tt][0] + bv.x) * g01.x, (g[0][tt][1] + bv.y) * g01.y, (g[0][tt][2] + bv.z) * g23.x, (g[0][tt][3] + bv.w) * g23.y};
        const int ch = 16 * wave + 4 * lg;
        *(uint2*)(hb + (16 * tt + li) * 128 + (((ch >> 3) ^ sA) << 4) + ((ch & 4) << 1)) = pk4(hv);
    };
    // chunk c: its W1 result is in `cur`; W1 (c + 1) goes to `nxt` with GEGLU (c) inside, then the barrier, then W2 (c)
    auto ff_step = [&](const int c, f32x4 (&cur)[2][8], f32x4 (&nxt)[2][8]) {
        unsigned char* hb = hbuf + (c & 1) * 16384;
        const int u = 64 * c + 16 * wave + 4 * lg;
        const float4 bv = *(const float4*)(prm + P_B1V + u), bg = *(const float4*)(prm + P_B1G + u);
        if (c + 1 < 20) {
            zero_g(nxt);
            layer_cb(I2{}, I10{}, I0{}, &nxt[0][0], rdA, [&](const int ks) { if (ks < 8) geglu_tile(cur, ks, bv, bg, hb); });
        } else {
#pragma unroll
            for (int tt = 0; tt < 8; ++tt) geglu_tile(cur, tt, bv, bg, hb);
        }
        bar();
        layer(I5{}, I2{}, I0{}, &t[0][0], [&](const int ks, const int tt) -> uint4 { return *(const uint4*)(hb + (16 * tt + li) * 128 + (((4 * ks + lg) ^ sA) << 4)); });
    };
    {
        f32x4 gA[2][8], gB[2][8];
        zero_g(gA);
        layer(I2{}, I10{}, I0{}, &gA[0][0], rdA);
#pragma unroll 1
        for (int c = 0; c < 20; c += 2) { ff_step(c, gA, gB); ff_step(c + 1, gB, gA); }
    }
    add_bias(P_B2);

    // ---- proj_out (+ bias + x [+ res1]); the trunk goes to the activation buffer as h16 first
    bar();                    // everybody is done with LN3's output
#pragma unroll
    for (int rt = 0; rt < 5; ++rt)
#pragma unroll
        for (int tt = 0; tt < 8; ++tt) { act_wr(80 * wave + 16 * rt + 4 * lg, tt, pk4(t[rt][tt])); t[rt][tt] = z4; }
    bar();
    layer(I5{}, I10{}, I0{}, &t[0][0], rdA);
#pragma unroll
    for (int tt = 0; tt < 8; ++tt) {
        const int64_t r = row0 + tt * 16 + li;
        if (r < p.M) {
            const int64_t row = r * CC;
#pragma unroll
            for (int rt = 0; rt < 5; ++rt) {
                const int c = 80 * wave + 16 * rt + 4 * lg;
                const float4 b = *(const float4*)(prm + P_BOUT + c);
                const float4 xr = *(const float4*)(p.x + row + c);
                float v0 = t[rt][tt][0] + b.x + xr.x, v1 = t[rt][tt][1] + b.y + xr.y, v2 = t[rt][tt][2] + b.z + xr.z, v3 = t[rt][tt][3] + b.w + xr.w;
                if (p.res1) { const float4 r4 = *(const float4*)(p.res1 + row + c); v0 += r4.x; v1 += r4.y; v2 += r4.z; v3 += r4.w; }
                if (p.out_dtype == VV_F32) *(float4*)((float*)p.out + row + c) = make_float4(v0, v1, v2, v3);
                else *(uint2*)((unsigned short*)p.out + row + c) = make_uint2(pack2<T>(v0, v1), pack2<T>(v2, v3));
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Front half of the same block: everything BEFORE the self-attention core is per token too --
//   t = Win GN(x) + bin        (GroupNorm apply with per-frame statistics + proj_in; t = the block's fp32 residual stream, written out for the tail)
//   qkv = Wqkv LN1(t)          (fused q | k | v projection, stored head-major [frame][q|k|v][head][token][40] for vv_attention, q pre-scaled)
// in one kernel per 128 tokens (100 slabs: 25 + 15 row blocks x 5), replacing vv_groupnorm's apply pass, two vv_conv_gemm and one vv_layernorm.
constexpr int F_BIN = 0, F_LN1G = 320, F_LN1B = 640, F_TOTAL = 960;
constexpr int NF_SLABS = 25 + 15 * 5;      // 100

template <typename T>
__global__ __launch_bounds__(256, 1) void chain_front_c320_kernel(const vv_chain_front_params p) {
    __shared__ __attribute__((aligned(1024))) unsigned char ring[NSLOT * SLAB];
    __shared__ __attribute__((aligned(16))) float prm[F_TOTAL];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int64_t row0 = (int64_t)blockIdx.x * 128 + wave * 32;

    for (int i = tid * 4; i < F_TOTAL; i += 256 * 4) *(float4*)(prm + i) = *(const float4*)(p.params + i);
    const unsigned char* sbase = (const unsigned char*)p.stream + (wave * 2) * 1024 + lane * 16;
    int issued = 0, consumed = 0;
    auto issue = [&]() {
        unsigned char* dst = ring + (issued % NSLOT) * SLAB + (wave * 2) * 1024;
        const unsigned char* src = sbase + (int64_t)issued * SLAB;
        glds16_asm(src, dst);
        glds16_asm(src + 1024, dst + 1024);
        ++issued;
    };
    auto next_slab = [&](auto even_tag, auto tail_tag) -> const unsigned char* {
        if constexpr (decltype(even_tag)::value) {
            if (!decltype(tail_tag)::value || issued < NF_SLABS) { issue(); issue(); asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); }
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        const unsigned char* s = ring + (consumed % NSLOT) * SLAB;
        ++consumed;
        return s;
    };

    // ---- x -> GroupNorm apply (scale / shift of the token's own frame) -> activation fragments a[ks][tt]
    uint4 a[10][2];
    int64_t rows[2];
    {
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            int64_t row = row0 + tt * 16 + li;
            if (row >= p.M) row = p.M - 1;
            rows[tt] = row;
            const float* xrow = p.x + row * CC;
            const float* aff = p.gn_affine + (row / p.HW) * (2 * CC);
#pragma unroll
            for (int s = 0; s < 10; ++s) {
                const int c0 = 32 * s + 4 * lg, c1 = c0 + 16;
                const float4 x0 = *(const float4*)(xrow + c0), x1 = *(const float4*)(xrow + c1);
                const float4 a0 = *(const float4*)(aff + c0), b0 = *(const float4*)(aff + CC + c0);
                const float4 a1 = *(const float4*)(aff + c1), b1 = *(const float4*)(aff + CC + c1);
                a[s][tt] = make_uint4(pack2<T>(x0.x * a0.x + b0.x, x0.y * a0.y + b0.y), pack2<T>(x0.z * a0.z + b0.z, x0.w * a0.w + b0.w),
                                      pack2<T>(x1.x * a1.x + b1.x, x1.y * a1.y + b1.y), pack2<T>(x1.z * a1.z + b1.z, x1.w * a1.w + b1.w));
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();       // parameter block visible
#pragma unroll 1
        for (int i = 0; i < AHEAD; ++i) issue();
    }

    struct WF { uint4 w[2][4]; };
    auto slab_load = [&](const unsigned char* s, WF& f) {
        const int sw = li & 7;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int off = ((kk * 4 + lg) ^ sw) << 4;
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) f.w[kk][rt] = *(const uint4*)(s + (rt * 16 + li) * 128 + off);
        }
    };
    auto slab_fma = [&](const WF& f, f32x4* acc /* [4][2] */, const uint4 (&x0)[2], const uint4 (&x1)[2]) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) acc[rt * 2 + tt] = T::mfma(f.w[kk][rt], kk ? x1[tt] : x0[tt], acc[rt * 2 + tt]);
    };
    using EVEN = std::true_type; using ODD = std::false_type; using BODY = std::false_type; using TAIL = std::true_type;
    // N slabs [64 rows x 64 k] (5 k tiles per 64-row block) starting at stream-index parity P0
    auto slab_group = [&](auto p0_tag, auto n_tag, auto&& acc_of, auto tail) {
        constexpr int P0 = decltype(p0_tag)::value, N = decltype(n_tag)::value;
        WF f[2];
        slab_load(next_slab(std::bool_constant<P0 == 0>{}, tail), f[0]);
#pragma unroll
        for (int i = 0; i < N; ++i) {
            if (i + 1 < N) {
                if (((P0 + i + 1) & 1) == 0) slab_load(next_slab(EVEN{}, tail), f[(i + 1) & 1]);
                else slab_load(next_slab(ODD{}, tail), f[(i + 1) & 1]);
            }
            slab_fma(f[i & 1], acc_of(i), a[2 * (i % 5)], a[2 * (i % 5) + 1]);
            if (i + 1 < N) __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
        }
    };
    using P0E = std::integral_constant<int, 0>; using P0O = std::integral_constant<int, 1>;
    using N5 = std::integral_constant<int, 5>; using N25 = std::integral_constant<int, 25>;
    auto frag = [&](const f32x4& lo, const f32x4& hi) -> uint4 {
        return make_uint4(pack2<T>(lo[0], lo[1]), pack2<T>(lo[2], lo[3]), pack2<T>(hi[0], hi[1]), pack2<T>(hi[2], hi[3]));
    };

    // ---- proj_in: t = Win a + bin, stored (the tail kernel and the residual read it back)
    f32x4 t[20][2];
#pragma unroll
    for (int j = 0; j < 20; ++j)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) t[j][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
    slab_group(P0E{}, N25{}, [&](int i) { return &t[(i / 5) * 4][0]; }, BODY{});
#pragma unroll
    for (int j = 0; j < 20; ++j) {
        const float4 b = *(const float4*)(prm + F_BIN + 16 * j + 4 * lg);
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) { t[j][tt][0] += b.x; t[j][tt][1] += b.y; t[j][tt][2] += b.z; t[j][tt][3] += b.w; }
    }
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        if (row0 + tt * 16 + li < p.M) {
            float* trow = p.t_out + rows[tt] * CC;
#pragma unroll
            for (int j = 0; j < 20; ++j) *(float4*)(trow + 16 * j + 4 * lg) = make_float4(t[j][tt][0], t[j][tt][1], t[j][tt][2], t[j][tt][3]);
        }
    }
    // ---- LN1 -> a
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 20; ++j) s += (t[j][tt][0] + t[j][tt][1]) + (t[j][tt][2] + t[j][tt][3]);
        s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);
        const float mean = s * (1.0f / CC);
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < 20; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float d = t[j][tt][r] - mean; q += d * d; }
        q += __shfl_xor(q, 16); q += __shfl_xor(q, 32);
        const float rstd = rsqrtf(q * (1.0f / CC) + 1e-5f);
#pragma unroll
        for (int s2 = 0; s2 < 10; ++s2) {
            f32x4 y[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int j = 2 * s2 + h, c = 16 * j + 4 * lg;
                const float4 g = *(const float4*)(prm + F_LN1G + c), b = *(const float4*)(prm + F_LN1B + c);
                y[h][0] = (t[j][tt][0] - mean) * rstd * g.x + b.x; y[h][1] = (t[j][tt][1] - mean) * rstd * g.y + b.y;
                y[h][2] = (t[j][tt][2] - mean) * rstd * g.z + b.z; y[h][3] = (t[j][tt][3] - mean) * rstd * g.w + b.w;
            }
            a[s2][tt] = frag(y[0], y[1]);
        }
    }
    // ---- fused q | k | v projection: 15 blocks of 64 output channels, stored head-major.  Channel c = 64 rb + 16 rt + 4 lg + r is element
    //      (which = c / 320, head = (c % 320) / 40, d = c % 40) of the token's row; 4 consecutive channels never straddle a head (40 % 4 == 0)
    unsigned short* qkv = (unsigned short*)p.qkv;
    int64_t tokbase[2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        const int64_t fr = rows[tt] / p.HW, tk = rows[tt] - fr * p.HW;
        tokbase[tt] = fr * (3 * (int64_t)p.HW * CC) + tk * CD;          // + which * HW * 320 + head * HW * 40 + d
    }
    auto qkv_block = [&](const int rb, auto p0_tag, auto tail) {
        f32x4 acc[4][2];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) acc[rt][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
        slab_group(p0_tag, N5{}, [&](int) { return &acc[0][0]; }, tail);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            const int c = 64 * rb + 16 * rt + 4 * lg;
            const int which = c / CC, cc = c - which * CC, head = cc / CD, d = cc - head * CD;
            const int64_t off = ((int64_t)which * CC + (int64_t)head * CD) * p.HW + d;
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
                if (row0 + tt * 16 + li < p.M)
                    *(uint2*)(qkv + tokbase[tt] + off) = make_uint2(pack2<T>(acc[rt][tt][0], acc[rt][tt][1]), pack2<T>(acc[rt][tt][2], acc[rt][tt][3]));
        }
    };
    // (row block rb is 5 slabs and starts at stream index 25 + 5 rb: odd for even rb -> two row blocks per loop iteration)
#pragma unroll 1
    for (int rb = 0; rb < 12; rb += 2) { qkv_block(rb, P0O{}, BODY{}); qkv_block(rb + 1, P0E{}, BODY{}); }
    qkv_block(12, P0O{}, TAIL{}); qkv_block(13, P0E{}, TAIL{}); qkv_block(14, P0O{}, TAIL{});
}

// per-frame GroupNorm affine: out[f][0][c] = rstd * gamma[c], out[f][1][c] = beta[c] - mean * rstd * gamma[c]
__global__ void gn_affine_frames_kernel(const float* fin /* [F][groups][2] */, const float* gamma, const float* beta, int C, int groups, int F, float* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= F * C) return;
    const int f = i / C, c = i - f * C, g = c / (C / groups);
    const float a = fin[((int64_t)f * groups + g) * 2 + 1] * gamma[c];
    out[(int64_t)f * 2 * C + c] = a;
    out[(int64_t)f * 2 * C + C + c] = beta[c] - fin[((int64_t)f * groups + g) * 2] * a;
}

}  // namespace

extern "C" int vv_spatial_chain_c320(const vv_chain_params* pp, int dtype, void* stream) {
    if (!pp) VV_FAIL(VV_E_ARG, "vv_spatial_chain_c320: null params");
    const vv_chain_params& p = *pp;
    if (!p.o || !p.t_in || !p.x || !p.out || !p.stream || !p.params) VV_FAIL(VV_E_ARG, "vv_spatial_chain_c320: null pointer");
    if (p.C != CC || p.heads != CH || p.text_len != NKEY) VV_FAIL(VV_E_UNSUPPORTED, "vv_spatial_chain_c320: built for C = 320, 8 heads, 77 text tokens (got %d, %d, %d)", p.C, p.heads, p.text_len);
    if (p.M <= 0) VV_FAIL(VV_E_ARG, "vv_spatial_chain_c320: empty input");
    if (p.out_dtype != VV_F32 && p.out_dtype != dtype) VV_FAIL(VV_E_ARG, "vv_spatial_chain_c320: out_dtype mismatch");
#ifndef VV_CHAIN_FORM
#define VV_CHAIN_FORM 1    // 1 = row-split pairs (chain_rs_c320_kernel, packing layout "rowsplit": the product form); 2 = column-split (chain_cs_c320_kernel, layout "columns": lab); 0 = token-split forms (layout "tokens": lab)
#endif
    const int want_slabs = VV_CHAIN_FORM == 2 ? 4 * CS_FRAGS : N_SLABS;
    if (p.n_slabs != want_slabs || p.n_params != Q_TOTAL) VV_FAIL(VV_E_ARG, "vv_spatial_chain_c320: stream / parameter block size mismatch (%d slabs, %d floats)", p.n_slabs, p.n_params);
    const int64_t nblk = (p.M + 127) / 128;
    if (nblk > 0x7fffffff) VV_FAIL(VV_E_ARG, "vv_spatial_chain_c320: grid too large");
    hipStream_t st = (hipStream_t)stream;
#ifndef VV_CHAIN_TT
#define VV_CHAIN_TT 1      // token tiles per wave: 1 = 8 waves x 16 tokens, two per SIMD, staggered (round 5); 2 = the round-3 form (lab A/B: -DVV_CHAIN_TT=2)
#endif
#ifndef VV_CHAIN_LAG
#define VV_CHAIN_LAG 0     // slab pairs the second half of the block runs behind the first
#endif
#ifndef VV_CHAIN_AHEAD
#define VV_CHAIN_AHEAD 6
#endif
#if VV_CHAIN_FORM == 2
    if (dtype == VV_BF16) hipLaunchKernelGGL((chain_cs_c320_kernel<BF16>), dim3((unsigned)nblk), dim3(256), 0, st, p);
    else if (dtype == VV_F16) hipLaunchKernelGGL((chain_cs_c320_kernel<F16>), dim3((unsigned)nblk), dim3(256), 0, st, p);
#elif VV_CHAIN_FORM == 1
    if (dtype == VV_BF16) hipLaunchKernelGGL((chain_rs_c320_kernel<BF16>), dim3((unsigned)nblk), dim3(512), 0, st, p);
    else if (dtype == VV_F16) hipLaunchKernelGGL((chain_rs_c320_kernel<F16>), dim3((unsigned)nblk), dim3(512), 0, st, p);
#else
    constexpr int TT = VV_CHAIN_TT, LAG = TT == 2 ? 0 : VV_CHAIN_LAG, NT = 128 / (16 * TT) * 64, AH = VV_CHAIN_AHEAD;
    if (dtype == VV_BF16) hipLaunchKernelGGL((chain_c320_kernel<BF16, TT, LAG, AH>), dim3((unsigned)nblk), dim3(NT), 0, st, p);
    else if (dtype == VV_F16) hipLaunchKernelGGL((chain_c320_kernel<F16, TT, LAG, AH>), dim3((unsigned)nblk), dim3(NT), 0, st, p);
#endif
    else VV_FAIL(VV_E_ARG, "vv_spatial_chain_c320: bad dtype");
    VV_CHECK_LAUNCH("vv_spatial_chain_c320");
    return VV_OK;
}

extern "C" int vv_gn_affine_frames(const float* mean_rstd, const float* gamma, const float* beta, int C, int groups, int F, float* out, void* stream) {
    if (!mean_rstd || !gamma || !beta || !out || C <= 0 || groups <= 0 || C % groups || F <= 0) VV_FAIL(VV_E_ARG, "vv_gn_affine_frames: bad args");
    hipLaunchKernelGGL(gn_affine_frames_kernel, dim3((F * C + 255) / 256), dim3(256), 0, (hipStream_t)stream, mean_rstd, gamma, beta, C, groups, F, out);
    VV_CHECK_LAUNCH("vv_gn_affine_frames");
    return VV_OK;
}

extern "C" int vv_spatial_chain_front_c320(const vv_chain_front_params* pp, int dtype, void* stream) {
    if (!pp) VV_FAIL(VV_E_ARG, "vv_spatial_chain_front_c320: null params");
    const vv_chain_front_params& p = *pp;
    if (!p.x || !p.gn_affine || !p.t_out || !p.qkv || !p.stream || !p.params) VV_FAIL(VV_E_ARG, "vv_spatial_chain_front_c320: null pointer");
    if (p.C != CC || p.heads != CH) VV_FAIL(VV_E_UNSUPPORTED, "vv_spatial_chain_front_c320: built for C = 320, 8 heads (got %d, %d)", p.C, p.heads);
    if (p.M <= 0 || p.HW <= 0 || p.M % p.HW) VV_FAIL(VV_E_ARG, "vv_spatial_chain_front_c320: M must be a positive multiple of HW");
    if (p.n_slabs != NF_SLABS || p.n_params != F_TOTAL) VV_FAIL(VV_E_ARG, "vv_spatial_chain_front_c320: stream / parameter block size mismatch (%d slabs, %d floats)", p.n_slabs, p.n_params);
    const int64_t nblk = (p.M + 127) / 128;
    if (nblk > 0x7fffffff) VV_FAIL(VV_E_ARG, "vv_spatial_chain_front_c320: grid too large");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == VV_BF16) hipLaunchKernelGGL(chain_front_c320_kernel<BF16>, dim3((unsigned)nblk), dim3(256), 0, st, p);
    else if (dtype == VV_F16) hipLaunchKernelGGL(chain_front_c320_kernel<F16>, dim3((unsigned)nblk), dim3(256), 0, st, p);
    else VV_FAIL(VV_E_ARG, "vv_spatial_chain_front_c320: bad dtype");
    VV_CHECK_LAUNCH("vv_spatial_chain_front_c320");
    return VV_OK;
}
