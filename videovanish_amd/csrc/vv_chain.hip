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
// which GELU the fused kernel evaluates: the A&S form above.  gelu_poly2 (vv_common.h: packed fp32 polynomial, no v_rcp / v_exp -- the GEMM kernels' GEGLU epilogue since
// round 6, +4.5..6.6 % there) was measured here too (-DVV_GELU2_POLY, lab): the motion module LOSES 3 % (3.04 -> 3.13 ms), the chain tail is unchanged -- these kernels
// run one or two waves per SIMD beside the matrix pipe, the transcendental unit is otherwise idle and the polynomial's 14 extra packed FMAs are not (profiles/r6_gelu_ab.txt)
#ifdef VV_GELU2_POLY
#define VV_GELU2 gelu_poly2
#else
#define VV_GELU2 gelu2
#endif

#ifndef VV_CHAIN_FORM
#define VV_CHAIN_FORM 1    // 1 = row-split pairs (chain_rs_c320_kernel, packing layout "rowsplit": the product form); lab builds: 2 = column-split (layout "columns"), 0 = token-split forms (layout "tokens")
#endif
#if VV_CHAIN_FORM != 1
#include "vv_chain_lab.h"      // the lab forms (token-split, column-split): not part of the product library
#endif

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
#define VV_WAIT6 asm volatile("s_waitcnt vmcnt(6)" ::: "memory")

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
        glds16_asm(sbase + (int64_t)issued * SLAB, ring + islot * SLAB + wave * 1024);
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
            const float* trow = p.t_in + row * CC;
            // o: row-major [M][320], or head-major [frame][head][token][40] (o_hw tokens per frame; what vv_attention stores as whole contiguous
            // 80-byte records): channel c of the row = head c / 40, d = c % 40 -- every 4-channel piece lies inside one head
            const unsigned short* obase = (const unsigned short*)p.o;
            int64_t ofr = row * CC, otk = 0;
            if (p.o_hw > 0) { const int64_t fr = row / p.o_hw; ofr = fr * p.o_hw * CC; otk = (row - fr * p.o_hw) * CD; }
            auto opiece = [&](const int c0) -> uint2 {      // channels c0 + 4 lg .. + 3
                const int c = c0 + 4 * lg;
                if (p.o_hw > 0) { const int hd = c / CD; return *(const uint2*)(obase + ofr + (int64_t)hd * p.o_hw * CD + otk + (c - hd * CD)); }
                return *(const uint2*)(obase + ofr + c);
            };
#pragma unroll
            for (int kt = 0; kt < 5; ++kt) {
                const uint2 l0 = opiece(64 * kt), h0 = opiece(64 * kt + 16);
                const uint2 l1 = opiece(64 * kt + 32), h1 = opiece(64 * kt + 48);
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
                const vv_f32x2 g01 = VV_GELU2((vv_f32x2){g[2 * i + 1][tt][0] + bg.x, g[2 * i + 1][tt][1] + bg.y});
                const vv_f32x2 g23 = VV_GELU2((vv_f32x2){g[2 * i + 1][tt][2] + bg.z, g[2 * i + 1][tt][3] + bg.w});
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
    // Block residual: ALL 20 float4 of x (and of res1) are requested before the first store.  Read in place -- load, s_waitcnt vmcnt(0), store, per tile --
    // every wait also drained the previous tile's store: 20 serialised memory round trips at the end of every block (round 5, second session; the same
    // finding as in vv_gemm_epilogue.h).  The activation row registers are dead here, the 80 extra registers are free.
    float4 xr[2][10];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        const int64_t r = row0 + tt * 16 + li;
        const int64_t row = (r < p.M ? r : 0) * CC;
#pragma unroll
        for (int j = 0; j < 10; ++j) xr[tt][j] = r < p.M ? *(const float4*)(p.x + row + chan(j)) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (p.res1) {
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int64_t r = row0 + tt * 16 + li;
            const int64_t row = (r < p.M ? r : 0) * CC;
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                const float4 r4 = r < p.M ? *(const float4*)(p.res1 + row + chan(j)) : make_float4(0.f, 0.f, 0.f, 0.f);
                xr[tt][j].x += r4.x; xr[tt][j].y += r4.y; xr[tt][j].z += r4.z; xr[tt][j].w += r4.w;
            }
        }
    }
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        const int64_t r = row0 + tt * 16 + li;
        if (r < p.M) {
            const int64_t row = r * CC;
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                const int c = chan(j);
                const float4 b = *(const float4*)(prm + Q_BOUT + c);
                const float v0 = t[j][tt][0] + b.x + xr[tt][j].x, v1 = t[j][tt][1] + b.y + xr[tt][j].y, v2 = t[j][tt][2] + b.z + xr[tt][j].z, v3 = t[j][tt][3] + b.w + xr[tt][j].w;
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

// ROW-SPLIT pair form of the block front (round 5; design: chain_rs_c320_kernel above).  All 100 slabs are dense layers, so every slab is read as two row
// tiles per wave (4 fragment reads feed 8 MFMAs) at two waves per SIMD; the partners swap the LayerNorm output once.  Same stream as the 4 x 32 form.
template <typename T>
__global__ __launch_bounds__(512, 2) void chain_front_rs_c320_kernel(const vv_chain_front_params p) {
    __shared__ __attribute__((aligned(1024))) unsigned char ring[RS_NS * SLAB];
    __shared__ __attribute__((aligned(16))) unsigned char xbuf[RS_XBUF];
    __shared__ __attribute__((aligned(16))) float sbuf[8 * 64 * 4];
    __shared__ __attribute__((aligned(16))) float prm[F_TOTAL];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int grp = wave & 3, hf = wave >> 2, pw = wave ^ 4;
    const bool hi = hf != 0;
    const int64_t row0 = (int64_t)blockIdx.x * 128 + grp * 32;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};

    for (int i = tid * 4; i < F_TOTAL; i += 512 * 4) *(float4*)(prm + i) = *(const float4*)(p.params + i);
    const unsigned char* sbase = (const unsigned char*)p.stream + wave * 1024 + lane * 16;
    int issued = 0, islot = 0, cslot = 0;
    auto issue = [&]() {
        glds16_asm(sbase + (int64_t)issued * SLAB, ring + islot * SLAB + wave * 1024);
        ++issued;
        islot = islot + 1 == RS_NS ? 0 : islot + 1;
    };
    using BODY = std::false_type; using TAIL = std::true_type;
    using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
    auto sync = [&](auto ni_tag, auto tail_tag) {
        constexpr int NI = decltype(ni_tag)::value;
        if constexpr (decltype(tail_tag)::value) {
            if (issued + NI <= NF_SLABS) {
#pragma unroll
                for (int i = 0; i < NI; ++i) issue();
                asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            } else {
                while (issued < NF_SLABS) issue();
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        } else {
#pragma unroll
            for (int i = 0; i < NI; ++i) issue();
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
    };
    auto slab = [&]() -> const unsigned char* {
        const unsigned char* s = ring + cslot * SLAB;
        cslot = cslot + 1 == RS_NS ? 0 : cslot + 1;
        return s;
    };
    auto meet = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); };
    auto frag = [&](const f32x4& lo, const f32x4& hi_) -> uint4 {
        return make_uint4(pack2<T>(lo[0], lo[1]), pack2<T>(lo[2], lo[3]), pack2<T>(hi_[0], hi_[1]), pack2<T>(hi_[2], hi_[3]));
    };
    auto sel = [&](const uint4& a_, const uint4& b_) -> uint4 { return hi ? b_ : a_; };
    struct WF2 { uint4 w[2][2]; };
    const int rs_off = hf * 4096 + li * 128, sw = li & 7;
    auto load_rs = [&](const unsigned char* s, WF2& f) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int off = ((kk * 4 + lg) ^ sw) << 4;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) f.w[kk][rt] = *(const uint4*)(s + rs_off + rt * 2048 + off);
        }
    };
    uint4 a0[5][2], a1[5][2];
    // N slabs (5 k tiles per 64-row block), a step in front of every pair: acc_of(i)[rt][tt] += own row tiles x the full activation row
    auto group_rs = [&](auto n_tag, auto&& acc_of, auto tail) {
        constexpr int N = decltype(n_tag)::value;
        WF2 f[2];
        if constexpr (N >= 2) sync(I2{}, tail); else sync(I1{}, tail);
        load_rs(slab(), f[0]);
#pragma unroll
        for (int i = 0; i < N; ++i) {
            if (i + 1 < N) {
                if (((i + 1) & 1) == 0) { if (i + 2 < N) sync(I2{}, tail); else sync(I1{}, tail); }
                load_rs(slab(), f[(i + 1) & 1]);
            }
            f32x4* acc = acc_of(i);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) acc[rt * 2 + tt] = T::mfma(f[i & 1].w[kk][rt], kk ? a1[i % 5][tt] : a0[i % 5][tt], acc[rt * 2 + tt]);
            if (i + 1 < N) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
        }
    };
    using N5 = std::integral_constant<int, 5>; using N25 = std::integral_constant<int, 25>;
    auto chan = [&](const int j) -> int { return 64 * (j >> 1) + 32 * hf + 16 * (j & 1) + 4 * lg; };

    unsigned char* const xmine = xbuf + wave * 5120 + lane * 16;
    const unsigned char* const xpart = xbuf + pw * 5120 + lane * 16;
    // full swap of the partners' halves: a0 / a1 <- (own k steps, partner's) for both token tiles (two rounds through the 40 KB buffer)
    auto swap_full = [&](const uint4 (&own)[5][2]) {
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            if (tt) meet();
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
    // ---- x -> GroupNorm apply (scale / shift of the token's own frame) -> activation fragments.  Each wave converts ITS k steps (2 rb + hf) of both
    //      token tiles and the partners swap (no duplicate reads of x); the per-frame affine rows the block touches are staged in LDS first
    int64_t rows[2];
    {
        const int64_t r_lo = (int64_t)blockIdx.x * 128, r_hi = (r_lo + 127 < p.M ? r_lo + 127 : p.M - 1);
        const int f_lo = (int)(r_lo / p.HW), nfr = (int)(r_hi / p.HW) - f_lo + 1;      // <= 16 frames (launcher)
        float* affs = (float*)xbuf;
        for (int i = tid * 4; i < nfr * 2 * CC; i += 512 * 4) *(float4*)(affs + i) = *(const float4*)(p.gn_affine + (int64_t)f_lo * (2 * CC) + i);
        __syncthreads();       // affine rows and parameter block visible
        uint4 own[5][2];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            int64_t row = row0 + tt * 16 + li;
            if (row >= p.M) row = p.M - 1;
            rows[tt] = row;
            const float* xrow = p.x + row * CC;
            const float* aff = affs + ((int)(row / p.HW) - f_lo) * (2 * CC);
#pragma unroll
            for (int rb = 0; rb < 5; ++rb) {
                const int c0 = 64 * rb + 32 * hf + 4 * lg, c1 = c0 + 16;
                const float4 x0 = *(const float4*)(xrow + c0), x1 = *(const float4*)(xrow + c1);
                const float4 g0 = *(const float4*)(aff + c0), b0 = *(const float4*)(aff + CC + c0);
                const float4 g1 = *(const float4*)(aff + c1), b1 = *(const float4*)(aff + CC + c1);
                own[rb][tt] = make_uint4(pack2<T>(x0.x * g0.x + b0.x, x0.y * g0.y + b0.y), pack2<T>(x0.z * g0.z + b0.z, x0.w * g0.w + b0.w),
                                         pack2<T>(x1.x * g1.x + b1.x, x1.y * g1.y + b1.y), pack2<T>(x1.z * g1.z + b1.z, x1.w * g1.w + b1.w));
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll 1
        for (int i = 0; i < RS_AH; ++i) issue();
        meet();                // everybody has read its affine rows: the buffer turns into the exchange buffer
        swap_full(own);
    }
    // ---- proj_in: t = Win a + bin (own channels), stored for the tail kernel and the residual
    f32x4 t[10][2];
#pragma unroll
    for (int j = 0; j < 10; ++j)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) t[j][tt] = z4;
    group_rs(N25{}, [&](int i) { return &t[(i / 5) * 2][0]; }, BODY{});
#pragma unroll
    for (int j = 0; j < 10; ++j) {
        const float4 b = *(const float4*)(prm + F_BIN + chan(j));
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) { t[j][tt][0] += b.x; t[j][tt][1] += b.y; t[j][tt][2] += b.z; t[j][tt][3] += b.w; }
    }
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        if (row0 + tt * 16 + li < p.M) {
            float* trow = p.t_out + rows[tt] * CC;
#pragma unroll
            for (int j = 0; j < 10; ++j) *(float4*)(trow + chan(j)) = make_float4(t[j][tt][0], t[j][tt][1], t[j][tt][2], t[j][tt][3]);
        }
    }
    // ---- LN1 (statistics merged with the partner's) -> own k steps -> swap -> full rows
    {
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
        uint4 own[5][2];
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
                    const float4 g = *(const float4*)(prm + F_LN1G + c), b = *(const float4*)(prm + F_LN1B + c);
                    y[h][0] = (t[j][tt][0] - mean) * rstd * g.x + b.x; y[h][1] = (t[j][tt][1] - mean) * rstd * g.y + b.y;
                    y[h][2] = (t[j][tt][2] - mean) * rstd * g.z + b.z; y[h][3] = (t[j][tt][3] - mean) * rstd * g.w + b.w;
                }
                own[rb][tt] = frag(y[0], y[1]);
            }
        }
        swap_full(own);
    }
    // ---- fused q | k | v projection: 15 blocks of 64 output channels (this wave: 32 of them), stored head-major
    unsigned short* qkv = (unsigned short*)p.qkv;
    int64_t tokbase[2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        const int64_t fr = rows[tt] / p.HW, tk = rows[tt] - fr * p.HW;
        tokbase[tt] = fr * (3 * (int64_t)p.HW * CC) + tk * CD;
    }
    // Stores of the QKV phase: block by block (4 stores per wave and 64-channel block).  The kernel writes 1.574 GB per 32-frame launch against 1.475 GB algorithmic (t fp32 0.590 GB +
    // qkv 0.885 GB): 1.07x -- round 5's "1.78x" divided by the QKV bytes alone.  Round 6 built a one-burst form anyway (one `which` = q, k or v = 5 blocks held packed in 40 registers
    // and stored together; lab form, -DVV_FRONT_BURST) and measured it 3 % SLOWER with no fewer bytes written (profiles/r6_front_store_ab.txt): the product keeps this form.
#ifdef VV_FRONT_BURST
    auto qkv_which = [&](const int which, auto tail) {
        uint2 hold[5][2][2];
#pragma unroll
        for (int b5 = 0; b5 < 5; ++b5) {
            f32x4 acc[2][2] = {{z4, z4}, {z4, z4}};
            if (b5 < 2) group_rs(N5{}, [&](int) { return &acc[0][0]; }, BODY{});      // (the stream's last 3 blocks are the only ones that can run out of slabs)
            else group_rs(N5{}, [&](int) { return &acc[0][0]; }, tail);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) hold[b5][rt][tt] = make_uint2(pack2<T>(acc[rt][tt][0], acc[rt][tt][1]), pack2<T>(acc[rt][tt][2], acc[rt][tt][3]));
        }
        const int64_t wbase = (int64_t)which * CC * p.HW;
#pragma unroll
        for (int b5 = 0; b5 < 5; ++b5)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const int cc = 64 * b5 + 32 * hf + 16 * rt + 4 * lg, head = cc / CD, d = cc - head * CD;
                const int64_t off = wbase + (int64_t)head * CD * p.HW + d;
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
                    if (row0 + tt * 16 + li < p.M) *(uint2*)(qkv + tokbase[tt] + off) = hold[b5][rt][tt];
            }
    };
#pragma unroll 1
    for (int which = 0; which < 2; ++which) qkv_which(which, BODY{});
    qkv_which(2, TAIL{});
#else
    auto qkv_block = [&](const int rb, auto tail) {
        f32x4 acc[2][2] = {{z4, z4}, {z4, z4}};
        group_rs(N5{}, [&](int) { return &acc[0][0]; }, tail);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const int c = 64 * rb + 32 * hf + 16 * rt + 4 * lg;
            const int which = c / CC, cc = c - which * CC, head = cc / CD, d = cc - head * CD;
            const int64_t off = ((int64_t)which * CC + (int64_t)head * CD) * p.HW + d;
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
                if (row0 + tt * 16 + li < p.M)
                    *(uint2*)(qkv + tokbase[tt] + off) = make_uint2(pack2<T>(acc[rt][tt][0], acc[rt][tt][1]), pack2<T>(acc[rt][tt][2], acc[rt][tt][3]));
        }
    };
#pragma unroll 1
    for (int rb = 0; rb < 12; ++rb) qkv_block(rb, BODY{});
    qkv_block(12, TAIL{}); qkv_block(13, TAIL{}); qkv_block(14, TAIL{});
#endif
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
    const int want_slabs = VV_CHAIN_FORM == 2 ? 4 * 870 : N_SLABS;
    if (VV_CHAIN_FORM != 1 && p.o_hw) VV_FAIL(VV_E_UNSUPPORTED, "vv_spatial_chain_c320: the lab forms read o row-major only");
    if (p.o_hw < 0 || (p.o_hw > 0 && p.M % p.o_hw)) VV_FAIL(VV_E_ARG, "vv_spatial_chain_c320: o_hw must be 0 (row-major o) or divide M (head-major o)");
    if (p.layout != VV_CHAIN_FORM) VV_FAIL(VV_E_ARG, "vv_spatial_chain_c320: weight stream packed in layout %d, this library's kernel consumes layout %d (VV_CHAIN_LAYOUT_*)", p.layout, (int)VV_CHAIN_FORM);
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
#ifndef VV_FRONT_FORM
#define VV_FRONT_FORM 1      // 1 = row-split pairs (chain_front_rs_c320_kernel); 0 = 4 waves x 32 tokens (round 3; same stream)
#endif
#if VV_FRONT_FORM == 1
    if (127 / p.HW + 2 <= 16) {      // the block's per-frame affine rows fit the staging buffer (always, beyond toy frame sizes)
        if (dtype == VV_BF16) hipLaunchKernelGGL(chain_front_rs_c320_kernel<BF16>, dim3((unsigned)nblk), dim3(512), 0, st, p);
        else if (dtype == VV_F16) hipLaunchKernelGGL(chain_front_rs_c320_kernel<F16>, dim3((unsigned)nblk), dim3(512), 0, st, p);
        else VV_FAIL(VV_E_ARG, "vv_spatial_chain_front_c320: bad dtype");
        VV_CHECK_LAUNCH("vv_spatial_chain_front_c320");
        return VV_OK;
    }
    if (dtype == VV_BF16) hipLaunchKernelGGL(chain_front_c320_kernel<BF16>, dim3((unsigned)nblk), dim3(256), 0, st, p);
    else if (dtype == VV_F16) hipLaunchKernelGGL(chain_front_c320_kernel<F16>, dim3((unsigned)nblk), dim3(256), 0, st, p);
#else
    if (dtype == VV_BF16) hipLaunchKernelGGL(chain_front_c320_kernel<BF16>, dim3((unsigned)nblk), dim3(256), 0, st, p);
    else if (dtype == VV_F16) hipLaunchKernelGGL(chain_front_c320_kernel<F16>, dim3((unsigned)nblk), dim3(256), 0, st, p);
#endif
    else VV_FAIL(VV_E_ARG, "vv_spatial_chain_front_c320: bad dtype");
    VV_CHECK_LAUNCH("vv_spatial_chain_front_c320");
    return VV_OK;
}
