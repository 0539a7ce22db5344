// K5 (fused form): one kernel for a whole AnimateDiff temporal transformer ("motion module") at C = 320, F = 32 frames:
//   out = x + res1 + proj_out( FF( A2( A1( proj_in( GN(x) ) ) ) ) ),   A(t) = t + out_proj(attn_over_frames(LN(t) + pe)),
//   FF(t) = t + W2 (GEGLU(W1 LN(t)))                      (oracle: oracle/model_ref.py::motion_module)
// The unfused path (nn.MotionModule) runs 17 kernels and moves 17.7 GB of intermediates through HBM for 2.08 TFLOP at level 0
// (DESIGN.md 5); every operation of the block is per PIXEL over its 32 frames, so here ONE WAVE OWNS ONE PIXEL END TO END:
//   * the fp32 trunk [32 tokens x 320 channels] lives in 160 accumulator registers for the whole kernel, the current h16
//     activations in 80 registers laid out as MFMA B-operand fragments; a block = 4 waves = 4 adjacent pixels, one wave per SIMD
//     with the full 512-entry register file;
//   * every GEMM is D = W * X^T (swapped operands): the accumulator of one layer (lane = token, registers = 4 consecutive
//     channels) packs straight into the B fragments of the next layer -- the k order this implies (PERM32 below) is applied to
//     the weights once, when the host packs them -- so activations never leave registers: LayerNorm is an in-lane sum + 2
//     cross-lane adds, the 32 x 32 attention of a (pixel, head) is 8 + 6 MFMAs on fragments built from the QKV accumulators
//     (V through the un-swapped product, which yields V^T in A-operand layout directly), GEGLU output feeds FF2 from registers;
//   * only the WEIGHTS move: the module's 670 weight slabs ([64 rows x 64 k] h16, 8 KB, pre-swizzled, in consumption order:
//     packing.pack_motion_stream) stream through an 8-slot LDS ring by LDS-DMA (inline asm, counted vmcnt, raw s_barrier: 6 slabs
//     in flight across the barriers), shared by the 4 waves: 5.5 MB per 128 tokens = 1.07 ms per module at the measured 15 TB/s
//     L2->LDS rate -- the kernel's floor next to 0.83 ms of MFMA time at peak;
//   * small parameters (biases, LayerNorm affine, the sinusoidal table, the per-call GroupNorm scale/shift) sit in 68 KB of LDS.
// Fragment conventions (v_mfma_f32_16x16x32, lane = (li = lane & 15, lg = lane >> 4)):
//   A[m][k]: lane holds row m = li, k = 8 lg .. 8 lg + 7;  B[k][n]: column n = li, same k;  D[m][n]: n = li, m = 4 lg + r (r = 0..3).
//   PERM32: position p = 8 lg + e of a 32-wide k step holds logical index 16 (e >> 2) + 4 lg + (e & 3)  (two D tiles -> one operand).
#include <type_traits>
#include "vv_common.h"

namespace {

[[maybe_unused]] constexpr int MC = 320, MF = 32, MH = 8, MD = 40;
constexpr int NSLOT = 10, AHEAD = 6, SLAB = 8192;     // slabs are consumed in PAIRS: one wait + barrier per 16 KB
// fp32 parameter block (floats): offsets
constexpr int P_GN_A = 0, P_GN_B = 320, P_BIN = 640, P_LN1G = 960, P_LN1B = 1280, P_BO1 = 1600, P_LN2G = 1920, P_LN2B = 2240, P_BO2 = 2560,
              P_LN3G = 2880, P_LN3B = 3200, P_B1 = 3520, P_B2 = 6080, P_BOUT = 6400, P_PE = 6720, P_TOTAL = 6720 + MF * MC;   // 16960 floats
constexpr int N_SLABS = 25 + 2 * MH * 20 + 20 * 15 + 25;      // 670

__device__ __forceinline__ void glds16_asm(const void* gptr, void* lds_wave_base) {
    typedef void __attribute__((address_space(3))) * lp_t;
    const unsigned dst = (unsigned)(size_t)(lp_t)lds_wave_base;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gptr), "s"(dst) : "memory");
}

// exact (erf) GELU through Abramowitz-Stegun 7.1.26 (|erf error| <= 1.5e-7, far below the h16 rounding that follows): 2 transcendentals
// + ~12 VALU instead of the ~30 of erff -- with ONE wave per SIMD the activation is not hidden behind another wave's MFMAs
__device__ __forceinline__ float gelu_fast(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float e = poly * __builtin_amdgcn_exp2f(-z * z * 1.4426950408889634f);      // = 1 - erf(z)
    const float erf_abs = 1.0f - e;
    return 0.5f * x * (1.0f + copysignf(erf_abs, x));
}

// two GELUs at once on packed fp32 math (v_pk_fma_f32)
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
    const vv_f32x2 erfc = q * e;                                       // 1 - erf(|x| / sqrt 2)
    return __builtin_elementwise_fma(ax * 0.5f, (vv_f32x2){1.0f, 1.0f} - erfc, x * 0.5f);      // 0.5 x (1 + sign(x) (1 - erfc))
}
// which GELU the fused kernel evaluates: the A&S form above.  gelu_poly2 (vv_common.h: packed fp32 polynomial, no v_rcp / v_exp -- the GEMM kernels' GEGLU epilogue since
// round 6, +4.5..6.6 % there) was measured here too (-DVV_GELU2_POLY, lab): the motion module LOSES 3 % (3.04 -> 3.13 ms), the chain tail is unchanged -- these kernels
// run one or two waves per SIMD beside the matrix pipe, the transcendental unit is otherwise idle and the polynomial's 14 extra packed FMAs are not (profiles/r6_gelu_ab.txt)
#ifdef VV_GELU2_POLY
#define VV_GELU2 gelu_poly2
#else
#define VV_GELU2 gelu2
#endif

template <typename T>
__global__ __launch_bounds__(256, 1) void motion_c320_kernel(const vv_motion_params p) {
    __shared__ __attribute__((aligned(1024))) unsigned char ring[NSLOT * SLAB];
    __shared__ __attribute__((aligned(16))) float prm[P_TOTAL];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int pixel = blockIdx.x * 4 + wave;
    const int64_t HW = p.HW;

    // ---- parameter block -> LDS (plain loads: 16960 floats per block), then the ring prologue
    for (int i = tid * 4; i < P_TOTAL; i += 256 * 4) {
        const float4 v = i < 640 ? *(const float4*)(p.gn_affine + i) : *(const float4*)(p.params + (i - 640));
        *(float4*)(prm + i) = v;
    }
    // ---- weight stream: slab s of the module at p.stream + s * SLAB; each wave copies 2 KB of every slab
    const unsigned char* sbase = (const unsigned char*)p.stream + (wave * 2) * 1024 + lane * 16;
    int issued = 0, consumed = 0;
    auto issue = [&]() {
        unsigned char* dst = ring + (issued % NSLOT) * SLAB + (wave * 2) * 1024;
        const unsigned char* src = sbase + (int64_t)issued * SLAB;
        glds16_asm(src, dst);
        glds16_asm(src + 1024, dst + 1024);
        ++issued;
    };
    // next slab of the stream, ready to be read by every wave of the block.  Slabs are synchronised in PAIRS (N_SLABS is even): the EVEN slab
    // issues two more slabs, waits until all but the newest AHEAD have landed (this wave's share) and joins the barrier (everybody's share has
    // landed, everybody is done with the pair before the previous one: the ring keeps AHEAD + 4 slots); the odd one just advances.  The parity of a
    // slab's stream index is a compile-time property of its call site (`even_tag`; a run-time test would cut the instruction stream into one basic
    // block per slab); `tail_tag`: only the last group of the stream (proj_out) can run out of slabs to issue, everywhere else the issue is unconditional.
    auto next_slab = [&](auto even_tag, auto tail_tag) -> const unsigned char* {
        if constexpr (decltype(even_tag)::value) {
            if (!decltype(tail_tag)::value || issued < N_SLABS) { issue(); issue(); asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); }     // 2 DMA x AHEAD slabs may stay in flight
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        const unsigned char* s = ring + (consumed % NSLOT) * SLAB;
        ++consumed;
        return s;
    };

    // ---- x -> GroupNorm apply -> activation fragments a[ks][tt] (token li + 16 tt; k positions per PERM32)
    uint4 a[10][2];
    f32x4 t[20][2];
    {
        __syncthreads();       // parameter block visible
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int fr = min(tt * 16 + li, p.F - 1);                       // clips shorter than 32 frames: the rows past F repeat the last frame (masked as keys, never stored)
            const float* xrow = p.x + ((int64_t)fr * HW + pixel) * MC;
#pragma unroll
            for (int s = 0; s < 10; ++s) {
                const int c0 = 32 * s + 4 * lg, c1 = c0 + 16;
                const float4 x0 = *(const float4*)(xrow + c0), x1 = *(const float4*)(xrow + c1);
                const float4 a0 = *(const float4*)(prm + P_GN_A + c0), b0 = *(const float4*)(prm + P_GN_B + c0);
                const float4 a1 = *(const float4*)(prm + P_GN_A + c1), b1 = *(const float4*)(prm + P_GN_B + c1);
                a[s][tt] = make_uint4(pack2<T>(x0.x * a0.x + b0.x, x0.y * a0.y + b0.y), pack2<T>(x0.z * a0.z + b0.z, x0.w * a0.w + b0.w),
                                      pack2<T>(x1.x * a1.x + b1.x, x1.y * a1.y + b1.y), pack2<T>(x1.z * a1.z + b1.z, x1.w * a1.w + b1.w));
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll 1
        for (int i = 0; i < AHEAD; ++i) issue();
    }

    // D += W_slab * X^T for RT row tiles of one [RT*16 rows x 64 k] slab, split into the fragment reads (LDS -> registers) and the MFMAs so that
    // a group of slabs runs software pipelined: the reads of slab i+1 are in flight under the MFMAs of slab i (with ONE wave per SIMD and the
    // four waves of a block released by the same barrier, reads waited for right before their MFMAs leave the matrix pipe idle for the whole
    // LDS round trip on every slab).  TR: operands exchanged, D = X * W^T (lane = output channel li, registers = tokens 4 lg + r): V^T.
    struct WF { uint4 w[2][4]; };
    auto slab_load = [&](const unsigned char* s, auto rt_tag, WF& f) {
        constexpr int RT = decltype(rt_tag)::value;
        const int sw = li & 7;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int off = ((kk * 4 + lg) ^ sw) << 4;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) f.w[kk][rt] = *(const uint4*)(s + (rt * 16 + li) * 128 + off);
        }
    };
    auto slab_fma = [&](const WF& f, auto rt_tag, auto tr_tag, f32x4* acc /* [RT][2] */, const uint4 (&x0)[2], const uint4 (&x1)[2]) {
        constexpr int RT = decltype(rt_tag)::value;
        constexpr bool TR = decltype(tr_tag)::value;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
                    acc[rt * 2 + tt] = TR ? T::mfma(kk ? x1[tt] : x0[tt], f.w[kk][rt], acc[rt * 2 + tt]) : T::mfma(f.w[kk][rt], kk ? x1[tt] : x0[tt], acc[rt * 2 + tt]);
    };
    using I3 = std::integral_constant<int, 3>; using I4 = std::integral_constant<int, 4>;
    using EVEN = std::true_type; using ODD = std::false_type; using BODY = std::false_type; using TAIL = std::true_type;
    using PLAIN = std::false_type; using TRANSP = std::true_type;
    // a group of N slabs of the same shape whose first slab has stream-index parity P0 (0 = even)
    auto slab_group = [&](auto p0_tag, auto n_tag, auto rt_tag, auto tr_tag, auto&& acc_of, auto&& x0_of, auto&& x1_of, auto tail) {
        constexpr int P0 = decltype(p0_tag)::value, N = decltype(n_tag)::value, RT = decltype(rt_tag)::value;
        WF f[2];
        slab_load(next_slab(std::bool_constant<P0 == 0>{}, tail), rt_tag, f[0]);
#pragma unroll
        for (int i = 0; i < N; ++i) {
            if (i + 1 < N) {
                if (((P0 + i + 1) & 1) == 0) slab_load(next_slab(EVEN{}, tail), rt_tag, f[(i + 1) & 1]);
                else slab_load(next_slab(ODD{}, tail), rt_tag, f[(i + 1) & 1]);
            }
            slab_fma(f[i & 1], rt_tag, tr_tag, acc_of(i), x0_of(i), x1_of(i));
            if (i + 1 < N) __builtin_amdgcn_sched_group_barrier(0x100, 2 * RT, 0);      // next slab's reads first ...
            __builtin_amdgcn_sched_group_barrier(0x008, 4 * RT, 0);                     // ... then this slab's MFMAs
            // (3.08 -> 3.025 ms at level 0, profiles/r5_chain_forms.txt section 5)
            // nothing crosses into the next slab's region (round 5): without this fence hipcc fills the MFMA group with the MFMAs of the slab whose reads
            // it has just issued -- the group barriers order instruction types, not instances -- and the fragment double buffer collapses into one
            // register set (profiles/r5_chain_forms.txt, section 1)
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    using P0E = std::integral_constant<int, 0>; using P0O = std::integral_constant<int, 1>;
    using N5 = std::integral_constant<int, 5>; using N10 = std::integral_constant<int, 10>; using N25 = std::integral_constant<int, 25>;
    // full-width layer: t[20][2] (+)= W [320 x 320] * a   (5 row blocks x 5 k tiles = 25 slabs)
    auto dense320 = [&](auto p0_tag, f32x4 (&acc)[20][2], auto tail) {
        slab_group(p0_tag, N25{}, I4{}, PLAIN{}, [&](int i) { return &acc[(i / 5) * 4][0]; }, [&](int i) -> const uint4 (&)[2] { return a[2 * (i % 5)]; },
                   [&](int i) -> const uint4 (&)[2] { return a[2 * (i % 5) + 1]; }, tail);
    };
    auto frag = [&](const f32x4& lo, const f32x4& hi) -> uint4 {
        return make_uint4(pack2<T>(lo[0], lo[1]), pack2<T>(lo[2], lo[3]), pack2<T>(hi[0], hi[1]), pack2<T>(hi[2], hi[3]));
    };
    auto add_bias = [&](const int off) {
#pragma unroll
        for (int j = 0; j < 20; ++j) {
            const float4 b = *(const float4*)(prm + off + 16 * j + 4 * lg);
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) { t[j][tt][0] += b.x; t[j][tt][1] += b.y; t[j][tt][2] += b.z; t[j][tt][3] += b.w; }
        }
    };
    // a = h16( LN(t) * g + b (+ pe[frame]) ) as fragments
    auto layer_norm = [&](const int goff, const int boff, const bool with_pe) {
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 20; ++j) s += (t[j][tt][0] + t[j][tt][1]) + (t[j][tt][2] + t[j][tt][3]);
            s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);
            const float mean = s * (1.0f / MC);
            float q = 0.f;
#pragma unroll
            for (int j = 0; j < 20; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float d = t[j][tt][r] - mean; q += d * d; }
            q += __shfl_xor(q, 16); q += __shfl_xor(q, 32);
            const float rstd = rsqrtf(q * (1.0f / MC) + 1e-5f);
            const float* pe = prm + P_PE + (tt * 16 + li) * MC;
#pragma unroll
            for (int s2 = 0; s2 < 10; ++s2) {
                f32x4 y[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int j = 2 * s2 + h, c = 16 * j + 4 * lg;
                    const float4 g = *(const float4*)(prm + goff + c), b = *(const float4*)(prm + boff + c);
                    y[h][0] = (t[j][tt][0] - mean) * rstd * g.x + b.x; y[h][1] = (t[j][tt][1] - mean) * rstd * g.y + b.y;
                    y[h][2] = (t[j][tt][2] - mean) * rstd * g.z + b.z; y[h][3] = (t[j][tt][3] - mean) * rstd * g.w + b.w;
                    if (with_pe) { const float4 e = *(const float4*)(pe + c); y[h][0] += e.x; y[h][1] += e.y; y[h][2] += e.z; y[h][3] += e.w; }
                }
                a[s2][tt] = frag(y[0], y[1]);
            }
        }
    };
    // t += out_proj( attention over the 32 frames of this pixel ( a ) ), 8 heads, 20 slabs per head
    const float sc = 0.15811388300841897f * 1.4426950408889634f;      // 40^-1/2 * log2(e)
    auto attention = [&](const int bias_off) {
#pragma unroll 1
        for (int h = 0; h < MH; ++h) {
            f32x4 qa[3][2], ka[3][2], va[3][2];          // [d tile][token tile]; va: [d tile][token tile] un-swapped (lane = d)
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) { qa[i][tt] = f32x4{0.f, 0.f, 0.f, 0.f}; ka[i][tt] = f32x4{0.f, 0.f, 0.f, 0.f}; va[i][tt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            // (a head is 20 slabs and starts at an odd stream index: 25 + 160 A + 20 h)
            auto ak0 = [&](int i) -> const uint4 (&)[2] { return a[2 * i]; };
            auto ak1 = [&](int i) -> const uint4 (&)[2] { return a[2 * i + 1]; };
            slab_group(P0O{}, N5{}, I3{}, PLAIN{}, [&](int) { return &qa[0][0]; }, ak0, ak1, BODY{});
            slab_group(P0E{}, N5{}, I3{}, PLAIN{}, [&](int) { return &ka[0][0]; }, ak0, ak1, BODY{});
            slab_group(P0O{}, N5{}, I3{}, TRANSP{}, [&](int) { return &va[0][0]; }, ak0, ak1, BODY{});
            const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
            uint4 qf[2][2], kf[2][2];                     // [token tile][k step]
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                qf[tt][0] = frag(qa[0][tt], qa[1][tt]); qf[tt][1] = frag(qa[2][tt], z4);
                kf[tt][0] = frag(ka[0][tt], ka[1][tt]); kf[tt][1] = frag(ka[2][tt], z4);
            }
            // S^T[key tile][query tile] = K Q^T
            f32x4 sT[2][2];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) {
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                    acc = T::mfma(kf[kt][0], qf[qt][0], acc);
                    acc = T::mfma(kf[kt][1], qf[qt][1], acc);
                    sT[kt][qt] = acc;
                }
            // softmax over the 32 keys of every query (query = lane column li of tile qt): 8 values in the lane, 4 lane groups
            uint4 pf[2];
            float inv[2];
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                // keys (frames) past the end of a short clip: selects, not a branch (a run-time branch here would cut the head into basic blocks)
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sT[kt][qt][r] = (kt * 16 + 4 * lg + r >= p.F) ? -1e30f : sT[kt][qt][r];
                float m = fmaxf(fmaxf(fmaxf(sT[0][qt][0], sT[0][qt][1]), fmaxf(sT[0][qt][2], sT[0][qt][3])),
                                fmaxf(fmaxf(sT[1][qt][0], sT[1][qt][1]), fmaxf(sT[1][qt][2], sT[1][qt][3])));
                m = fmaxf(m, __shfl_xor(m, 16)); m = fmaxf(m, __shfl_xor(m, 32));
                const float mc = m * sc;
                float l = 0.f;
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { const float e = __builtin_amdgcn_exp2f(sT[kt][qt][r] * sc - mc); sT[kt][qt][r] = e; l += e; }
                l += __shfl_xor(l, 16); l += __shfl_xor(l, 32);
                inv[qt] = 1.0f / l;
                pf[qt] = frag(sT[0][qt], sT[1][qt]);
            }
            // O^T[d tile][query tile] = V^T P^T (one k step = the 32 keys), then / l
            uint4 of[2][2];                               // [token tile][k step] fragments of the out projection's input
            f32x4 oT[3][2];
#pragma unroll
            for (int dt = 0; dt < 3; ++dt) {
                const uint4 vf = frag(va[dt][0], va[dt][1]);
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) {
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                    acc = T::mfma(vf, pf[qt], acc);
                    oT[dt][qt] = acc * inv[qt];
                }
            }
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) { of[tt][0] = frag(oT[0][tt], oT[1][tt]); of[tt][1] = frag(oT[2][tt], z4); }
            // t += Wo[:, head] * O   (5 row blocks, one k tile: d padded 40 -> 64)
            const uint4 o0[2] = {of[0][0], of[1][0]}, o1[2] = {of[0][1], of[1][1]};
            slab_group(P0E{}, N5{}, I4{}, PLAIN{}, [&](int i) { return &t[i * 4][0]; }, [&](int) -> const uint4 (&)[2] { return o0; },
                       [&](int) -> const uint4 (&)[2] { return o1; }, BODY{});
        }
        add_bias(bias_off);
    };

    // ---- proj_in
#pragma unroll
    for (int j = 0; j < 20; ++j)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) t[j][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
    dense320(P0E{}, t, BODY{});          // stream slabs 0..24
    add_bias(P_BIN);
    // ---- two temporal self-attentions
    layer_norm(P_LN1G, P_LN1B, true);
    attention(P_BO1);
    layer_norm(P_LN2G, P_LN2B, true);
    attention(P_BO2);
    // ---- GEGLU feed-forward, 20 chunks of 64 hidden units: 10 slabs of W1 (value/gate rows interleaved per 16), 5 slabs of W2
    layer_norm(P_LN3G, P_LN3B, false);
    // (chunk c is 15 slabs and starts at stream index 345 + 15 c: odd for even c, even for odd c -> two chunks per loop iteration)
    auto ff_chunk = [&](const int c, auto p0_tag) {
        f32x4 g[8][2];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) g[i][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
        slab_group(p0_tag, N10{}, I4{}, PLAIN{}, [&](int i) { return &g[(i / 5) * 4][0]; }, [&](int i) -> const uint4 (&)[2] { return a[2 * (i % 5)]; },
                   [&](int i) -> const uint4 (&)[2] { return a[2 * (i % 5) + 1]; }, BODY{});
        uint4 hf0[2], hf1[2];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            f32x4 hv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 bv = *(const float4*)(prm + P_B1 + c * 128 + (2 * i) * 16 + 4 * lg), bg = *(const float4*)(prm + P_B1 + c * 128 + (2 * i + 1) * 16 + 4 * lg);
                const vv_f32x2 g01 = VV_GELU2((vv_f32x2){g[2 * i + 1][tt][0] + bg.x, g[2 * i + 1][tt][1] + bg.y});
                const vv_f32x2 g23 = VV_GELU2((vv_f32x2){g[2 * i + 1][tt][2] + bg.z, g[2 * i + 1][tt][3] + bg.w});
                hv[i][0] = (g[2 * i][tt][0] + bv.x) * g01.x; hv[i][1] = (g[2 * i][tt][1] + bv.y) * g01.y;
                hv[i][2] = (g[2 * i][tt][2] + bv.z) * g23.x; hv[i][3] = (g[2 * i][tt][3] + bv.w) * g23.y;
            }
            hf0[tt] = frag(hv[0], hv[1]); hf1[tt] = frag(hv[2], hv[3]);
        }
        slab_group(p0_tag, N5{}, I4{}, PLAIN{}, [&](int i) { return &t[i * 4][0]; }, [&](int) -> const uint4 (&)[2] { return hf0; },
                   [&](int) -> const uint4 (&)[2] { return hf1; }, BODY{});
    };
#pragma unroll 1
    for (int c = 0; c < 20; c += 2) { ff_chunk(c, P0O{}); ff_chunk(c + 1, P0E{}); }
    add_bias(P_B2);
    // ---- proj_out (+ bias + x + res1)
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int s2 = 0; s2 < 10; ++s2) a[s2][tt] = frag(t[2 * s2][tt], t[2 * s2 + 1][tt]);
#pragma unroll
    for (int j = 0; j < 20; ++j)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) t[j][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
    dense320(P0O{}, t, TAIL{});          // stream slabs 645..669
    // Block residual: all 40 float4 of x (and of res1) are requested before the first store (read in place, every tile's s_waitcnt vmcnt(0) also drained the
    // previous tile's store: 40 serialised memory round trips at the end of a block that owns its CU alone -- round 5, second session).  The operand
    // registers a[][] are dead here.
    float4 xr[2][20];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        const bool on = tt * 16 + li < p.F;
        const int64_t row = on ? ((int64_t)(tt * 16 + li) * HW + pixel) * MC : 0;
#pragma unroll
        for (int j = 0; j < 20; ++j) xr[tt][j] = on ? *(const float4*)(p.x + row + 16 * j + 4 * lg) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (p.res1) {
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const bool on = tt * 16 + li < p.F;
            const int64_t row = on ? ((int64_t)(tt * 16 + li) * HW + pixel) * MC : 0;
#pragma unroll
            for (int j = 0; j < 20; ++j) {
                const float4 r4 = on ? *(const float4*)(p.res1 + row + 16 * j + 4 * lg) : make_float4(0.f, 0.f, 0.f, 0.f);
                xr[tt][j].x += r4.x; xr[tt][j].y += r4.y; xr[tt][j].z += r4.z; xr[tt][j].w += r4.w;
            }
        }
    }
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        const int64_t row = ((int64_t)(tt * 16 + li) * HW + pixel) * MC;
        if (tt * 16 + li < p.F)
#pragma unroll
        for (int j = 0; j < 20; ++j) {
            const int c = 16 * j + 4 * lg;
            const float4 b = *(const float4*)(prm + P_BOUT + c);
            const float v0 = t[j][tt][0] + b.x + xr[tt][j].x, v1 = t[j][tt][1] + b.y + xr[tt][j].y, v2 = t[j][tt][2] + b.z + xr[tt][j].z, v3 = t[j][tt][3] + b.w + xr[tt][j].w;
            if (p.out_dtype == VV_F32) *(float4*)((float*)p.out + row + c) = make_float4(v0, v1, v2, v3);
            else *(uint2*)((unsigned short*)p.out + row + c) = make_uint2(pack2<T>(v0, v1), pack2<T>(v2, v3));
        }
    }
}

#ifndef VV_MOTION_FORM
#define VV_MOTION_FORM 0      // 0 = 4 waves x 32 tokens (the product form; stream layout "tokens"); 1 = row-split pairs (lab: vv_motion_lab.h, layout "rowsplit")
#endif
#if VV_MOTION_FORM == 1
#include "vv_motion_lab.h"
#endif

// per-channel GroupNorm affine of a clip-pooled GroupNorm: a[c] = rstd_g * gamma_c, b[c] = beta_c - mean_g * a[c]   ([2][C] floats)
__global__ void gn_affine_kernel(const float* fin /* [groups][2] */, const float* gamma, const float* beta, int C, int groups, float* out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const int g = c / (C / groups);
    const float a = fin[2 * g + 1] * gamma[c];
    out[c] = a;
    out[C + c] = beta[c] - fin[2 * g] * a;
}

}  // namespace

extern "C" int vv_gn_affine(const float* mean_rstd, const float* gamma, const float* beta, int C, int groups, float* out, void* stream) {
    if (!mean_rstd || !gamma || !beta || !out || C <= 0 || groups <= 0 || C % groups) VV_FAIL(VV_E_ARG, "vv_gn_affine: bad args");
    hipLaunchKernelGGL(gn_affine_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, mean_rstd, gamma, beta, C, groups, out);
    VV_CHECK_LAUNCH("vv_gn_affine");
    return VV_OK;
}

extern "C" int vv_motion_module_c320(const vv_motion_params* pp, int dtype, void* stream) {
    if (!pp) VV_FAIL(VV_E_ARG, "vv_motion_module_c320: null params");
    const vv_motion_params& p = *pp;
    if (!p.x || !p.out || !p.stream || !p.params || !p.gn_affine) VV_FAIL(VV_E_ARG, "vv_motion_module_c320: null pointer");
    if (p.C != MC || p.F < 1 || p.F > MF || p.heads != MH) VV_FAIL(VV_E_UNSUPPORTED, "vv_motion_module_c320: built for C = 320, F <= 32, 8 heads (got %d, %d, %d)", p.C, p.F, p.heads);
    if (p.HW <= 0 || (p.HW & 3)) VV_FAIL(VV_E_UNSUPPORTED, "vv_motion_module_c320: HW = %d must be a positive multiple of 4", p.HW);
    if (p.out_dtype != VV_F32 && p.out_dtype != dtype) VV_FAIL(VV_E_ARG, "vv_motion_module_c320: out_dtype mismatch");
    if (p.n_slabs != N_SLABS || p.n_params != P_TOTAL - 640) VV_FAIL(VV_E_ARG, "vv_motion_module_c320: stream / parameter block size mismatch (%d slabs, %d floats)", p.n_slabs, p.n_params);
    hipStream_t st = (hipStream_t)stream;
#if VV_MOTION_FORM == 1
    if (dtype == VV_BF16) hipLaunchKernelGGL(motion_rs_c320_kernel<BF16>, dim3(p.HW / 4), dim3(512), 0, st, p);
    else if (dtype == VV_F16) hipLaunchKernelGGL(motion_rs_c320_kernel<F16>, dim3(p.HW / 4), dim3(512), 0, st, p);
#else
    if (dtype == VV_BF16) hipLaunchKernelGGL(motion_c320_kernel<BF16>, dim3(p.HW / 4), dim3(256), 0, st, p);
    else if (dtype == VV_F16) hipLaunchKernelGGL(motion_c320_kernel<F16>, dim3(p.HW / 4), dim3(256), 0, st, p);
#endif
    else VV_FAIL(VV_E_ARG, "vv_motion_module_c320: bad dtype");
    VV_CHECK_LAUNCH("vv_motion_module_c320");
    return VV_OK;
}
