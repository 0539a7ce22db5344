// LAB forms of the fused chain tail (NOT compiled into the product library: vv_chain.hip includes this file only under -DVV_CHAIN_FORM=0 / 2).
// Kept as the measured record of round 5 (profiles/r5_chain_forms.txt): the token-split forms (4 waves x 32 tokens = rounds 3-4, 8 waves x 16 tokens) and
// the column-split form without any weight in LDS.  Both are correct against tests/test_chain_gpu.py with the matching stream layout
// (packing.pack_chain_stream(layout="tokens" / "columns")).  Included inside vv_chain.hip's anonymous namespace, after its shared helpers.
#pragma once

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

