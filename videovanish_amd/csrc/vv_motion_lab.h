// LAB form of the fused motion module (NOT compiled into the product library: vv_motion.hip includes this file only under -DVV_MOTION_FORM=1).
// Correct against tests/test_motion_gpu.py with packing.pack_motion_stream(layout="rowsplit"), measured 3.12-3.13 ms against 3.057 ms of the product
// form at level 0 (profiles/r5_chain_forms.txt section 7): 240 of the module's 670 slabs are q / k / v projections that run on one token tile per wave
// (one fragment read per MFMA), and the weight ring's LDS-DMA bounds both forms.  Included inside vv_motion.hip's anonymous namespace.
#pragma once

// ---------------------------------------------------------------------------------------------------------------------------
// ROW-SPLIT pair form of the motion module (round 5; the design and the measurements behind it: vv_chain.hip::chain_rs_c320_kernel,
// profiles/r5_chain_forms.txt).  A block is still 4 pixels x 32 frames behind one weight ring, but its 8 waves are 4 pixels x 2 ROW HALVES, two per
// SIMD at <= 256 architectural registers (no AGPR copies): the two waves of a pair own the same pixel (32 tokens), wave hf reads row tiles
// 2 hf, 2 hf + 1 of every [64 x 64] slab of the dense layers (4 fragment reads feed 8 MFMAs), holds the fp32 trunk of its 160 channels and the full
// h16 activation row, of which it produces k steps 2 kt + hf; partners swap halves through LDS lane for lane at every layer end.
// Temporal attention: per head the k, v^T, q projections run on ONE token tile per wave (tile hf = frames 16 hf .. 16 hf + 15: 15 of a head's 20
// slabs, stream order k | v | q | Wo), the partners swap their key-tile fragments (k as A fragments, v^T packed) while the q slabs stream, every wave
// then has all 32 keys for its 16 queries; O goes through LDS and the output projection is row-split again.
// The sinusoidal table is read from global memory (2 x 20 float4 per lane and LayerNorm) instead of 40 KB of LDS.
constexpr int MR_NS = 10, MR_AH = 6, MR_XBUF = 45056, MR_PRM = 6720;

template <typename T>
__global__ __launch_bounds__(512, 2) void motion_rs_c320_kernel(const vv_motion_params p) {
    __shared__ __attribute__((aligned(1024))) unsigned char ring[MR_NS * SLAB];
    __shared__ __attribute__((aligned(16))) unsigned char xbuf[MR_XBUF];
    __shared__ __attribute__((aligned(16))) float sbuf[8 * 64 * 4];
    __shared__ __attribute__((aligned(16))) float prm[MR_PRM];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int grp = wave & 3, hf = wave >> 2, pw = wave ^ 4;
    const bool hi = hf != 0;
    const int pixel = blockIdx.x * 4 + grp;
    const int64_t HW = p.HW;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};

    for (int i = tid * 4; i < MR_PRM; i += 512 * 4) {
        const float4 v = i < 640 ? *(const float4*)(p.gn_affine + i) : *(const float4*)(p.params + (i - 640));
        *(float4*)(prm + i) = v;
    }
    const float* pe_g = p.params + (P_PE - 640);      // [32][320] sinusoidal table in global memory
    // ---- weight stream: every wave copies 1 KB of every slab
    const unsigned char* sbase = (const unsigned char*)p.stream + wave * 1024 + lane * 16;
    int issued = 0, islot = 0, cslot = 0;
    auto issue = [&]() {
        glds16_asm(sbase + (int64_t)issued * SLAB, ring + islot * SLAB + wave * 1024);
        ++issued;
        islot = islot + 1 == MR_NS ? 0 : islot + 1;
    };
    using BODY = std::false_type; using TAIL = std::true_type;
    using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
    // one synchronisation step in front of NI slabs (vv_chain.hip): issue NI more, wait until all but the newest MR_AH have landed, meet.  XCH: the
    // step also publishes LDS writes of this wave.  (Plain global loads -- the sinusoidal table -- count in vmcnt too: they only make the wait stricter.)
    auto sync = [&](auto ni_tag, auto tail_tag, auto xch_tag) {
        constexpr int NI = decltype(ni_tag)::value;
        if constexpr (decltype(tail_tag)::value) {
            if (issued + NI <= N_SLABS) {
#pragma unroll
                for (int i = 0; i < NI; ++i) issue();
                asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            } else {
                while (issued < N_SLABS) issue();
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        } else {
#pragma unroll
            for (int i = 0; i < NI; ++i) issue();
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        }
        if constexpr (decltype(xch_tag)::value) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    using NOX = std::false_type; using XCH = std::true_type;
    auto slab = [&]() -> const unsigned char* {
        const unsigned char* s = ring + cslot * SLAB;
        cslot = cslot + 1 == MR_NS ? 0 : cslot + 1;
        return s;
    };
    auto meet = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); };
    auto frag = [&](const f32x4& lo, const f32x4& hi_) -> uint4 {
        return make_uint4(pack2<T>(lo[0], lo[1]), pack2<T>(lo[2], lo[3]), pack2<T>(hi_[0], hi_[1]), pack2<T>(hi_[2], hi_[3]));
    };
    auto sel = [&](const uint4& a_, const uint4& b_) -> uint4 { return hi ? b_ : a_; };      // wave-uniform select

    // ---- row-split slab groups: this wave's two row tiles (2 hf, 2 hf + 1) of N [64 x 64] slabs
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
    auto fma_rs = [&](const WF2& f, f32x4* acc /* [2][2] = [rt][tt] */, const uint4 (&x0)[2], const uint4 (&x1)[2]) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) acc[rt * 2 + tt] = T::mfma(f.w[kk][rt], kk ? x1[tt] : x0[tt], acc[rt * 2 + tt]);
    };
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
        }
    };
    using N5 = std::integral_constant<int, 5>; using N10 = std::integral_constant<int, 10>; using N25 = std::integral_constant<int, 25>;
    using NOPRE = std::false_type; using PRE = std::true_type;

    // ---- state: trunk t[2 rb + rt][tt] (own channels), activations a0[kt][tt] / a1[kt][tt] = k steps 2 kt / 2 kt + 1 of the full row
    f32x4 t[10][2];
    uint4 a0[5][2], a1[5][2];
    auto chan = [&](const int j) -> int { return 64 * (j >> 1) + 32 * hf + 16 * (j & 1) + 4 * lg; };
    int64_t xrow[2];
    {
        __syncthreads();       // parameter block visible
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int fr = min(tt * 16 + li, p.F - 1);                       // clips shorter than 32 frames: rows past F repeat the last frame (masked as keys, never stored)
            xrow[tt] = ((int64_t)fr * HW + pixel) * MC;
            const float* xr = p.x + xrow[tt];
#pragma unroll
            for (int s = 0; s < 10; ++s) {
                const int c0 = 32 * s + 4 * lg, c1 = c0 + 16;
                const float4 x0 = *(const float4*)(xr + c0), x1 = *(const float4*)(xr + c1);
                const float4 g0 = *(const float4*)(prm + P_GN_A + c0), b0 = *(const float4*)(prm + P_GN_B + c0);
                const float4 g1 = *(const float4*)(prm + P_GN_A + c1), b1 = *(const float4*)(prm + P_GN_B + c1);
                const uint4 v = make_uint4(pack2<T>(x0.x * g0.x + b0.x, x0.y * g0.y + b0.y), pack2<T>(x0.z * g0.z + b0.z, x0.w * g0.w + b0.w),
                                           pack2<T>(x1.x * g1.x + b1.x, x1.y * g1.y + b1.y), pack2<T>(x1.z * g1.z + b1.z, x1.w * g1.w + b1.w));
                if (s & 1) a1[s >> 1][tt] = v; else a0[s >> 1][tt] = v;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll 1
        for (int i = 0; i < MR_AH; ++i) issue();
    }
    auto add_bias = [&](const int off) {
#pragma unroll
        for (int j = 0; j < 10; ++j) {
            const float4 b = *(const float4*)(prm + off + chan(j));
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) { t[j][tt][0] += b.x; t[j][tt][1] += b.y; t[j][tt][2] += b.z; t[j][tt][3] += b.w; }
        }
    };
    auto dense320 = [&](f32x4 (&acc)[10][2], auto tail) {
        group_rs(N25{}, NOPRE{}, [&](int i) { return &acc[(i / 5) * 2][0]; }, [&](int i) -> const uint4 (&)[2] { return a0[i % 5]; },
                 [&](int i) -> const uint4 (&)[2] { return a1[i % 5]; }, tail);
    };
    // own[rb][tt] = h16(LN(t) g + b (+ pe[frame])) of this wave's channels = k step 2 rb + hf of the row; statistics merged with the partner's (Chan)
    auto layer_norm = [&](const int goff, const int boff, const bool with_pe, uint4 (&own)[5][2]) {
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
            const float rstd = rsqrtf((m2loc[tt] + oq[tt] + 80.0f * dm * dm) * (1.0f / MC) + 1e-5f);
            const float* pe = pe_g + (tt * 16 + li) * MC;
#pragma unroll
            for (int rb = 0; rb < 5; ++rb) {
                f32x4 y[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int j = 2 * rb + h, c = chan(j);
                    const float4 g = *(const float4*)(prm + goff + c), b = *(const float4*)(prm + boff + c);
                    y[h][0] = (t[j][tt][0] - mean) * rstd * g.x + b.x; y[h][1] = (t[j][tt][1] - mean) * rstd * g.y + b.y;
                    y[h][2] = (t[j][tt][2] - mean) * rstd * g.z + b.z; y[h][3] = (t[j][tt][3] - mean) * rstd * g.w + b.w;
                    if (with_pe) { const float4 e = *(const float4*)(pe + c); y[h][0] += e.x; y[h][1] += e.y; y[h][2] += e.z; y[h][3] += e.w; }
                }
                own[rb][tt] = frag(y[0], y[1]);
            }
        }
    };
    unsigned char* const xmine = xbuf + wave * 5120 + lane * 16;
    const unsigned char* const xpart = xbuf + pw * 5120 + lane * 16;
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

    // ---- temporal self-attention over the 32 frames of the pair's pixel: t += Wo attn(LN(t) + pe) + bo
    const float sc = 0.15811388300841897f * 1.4426950408889634f;      // 40^-1/2 * log2(e)
    struct WF { uint4 w[2][3]; };
    auto load_full = [&](const unsigned char* s, WF& f) {      // all 48 rows of a [48 x 64] slab
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int off = ((kk * 4 + lg) ^ sw) << 4;
#pragma unroll
            for (int rt = 0; rt < 3; ++rt) f.w[kk][rt] = *(const uint4*)(s + (rt * 16 + li) * 128 + off);
        }
    };
    auto attention = [&](const int goff, const int boff, const int bias_off) {
        uint4 x0[5], x1[5];      // the wave's token tile hf: full row as B fragments (k steps 2 kt / 2 kt + 1)
        {
            uint4 own[5][2];
            layer_norm(goff, boff, true, own);
#pragma unroll
            for (int rb = 0; rb < 5; ++rb) *(uint4*)(xmine + rb * 1024) = sel(own[rb][1], own[rb][0]);      // the partner's tile of my k steps
            meet();
#pragma unroll
            for (int rb = 0; rb < 5; ++rb) {
                const uint4 o = *(const uint4*)(xpart + rb * 1024), m = sel(own[rb][0], own[rb][1]);
                x0[rb] = sel(m, o);
                x1[rb] = sel(o, m);
            }
            meet();      // the buffer is free for the heads' exchanges
        }
        // 5 slabs of 48 rows against the tile's row; TR: operands exchanged (D = X W^T: lane = output channel, registers = tokens 4 lg + r)
        auto proj = [&](f32x4 (&acc)[3], auto tr_tag, auto first_xch) {
            constexpr bool TR = decltype(tr_tag)::value;
            WF f[2];
            sync(I2{}, BODY{}, first_xch);
            load_full(slab(), f[0]);
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                if (i + 1 < 5) {
                    if (((i + 1) & 1) == 0) { if (i + 2 < 5) sync(I2{}, BODY{}, NOX{}); else sync(I1{}, BODY{}, NOX{}); }
                    load_full(slab(), f[(i + 1) & 1]);
                }
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int rt = 0; rt < 3; ++rt)
                        acc[rt] = TR ? T::mfma(kk ? x1[i] : x0[i], f[i & 1].w[kk][rt], acc[rt]) : T::mfma(f[i & 1].w[kk][rt], kk ? x1[i] : x0[i], acc[rt]);
                if (i + 1 < 5) __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
            }
        };
        using PLAIN = std::false_type; using TRANSP = std::true_type;
        unsigned char* const kvmine = xbuf + wave * 3584 + lane * 16;             // kf0 | kf1 (1 KB each) | v^T packed 3 x 512 B
        const unsigned char* const kvpart = xbuf + pw * 3584 + lane * 16;
        unsigned char* const obase = xbuf + 8 * 3584;                             // O fragments: 8 waves x 2 KB
#pragma unroll 1
        for (int h = 0; h < MH; ++h) {
            f32x4 ka[3] = {z4, z4, z4}, va[3] = {z4, z4, z4}, qa[3] = {z4, z4, z4};
            proj(ka, PLAIN{}, NOX{});
            proj(va, TRANSP{}, NOX{});
            const uint4 kf0 = frag(ka[0], ka[1]), kf1 = frag(ka[2], z4);
            uint2 vp[3];
#pragma unroll
            for (int dt = 0; dt < 3; ++dt) vp[dt] = make_uint2(pack2<T>(va[dt][0], va[dt][1]), pack2<T>(va[dt][2], va[dt][3]));
            *(uint4*)kvmine = kf0;
            *(uint4*)(kvmine + 1024) = kf1;
#pragma unroll
            for (int dt = 0; dt < 3; ++dt) *(uint2*)(xbuf + wave * 3584 + 2048 + dt * 512 + lane * 8) = vp[dt];
            proj(qa, PLAIN{}, XCH{});          // its first step publishes the key-tile fragments; they have landed long before the q slabs are through
            const uint4 pk0 = *(const uint4*)kvpart, pk1 = *(const uint4*)(kvpart + 1024);
            uint2 pv[3];
#pragma unroll
            for (int dt = 0; dt < 3; ++dt) pv[dt] = *(const uint2*)(xbuf + pw * 3584 + 2048 + dt * 512 + lane * 8);
            const uint4 qf0 = frag(qa[0], qa[1]), qf1 = frag(qa[2], z4);
            // S^T[key tile][own query tile]; key tile 0 = frames 0..15 = the hf = 0 wave's
            f32x4 sT[2];
            sT[0] = T::mfma(sel(kf0, pk0), qf0, z4); sT[0] = T::mfma(sel(kf1, pk1), qf1, sT[0]);
            sT[1] = T::mfma(sel(pk0, kf0), qf0, z4); sT[1] = T::mfma(sel(pk1, kf1), qf1, sT[1]);
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) sT[kt][r] = (kt * 16 + 4 * lg + r >= p.F) ? -1e30f : sT[kt][r];
            float m = fmaxf(fmaxf(fmaxf(sT[0][0], sT[0][1]), fmaxf(sT[0][2], sT[0][3])), fmaxf(fmaxf(sT[1][0], sT[1][1]), fmaxf(sT[1][2], sT[1][3])));
            m = fmaxf(m, __shfl_xor(m, 16)); m = fmaxf(m, __shfl_xor(m, 32));
            const float mc = m * sc;
            float l = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float e = __builtin_amdgcn_exp2f(sT[kt][r] * sc - mc); sT[kt][r] = e; l += e; }
            l += __shfl_xor(l, 16); l += __shfl_xor(l, 32);
            const float inv = 1.0f / l;
            const uint4 pf = frag(sT[0], sT[1]);
            // O^T[d tile][own query tile] = V^T P^T: V^T fragment = (frames 0..15 | frames 16..31) of lane d
            f32x4 oT[3];
#pragma unroll
            for (int dt = 0; dt < 3; ++dt) {
                const uint4 vf = hi ? make_uint4(pv[dt].x, pv[dt].y, vp[dt].x, vp[dt].y) : make_uint4(vp[dt].x, vp[dt].y, pv[dt].x, pv[dt].y);
                oT[dt] = T::mfma(vf, pf, z4) * inv;
            }
            unsigned char* dst = obase + wave * 2048 + lane * 16;
            *(uint4*)dst = frag(oT[0], oT[1]);
            *(uint4*)(dst + 1024) = frag(oT[2], z4);
            // t += Wo[:, head] O for both token tiles of the pixel (row-split)
            sync(I2{}, BODY{}, XCH{});
            uint4 o0[2], o1[2];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const unsigned char* src = obase + (grp + 4 * tt) * 2048 + lane * 16;
                o0[tt] = *(const uint4*)src;
                o1[tt] = *(const uint4*)(src + 1024);
            }
            group_rs(N5{}, PRE{}, [&](int i) { return &t[i * 2][0]; }, [&](int) -> const uint4 (&)[2] { return o0; },
                     [&](int) -> const uint4 (&)[2] { return o1; }, BODY{});
        }
        add_bias(bias_off);
    };

    // ---- proj_in
#pragma unroll
    for (int j = 0; j < 10; ++j)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) t[j][tt] = z4;
    dense320(t, BODY{});
    add_bias(P_BIN);
    attention(P_LN1G, P_LN1B, P_BO1);
    attention(P_LN2G, P_LN2B, P_BO2);
    // ---- GEGLU feed-forward (20 chunks of 64 hidden units: vv_chain.hip::chain_rs_c320_kernel)
    {
        uint4 own[5][2];
        layer_norm(P_LN3G, P_LN3B, false, own);
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
                const float* bp = prm + P_B1 + c * 128 + i * 64 + hf * 32 + 4 * lg;
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
    add_bias(P_B2);
    // ---- proj_out (+ bias + x + res1)
    {
        uint4 own[5][2];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int rb = 0; rb < 5; ++rb) own[rb][tt] = frag(t[2 * rb][tt], t[2 * rb + 1][tt]);
        meet();
        swap_full(own);
    }
#pragma unroll
    for (int j = 0; j < 10; ++j)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) t[j][tt] = z4;
    dense320(t, TAIL{});
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        if (tt * 16 + li < p.F) {
            const int64_t row = xrow[tt];
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                const int c = chan(j);
                const float4 b = *(const float4*)(prm + P_BOUT + c);
                const float4 xr = *(const float4*)(p.x + row + c);
                float v0 = t[j][tt][0] + b.x + xr.x, v1 = t[j][tt][1] + b.y + xr.y, v2 = t[j][tt][2] + b.z + xr.z, v3 = t[j][tt][3] + b.w + xr.w;
                if (p.res1) { const float4 r4 = *(const float4*)(p.res1 + row + c); v0 += r4.x; v1 += r4.y; v2 += r4.z; v3 += r4.w; }
                if (p.out_dtype == VV_F32) *(float4*)((float*)p.out + row + c) = make_float4(v0, v1, v2, v3);
                else *(uint2*)((unsigned short*)p.out + row + c) = make_uint2(pack2<T>(v0, v1), pack2<T>(v2, v3));
            }
        }
    }
}

