// Epilogue shared by the MFMA GEMM / convolution kernels (vv_gemm.hip, vv_gemm256.hip, vv_conv3.hip): accumulators in the swapped-operand layout
// (lane (lr, lq) owns output row row0 + i*16 + lr, channels ncol0 + j*16 + 4*lq .. +3) -> bias, out_scale, time-embedding row
// vector, up to two residuals, ReLU, GEGLU, cast, 16-byte stores.  row_m(tile_row, ok&) maps a tile row to the output row.
//
// Two forms of the 16-byte vector path (round 5, second session; A/B record: profiles/r5_epilogue_ab.txt):
//   * LEAN = false: per (strip, column tile) the bias float4 is re-read and waited for with s_waitcnt vmcnt(0) -- which on gfx950 also drains the STORE
//     of the previous tile: MT x NT serialised load -> wait -> store steps per wave.  It holds no operand across tiles, which is what lets the
//     128 x 160 LIN / FAST9 loaders fit the 128-VGPR budget of a fourth block per CU (whose MFMAs then cover the steps).
//   * LEAN = true: the lane's NT bias float4 are read once and folded into the accumulators (same fp32 sum, same order: bit-identical results), the
//     fp32 residual of a strip is requested in one batch, and the strip's stores issue back to back: MT waits per wave instead of MT x NT.  Used by
//     the 256-row kernels (vv_gemm256.hip: ONE block per CU, nothing else covers the epilogue; qkv L2 0.339 -> 0.303 ms, ff2 L2 0.390 -> 0.376,
//     8192^3 1225 -> 1250 TFLOP/s -- and the 2-phase form now wins on the level-1 / level-2 linears with K >= 640, vv_gemm256_try) and by the halo-tile
//     3x3 kernels (MODE_HALO: +1.5..3.4 % with a residual).  Measured losers: the LIN / FAST9 loaders (55 spilled registers, -30 %), the fp32-operand
//     loader with the lean form ALONE (FAST32, -8..-10 % at levels 1 / 2; it gains with STAGED on top), and a column-by-column order of the same work (-8..-12 % on every fp32-residual shape: the 64-byte
//     pieces of an output row are then written microseconds apart).
//   * STAGED (with LEAN; the 2-phase 256-row kernel, the 128 x 160 halo-tile kernel and the fp32-operand loader; fp32 output with at most the fp32 residual): in the accumulator layout a wave
//     instruction touches 16 rows x 64 bytes -- half a cache line per row and request.  A strip goes through a wave-private LDS tile (the operand stages are dead) and
//     comes back row-major: 256 / W rows x W * 4 contiguous bytes per instruction.  out-proj L1 0.210 -> 0.189 ms, L2 0.146 -> 0.134; zero convolutions (fp32 operand) +9..15 %;
//     3x3 + residual +1.1 %.  Since round 6 also for h16 outputs (second branch below).
#pragma once
#include "vv_common.h"

// Order pin for the wave-private LDS exchange of the STAGED forms: lanes write the strip in the accumulator layout and read it back row-major --
// OTHER lanes' data -- so the reads must stay behind the writes, and the next strip's writes behind this strip's reads.  The hardware keeps a wave's LDS
// operations in order; this keeps the COMPILER from reordering may-alias LDS accesses it could one day prove independent per lane.  No instruction is emitted.
__device__ __forceinline__ void stage_order_pin() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <typename T, int MT, int NT, bool LEAN = false, bool STAGED = false, bool GN = false, typename RowMap>
__device__ __forceinline__ void gemm_epilogue(const vv_conv_params& p, f32x4 (&acc)[MT][NT], const int row0, const int ncol0, const int lr,
                                              const int lq, const int HWo, RowMap row_m, const float* sbias = nullptr, float* stage = nullptr, float* gn_row = nullptr) {
    // ---- epilogue: lane owns out[m][n .. n+3].  All bias / time-embedding / residual loads of one 16-row strip are issued
    // back to back into registers BEFORE their first use (one wait per strip instead of one per load).
    const bool geglu = p.epilogue == VV_EPI_GEGLU;
    const int N = p.N;
    const bool vec = (N & 3) == 0 && (p.ldo & 3) == 0;
    const bool r0f32 = p.res_dtype == VV_F32;
    if (geglu) {
        if constexpr (NT % 2 == 0) {
            float4 bv[NT / 2], bg[NT / 2];
#pragma unroll
            for (int j = 0; j < NT; j += 2) {
                const int n = ncol0 + j * 16 + 4 * lq;
                bv[j / 2] = (p.bias && n < N) ? *(const float4*)(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
                bg[j / 2] = (p.bias && n < N) ? *(const float4*)(p.bias + n + 16) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                bool mok;
                const int m = row_m(row0 + i * 16 + lr, mok);
                if (!mok) continue;
#pragma unroll
                for (int j = 0; j < NT; j += 2) {
                    const int nt0 = ncol0 + j * 16;
                    if (nt0 + 4 * lq >= N) continue;
                    const float bvv[4] = {bv[j / 2].x, bv[j / 2].y, bv[j / 2].z, bv[j / 2].w};
                    const float bgg[4] = {bg[j / 2].x, bg[j / 2].y, bg[j / 2].z, bg[j / 2].w};
                    float o[4];
#ifdef VV_GELU_SCALAR      // lab: the scalar A&S form (v_rcp + v_exp per element) the epilogue used until round 6
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = (acc[i][j][r] + bvv[r]) * gelu_f(acc[i][j + 1][r] + bgg[r]);
#else
#pragma unroll
                    for (int r = 0; r < 4; r += 2) {
                        const vv_f32x2 ge = gelu_poly2((vv_f32x2){acc[i][j + 1][r] + bgg[r], acc[i][j + 1][r + 1] + bgg[r + 1]});
                        o[r] = (acc[i][j][r] + bvv[r]) * ge.x; o[r + 1] = (acc[i][j][r + 1] + bvv[r + 1]) * ge.y;
                    }
#endif
                    const int64_t oc = (int64_t)m * p.ldo + (nt0 >> 1) + 4 * lq;
                    if (p.out_dtype == VV_F32) *(float4*)((float*)p.out + oc) = make_float4(o[0], o[1], o[2], o[3]);
                    else *(uint2*)((unsigned short*)p.out + oc) = make_uint2(pack2<T>(o[0], o[1]), pack2<T>(o[2], o[3]));
                }
            }
        }
        return;
    }
#ifdef VV_NO_STAGE_F32      // lab: the fp32 strips stay in the accumulator layout (the form before round 5's second session) -- the in-pipeline A/B of the staged form, profiles/r6_stage_f32_pipeline_ab.txt
    constexpr bool STAGE_F32 = false;
#else
    constexpr bool STAGE_F32 = true;
#endif
    if (STAGE_F32 && LEAN && STAGED && vec && stage && p.out_dtype == VV_F32 && p.split_heads <= 0 && !p.rowvec && !p.res1 && p.act == VV_ACT_NONE && (!p.res0 || r0f32)) {
        // STAGED form of the lean path (fp32 trunk out, at most the fp32 residual: the out-projections and FF outputs of levels 1 / 2 -- streaming kernels, 60 % of
        // their time in this epilogue).  In the accumulator layout a wave instruction touches 16 rows x 64 bytes: half a cache line per row and request.  A strip
        // (16 rows x W columns) goes through a wave-private LDS tile instead and comes back row-major: each instruction then covers 256 / W rows x W * 4 contiguous bytes
        // (320 bytes per row for the 80-column wave tile), residual read and store alike.  Same arithmetic, same order: bit-identical.
        constexpr int W = NT * 16, PITCH = W + 4;
        const int lane = lq * 16 + lr;
        // GroupNorm statistics of the layer's output (vv_conv_params.gn_partials, round 6): lane c sums column c (and c + 64 while < W) of every strip of this
        // wave's rows -- the final values go back into the wave-private tile (it is row-major and dead after the store) and are read column-wise: fixed order,
        // no atomics, rows outside the image count as zero.  gn_row = this wave's [N][2] slot, offset to its first column.
        float gs0 = 0.f, gq0 = 0.f, gs1 = 0.f, gq1 = 0.f;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = ncol0 + j * 16 + 4 * lq;
            const float4 b = (p.bias && n < N) ? *(const float4*)(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int i = 0; i < MT; ++i) { acc[i][j][0] += b.x; acc[i][j][1] += b.y; acc[i][j][2] += b.z; acc[i][j][3] += b.w; }
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#pragma unroll
            for (int j = 0; j < NT; ++j)
                *(float4*)(stage + lr * PITCH + j * 16 + 4 * lq) = make_float4(acc[i][j][0] * p.out_scale, acc[i][j][1] * p.out_scale, acc[i][j][2] * p.out_scale, acc[i][j][3] * p.out_scale);
            stage_order_pin();
            float4 v[NT], r4[NT];
            int64_t off[NT];
            bool on[NT];
#pragma unroll
            for (int q = 0; q < NT; ++q) {
                const int idx = (q * 64 + lane) * 4, rr = idx / W, cc = idx - rr * W;
                bool mok;
                const int m = row_m(row0 + i * 16 + rr, mok);
                on[q] = mok && ncol0 + cc < N;
                off[q] = (int64_t)(mok ? m : 0);
                r4[q] = (p.res0 && on[q]) ? *(const float4*)((const float*)p.res0 + off[q] * N + ncol0 + cc) : make_float4(0.f, 0.f, 0.f, 0.f);
                off[q] = off[q] * p.ldo + ncol0 + cc;
                v[q] = *(const float4*)(stage + rr * PITCH + cc);
            }
            stage_order_pin();
#pragma unroll
            for (int q = 0; q < NT; ++q) {
                const float4 o = make_float4(v[q].x + r4[q].x, v[q].y + r4[q].y, v[q].z + r4[q].z, v[q].w + r4[q].w);
                if (on[q]) *(float4*)((float*)p.out + off[q]) = o;
                if constexpr (GN) {
                    const int idx = (q * 64 + lane) * 4, rr = idx / W, cc = idx - rr * W;
                    *(float4*)(stage + rr * PITCH + cc) = on[q] ? o : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
            if constexpr (GN) {
                stage_order_pin();
#pragma unroll 4
                for (int r = 0; r < 16; ++r) {
                    const float a = stage[r * PITCH + lane];
                    gs0 += a; gq0 += a * a;
                    if (W > 64) {
                        const float b = lane < W - 64 ? stage[r * PITCH + 64 + lane] : 0.f;
                        gs1 += b; gq1 += b * b;
                    }
                }
                stage_order_pin();
            }
        }
        if constexpr (GN) {
            if (lane < W && ncol0 + lane < N) *(float2*)(gn_row + 2 * lane) = make_float2(gs0, gq0);
            if (W > 64 && lane < W - 64 && ncol0 + 64 + lane < N) *(float2*)(gn_row + 2 * (64 + lane)) = make_float2(gs1, gq1);
        }
        return;
    }
    // STAGED form for h16 outputs (written at the end of round 5, timed and adopted in round 6: +0.13 % on the bench line in an interleaved in-pipeline A/B, bit-equal
    // outputs -- profiles/r6_stage_h16_pipeline_ab.txt): QKV with the head-major store when a wave tile is exactly one head, proj_in, FF outputs stored h16.  In the
    // accumulator layout a wave instruction writes 16 rows x 32 bytes; through the same fp32 LDS tile a lane takes 8 consecutive columns of a row instead (two
    // ds_read_b128, at most two residual float4, one 16-byte store): W * 2 contiguous bytes per row.
    if (LEAN && STAGED && vec && stage && p.out_dtype != VV_F32 && !p.rowvec && !p.res1 && p.act == VV_ACT_NONE && (!p.res0 || r0f32) && (N & 7) == 0 && (p.ldo & 7) == 0 &&
        (p.split_heads <= 0 || p.split_dim == NT * 16)) {
        constexpr int W = NT * 16, PITCH = W + 4, NQ = (16 * W + 511) / 512;
        const int lane = lq * 16 + lr;
        const bool split = p.split_heads > 0;
        const int stok = p.split_tokens < 0 ? -p.split_tokens : p.split_tokens;
        const int snb = p.split_tokens < 0 ? (p.F * HWo) / stok : 0;
        const int64_t colpart0 = split ? (int64_t)(ncol0 / p.split_dim) * stok * p.split_dim : 0;      // the wave tile is one (which, head) block
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = ncol0 + j * 16 + 4 * lq;
            const float4 b = (p.bias && n < N) ? *(const float4*)(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int i = 0; i < MT; ++i) { acc[i][j][0] += b.x; acc[i][j][1] += b.y; acc[i][j][2] += b.z; acc[i][j][3] += b.w; }
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#pragma unroll
            for (int j = 0; j < NT; ++j)
                *(float4*)(stage + lr * PITCH + j * 16 + 4 * lq) = make_float4(acc[i][j][0] * p.out_scale, acc[i][j][1] * p.out_scale, acc[i][j][2] * p.out_scale, acc[i][j][3] * p.out_scale);
            stage_order_pin();
            float v[NQ][8];
            int64_t off[NQ];
            bool on[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int idx = (q * 64 + lane) * 8;
                const bool valid = idx < 16 * W;
                const int rr = valid ? idx / W : 0, cc = valid ? idx - rr * W : 0;
                bool mok;
                const int m = row_m(row0 + i * 16 + rr, mok);
                on[q] = valid && mok && ncol0 + cc < N;
                const int64_t mm = mok ? m : 0;
                *(float4*)&v[q][0] = *(const float4*)(stage + rr * PITCH + cc);
                *(float4*)&v[q][4] = *(const float4*)(stage + rr * PITCH + cc + 4);
                if (p.res0 && on[q]) {
                    const float* rp = (const float*)p.res0 + mm * N + ncol0 + cc;
                    const float4 a4 = *(const float4*)rp, b4 = *(const float4*)(rp + 4);
                    v[q][0] += a4.x; v[q][1] += a4.y; v[q][2] += a4.z; v[q][3] += a4.w; v[q][4] += b4.x; v[q][5] += b4.y; v[q][6] += b4.z; v[q][7] += b4.w;
                }
                if (split) {
                    const int mi = (int)mm;      // (32-bit division: a row index is < 2^31; as int64 every (strip, q) paid a ~100-instruction 64-bit division -- round 6)
                    int b, tok;
                    if (snb) { tok = mi / snb; b = mi - tok * snb; } else { b = mi / stok; tok = mi - b * stok; }
                    off[q] = ((int64_t)b * 3 * p.split_heads * stok + tok) * p.split_dim + colpart0 + cc;
                } else off[q] = mm * p.ldo + ncol0 + cc;
            }
            stage_order_pin();
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                if (on[q]) *(uint4*)((unsigned short*)p.out + off[q]) = pack8<T>(v[q]);
        }
        return;
    }
    if (LEAN && vec) {
        int colpart[NT];          // split_heads store: the column's (which, head, d) part of the output index (< 3*C*tokens)
        const bool split = p.split_heads > 0;
        const int stok = p.split_tokens < 0 ? -p.split_tokens : p.split_tokens;      // tokens per batch element
        const int snb = p.split_tokens < 0 ? (p.F * HWo) / stok : 0;                 // token-major rows: number of batch elements
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = ncol0 + j * 16 + 4 * lq;
            colpart[j] = 0;
            if (split) {
                const int wh = n / p.split_dim;                       // which * heads + head
                colpart[j] = wh * stok * p.split_dim + (n - wh * p.split_dim);
            }
        }
        {      // the lane's NT bias float4 are read ONCE and folded into the accumulators (the sum keeps its order: products first, bias last)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int n = ncol0 + j * 16 + 4 * lq;
                const float4 b = (p.bias && n < N) ? *(const float4*)(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int i = 0; i < MT; ++i) { acc[i][j][0] += b.x; acc[i][j][1] += b.y; acc[i][j][2] += b.z; acc[i][j][3] += b.w; }
            }
        }
        const bool pre0 = p.res0 && r0f32;          // the common residual: the loads of one strip are issued back to back before their first use
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            __builtin_amdgcn_sched_barrier(0);      // one strip at a time: hoisting the next strips' address arithmetic over this one's stores costs the 128-VGPR kernels their budget
            bool mok;
            const int m = row_m(row0 + i * 16 + lr, mok);
            float4 ra4[NT];
            if (pre0) {
                const float* rp = (const float*)p.res0 + (int64_t)(mok ? m : 0) * N + ncol0 + 4 * lq;
#pragma unroll
                for (int j = 0; j < NT; ++j) ra4[j] = (mok && ncol0 + j * 16 + 4 * lq < N) ? *(const float4*)(rp + j * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            if (!mok) continue;
            const int64_t rbase = (int64_t)(mok ? m : 0) * N;
            const float* rowv = (p.rowvec && mok) ? p.rowvec + (int64_t)(m / HWo) * N : nullptr;
            int64_t rowpart = 0;
            if (split) {
                int b, tok;
                if (snb) { tok = m / snb; b = m - tok * snb; } else { b = m / stok; tok = m - b * stok; }
                rowpart = ((int64_t)b * 3 * p.split_heads * stok + tok) * p.split_dim;
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int n = ncol0 + j * 16 + 4 * lq;
                const bool on = mok && n < N;
                float v[4] = {acc[i][j][0] * p.out_scale, acc[i][j][1] * p.out_scale, acc[i][j][2] * p.out_scale, acc[i][j][3] * p.out_scale};
                if (rowv && on) { const float4 t4 = *(const float4*)(rowv + n); v[0] += t4.x; v[1] += t4.y; v[2] += t4.z; v[3] += t4.w; }
                if (pre0) { v[0] += ra4[j].x; v[1] += ra4[j].y; v[2] += ra4[j].z; v[3] += ra4[j].w; }
                if (!on) continue;
                if (p.res0 && !r0f32) { const uint2 r2 = *(const uint2*)((const unsigned short*)p.res0 + rbase + n); v[0] += T::to_f32(r2.x & 0xffff); v[1] += T::to_f32(r2.x >> 16); v[2] += T::to_f32(r2.y & 0xffff); v[3] += T::to_f32(r2.y >> 16); }
                if (p.res1) {
                    if (r0f32) { const float4 r4 = *(const float4*)((const float*)p.res1 + rbase + n); v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w; }
                    else { const uint2 r2 = *(const uint2*)((const unsigned short*)p.res1 + rbase + n); v[0] += T::to_f32(r2.x & 0xffff); v[1] += T::to_f32(r2.x >> 16); v[2] += T::to_f32(r2.y & 0xffff); v[3] += T::to_f32(r2.y >> 16); }
                }
                if (p.act == VV_ACT_RELU) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                else if (p.act == VV_ACT_LRELU) { v[0] = v[0] > 0.f ? v[0] : v[0] * p.act_slope; v[1] = v[1] > 0.f ? v[1] : v[1] * p.act_slope; v[2] = v[2] > 0.f ? v[2] : v[2] * p.act_slope; v[3] = v[3] > 0.f ? v[3] : v[3] * p.act_slope; }
                // head-major QKV store: out[b][which][head][token][d] = row part + column part
                const int64_t oc = split ? rowpart + colpart[j] : (int64_t)m * p.ldo + n;
                if (p.out_dtype == VV_F32) *(float4*)((float*)p.out + oc) = make_float4(v[0], v[1], v[2], v[3]);
                else *(uint2*)((unsigned short*)p.out + oc) = make_uint2(pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3]));
            }
        }
        return;
    }
    if (vec) {
        // (the bias is re-read per strip instead of being held in NT float4 registers: it keeps the kernels under 128 VGPRs)
        int colpart[NT];          // split_heads store: the column's (which, head, d) part of the output index (< 3*C*tokens)
        const bool split = p.split_heads > 0;
        const int stok = p.split_tokens < 0 ? -p.split_tokens : p.split_tokens;      // tokens per batch element
        const int snb = p.split_tokens < 0 ? (p.F * HWo) / stok : 0;                 // token-major rows: number of batch elements
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = ncol0 + j * 16 + 4 * lq;
            colpart[j] = 0;
            if (split) {
                const int wh = n / p.split_dim;                       // which * heads + head
                colpart[j] = wh * stok * p.split_dim + (n - wh * p.split_dim);
            }
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            bool mok;
            const int m = row_m(row0 + i * 16 + lr, mok);
            const int64_t rbase = (int64_t)(mok ? m : 0) * N;
            float4 ra4[NT];
            // issue every res0 load of this strip first (the common residual); rarer addends are read in place
            if (p.res0 && r0f32) {
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int n = ncol0 + j * 16 + 4 * lq;
                    ra4[j] = (mok && n < N) ? *(const float4*)((const float*)p.res0 + rbase + n) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
            if (!mok) continue;
            const float* rowv = p.rowvec ? p.rowvec + (int64_t)(m / HWo) * N : nullptr;
            int64_t rowpart = 0;
            if (split) {
                int b, tok;
                if (snb) { tok = m / snb; b = m - tok * snb; } else { b = m / stok; tok = m - b * stok; }
                rowpart = ((int64_t)b * 3 * p.split_heads * stok + tok) * p.split_dim;
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int n = ncol0 + j * 16 + 4 * lq;
                if (n >= N) continue;
                // (sbias: the block's bias columns staged in LDS by the kernel -- an LDS read waits on lgkmcnt, not on vmcnt: the previous tile's store stays in flight)
                const float4 bj = sbias ? *(const float4*)(sbias + j * 16 + 4 * lq) : (p.bias ? *(const float4*)(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f));
                float v[4] = {(acc[i][j][0] + bj.x) * p.out_scale, (acc[i][j][1] + bj.y) * p.out_scale,
                              (acc[i][j][2] + bj.z) * p.out_scale, (acc[i][j][3] + bj.w) * p.out_scale};
                if (rowv) { const float4 t4 = *(const float4*)(rowv + n); v[0] += t4.x; v[1] += t4.y; v[2] += t4.z; v[3] += t4.w; }
                if (p.res0) {
                    if (r0f32) { v[0] += ra4[j].x; v[1] += ra4[j].y; v[2] += ra4[j].z; v[3] += ra4[j].w; }
                    else { const uint2 r2 = *(const uint2*)((const unsigned short*)p.res0 + rbase + n); v[0] += T::to_f32(r2.x & 0xffff); v[1] += T::to_f32(r2.x >> 16); v[2] += T::to_f32(r2.y & 0xffff); v[3] += T::to_f32(r2.y >> 16); }
                }
                if (p.res1) {
                    if (r0f32) { const float4 r4 = *(const float4*)((const float*)p.res1 + rbase + n); v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w; }
                    else { const uint2 r2 = *(const uint2*)((const unsigned short*)p.res1 + rbase + n); v[0] += T::to_f32(r2.x & 0xffff); v[1] += T::to_f32(r2.x >> 16); v[2] += T::to_f32(r2.y & 0xffff); v[3] += T::to_f32(r2.y >> 16); }
                }
                if (p.act == VV_ACT_RELU) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                else if (p.act == VV_ACT_LRELU) { v[0] = v[0] > 0.f ? v[0] : v[0] * p.act_slope; v[1] = v[1] > 0.f ? v[1] : v[1] * p.act_slope; v[2] = v[2] > 0.f ? v[2] : v[2] * p.act_slope; v[3] = v[3] > 0.f ? v[3] : v[3] * p.act_slope; }
                // head-major QKV store: out[b][which][head][token][d] = row part + column part
                const int64_t oc = split ? rowpart + colpart[j] : (int64_t)m * p.ldo + n;
                if (p.out_dtype == VV_F32) *(float4*)((float*)p.out + oc) = make_float4(v[0], v[1], v[2], v[3]);
                else *(uint2*)((unsigned short*)p.out + oc) = make_uint2(pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3]));
            }
        }
        return;
    }
    // scalar tail path (N or ldo not a multiple of 4: e.g. the 3-channel VAE conv_out)
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        bool mok;
        const int m = row_m(row0 + i * 16 + lr, mok);
        if (!mok) continue;
        const float* rowv = p.rowvec ? p.rowvec + (int64_t)(m / HWo) * N : nullptr;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = ncol0 + j * 16 + 4 * lq;
            const int64_t ri = (int64_t)m * N + n;
            const int64_t oc = (int64_t)m * p.ldo + n;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (n + r >= N) continue;
                float x = acc[i][j][r];
                if (p.bias) x += p.bias[n + r];
                x *= p.out_scale;
                if (rowv) x += rowv[n + r];
                if (p.res0) x += r0f32 ? ((const float*)p.res0)[ri + r] : T::to_f32(((const unsigned short*)p.res0)[ri + r]);
                if (p.res1) x += r0f32 ? ((const float*)p.res1)[ri + r] : T::to_f32(((const unsigned short*)p.res1)[ri + r]);
                if (p.act == VV_ACT_RELU) x = fmaxf(x, 0.f);
                else if (p.act == VV_ACT_LRELU) x = x > 0.f ? x : x * p.act_slope;
                if (p.out_dtype == VV_F32) ((float*)p.out)[oc + r] = x;
                else ((unsigned short*)p.out)[oc + r] = T::from_f32(x);
            }
        }
    }
}
