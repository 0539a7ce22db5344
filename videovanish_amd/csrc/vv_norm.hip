// K2: GroupNorm(+SiLU) and LayerNorm(+positional embedding) for frames-major NHWC activations.  HBM-bound:
// every element is read twice (statistics pass, apply pass) with 16/32-byte vector loads and written once as h16.
#include "vv_common.h"

namespace {

struct GNGeom {
    int C8;          // 8-channel chunks per pixel
    int krows;      // pixel rows handled per block pass
    int threads;     // active threads = krows * C8
    int rows_per_split;
    int nsplit;
};

__host__ __device__ inline GNGeom gn_geom(int HW, int C) {
    GNGeom g;
    g.C8 = C / 8;
    g.krows = g.C8 >= 256 ? 1 : 256 / g.C8;
    g.threads = g.krows * g.C8;
    int rows = 32768 / C; if (rows < g.krows) rows = g.krows;
    rows = (rows + g.krows - 1) / g.krows * g.krows;
    int ns = (HW + rows - 1) / rows;
    if (ns > 128) { ns = 128; rows = ((HW + ns - 1) / ns + g.krows - 1) / g.krows * g.krows; ns = (HW + rows - 1) / rows; }
    g.rows_per_split = rows; g.nsplit = ns;
    return g;
}

// Partial statistics of one (frame, row-split): DETERMINISTIC reduction (no float atomics): every thread parks its 8
// per-channel sums in LDS, then one thread per group adds the group's channels over the block's row lanes in a fixed
// order.  (The multi-GPU path promises bit-identical output for every world size: nothing may depend on arrival order.)
template <typename T>
__global__ void gn_stats_kernel(const vv_groupnorm_params p, const GNGeom g) {
    extern __shared__ float part[];          // [2][krows][C]
    const int t = threadIdx.x, split = blockIdx.x, f = blockIdx.y;
    const int C = p.C0 + p.C1, cpg = C / p.groups;
    float* ps = part;
    float* pq = part + g.krows * C;
    if (t < g.threads) {
        const int chunk = t % g.C8, r0 = t / g.C8;
        float s[8], q[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
        const int rbeg = split * g.rows_per_split;
        const int rend = min(rbeg + g.rows_per_split, p.HW);
        int r = rbeg + r0;
        // Four rows per pass, all loads first (round 6).  The one-row loop kept ONE 32-byte load per thread in flight (load, s_waitcnt vmcnt(0), accumulate: ~64 KB
        // per CU, what Little's law asks for at HBM latency with nothing to spare), and gn_load8's source / type branches sit between the loads of an unrolled
        // loop -- so the thread's source, row stride and element type are resolved ONCE, outside the loop.  The order of the additions per thread is unchanged
        // (row r, r + krows, ...): bit-identical statistics.
        const unsigned char* src = (const unsigned char*)p.in0;
        int Cs = p.C0, c = chunk * 8;
        if (c >= p.C0) { src = (const unsigned char*)p.in1; c -= p.C0; Cs = p.C1; }
        const int64_t pix0 = (int64_t)f * p.HW;
        auto acc8 = [&](const float* v) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { s[e] += v[e]; q[e] += v[e] * v[e]; }
        };
        if (p.in_dtype == VV_F32) {
            const float* b32 = (const float*)src + c;
            for (; r + 3 * g.krows < rend; r += 4 * g.krows) {
                float4 v[4][2];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const float4* gp = (const float4*)(b32 + (pix0 + r + u * g.krows) * Cs); v[u][0] = gp[0]; v[u][1] = gp[1]; }
#pragma unroll
                for (int u = 0; u < 4; ++u) acc8((const float*)&v[u][0]);
            }
            for (; r < rend; r += g.krows) {
                const float4* gp = (const float4*)(b32 + (pix0 + r) * Cs);
                float4 v[2] = {gp[0], gp[1]};
                acc8((const float*)&v[0]);
            }
        } else {
            const unsigned short* b16 = (const unsigned short*)src + c;
            for (; r + 3 * g.krows < rend; r += 4 * g.krows) {
                uint4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = *(const uint4*)(b16 + (pix0 + r + u * g.krows) * Cs);
#pragma unroll
                for (int u = 0; u < 4; ++u) { float w[8]; unpack8<T>(v[u], w); acc8(w); }
            }
            for (; r < rend; r += g.krows) {
                float w[8];
                unpack8<T>(*(const uint4*)(b16 + (pix0 + r) * Cs), w);
                acc8(w);
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) { ps[r0 * C + chunk * 8 + e] = s[e]; pq[r0 * C + chunk * 8 + e] = q[e]; }
    }
    __syncthreads();
    float* ws = p.stats_ws + ((int64_t)f * g.nsplit + split) * p.groups * 2;
    for (int grp = t; grp < p.groups; grp += blockDim.x) {
        float ss = 0.f, qq = 0.f;
        for (int r = 0; r < g.krows; ++r)
            for (int c = grp * cpg; c < (grp + 1) * cpg; ++c) { ss += ps[r * C + c]; qq += pq[r * C + c]; }
        ws[2 * grp] = ss; ws[2 * grp + 1] = qq;
    }
}

// one block per frame (or one block when statistics pool over the clip): partials -> mean / rstd.
// 256 threads: thread (grp, lane8) sums a strided share of the partials in double, then an 8-lane shuffle reduce.
__global__ __launch_bounds__(256) void gn_finalize_kernel(const vv_groupnorm_params p, const GNGeom g) {
    const int C = p.C0 + p.C1, cpg = C / p.groups;
    const int f0 = p.pool_frames ? 0 : blockIdx.x, f1 = p.pool_frames ? p.F : blockIdx.x + 1;
    const int sub = threadIdx.x & 7;
    float* fin = p.stats_ws + (int64_t)p.F * g.nsplit * p.groups * 2;
    for (int grp = threadIdx.x >> 3; grp < p.groups; grp += 32) {   // 256 threads = 32 groups per pass
        double s = 0.0, q = 0.0;
        const int nparts = (f1 - f0) * g.nsplit;
        for (int i = sub; i < nparts; i += 8) {
            const float* ws = p.stats_ws + (((int64_t)f0 * g.nsplit + i) * p.groups + grp) * 2;
            s += ws[0]; q += ws[1];
        }
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
        const double n = (double)(f1 - f0) * p.HW * cpg;
        const double mean = s / n;
        double var = q / n - mean * mean;
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)p.eps));
        if (p.pool_frames) {
            for (int f = sub; f < p.F; f += 8) { fin[((int64_t)f * p.groups + grp) * 2] = (float)mean; fin[((int64_t)f * p.groups + grp) * 2 + 1] = rstd; }
        } else if (sub == 0) {
            fin[((int64_t)blockIdx.x * p.groups + grp) * 2] = (float)mean; fin[((int64_t)blockIdx.x * p.groups + grp) * 2 + 1] = rstd;
        }
    }
}

// pooled statistics (motion modules: one (mean, rstd) per group over the whole clip): ONE BLOCK PER GROUP sums the F * nsplit partials of its
// group -- thread t takes partials t, t + 256, ... in double, then a fixed-order LDS tree -- instead of the single block of gn_finalize_kernel
// walking 4096 partials with 8 lanes per group (0.17 ms of pure latency per launch at level 0).  Same order for every launch: deterministic.
__global__ __launch_bounds__(256) void gn_finalize_pooled_kernel(const vv_groupnorm_params p, const GNGeom g) {
    __shared__ double ss[256], sq[256];
    const int C = p.C0 + p.C1, cpg = C / p.groups, grp = blockIdx.x, t = threadIdx.x;
    const int nparts = p.F * g.nsplit;
    double s = 0.0, q = 0.0;
    for (int i = t; i < nparts; i += 256) {
        const float* ws = p.stats_ws + ((int64_t)i * p.groups + grp) * 2;
        s += ws[0]; q += ws[1];
    }
    ss[t] = s; sq[t] = q;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) { ss[t] += ss[t + o]; sq[t] += sq[t + o]; }
        __syncthreads();
    }
    const double n = (double)p.F * p.HW * cpg;
    const double mean = ss[0] / n;
    double var = sq[0] / n - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)p.eps));
    float* fin = p.stats_ws + (int64_t)p.F * g.nsplit * p.groups * 2;
    for (int f = t; f < p.F; f += 256) { fin[((int64_t)f * p.groups + grp) * 2] = (float)mean; fin[((int64_t)f * p.groups + grp) * 2 + 1] = rstd; }
}

template <typename T>
__global__ void gn_apply_kernel(const vv_groupnorm_params p, const GNGeom g, const float* fin_all /* [F][groups][2] (mean, rstd) */) {
    const int t = threadIdx.x, split = blockIdx.x, f = blockIdx.y;
    if (t >= g.threads) return;
    const int C = p.C0 + p.C1, cpg = C / p.groups;
    const int chunk = t % g.C8, r0 = t / g.C8;
    const float* fin = fin_all + (int64_t)f * p.groups * 2;
    float a[8], b[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = chunk * 8 + e, grp = c / cpg;
        const float mean = fin[2 * grp], rstd = fin[2 * grp + 1];
        a[e] = rstd * p.gamma[c];
        b[e] = p.beta[c] - mean * a[e];
    }
    const int rbeg = split * g.rows_per_split;
    const int rend = min(rbeg + g.rows_per_split, p.HW);
    auto emit = [&](float* v, int64_t pix) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float y = v[e] * a[e] + b[e];
            v[e] = p.silu == VV_ACT_SILU ? silu_f(y) : (p.silu == VV_ACT_RELU ? fmaxf(y, 0.f) : y);
        }
        const int64_t o = pix * C + chunk * 8;
        if (p.out_dtype == VV_F32) { float4* d = (float4*)((float*)p.out + o); d[0] = *(float4*)&v[0]; d[1] = *(float4*)&v[4]; }
        else if (p.out_dtype == VV_SPLIT3) {
            // K-concatenated split-precision operand (vv_split3): [pix][hi | (y - hi) * 2^4 | hi * 2^-10], three C-channel groups
            float lo[8], g3[8];
            const uint4 hi = pack8<T>(v);
            float hf[8];
            unpack8<T>(hi, hf);
#pragma unroll
            for (int e = 0; e < 8; ++e) { lo[e] = (v[e] - hf[e]) * 16.0f; g3[e] = hf[e] * 0.0009765625f; }
            unsigned short* d = (unsigned short*)p.out + pix * 3 * C + chunk * 8;
            *(uint4*)d = hi; *(uint4*)(d + C) = pack8<T>(lo); *(uint4*)(d + 2 * C) = pack8<T>(g3);
        }
        else *(uint4*)((unsigned short*)p.out + o) = pack8<T>(v);
    };
    // Four rows per pass, all loads first: the input and the output may alias as far as the compiler knows, so in a one-row loop the load of row i + 1 follows
    // the store of row i and its s_waitcnt vmcnt(0) drains that store -- one serialised memory round trip per row and wave (round 5, second session)
    // (round 6) ... and the thread's source, row stride and element type are resolved ONCE, outside the loop: gn_load8's source / type branches sat between the
    // four loads of a pass, each in its own basic block behind an s_waitcnt vmcnt(0) -- the four loads were never in flight together
    int r = rbeg + r0;
    const int64_t base = (int64_t)f * p.HW;
    const unsigned char* src = (const unsigned char*)p.in0;
    int Cs = p.C0, c = chunk * 8;
    if (c >= p.C0) { src = (const unsigned char*)p.in1; c -= p.C0; Cs = p.C1; }
    if (p.in_dtype == VV_F32) {
        const float* b32 = (const float*)src + c;
        for (; r + 3 * g.krows < rend; r += 4 * g.krows) {
            float4 v[4][2];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const float4* gp = (const float4*)(b32 + (base + r + u * g.krows) * Cs); v[u][0] = gp[0]; v[u][1] = gp[1]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) emit((float*)&v[u][0], base + r + u * g.krows);
        }
        for (; r < rend; r += g.krows) {
            const float4* gp = (const float4*)(b32 + (base + r) * Cs);
            float4 v[2] = {gp[0], gp[1]};
            emit((float*)&v[0], base + r);
        }
    } else {
        const unsigned short* b16 = (const unsigned short*)src + c;
        for (; r + 3 * g.krows < rend; r += 4 * g.krows) {
            uint4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *(const uint4*)(b16 + (base + r + u * g.krows) * Cs);
#pragma unroll
            for (int u = 0; u < 4; ++u) { float w[8]; unpack8<T>(v[u], w); emit(w, base + r + u * g.krows); }
        }
        for (; r < rend; r += g.krows) {
            float w[8];
            unpack8<T>(*(const uint4*)(b16 + (base + r) * Cs), w);
            emit(w, base + r);
        }
    }
}

// LayerNorm: one wave per row, row held in registers (C <= 64*4*NCH)
template <typename T, int NCH>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int M, int C, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const float* __restrict__ pe, int rows_per_frame,
                                                        unsigned short* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int C4 = C >> 2;
    const float4* xr = (const float4*)(x + (int64_t)row * C);
    float4 v[NCH];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = lane + 64 * i;
        v[i] = c < C4 ? xr[c] : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = lane + 64 * i;
        if (c < C4) {
            const float dx = v[i].x - mean, dy = v[i].y - mean, dz = v[i].z - mean, dw = v[i].w - mean;
            q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = rsqrtf(q / (float)C + 1e-5f);
    const float4* per = pe ? (const float4*)(pe + (int64_t)(row / rows_per_frame) * C) : nullptr;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = lane + 64 * i;
        if (c < C4) {
            const float4 g = ((const float4*)gamma)[c], b = ((const float4*)beta)[c];
            float4 y;
            y.x = (v[i].x - mean) * rstd * g.x + b.x; y.y = (v[i].y - mean) * rstd * g.y + b.y;
            y.z = (v[i].z - mean) * rstd * g.z + b.z; y.w = (v[i].w - mean) * rstd * g.w + b.w;
            if (per) { const float4 e = per[c]; y.x += e.x; y.y += e.y; y.z += e.z; y.w += e.w; }
            uint2 o2 = make_uint2(pack2<T>(y.x, y.y), pack2<T>(y.z, y.w));
            *(uint2*)(out + (int64_t)row * C + 4 * c) = o2;
        }
    }
}

template <typename T>
int gn_launch(const vv_groupnorm_params& p, hipStream_t st, bool apply = true) {
    const int C = p.C0 + p.C1;
    const GNGeom g = gn_geom(p.HW, C);
    const int threads = (g.threads + 63) / 64 * 64;
    hipLaunchKernelGGL(gn_stats_kernel<T>, dim3(g.nsplit, p.F), dim3(threads), (size_t)2 * g.krows * C * sizeof(float), st, p, g);
    if (p.pool_frames) hipLaunchKernelGGL(gn_finalize_pooled_kernel, dim3(p.groups), dim3(256), 0, st, p, g);
    else hipLaunchKernelGGL(gn_finalize_kernel, dim3(p.F), dim3(256), 0, st, p, g);
    if (apply) hipLaunchKernelGGL(gn_apply_kernel<T>, dim3(g.nsplit, p.F), dim3(threads), 0, st, p, g, (const float*)(p.stats_ws + (int64_t)p.F * g.nsplit * p.groups * 2));
    VV_CHECK_LAUNCH("vv_groupnorm");
    return VV_OK;
}

// (mean, rstd) from per-channel partials [F][nblk][C][2] written by a producing layer's epilogue (vv_conv_params.gn_partials): one block per (group, frame) --
// or per group when the statistics pool over the clip -- thread t takes items t, t + 256, ... in double, then a fixed-order LDS tree: deterministic.
__global__ __launch_bounds__(256) void gn_finalize_partials_kernel(const float* part, int F, int nblk, int C, int HW, int groups, float eps, int pool, float* fin) {
    __shared__ double ss[256], sq[256];
    const int grp = blockIdx.x, t = threadIdx.x, cpg = C / groups;
    const int f0 = pool ? 0 : blockIdx.y, nf = pool ? F : 1;
    const int items = nf * nblk * cpg;
    double s = 0.0, q = 0.0;
    for (int i = t; i < items; i += 256) {
        const int c = i % cpg, b = i / cpg;           // b = (frame - f0) * nblk + row block
        const float* w = part + (((int64_t)f0 * nblk + b) * C + grp * cpg + c) * 2;
        s += w[0]; q += w[1];
    }
    ss[t] = s; sq[t] = q;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) { ss[t] += ss[t + o]; sq[t] += sq[t + o]; }
        __syncthreads();
    }
    const double n = (double)nf * HW * cpg;
    const double mean = ss[0] / n;
    double var = sq[0] / n - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    if (pool) { for (int f = t; f < F; f += 256) { fin[((int64_t)f * groups + grp) * 2] = (float)mean; fin[((int64_t)f * groups + grp) * 2 + 1] = rstd; } }
    else if (t == 0) { fin[((int64_t)f0 * groups + grp) * 2] = (float)mean; fin[((int64_t)f0 * groups + grp) * 2 + 1] = rstd; }
}

}  // namespace

extern "C" int vv_gn_finalize_partials(const float* partials, int F, int nblk, int C, int HW, int groups, float eps, int pool_frames, float* fin, void* stream) {
    if (!partials || !fin || F <= 0 || nblk <= 0 || C <= 0 || HW <= 0 || groups <= 0 || C % groups) VV_FAIL(VV_E_ARG, "vv_gn_finalize_partials: bad arguments");
    hipLaunchKernelGGL(gn_finalize_partials_kernel, dim3(groups, pool_frames ? 1 : F), dim3(256), 0, (hipStream_t)stream, partials, F, nblk, C, HW, groups, eps, pool_frames, fin);
    VV_CHECK_LAUNCH("vv_gn_finalize_partials");
    return VV_OK;
}

extern "C" int vv_groupnorm_apply_fin(const vv_groupnorm_params* pp, const float* fin, int dtype, void* stream) {
    if (!pp || !fin) VV_FAIL(VV_E_ARG, "vv_groupnorm_apply_fin: null pointer");
    const vv_groupnorm_params& p = *pp;
    const int C = p.C0 + p.C1;
    if (dtype != VV_BF16 && dtype != VV_F16) VV_FAIL(VV_E_ARG, "vv_groupnorm_apply_fin: bad dtype");
    if (!p.in0 || !p.out || !p.gamma || !p.beta) VV_FAIL(VV_E_ARG, "vv_groupnorm_apply_fin: null pointer");
    if (p.C0 <= 0 || p.C0 % 8 || p.C1 % 8 || (p.C1 > 0 && !p.in1) || p.groups <= 0 || p.groups > 256 || C % p.groups || C / 8 > 1024) VV_FAIL(VV_E_ARG, "vv_groupnorm_apply_fin: channels / groups");
    if ((p.in_dtype != VV_F32 && p.in_dtype != dtype) || (p.out_dtype != VV_F32 && p.out_dtype != dtype && p.out_dtype != VV_SPLIT3)) VV_FAIL(VV_E_ARG, "vv_groupnorm_apply_fin: dtype mismatch");
    if (p.F <= 0 || p.HW <= 0) VV_FAIL(VV_E_ARG, "vv_groupnorm_apply_fin: empty input");
    const GNGeom g = gn_geom(p.HW, C);
    const int threads = (g.threads + 63) / 64 * 64;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == VV_BF16) hipLaunchKernelGGL(gn_apply_kernel<BF16>, dim3(g.nsplit, p.F), dim3(threads), 0, st, p, g, fin);
    else hipLaunchKernelGGL(gn_apply_kernel<F16>, dim3(g.nsplit, p.F), dim3(threads), 0, st, p, g, fin);
    VV_CHECK_LAUNCH("vv_groupnorm_apply_fin");
    return VV_OK;
}

extern "C" int vv_groupnorm_nsplit(int HW, int C) { return gn_geom(HW, C).nsplit; }

extern "C" int vv_groupnorm(const vv_groupnorm_params* pp, int dtype, void* stream) {
    if (!pp) VV_FAIL(VV_E_ARG, "vv_groupnorm: null params");
    const vv_groupnorm_params& p = *pp;
    const int C = p.C0 + p.C1;
    if (dtype != VV_BF16 && dtype != VV_F16) VV_FAIL(VV_E_ARG, "vv_groupnorm: bad dtype");
    if (!p.in0 || !p.out || !p.gamma || !p.beta || !p.stats_ws) VV_FAIL(VV_E_ARG, "vv_groupnorm: null pointer");
    if (p.C0 <= 0 || p.C0 % 8 || p.C1 % 8 || (p.C1 > 0 && !p.in1)) VV_FAIL(VV_E_ARG, "vv_groupnorm: channels must be multiples of 8 (C0=%d C1=%d)", p.C0, p.C1);
    if (p.groups <= 0 || p.groups > 256 || C % p.groups) VV_FAIL(VV_E_ARG, "vv_groupnorm: groups=%d C=%d", p.groups, C);
    if (C / 8 > 1024) VV_FAIL(VV_E_UNSUPPORTED, "vv_groupnorm: C=%d too large", C);
    if (p.in_dtype != VV_F32 && p.in_dtype != dtype) VV_FAIL(VV_E_ARG, "vv_groupnorm: in_dtype mismatch");
    if (p.out_dtype != VV_F32 && p.out_dtype != dtype && p.out_dtype != VV_SPLIT3) VV_FAIL(VV_E_ARG, "vv_groupnorm: out_dtype mismatch");
    if (p.F <= 0 || p.HW <= 0) VV_FAIL(VV_E_ARG, "vv_groupnorm: empty input");
    return dtype == VV_BF16 ? gn_launch<BF16>(p, (hipStream_t)stream) : gn_launch<F16>(p, (hipStream_t)stream);
}

// statistics only: (mean, rstd) of frame f, group g at stats_ws[(F * nsplit + f) * groups * 2 + 2 g] (pooled: identical for every f)
extern "C" int vv_groupnorm_stats(const vv_groupnorm_params* pp, int dtype, void* stream) {
    if (!pp) VV_FAIL(VV_E_ARG, "vv_groupnorm_stats: null params");
    const vv_groupnorm_params& p = *pp;
    const int C = p.C0 + p.C1;
    if (dtype != VV_BF16 && dtype != VV_F16) VV_FAIL(VV_E_ARG, "vv_groupnorm_stats: bad dtype");
    if (!p.in0 || !p.stats_ws) VV_FAIL(VV_E_ARG, "vv_groupnorm_stats: null pointer");
    if (p.C0 <= 0 || p.C0 % 8 || p.C1 % 8 || (p.C1 > 0 && !p.in1)) VV_FAIL(VV_E_ARG, "vv_groupnorm_stats: channels must be multiples of 8");
    if (p.groups <= 0 || p.groups > 256 || C % p.groups || C / 8 > 1024) VV_FAIL(VV_E_ARG, "vv_groupnorm_stats: groups=%d C=%d", p.groups, C);
    if (p.in_dtype != VV_F32 && p.in_dtype != dtype) VV_FAIL(VV_E_ARG, "vv_groupnorm_stats: in_dtype mismatch");
    if (p.F <= 0 || p.HW <= 0) VV_FAIL(VV_E_ARG, "vv_groupnorm_stats: empty input");
    return dtype == VV_BF16 ? gn_launch<BF16>(p, (hipStream_t)stream, false) : gn_launch<F16>(p, (hipStream_t)stream, false);
}

extern "C" int vv_layernorm(const float* x, int M, int C, const float* gamma, const float* beta, const float* pe, int rows_per_frame,
                            void* out, int dtype, void* stream) {
    if (!x || !gamma || !beta || !out) VV_FAIL(VV_E_ARG, "vv_layernorm: null pointer");
    if (M <= 0 || C <= 0 || C % 4) VV_FAIL(VV_E_ARG, "vv_layernorm: M=%d C=%d (C must be a multiple of 4)", M, C);
    if (C > 2048) VV_FAIL(VV_E_UNSUPPORTED, "vv_layernorm: C=%d > 2048", C);
    if (pe && rows_per_frame <= 0) VV_FAIL(VV_E_ARG, "vv_layernorm: rows_per_frame");
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((M + 3) / 4), blk(256);
    unsigned short* o = (unsigned short*)out;
#define LN_LAUNCH(TT, N) hipLaunchKernelGGL((layernorm_kernel<TT, N>), grid, blk, 0, st, x, M, C, gamma, beta, pe, rows_per_frame, o)
    if (dtype == VV_BF16) { if (C <= 512) LN_LAUNCH(BF16, 2); else if (C <= 1280) LN_LAUNCH(BF16, 5); else LN_LAUNCH(BF16, 8); }
    else if (dtype == VV_F16) { if (C <= 512) LN_LAUNCH(F16, 2); else if (C <= 1280) LN_LAUNCH(F16, 5); else LN_LAUNCH(F16, 8); }
    else VV_FAIL(VV_E_ARG, "vv_layernorm: bad dtype");
#undef LN_LAUNCH
    VV_CHECK_LAUNCH("vv_layernorm");
    return VV_OK;
}
