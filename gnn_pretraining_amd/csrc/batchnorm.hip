// Segment-wise BatchNorm1d fused with residual add, ReLU and dropout.
//
// A "segment" = the rows of one reference forward() call, so several independent
// forwards (7 per domain per step in the s4 scheme) share a launch while keeping
// their own batch statistics.  Two regimes, picked from the longest segment:
//   short segments (<= SHORT_MAX rows): one workgroup owns (segment, 32 columns) and
//     keeps that tile in registers: one read, exact two-pass statistics (mean, then
//     centred variance), apply, one write -- no atomics, one launch;
//   long segments (Cora, the roofline ladder): row chunks produce (mean, M2)
//     partials, combined in chunk order with Chan's formula, then an apply pass.
// Both are deterministic.  Memory-bound; a thread moves float4.
#include "gnnmp_internal.h"

namespace {

constexpr int THREADS = 256;
constexpr int COLS = 64;          // columns per block
constexpr int CQ = COLS / 4;      // float4 column quads per block (16)
constexpr int RL = THREADS / CQ;  // row lanes per block (16)
constexpr int SHORT_MAX = 1024;
// Medium segments (SHORT_MAX < rows <= MID_MAX: the 2,708-node Cora graph is ONE segment): still one read per element, by a 1,024-thread
// workgroup per (segment, 32 columns) -- 16 columns x 256 row lanes, up to 12 rows per thread in registers.  The chunked long regime took four launches
// forward and four backward for it (partial / finalize / apply / running: 712 of the fine-tune step's 1,468 us of kernel time, r03 trace).
constexpr int MID_MAX = 3072, MID_RL = 256, MID_CQ = 4, MID_COLS = 4 * MID_CQ;      // 16-column tiles: at 32 columns x 128 row lanes the tile spilled
#define GMP_BN_MID_LAUNCH(KERNEL, MAXROWS, S_, C_, ST, ARGS)                                                             \
    do {                                                                                                                 \
        const dim3 gridm(S_, (C_) / MID_COLS), blkm(MID_RL * MID_CQ);                                                    \
        if ((MAXROWS) <= 6 * MID_RL) hipLaunchKernelGGL((KERNEL<6, MID_RL, MID_CQ>), gridm, blkm, 0, ST, ARGS);          \
        else if ((MAXROWS) <= 8 * MID_RL) hipLaunchKernelGGL((KERNEL<8, MID_RL, MID_CQ>), gridm, blkm, 0, ST, ARGS);     \
        else if ((MAXROWS) <= 10 * MID_RL) hipLaunchKernelGGL((KERNEL<10, MID_RL, MID_CQ>), gridm, blkm, 0, ST, ARGS);   \
        else hipLaunchKernelGGL((KERNEL<12, MID_RL, MID_CQ>), gridm, blkm, 0, ST, ARGS);                                 \
    } while (0)
constexpr int CHUNK = 256;        // rows per chunk in the long regime

struct BnArgs {
    const float* x;
    const float* res;
    const float* gy;
    const int* seg_ptr;
    const int* seg_group;   // nullable: parameter group of each segment (per-domain encoders); gamma/beta/running are [groups][C]
    int S;
    int C;
    const float* gamma;
    const float* beta;
    float* running_mean;
    float* running_var;
    float* save_mean;
    float* save_rstd;
    float* y;       // fwd: output; bwd: g_u
    float* part;    // long regime: chunk partials
    float* segsum;  // bwd: [S][2][C] (sum g, sum g*xhat)
    int chunks;     // chunks per segment (long regime)
    int slabs;      // row slabs per segment (slab regime)
    float* pg_gamma;   // slab regime, one segment = one gradient group: the parameter gradients, written by the kernel itself (nullable)
    float* pg_beta;
    int fold_running;  // slab regime, one segment: the forward kernel also folds the batch statistics into running_mean / running_var
    gmp_bn_config cfg;
};

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 sub4(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 mul4(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 scl4(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// reduce a float4 over the RL row lanes of the block (same column quad); result to all
__device__ __forceinline__ float4 block_colsum(float4 v, float4 (*sh)[CQ], int rl, int cq) {
    __syncthreads();
    sh[rl][cq] = v;
    __syncthreads();
    float4 s = sh[0][cq];
#pragma unroll
    for (int i = 1; i < RL; ++i) s = add4(s, sh[i][cq]);
    return s;
}

__device__ __forceinline__ float4 load_u(const BnArgs& a, int64_t off) {
    float4 u = ld4(a.x + off);
    if (a.res) u = add4(u, ld4(a.res + off));
    return u;
}

// dropout seed of this launch: the configured one, plus the device word when the caller passed one (hipGraph replays: gnnmp.h gmp_bn_config)
__device__ __forceinline__ uint64_t bn_seed(const BnArgs& a) { return a.cfg.seed_dev ? a.cfg.seed + *a.cfg.seed_dev : a.cfg.seed; }

// y = dropout(relu(gamma*xhat+beta)) for one float4; returns the pre-dropout activation mask in `pos`
__device__ __forceinline__ float4 bn_apply(const BnArgs& a, float4 u, float4 mean, float4 rstd, float4 gam, float4 bet,
                                           int64_t elem_quad, float4* gate) {
    float4 xh = mul4(sub4(u, mean), rstd);
    float4 y = make_float4(fmaf(gam.x, xh.x, bet.x), fmaf(gam.y, xh.y, bet.y), fmaf(gam.z, xh.z, bet.z), fmaf(gam.w, xh.w, bet.w));
    float4 g = make_float4(1.f, 1.f, 1.f, 1.f);
    if (a.cfg.relu) {
        g = make_float4(y.x > 0.f, y.y > 0.f, y.z > 0.f, y.w > 0.f);
        y = make_float4(fmaxf(y.x, 0.f), fmaxf(y.y, 0.f), fmaxf(y.z, 0.f), fmaxf(y.w, 0.f));
    }
    if (a.cfg.dropout_p > 0.f) {
        float4 d = gmp::dropout_scale4(bn_seed(a), a.cfg.stream_id, (uint64_t)elem_quad, a.cfg.dropout_p,
                                       1.f / (1.f - a.cfg.dropout_p));
        y = mul4(y, d);
        g = mul4(g, d);
    }
    *gate = g;
    return y;
}

__device__ __forceinline__ int64_t pgrp(const BnArgs& a, int s) { return a.seg_group ? (int64_t)a.seg_group[s] * a.C : 0; }

__device__ __forceinline__ void stats_for(const BnArgs& a, int s, int c, float4* mean, float4* rstd) {
    if (a.cfg.training) {
        *mean = ld4(a.save_mean + (int64_t)s * a.C + c);
        *rstd = ld4(a.save_rstd + (int64_t)s * a.C + c);
    } else {
        *mean = ld4(a.running_mean + pgrp(a, s) + c);
        float4 v = ld4(a.running_var + pgrp(a, s) + c);
        *rstd = make_float4(rsqrtf(v.x + a.cfg.eps), rsqrtf(v.y + a.cfg.eps), rsqrtf(v.z + a.cfg.eps), rsqrtf(v.w + a.cfg.eps));
    }
}

// ------------------------------------------------------------ short regime, fwd
// Register-resident: a workgroup owns (segment, 32 columns); its 256 threads are 8 column quads x 32 row
// lanes and each thread keeps its <= RPT rows of the tile in registers, so the tile is read from memory
// ONCE with every load in flight together, instead of three latency-bound sweeps.
constexpr int SCOLS = 32, SCQ = SCOLS / 4, SRL = THREADS / SCQ;   // 8 quads x 32 row lanes

// Column sum over the RL_ row lanes of a (segment, 32 columns) block: thread (rl, cq) sits at threadIdx.x = rl * 8 + cq, so a wave holds
// eight row lanes of every column quad -- three shuffle steps fold those, the RL_ / 8 waves meet in LDS.  (The first version summed RL_
// LDS entries per thread: a 32- or 64-deep chain of dependent reads, twice per BatchNorm forward, on the step's critical path.)
template <int RL_, int CQ_>
__device__ __forceinline__ float4 colsum_t(float4 v, float4 (*sh)[CQ_], int rl, int cq) {
    constexpr int RPW = 64 / CQ_;                     // row lanes of one wave (8 column quads: 8, 4 column quads: 16)
    static_assert((CQ_ == 8 || CQ_ == 4) && RL_ % RPW == 0, "layout: CQ_ column quads x RL_ row lanes, 64 / CQ_ row lanes per wave");
#pragma unroll
    for (int o = CQ_; o < 64; o <<= 1)
        v = make_float4(v.x + __shfl_xor(v.x, o, 64), v.y + __shfl_xor(v.y, o, 64), v.z + __shfl_xor(v.z, o, 64), v.w + __shfl_xor(v.w, o, 64));
    __syncthreads();                                  // (sh may still be read from the previous call)
    if ((rl & (RPW - 1)) == 0) sh[rl / RPW][cq] = v;
    __syncthreads();
    float4 s = sh[0][cq];
#pragma unroll
    for (int i = 1; i < RL_ / RPW; ++i) s = add4(s, sh[i][cq]);
    return s;
}

// RL row lanes per column quad: 32 (256 threads, segments up to 16 * 32 = 512 rows) or 64 (512 threads, up to 1,024 rows: the
// 32-graph segments of the single-domain scheme and of validation batches stay in this one-read regime)
// CQT column quads per tile: 8 (32 columns) everywhere but the medium regime, which takes 4 (16 columns) x 256 row lanes
template <int RPT, int RL = SRL, int CQT = SCQ>
__global__ __launch_bounds__(RL * CQT) void bn_fwd_short_kernel(BnArgs a) {
    constexpr int SRL = RL, SCQ = CQT, SCOLS = 4 * CQT;          // shadow the 32-lane / 32-column defaults
    __shared__ float4 sh[SRL][SCQ];
    __builtin_amdgcn_s_setprio(3);
    const int s = blockIdx.x, cq = threadIdx.x % SCQ, rl = threadIdx.x / SCQ;
    const int c = blockIdx.y * SCOLS + cq * 4;
    const int r0 = a.seg_ptr[s], r1 = a.seg_ptr[s + 1], n = r1 - r0;
    if (n <= 0) return;
    float4 u[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int r = r0 + rl + i * SRL;
        u[i] = r < r1 ? load_u(a, (int64_t)r * a.C + c) : zero4();
    }
    float4 mean, rstd;
    if (a.cfg.training) {
        float4 acc = zero4();
#pragma unroll
        for (int i = 0; i < RPT; ++i) acc = add4(acc, u[i]);
        mean = scl4(colsum_t<SRL, SCQ>(acc, sh, rl, cq), 1.f / n);
        acc = zero4();
#pragma unroll
        for (int i = 0; i < RPT; ++i)
            if (r0 + rl + i * SRL < r1) {
                float4 d = sub4(u[i], mean);
                acc = add4(acc, mul4(d, d));
            }
        float4 var = scl4(colsum_t<SRL, SCQ>(acc, sh, rl, cq), 1.f / n);
        rstd = make_float4(rsqrtf(var.x + a.cfg.eps), rsqrtf(var.y + a.cfg.eps), rsqrtf(var.z + a.cfg.eps), rsqrtf(var.w + a.cfg.eps));
        if (rl == 0) {
            st4(a.save_mean + (int64_t)s * a.C + c, mean);
            st4(a.save_rstd + (int64_t)s * a.C + c, rstd);
        }
    } else {
        stats_for(a, s, c, &mean, &rstd);
    }
    const float4 gam = ld4(a.gamma + pgrp(a, s) + c), bet = ld4(a.beta + pgrp(a, s) + c);
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int r = r0 + rl + i * SRL;
        if (r < r1) {
            const int64_t off = (int64_t)r * a.C + c;
            float4 gate;
            st4(a.y + off, bn_apply(a, u[i], mean, rstd, gam, bet, off >> 2, &gate));
        }
    }
}

// running statistics: segments applied in order, exactly like S successive forward() calls
__global__ __launch_bounds__(THREADS) void bn_running_kernel(BnArgs a) {
    const int c = blockIdx.x * THREADS + threadIdx.x;
    if (c >= a.C) return;
    const float m = a.cfg.momentum;
    if (!a.seg_group) {                  // one parameter set: keep the running pair in registers, loads stay independent
        float rm = a.running_mean[c], rv = a.running_var[c];
        for (int s0 = 0; s0 < a.S; s0 += 8) {           // 8 segments' statistics in flight, folded in order
            float mean[8], rstd[8];
            int n[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int s = s0 + u;
                n[u] = s < a.S ? a.seg_ptr[s + 1] - a.seg_ptr[s] : 0;
                mean[u] = n[u] > 0 ? a.save_mean[(int64_t)s * a.C + c] : 0.f;
                rstd[u] = n[u] > 0 ? a.save_rstd[(int64_t)s * a.C + c] : 1.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (n[u] <= 0) continue;
                const float var = 1.f / (rstd[u] * rstd[u]) - a.cfg.eps;
                rm = (1.f - m) * rm + m * mean[u];
                rv = (1.f - m) * rv + m * (n[u] > 1 ? var * ((float)n[u] / (float)(n[u] - 1)) : var);
            }
        }
        a.running_mean[c] = rm;
        a.running_var[c] = rv;
        return;
    }
    for (int s = 0; s < a.S; ++s) {      // one thread owns column c of every group: sequential, race-free
        const int n = a.seg_ptr[s + 1] - a.seg_ptr[s];
        if (n <= 0) continue;
        const int64_t o = pgrp(a, s) + c;
        const float mean = a.save_mean[(int64_t)s * a.C + c], rstd = a.save_rstd[(int64_t)s * a.C + c];
        const float var = 1.f / (rstd * rstd) - a.cfg.eps;   // biased batch variance
        const float unb = n > 1 ? var * ((float)n / (float)(n - 1)) : var;
        a.running_mean[o] = (1.f - m) * a.running_mean[o] + m * mean;
        a.running_var[o] = (1.f - m) * a.running_var[o] + m * unb;
    }
}

// ------------------------------------------------------------- long regime, fwd
// partial layout: part[((s*chunks + j)*2 + {0:mean,1:M2}) * C + c]
__global__ __launch_bounds__(THREADS) void bn_partial_kernel(BnArgs a) {
    __shared__ float4 sh[RL][CQ];
    const int s = blockIdx.x / a.chunks, j = blockIdx.x % a.chunks;
    const int cq = threadIdx.x % CQ, rl = threadIdx.x / CQ, c = blockIdx.y * COLS + cq * 4;
    const int seg0 = a.seg_ptr[s], seg1 = a.seg_ptr[s + 1];
    const int r0 = seg0 + j * CHUNK, r1 = min(r0 + CHUNK, seg1);
    if (r0 >= seg1) return;   // block-uniform
    const int n = r1 - r0;
    float4 acc = zero4();
    for (int r = r0 + rl; r < r1; r += RL) acc = add4(acc, load_u(a, (int64_t)r * a.C + c));
    const float4 mean = scl4(block_colsum(acc, sh, rl, cq), 1.f / n);
    acc = zero4();
    for (int r = r0 + rl; r < r1; r += RL) {
        float4 d = sub4(load_u(a, (int64_t)r * a.C + c), mean);
        acc = add4(acc, mul4(d, d));
    }
    const float4 m2 = block_colsum(acc, sh, rl, cq);
    if (rl == 0) {
        float* p = a.part + ((int64_t)(s * a.chunks + j) * 2) * a.C + c;
        st4(p, mean);
        st4(p + a.C, m2);
    }
}

__global__ __launch_bounds__(THREADS) void bn_finalize_kernel(BnArgs a) {
    const int c = blockIdx.x * THREADS + threadIdx.x, s = blockIdx.y;
    if (c >= a.C) return;
    const int seg0 = a.seg_ptr[s], seg1 = a.seg_ptr[s + 1];
    if (seg1 <= seg0) return;
    float n = 0.f, mean = 0.f, m2 = 0.f;
    for (int j = 0; j * CHUNK < seg1 - seg0; ++j) {   // Chan et al. pairwise update, chunk order
        const float nb = (float)min(CHUNK, seg1 - seg0 - j * CHUNK);
        const float* p = a.part + ((int64_t)(s * a.chunks + j) * 2) * a.C + c;
        const float mb = p[0], m2b = p[a.C];
        const float nt = n + nb, d = mb - mean;
        mean += d * (nb / nt);
        m2 += m2b + d * d * (n * nb / nt);
        n = nt;
    }
    a.save_mean[(int64_t)s * a.C + c] = mean;
    a.save_rstd[(int64_t)s * a.C + c] = rsqrtf(m2 / n + a.cfg.eps);
}

__global__ __launch_bounds__(THREADS) void bn_apply_long_kernel(BnArgs a) {
    const int s = blockIdx.x / a.chunks, j = blockIdx.x % a.chunks;
    const int cq = threadIdx.x % CQ, rl = threadIdx.x / CQ, c = blockIdx.y * COLS + cq * 4;
    const int seg1 = a.seg_ptr[s + 1];
    const int r0 = a.seg_ptr[s] + j * CHUNK, r1 = min(r0 + CHUNK, seg1);
    if (r0 >= seg1) return;
    float4 mean, rstd;
    stats_for(a, s, c, &mean, &rstd);
    const float4 gam = ld4(a.gamma + pgrp(a, s) + c), bet = ld4(a.beta + pgrp(a, s) + c);
    for (int r = r0 + rl; r < r1; r += RL) {
        const int64_t off = (int64_t)r * a.C + c;
        float4 gate;
        st4(a.y + off, bn_apply(a, load_u(a, off), mean, rstd, gam, bet, off >> 2, &gate));
    }
}

// ------------------------------------------------------------------- backward
// g_aff = g_y * gate(relu, dropout); needs xhat.  returns g_aff and xhat.
__device__ __forceinline__ void bwd_elem(const BnArgs& a, int64_t off, float4 mean, float4 rstd, float4 gam, float4 bet,
                                         float4* gaff, float4* xhat) {
    const float4 u = load_u(a, off);
    float4 gate;
    (void)bn_apply(a, u, mean, rstd, gam, bet, off >> 2, &gate);
    *xhat = mul4(sub4(u, mean), rstd);
    *gaff = mul4(ld4(a.gy + off), gate);
}

__device__ __forceinline__ float4 bwd_input(const BnArgs& a, float4 gaff, float4 xhat, float4 s1, float4 s2, float4 gam,
                                            float4 rstd, float inv_n) {
    float4 k = mul4(gam, rstd);
    if (!a.cfg.training) return mul4(k, gaff);   // running statistics are constants
    return make_float4(k.x * (gaff.x - s1.x * inv_n - xhat.x * s2.x * inv_n), k.y * (gaff.y - s1.y * inv_n - xhat.y * s2.y * inv_n),
                       k.z * (gaff.z - s1.z * inv_n - xhat.z * s2.z * inv_n), k.w * (gaff.w - s1.w * inv_n - xhat.w * s2.w * inv_n));
}

// gate (ReLU mask x dropout scale) of one float4 from its normalised value: what bn_apply derives, without the output
__device__ __forceinline__ float4 gate_of(const BnArgs& a, float4 xh, float4 gam, float4 bet, int64_t elem_quad) {
    float4 g = make_float4(1.f, 1.f, 1.f, 1.f);
    if (a.cfg.relu) {
        const float4 y = make_float4(fmaf(gam.x, xh.x, bet.x), fmaf(gam.y, xh.y, bet.y), fmaf(gam.z, xh.z, bet.z), fmaf(gam.w, xh.w, bet.w));
        g = make_float4(y.x > 0.f, y.y > 0.f, y.z > 0.f, y.w > 0.f);
    }
    if (a.cfg.dropout_p > 0.f)
        g = mul4(g, gmp::dropout_scale4(bn_seed(a), a.cfg.stream_id, (uint64_t)elem_quad, a.cfg.dropout_p, 1.f / (1.f - a.cfg.dropout_p)));
    return g;
}

// Register-resident like the forward, but only the normalised inputs stay in registers (RPT float4): the upstream gradient is
// read twice (second time out of L2) and the gate is recomputed.  Keeping both operands resident needed 256 VGPRs at
// RPT = 16 -- one wave per SIMD, nothing to hide a load behind -- and made this the slowest kernel of the backward chain.
// KEEP: the gated gradient stays in registers too (no second read of g_y, no second Philox pass for the dropout gate): affordable at six
// rows per thread (122 VGPRs, still two 512-thread workgroups per CU) -- the stacked step's segments have at most ~340 rows.
template <int RPT, int RL = SRL, int CQT = SCQ, bool KEEP = false>
__global__ __launch_bounds__(RL * CQT) void bn_bwd_short_kernel(BnArgs a) {
    constexpr int SRL = RL, SCQ = CQT, SCOLS = 4 * CQT;
    __shared__ float4 sh[SRL][SCQ];
    __builtin_amdgcn_s_setprio(3);
    const int s = blockIdx.x, cq = threadIdx.x % SCQ, rl = threadIdx.x / SCQ;
    const int c = blockIdx.y * SCOLS + cq * 4;
    const int r0 = a.seg_ptr[s], r1 = a.seg_ptr[s + 1], n = r1 - r0;
    float* ss = a.segsum + (int64_t)s * 2 * a.C + c;
    if (n <= 0) {
        if (rl == 0) { st4(ss, zero4()); st4(ss + a.C, zero4()); }
        return;
    }
    float4 mean, rstd;
    stats_for(a, s, c, &mean, &rstd);
    const float4 gam = ld4(a.gamma + pgrp(a, s) + c), bet = ld4(a.beta + pgrp(a, s) + c);
    float4 xh[RPT], gk[KEEP ? RPT : 1];
    float4 a1 = zero4(), a2 = zero4();
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int r = r0 + rl + i * SRL;
        xh[i] = zero4();
        if (KEEP) gk[i] = zero4();
        if (r < r1) {
            const int64_t off = (int64_t)r * a.C + c;
            xh[i] = mul4(sub4(load_u(a, off), mean), rstd);
            const float4 ga = mul4(ld4(a.gy + off), gate_of(a, xh[i], gam, bet, off >> 2));
            if (KEEP) gk[i] = ga;
            a1 = add4(a1, ga);
            a2 = add4(a2, mul4(ga, xh[i]));
        }
    }
    const float4 s1 = colsum_t<SRL, SCQ>(a1, sh, rl, cq);
    const float4 s2 = colsum_t<SRL, SCQ>(a2, sh, rl, cq);
    if (rl == 0) { st4(ss, s1); st4(ss + a.C, s2); }
    const float inv_n = 1.f / n;
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int r = r0 + rl + i * SRL;
        if (r < r1) {
            const int64_t off = (int64_t)r * a.C + c;
            const float4 ga = KEEP ? gk[i] : mul4(ld4(a.gy + off), gate_of(a, xh[i], gam, bet, off >> 2));
            st4(a.y + off, bwd_input(a, ga, xh[i], s1, s2, gam, rstd, inv_n));
        }
    }
}

__global__ __launch_bounds__(THREADS) void bn_bwd_partial_kernel(BnArgs a) {
    __shared__ float4 sh[RL][CQ];
    const int s = blockIdx.x / a.chunks, j = blockIdx.x % a.chunks;
    const int cq = threadIdx.x % CQ, rl = threadIdx.x / CQ, c = blockIdx.y * COLS + cq * 4;
    const int seg1 = a.seg_ptr[s + 1];
    const int r0 = a.seg_ptr[s] + j * CHUNK, r1 = min(r0 + CHUNK, seg1);
    if (r0 >= seg1) return;
    float4 mean, rstd;
    stats_for(a, s, c, &mean, &rstd);
    const float4 gam = ld4(a.gamma + pgrp(a, s) + c), bet = ld4(a.beta + pgrp(a, s) + c);
    float4 a1 = zero4(), a2 = zero4();
    for (int r = r0 + rl; r < r1; r += RL) {
        float4 g, xh;
        bwd_elem(a, (int64_t)r * a.C + c, mean, rstd, gam, bet, &g, &xh);
        a1 = add4(a1, g);
        a2 = add4(a2, mul4(g, xh));
    }
    const float4 s1 = block_colsum(a1, sh, rl, cq);
    const float4 s2 = block_colsum(a2, sh, rl, cq);
    if (rl == 0) {
        float* p = a.part + ((int64_t)(s * a.chunks + j) * 2) * a.C + c;
        st4(p, s1);
        st4(p + a.C, s2);
    }
}

__global__ __launch_bounds__(THREADS) void bn_bwd_finalize_kernel(BnArgs a) {
    const int c = blockIdx.x * THREADS + threadIdx.x, s = blockIdx.y;
    if (c >= a.C) return;
    const int len = a.seg_ptr[s + 1] - a.seg_ptr[s];
    float s1 = 0.f, s2 = 0.f;
    for (int j = 0; j * CHUNK < len; ++j) {
        const float* p = a.part + ((int64_t)(s * a.chunks + j) * 2) * a.C + c;
        s1 += p[0];
        s2 += p[a.C];
    }
    a.segsum[(int64_t)s * 2 * a.C + c] = s1;
    a.segsum[(int64_t)s * 2 * a.C + a.C + c] = s2;
}

__global__ __launch_bounds__(THREADS) void bn_bwd_apply_long_kernel(BnArgs a) {
    const int s = blockIdx.x / a.chunks, j = blockIdx.x % a.chunks;
    const int cq = threadIdx.x % CQ, rl = threadIdx.x / CQ, c = blockIdx.y * COLS + cq * 4;
    const int seg0 = a.seg_ptr[s], seg1 = a.seg_ptr[s + 1];
    const int r0 = seg0 + j * CHUNK, r1 = min(r0 + CHUNK, seg1);
    if (r0 >= seg1) return;
    float4 mean, rstd;
    stats_for(a, s, c, &mean, &rstd);
    const float4 gam = ld4(a.gamma + pgrp(a, s) + c), bet = ld4(a.beta + pgrp(a, s) + c);
    const float4 s1 = ld4(a.segsum + (int64_t)s * 2 * a.C + c), s2 = ld4(a.segsum + (int64_t)s * 2 * a.C + a.C + c);
    const float inv_n = 1.f / (seg1 - seg0);
    for (int r = r0 + rl; r < r1; r += RL) {
        const int64_t off = (int64_t)r * a.C + c;
        float4 g, xh;
        bwd_elem(a, off, mean, rstd, gam, bet, &g, &xh);
        st4(a.y + off, bwd_input(a, g, xh, s1, s2, gam, rstd, inv_n));
    }
}

// ------------------------------------------------------------------ slab regime
// Segments of SHORT_MAX < rows <= SLAB_MAX when the caller passes rendezvous words (gmp_bn_config.sync): the 2,708-node Cora graph is
// ONE segment, and a (segment, column strip) tile owned by one workgroup is bound by ONE CU's bandwidth (18.5 us forward, 34 us
// backward for 5.5 MB, r03 trace; the three chunked launches were no faster).  Here the segment's rows are cut into slabs of
// SLAB_ROWS; workgroup (slab, strip) keeps its 128 x 32 tile in registers, publishes its partial statistics, collects the partials of the
// other slabs of its (segment, strip), combines all of them in a fixed order (so every workgroup derives bit-identical statistics,
// whatever the arrival order) and applies -- one read and one write per element on the whole chip, one launch.
// Progress: nobody waits before having published, and the slabs of a group are consecutive in dispatch order (slab = fastest grid
// index), so a group's workgroups become resident together as soon as earlier groups retire; a wait that lasts SLAB_TIMEOUT (the
// words were not zeroed, or two streams share them) gives up and raises sync[0].
constexpr int SLAB_RPT = 4, SLAB_ROWS = SLAB_RPT * SRL, SLAB_MAX = SRL * SLAB_ROWS;    // 128-row slabs, at most 32 of them (one per row lane)
constexpr unsigned long long SLAB_TIMEOUT_TICKS = 200000000ull;                         // 2 s of the 100 MHz wall clock

// Hand-off = tagged granules (MI355X_MICROARCH.md, persistent-kernel price list, "handoff-1to1"): a partial travels as 8-byte {value, tag}
// pairs written by 16-byte `sc1` (write-through, device-coherent) stores; a reader polls the granules it needs with `sc1` loads until all
// carry this launch's tag.  No flag, no counter, no barrier on the path: store -> visible -> seen is one hop (≈2 us of a 9 us launch), where
// {drain stores, barrier, counter add, poll the counter, barrier, load} cost 7.7 us, and `__threadfence()` on both sides 33.
// tag = generation + 1; the generation word sync[1] advances when the LAST workgroup of a launch has left (sync[2] counts them), so
// granules of earlier launches, whatever their shapes, never carry the current tag (the buffer starts zero-filled: tag >= 1).
constexpr int SYNC_HDR = 64;                                                               // int32 words: [0] error, [1] generation, [2] departures
typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float* slab_granules(const BnArgs& a, int s, int slab, int stat, int c) {   // float pairs [S][32][2][C]
    return reinterpret_cast<float*>(a.cfg.sync + SYNC_HDR) + 2 * ((((int64_t)s * SRL + slab) * 2 + stat) * a.C + c);
}
__device__ __forceinline__ unsigned slab_tag(const BnArgs& a) {
    return (unsigned)__hip_atomic_load(a.cfg.sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
}
// lane (row lane 0, column quad cq) publishes its four columns of both statistics: 4 x 16 bytes, fire and forget
__device__ __forceinline__ void slab_publish(const BnArgs& a, int s, int j, int c, float4 v0, float4 v1, unsigned tag) {
    const float t = __uint_as_float(tag);
    float* p0 = slab_granules(a, s, j, 0, c);
    float* p1 = slab_granules(a, s, j, 1, c);
    const f4v x0 = {v0.x, t, v0.y, t}, x1 = {v0.z, t, v0.w, t}, y0 = {v1.x, t, v1.y, t}, y1 = {v1.z, t, v1.w, t};
    asm volatile("global_store_dwordx4 %0, %2, off sc1\n\tglobal_store_dwordx4 %0, %3, off offset:16 sc1\n\t"
                 "global_store_dwordx4 %1, %4, off sc1\n\tglobal_store_dwordx4 %1, %5, off offset:16 sc1"
                 ::"v"(p0), "v"(p1), "v"(x0), "v"(x1), "v"(y0), "v"(y1) : "memory");
}
// lane (row lane = slab, column quad) waits for that slab's granules; an exit every lane reaches: all tags seen, or the wall clock
__device__ __forceinline__ void slab_collect(const BnArgs& a, int s, int slab, int c, unsigned tag, float4* v0, float4* v1) {
    const float* p0 = slab_granules(a, s, slab, 0, c);
    const float* p1 = slab_granules(a, s, slab, 1, c);
    const unsigned long long t0 = wall_clock64();
    f4v x0, x1, y0, y1;
    for (unsigned i = 1;; ++i) {
        asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %4, off offset:16 sc1\n\t"
                     "global_load_dwordx4 %2, %5, off sc1\n\tglobal_load_dwordx4 %3, %5, off offset:16 sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(x0), "=&v"(x1), "=&v"(y0), "=&v"(y1) : "v"(p0), "v"(p1) : "memory");
        const bool ok = __float_as_uint(x0.y) == tag && __float_as_uint(x0.w) == tag && __float_as_uint(x1.y) == tag && __float_as_uint(x1.w) == tag &&
                        __float_as_uint(y0.y) == tag && __float_as_uint(y0.w) == tag && __float_as_uint(y1.y) == tag && __float_as_uint(y1.w) == tag;
        if (ok) break;
        __builtin_amdgcn_s_sleep(1);
        if ((i & 255u) == 0 && wall_clock64() - t0 > SLAB_TIMEOUT_TICKS) {
            atomicOr(a.cfg.sync, 1);
            break;
        }
    }
    *v0 = make_float4(x0.x, x0.z, x1.x, x1.z);
    *v1 = make_float4(y0.x, y0.z, y1.x, y1.z);
}
// every workgroup of the launch, when it has read what it needs (callers put a workgroup barrier in front): the last one advances the generation
__device__ __forceinline__ void slab_depart(const BnArgs& a, unsigned tag) {
    if (threadIdx.x == 0) {
        const int total = (int)(gridDim.x * gridDim.y);
        if (__hip_atomic_fetch_add(a.cfg.sync + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == total - 1) {
            __hip_atomic_store(a.cfg.sync + 2, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.cfg.sync + 1, (int)tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// (count, mean, centred sum of squares) of a set of rows; merge = Chan et al.  An empty side leaves the other unchanged, bit for bit.
struct Mom {
    float n;
    float4 mean, m2;
};
__device__ __forceinline__ Mom mom_merge(const Mom& p, const Mom& q) {
    const float nt = p.n + q.n;
    if (nt == 0.f) return p;
    const float fq = q.n / nt, w = p.n * fq;
    const float4 d = sub4(q.mean, p.mean);
    Mom r;
    r.n = nt;
    r.mean = make_float4(fmaf(d.x, fq, p.mean.x), fmaf(d.y, fq, p.mean.y), fmaf(d.z, fq, p.mean.z), fmaf(d.w, fq, p.mean.w));
    r.m2 = make_float4(p.m2.x + q.m2.x + d.x * d.x * w, p.m2.y + q.m2.y + d.y * d.y * w, p.m2.z + q.m2.z + d.z * d.z * w, p.m2.w + q.m2.w + d.w * d.w * w);
    return r;
}
__device__ __forceinline__ float4 shfl4(float4 v, int o) {
    return make_float4(__shfl_xor(v.x, o, 64), __shfl_xor(v.y, o, 64), __shfl_xor(v.z, o, 64), __shfl_xor(v.w, o, 64));
}

template <int RPT>
__global__ __launch_bounds__(THREADS) void bn_fwd_slab_kernel(BnArgs a) {
    __shared__ float4 sh[SRL / 8][SCQ];
    __shared__ float4 shm[SRL / 8][2][SCQ];
    __shared__ float shn[SRL / 8];
    const int s = blockIdx.x / a.slabs, j = blockIdx.x % a.slabs;
    const int cq = threadIdx.x % SCQ, rl = threadIdx.x / SCQ, c = blockIdx.y * SCOLS + cq * 4;
    const int seg0 = a.seg_ptr[s], seg1 = a.seg_ptr[s + 1], n = seg1 - seg0;
    constexpr int ROWS = RPT * SRL;
    const int nslab = (n + ROWS - 1) / ROWS;
    const unsigned tag = a.cfg.training ? slab_tag(a) : 0u;
    if (j >= nslab) {                                        // block-uniform: slabs past the segment's end have nothing to add
        if (a.cfg.training) slab_depart(a, tag);
        return;
    }
    const int r0 = seg0 + j * ROWS, r1 = min(r0 + ROWS, seg1);
    float4 u[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int r = r0 + rl + i * SRL;
        u[i] = r < r1 ? load_u(a, (int64_t)r * a.C + c) : zero4();
    }
    float4 mean, rstd;
    if (a.cfg.training) {
        float4 acc = zero4();
#pragma unroll
        for (int i = 0; i < RPT; ++i) acc = add4(acc, u[i]);
        const float4 lmean = scl4(colsum_t<SRL, SCQ>(acc, sh, rl, cq), 1.f / (r1 - r0));
        acc = zero4();
#pragma unroll
        for (int i = 0; i < RPT; ++i)
            if (r0 + rl + i * SRL < r1) {
                const float4 d = sub4(u[i], lmean);
                acc = add4(acc, mul4(d, d));
            }
        const float4 lm2 = colsum_t<SRL, SCQ>(acc, sh, rl, cq);
        if (rl == 0) slab_publish(a, s, j, c, lmean, lm2, tag);
        Mom m;                                               // row lane rl collects slab rl's partial
        m.n = rl < nslab ? (float)min(ROWS, n - rl * ROWS) : 0.f;
        m.mean = zero4();
        m.m2 = zero4();
        if (rl < nslab) slab_collect(a, s, rl, c, tag, &m.mean, &m.m2);
#pragma unroll
        for (int o = SCQ; o < 64; o <<= 1) {                 // the eight row lanes of a wave; (lower lane, higher lane) whoever computes
            Mom q;
            q.n = __shfl_xor(m.n, o, 64);
            q.mean = shfl4(m.mean, o);
            q.m2 = shfl4(m.m2, o);
            m = (threadIdx.x & o) ? mom_merge(q, m) : mom_merge(m, q);
        }
        if ((rl & 7) == 0) {
            shm[rl >> 3][0][cq] = m.mean;
            shm[rl >> 3][1][cq] = m.m2;
            if (cq == 0) shn[rl >> 3] = m.n;
        }
        __syncthreads();
        slab_depart(a, tag);                                 // (behind the barrier: every lane of this workgroup has collected)
        m.n = shn[0];
        m.mean = shm[0][0][cq];
        m.m2 = shm[0][1][cq];
#pragma unroll
        for (int w = 1; w < SRL / 8; ++w) {                  // waves in order: slabs 0-7, 8-15, ...
            Mom q;
            q.n = shn[w];
            q.mean = shm[w][0][cq];
            q.m2 = shm[w][1][cq];
            m = mom_merge(m, q);
        }
        mean = m.mean;
        const float inv_n = 1.f / n;
        rstd = make_float4(rsqrtf(m.m2.x * inv_n + a.cfg.eps), rsqrtf(m.m2.y * inv_n + a.cfg.eps), rsqrtf(m.m2.z * inv_n + a.cfg.eps),
                           rsqrtf(m.m2.w * inv_n + a.cfg.eps));
        if (j == 0 && rl == 0) {
            st4(a.save_mean + (int64_t)s * a.C + c, mean);
            st4(a.save_rstd + (int64_t)s * a.C + c, rstd);
            if (a.fold_running) {                            // one segment, one parameter set: bn_running_kernel's arithmetic
                const float mo = a.cfg.momentum, k = n > 1 ? (float)n / (float)(n - 1) : 1.f;
                const float4 rm = ld4(a.running_mean + c), rv = ld4(a.running_var + c);
                const float4 var = make_float4(1.f / (rstd.x * rstd.x) - a.cfg.eps, 1.f / (rstd.y * rstd.y) - a.cfg.eps,
                                               1.f / (rstd.z * rstd.z) - a.cfg.eps, 1.f / (rstd.w * rstd.w) - a.cfg.eps);
                st4(a.running_mean + c, make_float4((1.f - mo) * rm.x + mo * mean.x, (1.f - mo) * rm.y + mo * mean.y,
                                                    (1.f - mo) * rm.z + mo * mean.z, (1.f - mo) * rm.w + mo * mean.w));
                st4(a.running_var + c, make_float4((1.f - mo) * rv.x + mo * (var.x * k), (1.f - mo) * rv.y + mo * (var.y * k),
                                                   (1.f - mo) * rv.z + mo * (var.z * k), (1.f - mo) * rv.w + mo * (var.w * k)));
            }
        }
    } else {
        stats_for(a, s, c, &mean, &rstd);
    }
    const float4 gam = ld4(a.gamma + pgrp(a, s) + c), bet = ld4(a.beta + pgrp(a, s) + c);
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int r = r0 + rl + i * SRL;
        if (r < r1) {
            const int64_t off = (int64_t)r * a.C + c;
            float4 gate;
            st4(a.y + off, bn_apply(a, u[i], mean, rstd, gam, bet, off >> 2, &gate));
        }
    }
}

template <int RPT>
__global__ __launch_bounds__(THREADS) void bn_bwd_slab_kernel(BnArgs a) {
    __shared__ float4 sh[SRL / 8][SCQ];
    const int s = blockIdx.x / a.slabs, j = blockIdx.x % a.slabs;
    const int cq = threadIdx.x % SCQ, rl = threadIdx.x / SCQ, c = blockIdx.y * SCOLS + cq * 4;
    const int seg0 = a.seg_ptr[s], seg1 = a.seg_ptr[s + 1], n = seg1 - seg0;
    constexpr int ROWS = RPT * SRL;
    const int nslab = (n + ROWS - 1) / ROWS;
    float* ss = a.segsum + (int64_t)s * 2 * a.C + c;
    const unsigned tag = slab_tag(a);
    if (n <= 0 || j >= nslab) {
        if (n <= 0 && j == 0 && rl == 0) { st4(ss, zero4()); st4(ss + a.C, zero4()); }
        slab_depart(a, tag);
        return;
    }
    const int r0 = seg0 + j * ROWS, r1 = min(r0 + ROWS, seg1);
    float4 mean, rstd;
    stats_for(a, s, c, &mean, &rstd);
    const float4 gam = ld4(a.gamma + pgrp(a, s) + c), bet = ld4(a.beta + pgrp(a, s) + c);
    float4 xh[RPT], ga[RPT];
    float4 a1 = zero4(), a2 = zero4();
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int r = r0 + rl + i * SRL;
        xh[i] = zero4();
        ga[i] = zero4();
        if (r < r1) {
            const int64_t off = (int64_t)r * a.C + c;
            xh[i] = mul4(sub4(load_u(a, off), mean), rstd);
            ga[i] = mul4(ld4(a.gy + off), gate_of(a, xh[i], gam, bet, off >> 2));
            a1 = add4(a1, ga[i]);
            a2 = add4(a2, mul4(ga[i], xh[i]));
        }
    }
    a1 = colsum_t<SRL, SCQ>(a1, sh, rl, cq);
    a2 = colsum_t<SRL, SCQ>(a2, sh, rl, cq);
    if (rl == 0) slab_publish(a, s, j, c, a1, a2, tag);
    a1 = zero4();
    a2 = zero4();
    if (rl < nslab) slab_collect(a, s, rl, c, tag, &a1, &a2);   // row lane rl collects slab rl's partial sums
    const float4 s1 = colsum_t<SRL, SCQ>(a1, sh, rl, cq);
    const float4 s2 = colsum_t<SRL, SCQ>(a2, sh, rl, cq);
    slab_depart(a, tag);                                     // (behind colsum_t's barriers: every lane of this workgroup has collected)
    if (j == 0 && rl == 0) {
        st4(ss, s1);
        st4(ss + a.C, s2);
        if (a.pg_beta) st4(a.pg_beta + c, s1);
        if (a.pg_gamma) st4(a.pg_gamma + c, s2);
    }
    const float inv_n = 1.f / n;
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int r = r0 + rl + i * SRL;
        if (r < r1) st4(a.y + (int64_t)r * a.C + c, bwd_input(a, ga[i], xh[i], s1, s2, gam, rstd, inv_n));
    }
}

// g_gamma[grp] = sum over the group's segments of sum(g*xhat); g_beta likewise.  blockIdx.y = group;
// outputs land at base + out_off[grp] (offsets into the per-task gradient buffer).
struct BnGroups {
    int n;
    int seg[GMP_MAX_GROUPS + 1];
    int64_t off_gamma[GMP_MAX_GROUPS], off_beta[GMP_MAX_GROUPS];
};
__global__ __launch_bounds__(THREADS) void bn_param_grad_kernel(const float* __restrict__ segsum, int C, BnGroups grp,
                                                                float* g_gamma, float* g_beta) {
    const int c = blockIdx.x * THREADS + threadIdx.x, g = blockIdx.y;
    if (c >= C) return;
    float s1 = 0.f, s2 = 0.f;
    for (int s = grp.seg[g]; s < grp.seg[g + 1]; ++s) {
        s1 += segsum[(int64_t)s * 2 * C + c];
        s2 += segsum[(int64_t)s * 2 * C + C + c];
    }
    g_beta[grp.off_beta[g] + c] = s1;
    g_gamma[grp.off_gamma[g] + c] = s2;
}

size_t al(size_t x) { return (x + 255) & ~(size_t)255; }
int chunks_for(int64_t max_seg_rows) { return (int)((max_seg_rows + CHUNK - 1) / CHUNK); }
int slabs_for(int64_t max_seg_rows) { return (int)((max_seg_rows + SLAB_ROWS - 1) / SLAB_ROWS); }
int64_t slab_sync_words(int C, int S) { return SYNC_HDR + (int64_t)S * SRL * 2 * C * 2; }   // header + granules [S][32 slabs][2][C] of 8 bytes
// the slab regime runs when the caller passed rendezvous words (enough of them) and the longest segment is in its range
bool slab_regime(const gmp_bn_config* cfg, int64_t max_seg_rows, int C, int S) {
    return cfg->sync && max_seg_rows > SHORT_MAX && max_seg_rows <= SLAB_MAX && (int64_t)cfg->sync_words >= slab_sync_words(C, S);
}

int common_check(const char* who, int64_t rows, int C, int S, int64_t max_seg_rows, const gmp_bn_config* cfg) {
    if (!cfg) return gmp::fail(GMP_ERR_ARG, "%s: null config", who);
    if (rows < 0 || S < 0 || C <= 0 || C % COLS != 0)
        return gmp::fail(GMP_ERR_ARG, "%s: rows=%lld S=%d C=%d (C must be a multiple of %d)", who, (long long)rows, S, C, COLS);
    if (rows > INT32_MAX) return gmp::fail(GMP_ERR_ARG, "%s: rows beyond int32", who);
    if (max_seg_rows < 0 || max_seg_rows > rows) return gmp::fail(GMP_ERR_ARG, "%s: max_seg_rows %lld", who, (long long)max_seg_rows);
    if (cfg->dropout_p < 0.f || cfg->dropout_p >= 1.f) return gmp::fail(GMP_ERR_ARG, "%s: dropout_p %f", who, cfg->dropout_p);
    return GMP_OK;
}

}  // namespace

extern "C" size_t gmp_bn_workspace_bytes(int64_t rows, int C, int S, int64_t max_seg_rows) {
    (void)rows;
    size_t b = al((size_t)S * 2 * C * sizeof(float));
    if (max_seg_rows > SHORT_MAX) b += al((size_t)S * chunks_for(max_seg_rows) * 2 * C * sizeof(float));
    return b + 256;
}

extern "C" size_t gmp_bn_sync_bytes(int C, int S) { return (size_t)slab_sync_words(C, S) * sizeof(int32_t); }

// WIDE: 64 row lanes (512 threads) and half the rows per thread for the same (segment, 32 columns) tile.  The backward kernel
// at 16 rows per thread needs 204 VGPRs: beside the resident blocks of the weight-gradient GEMM it runs next to in the step
// (4 waves x 60 VGPRs per SIMD) only ONE such wave fits per SIMD, half the workgroups wait for a slot and the kernel takes
// 31 us instead of the 15 us it takes alone; at 8 rows per thread two waves fit.  The forward takes the wide form too since the
// column sums fold by shuffles (1.420 -> 1.405 ms per step; 1,024-thread blocks: no further gain).
#define GMP_BN_SHORT_LAUNCH(KERNEL, MAXROWS, GRID, BLK, ST, ARGS, WIDE)                \
    do {                                                                               \
        if (WIDE) {                                                                    \
            const dim3 blk2(2 * SRL * SCQ);                                            \
            if ((MAXROWS) <= 8 * SRL) hipLaunchKernelGGL((KERNEL<4, 2 * SRL>), GRID, blk2, 0, ST, ARGS);        \
            else if ((MAXROWS) <= 16 * SRL) hipLaunchKernelGGL((KERNEL<8, 2 * SRL>), GRID, blk2, 0, ST, ARGS);  \
            else hipLaunchKernelGGL((KERNEL<16, 2 * SRL>), GRID, blk2, 0, ST, ARGS);   \
        } else if ((MAXROWS) <= 4 * SRL) hipLaunchKernelGGL(KERNEL<4>, GRID, BLK, 0, ST, ARGS);      \
        else if ((MAXROWS) <= 8 * SRL) hipLaunchKernelGGL(KERNEL<8>, GRID, BLK, 0, ST, ARGS); \
        else if ((MAXROWS) <= 16 * SRL) hipLaunchKernelGGL(KERNEL<16>, GRID, BLK, 0, ST, ARGS); \
        else hipLaunchKernelGGL((KERNEL<16, 2 * SRL>), GRID, dim3(2 * SRL * SCQ), 0, ST, ARGS);  \
    } while (0)

extern "C" int gmp_bn_fwd(const float* x, const float* residual, const int32_t* seg_ptr, const int32_t* seg_group, int S,
                          int64_t max_seg_rows, int64_t rows, int C, const float* gamma, const float* beta, float* running_mean,
                          float* running_var, float* save_mean, float* save_rstd, float* y, const gmp_bn_config* cfg,
                          void* ws, size_t ws_bytes, gmp_stream_t stream) {
    if (int rc = common_check("bn_fwd", rows, C, S, max_seg_rows, cfg)) return rc;
    if (rows == 0 || S == 0) return GMP_OK;
    if (!x || !seg_ptr || !gamma || !beta || !y) return gmp::fail(GMP_ERR_ARG, "bn_fwd: null pointer");
    if (cfg->training && (!save_mean || !save_rstd)) return gmp::fail(GMP_ERR_ARG, "bn_fwd: training needs save_mean/save_rstd");
    if (!cfg->training && (!running_mean || !running_var)) return gmp::fail(GMP_ERR_ARG, "bn_fwd: eval needs running stats");
    if (ws_bytes < gmp_bn_workspace_bytes(rows, C, S, max_seg_rows) || !ws) return gmp::fail(GMP_ERR_WORKSPACE, "bn_fwd: workspace");
    hipStream_t st = (hipStream_t)stream;
    BnArgs a{};
    a.x = x; a.res = residual; a.seg_ptr = seg_ptr; a.seg_group = seg_group; a.S = S; a.C = C; a.gamma = gamma; a.beta = beta;
    a.running_mean = running_mean; a.running_var = running_var; a.save_mean = save_mean; a.save_rstd = save_rstd;
    a.y = y; a.cfg = *cfg;
    a.part = (float*)((char*)ws + al((size_t)S * 2 * C * sizeof(float)));
    a.chunks = chunks_for(max_seg_rows);
    const dim3 blk(THREADS);
    if (slab_regime(cfg, max_seg_rows, C, S)) {
        a.slabs = slabs_for(max_seg_rows);
        a.fold_running = cfg->training && running_mean && running_var && S == 1 && !seg_group;
        hipLaunchKernelGGL(bn_fwd_slab_kernel<SLAB_RPT>, dim3(S * a.slabs, C / SCOLS), blk, 0, st, a);
        if (a.fold_running) return gmp::check_launch("bn_fwd_slab_kernel");
    } else if (max_seg_rows <= SHORT_MAX) {
        GMP_BN_SHORT_LAUNCH(bn_fwd_short_kernel, max_seg_rows, dim3(S, C / SCOLS), blk, st, a, true);
    } else if (max_seg_rows <= MID_MAX) {
        GMP_BN_MID_LAUNCH(bn_fwd_short_kernel, max_seg_rows, S, C, st, a);
    } else {
        const dim3 grid(S * a.chunks, C / COLS);
        if (cfg->training) {
            hipLaunchKernelGGL(bn_partial_kernel, grid, blk, 0, st, a);
            hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + THREADS - 1) / THREADS, S), blk, 0, st, a);
        }
        hipLaunchKernelGGL(bn_apply_long_kernel, grid, blk, 0, st, a);
    }
    if (cfg->training && running_mean && running_var)
        hipLaunchKernelGGL(bn_running_kernel, dim3((C + THREADS - 1) / THREADS), blk, 0, st, a);
    return gmp::check_launch("bn_fwd kernels");
}

// The running-statistics half of a training-mode gmp_bn_fwd on its own: gmp_bn_fwd skips it when running_mean is NULL, so
// a caller can keep it off its critical path (nothing reads running statistics during training) and apply it later from
// the saved batch statistics -- same kernel, same order over segments, bit-identical result.
extern "C" int gmp_bn_running_update(const int32_t* seg_ptr, const int32_t* seg_group, int S, int C, float* running_mean,
                                     float* running_var, const float* save_mean, const float* save_rstd,
                                     const gmp_bn_config* cfg, gmp_stream_t stream) {
    if (!cfg || S < 0 || C <= 0) return gmp::fail(GMP_ERR_ARG, "bn_running_update: bad sizes");
    if (S == 0) return GMP_OK;
    if (!seg_ptr || !running_mean || !running_var || !save_mean || !save_rstd) return gmp::fail(GMP_ERR_ARG, "bn_running_update: null pointer");
    BnArgs a{};
    a.seg_ptr = seg_ptr; a.seg_group = seg_group; a.S = S; a.C = C;
    a.running_mean = running_mean; a.running_var = running_var;
    a.save_mean = (float*)save_mean; a.save_rstd = (float*)save_rstd; a.cfg = *cfg;
    hipLaunchKernelGGL(bn_running_kernel, dim3((C + THREADS - 1) / THREADS), dim3(THREADS), 0, (hipStream_t)stream, a);
    return gmp::check_launch("bn_running_kernel");
}

// Several BatchNorms' running statistics in ONE launch (the 11 of a pre-training step): blockIdx.y = entry.  Per entry the
// arithmetic is bn_running_kernel's (segments folded in order), so the results are bit-identical to separate calls.
namespace {
constexpr int RUN_MAX = 16;
struct BnRunBatch {
    int n, S;
    float eps, momentum;
    const int* seg_ptr;
    float* rm[RUN_MAX];
    float* rv[RUN_MAX];
    const float* mean[RUN_MAX];
    const float* rstd[RUN_MAX];
    const int* seg_group[RUN_MAX];
    int C[RUN_MAX];
};
__global__ __launch_bounds__(THREADS) void bn_running_batch_kernel(const BnRunBatch b) {
    const int e = blockIdx.y, c = blockIdx.x * THREADS + threadIdx.x, C = b.C[e];
    if (c >= C) return;
    const float m = b.momentum;
    float* rmp = b.rm[e];
    float* rvp = b.rv[e];
    const float* mp = b.mean[e];
    const float* rp = b.rstd[e];
    const int* sg = b.seg_group[e];
    if (!sg) {
        float rm = rmp[c], rv = rvp[c];
        for (int s0 = 0; s0 < b.S; s0 += 8) {           // 8 segments' statistics in flight, folded in order
            float mean[8], rstd[8];
            int n[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int s = s0 + u;
                n[u] = s < b.S ? b.seg_ptr[s + 1] - b.seg_ptr[s] : 0;
                mean[u] = n[u] > 0 ? mp[(int64_t)s * C + c] : 0.f;
                rstd[u] = n[u] > 0 ? rp[(int64_t)s * C + c] : 1.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (n[u] <= 0) continue;
                const float var = 1.f / (rstd[u] * rstd[u]) - b.eps;
                rm = (1.f - m) * rm + m * mean[u];
                rv = (1.f - m) * rv + m * (n[u] > 1 ? var * ((float)n[u] / (float)(n[u] - 1)) : var);
            }
        }
        rmp[c] = rm;
        rvp[c] = rv;
        return;
    }
    for (int s = 0; s < b.S; ++s) {
        const int n = b.seg_ptr[s + 1] - b.seg_ptr[s];
        if (n <= 0) continue;
        const int64_t o = (int64_t)sg[s] * C + c;
        const float mean = mp[(int64_t)s * C + c], rstd = rp[(int64_t)s * C + c];
        const float var = 1.f / (rstd * rstd) - b.eps;
        const float unb = n > 1 ? var * ((float)n / (float)(n - 1)) : var;
        rmp[o] = (1.f - m) * rmp[o] + m * mean;
        rvp[o] = (1.f - m) * rvp[o] + m * unb;
    }
}
}  // namespace

extern "C" int gmp_bn_running_update_batch(int count, const int32_t* seg_ptr, int S, const int32_t* const* seg_group, const int32_t* channels,
                                           float* const* running_mean, float* const* running_var, const float* const* save_mean,
                                           const float* const* save_rstd, const gmp_bn_config* cfg, gmp_stream_t stream) {
    if (!cfg || count < 1 || count > RUN_MAX || S < 0 || !seg_ptr || !channels || !running_mean || !running_var || !save_mean || !save_rstd)
        return gmp::fail(GMP_ERR_ARG, "bn_running_update_batch: bad argument (count=%d, max %d)", count, RUN_MAX);
    if (S == 0) return GMP_OK;
    BnRunBatch b{};
    b.n = count; b.S = S; b.eps = cfg->eps; b.momentum = cfg->momentum; b.seg_ptr = seg_ptr;
    int maxC = 0;
    for (int i = 0; i < count; ++i) {
        if (!running_mean[i] || !running_var[i] || !save_mean[i] || !save_rstd[i] || channels[i] <= 0)
            return gmp::fail(GMP_ERR_ARG, "bn_running_update_batch: entry %d", i);
        b.rm[i] = running_mean[i]; b.rv[i] = running_var[i]; b.mean[i] = save_mean[i]; b.rstd[i] = save_rstd[i];
        b.seg_group[i] = seg_group ? seg_group[i] : nullptr;
        b.C[i] = channels[i];
        if (channels[i] > maxC) maxC = channels[i];
    }
    hipLaunchKernelGGL(bn_running_batch_kernel, dim3((maxC + THREADS - 1) / THREADS, count), dim3(THREADS), 0, (hipStream_t)stream, b);
    return gmp::check_launch("bn_running_batch_kernel");
}

extern "C" int gmp_bn_bwd(const float* g_y, const float* x, const float* residual, const int32_t* seg_ptr,
                          const int32_t* seg_group, int S, int64_t max_seg_rows, int64_t rows, int C, const float* gamma,
                          const float* beta, const float* running_mean, const float* running_var, const float* save_mean,
                          const float* save_rstd, float* g_u, float* g_gamma, float* g_beta,
                          const int32_t* grp_seg_ptr_host, const int64_t* grp_off_gamma_host,
                          const int64_t* grp_off_beta_host, int G, const gmp_bn_config* cfg, void* ws, size_t ws_bytes,
                          gmp_stream_t stream) {
    if (int rc = common_check("bn_bwd", rows, C, S, max_seg_rows, cfg)) return rc;
    if (G < 0 || G > GMP_MAX_GROUPS || (G > 0 && (!grp_seg_ptr_host || !g_gamma || !g_beta)))
        return gmp::fail(GMP_ERR_ARG, "bn_bwd: group arguments (G=%d, max %d)", G, GMP_MAX_GROUPS);
    hipStream_t st = (hipStream_t)stream;
    if (rows == 0 || S == 0) {
        for (int g = 0; g < G; ++g) {
            (void)hipMemsetAsync(g_gamma + (grp_off_gamma_host ? grp_off_gamma_host[g] : (int64_t)g * C), 0, (size_t)C * sizeof(float), st);
            (void)hipMemsetAsync(g_beta + (grp_off_beta_host ? grp_off_beta_host[g] : (int64_t)g * C), 0, (size_t)C * sizeof(float), st);
        }
        return GMP_OK;
    }
    if (!g_y || !x || !seg_ptr || !gamma || !beta || !g_u) return gmp::fail(GMP_ERR_ARG, "bn_bwd: null pointer");
    if (cfg->training && (!save_mean || !save_rstd)) return gmp::fail(GMP_ERR_ARG, "bn_bwd: training needs saved stats");
    if (!cfg->training && (!running_mean || !running_var)) return gmp::fail(GMP_ERR_ARG, "bn_bwd: eval needs running stats");
    if (ws_bytes < gmp_bn_workspace_bytes(rows, C, S, max_seg_rows) || !ws) return gmp::fail(GMP_ERR_WORKSPACE, "bn_bwd: workspace");
    for (int g = 0; g < G; ++g)
        if (grp_seg_ptr_host[g] < 0 || grp_seg_ptr_host[g] > grp_seg_ptr_host[g + 1] || grp_seg_ptr_host[g + 1] > S)
            return gmp::fail(GMP_ERR_ARG, "bn_bwd: group %d covers segments [%d,%d) of %d", g, grp_seg_ptr_host[g], grp_seg_ptr_host[g + 1], S);
    BnArgs a{};
    a.x = x; a.res = residual; a.gy = g_y; a.seg_ptr = seg_ptr; a.seg_group = seg_group; a.S = S; a.C = C; a.gamma = gamma; a.beta = beta;
    a.running_mean = (float*)running_mean; a.running_var = (float*)running_var;
    a.save_mean = (float*)save_mean; a.save_rstd = (float*)save_rstd; a.y = g_u; a.cfg = *cfg;
    a.segsum = (float*)ws;
    a.part = (float*)((char*)ws + al((size_t)S * 2 * C * sizeof(float)));
    a.chunks = chunks_for(max_seg_rows);
    const dim3 blk(THREADS);
    if (slab_regime(cfg, max_seg_rows, C, S)) {
        a.slabs = slabs_for(max_seg_rows);
        if (G == 1 && S == 1 && grp_seg_ptr_host[0] == 0 && grp_seg_ptr_host[1] == 1) {   // one segment = one group: no reduction left to do
            a.pg_gamma = g_gamma + (grp_off_gamma_host ? grp_off_gamma_host[0] : 0);
            a.pg_beta = g_beta + (grp_off_beta_host ? grp_off_beta_host[0] : 0);
            G = 0;
        }
        hipLaunchKernelGGL(bn_bwd_slab_kernel<SLAB_RPT>, dim3(S * a.slabs, C / SCOLS), blk, 0, st, a);
    } else if (max_seg_rows <= 8 * SRL) {                                 // up to 256 rows: four rows per thread, the gated gradient kept in registers
        hipLaunchKernelGGL((bn_bwd_short_kernel<4, 2 * SRL, SCQ, true>), dim3(S, C / SCOLS), dim3(2 * SRL * SCQ), 0, st, a);
    } else if (max_seg_rows <= 12 * SRL) {                                // 257..384 rows (the stacked step): six rows per thread, likewise
        hipLaunchKernelGGL((bn_bwd_short_kernel<6, 2 * SRL, SCQ, true>), dim3(S, C / SCOLS), dim3(2 * SRL * SCQ), 0, st, a);
    } else if (max_seg_rows <= SHORT_MAX) {
        GMP_BN_SHORT_LAUNCH(bn_bwd_short_kernel, max_seg_rows, dim3(S, C / SCOLS), blk, st, a, true);
    } else if (max_seg_rows <= MID_MAX) {
        GMP_BN_MID_LAUNCH(bn_bwd_short_kernel, max_seg_rows, S, C, st, a);
    } else {
        const dim3 grid(S * a.chunks, C / COLS);
        hipLaunchKernelGGL(bn_bwd_partial_kernel, grid, blk, 0, st, a);
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + THREADS - 1) / THREADS, S), blk, 0, st, a);
        hipLaunchKernelGGL(bn_bwd_apply_long_kernel, grid, blk, 0, st, a);
    }
    if (G > 0) {
        BnGroups grp{};
        grp.n = G;
        for (int g = 0; g <= G; ++g) grp.seg[g] = grp_seg_ptr_host[g];
        for (int g = 0; g < G; ++g) {
            grp.off_gamma[g] = grp_off_gamma_host ? grp_off_gamma_host[g] : (int64_t)g * C;
            grp.off_beta[g] = grp_off_beta_host ? grp_off_beta_host[g] : (int64_t)g * C;
        }
        hipLaunchKernelGGL(bn_param_grad_kernel, dim3((C + THREADS - 1) / THREADS, G), blk, 0, st, (const float*)a.segsum, C, grp,
                           g_gamma, g_beta);
    }
    return gmp::check_launch("bn_bwd kernels");
}

// The parameter-gradient half of gmp_bn_bwd on its own: a gmp_bn_bwd call with num_groups = 0 leaves the per-segment sums
// (sum g, sum g*xhat) at the start of its workspace; this reduces them per group -- same kernel, same order, bit-identical --
// on any stream, so the caller can keep it off the input-gradient chain.
extern "C" int gmp_bn_param_grads(const void* bwd_workspace, int S, int C, float* g_gamma, float* g_beta,
                                  const int32_t* grp_seg_ptr_host, const int64_t* grp_off_gamma_host,
                                  const int64_t* grp_off_beta_host, int G, gmp_stream_t stream) {
    if (G < 1 || G > GMP_MAX_GROUPS || !grp_seg_ptr_host || !g_gamma || !g_beta || !bwd_workspace || C <= 0 || S < 0)
        return gmp::fail(GMP_ERR_ARG, "bn_param_grads: bad argument (G=%d, max %d)", G, GMP_MAX_GROUPS);
    BnGroups grp{};
    grp.n = G;
    for (int g = 0; g <= G; ++g) grp.seg[g] = grp_seg_ptr_host[g];
    for (int g = 0; g < G; ++g) {
        if (grp.seg[g] < 0 || grp.seg[g] > grp.seg[g + 1] || grp.seg[g + 1] > S)
            return gmp::fail(GMP_ERR_ARG, "bn_param_grads: group %d covers segments [%d,%d) of %d", g, grp.seg[g], grp.seg[g + 1], S);
        grp.off_gamma[g] = grp_off_gamma_host ? grp_off_gamma_host[g] : (int64_t)g * C;
        grp.off_beta[g] = grp_off_beta_host ? grp_off_beta_host[g] : (int64_t)g * C;
    }
    hipLaunchKernelGGL(bn_param_grad_kernel, dim3((C + THREADS - 1) / THREADS, G), dim3(THREADS), 0, (hipStream_t)stream,
                       (const float*)bwd_workspace, C, grp, g_gamma, g_beta);
    return gmp::check_launch("bn_param_grad_kernel");
}
