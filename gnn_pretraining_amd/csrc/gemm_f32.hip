// fp32 GEMM on the CDNA4 f32-input MFMA (v_mfma_f32_32x32x2_f32): exact fp32
// products and accumulation (gfx950 has no TF32/xf32), at the fp32 vector peak.
// The dense post-aggregation feature transform of GINConv and every head Linear.
//
// Block = 4 waves in a 2x2 arrangement over a BMxBN tile (128x128 or 64x64); each
// wave owns (BM/2)x(BN/2) as 32x32 MFMA tiles, accumulators stay in registers for
// the whole K loop.  Operands are staged through LDS in k-major order ([k][i]) so a
// wave's A/B fragment read (lane -> A[i = lane&31][k = lane>>5]) is 32 consecutive
// floats per half-wave: conflict-free ds_read_b32.  Global tiles are prefetched into
// registers one K-step ahead of the MFMAs.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "gnnmp_internal.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK_DEFAULT = 32;
constexpr int THREADS = 256;

// Load one BK x R operand tile into registers.
//  KMAJOR=false: memory is [R_total rows][K] (k contiguous, leading dim ld)  -> row-major activations / Linear weights
//  KMAJOR=true : memory is [K][R_total]      (row index contiguous)          -> already k-major
template <int R, bool KMAJOR, int BK>
struct TileLoader {
    static constexpr int NV = R * BK / 4 / THREADS;   // float4 per thread
    float4 v[NV];

    __device__ __forceinline__ void load(const float* __restrict__ P, int64_t ld, int64_t r0, int64_t rtot, int64_t k0,
                                         int64_t K, bool vec, int t) {
#pragma unroll
        for (int s = 0; s < NV; ++s) {
            const int f = t + s * THREADS;
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!KMAJOR) {
                const int row = f / (BK / 4), kq = f % (BK / 4);
                const int64_t gr = r0 + row, gk = k0 + 4 * kq;
                if (gr < rtot) {
                    const float* p = P + gr * ld + gk;
                    if (vec && gk + 3 < K) x = *reinterpret_cast<const float4*>(p);
                    else {
                        if (gk < K) x.x = p[0];
                        if (gk + 1 < K) x.y = p[1];
                        if (gk + 2 < K) x.z = p[2];
                        if (gk + 3 < K) x.w = p[3];
                    }
                }
            } else {
                const int k = f / (R / 4), q = f % (R / 4);
                const int64_t gk = k0 + k, gr = r0 + 4 * q;
                if (gk < K) {
                    const float* p = P + gk * ld + gr;
                    if (vec && gr + 3 < rtot) x = *reinterpret_cast<const float4*>(p);
                    else {
                        if (gr < rtot) x.x = p[0];
                        if (gr + 1 < rtot) x.y = p[1];
                        if (gr + 2 < rtot) x.z = p[2];
                        if (gr + 3 < rtot) x.w = p[3];
                    }
                }
            }
            v[s] = x;
        }
    }

    // Branch-free variant for the common case (16-B aligned operands, leading dims and extents multiples of 4, and for
    // k-contiguous operands K a multiple of BK): out-of-range rows are CLAMPED to the last valid row instead of guarded
    // (their products land in output rows/columns the epilogue never stores), k beyond the reduction range is zeroed
    // with a select.  One unconditional global_load_dwordx4 per slot -- the guarded generic loader above compiles to
    // ~50 scalar loads and ~100 branches per K-step.
    __device__ __forceinline__ void load_fast(const float* __restrict__ P, int64_t ld, int64_t r0, int64_t rtot, int64_t k0,
                                              int64_t K, int t) {
#pragma unroll
        for (int s = 0; s < NV; ++s) {
            const int f = t + s * THREADS;
            if (!KMAJOR) {
                const int row = f / (BK / 4), kq = f % (BK / 4);
                int64_t gr = r0 + row;
                gr = gr < rtot ? gr : rtot - 1;
                v[s] = *reinterpret_cast<const float4*>(P + gr * ld + k0 + 4 * kq);
            } else {
                const int k = f / (R / 4), q = f % (R / 4);
                const int64_t gk = k0 + k;
                int64_t gr = r0 + 4 * q;
                gr = gr + 3 < rtot ? gr : rtot - 4;
                const float4 x = *reinterpret_cast<const float4*>(P + (gk < K ? gk : K - 1) * ld + gr);
                const float m = gk < K ? 1.f : 0.f;
                v[s] = make_float4(x.x * m, x.y * m, x.z * m, x.w * m);
            }
        }
    }

    static constexpr int LD = KMAJOR ? R + 4 : R + 1;   // +4 keeps b128 stores aligned; +1 spreads the transposing b32 stores

    __device__ __forceinline__ void store(float* __restrict__ S, int t) const {
#pragma unroll
        for (int s = 0; s < NV; ++s) {
            const int f = t + s * THREADS;
            if (!KMAJOR) {
                const int row = f / (BK / 4), kq = f % (BK / 4);
                S[(4 * kq + 0) * LD + row] = v[s].x;
                S[(4 * kq + 1) * LD + row] = v[s].y;
                S[(4 * kq + 2) * LD + row] = v[s].z;
                S[(4 * kq + 3) * LD + row] = v[s].w;
            } else {
                const int k = f / (R / 4), q = f % (R / 4);
                *reinterpret_cast<float4*>(&S[k * LD + 4 * q]) = v[s];
            }
        }
    }
};

constexpr int MAXG = GMP_MAX_GROUPS;

struct GemmArgs {
    const float* A;
    const float* B;
    const float* bias;
    float* C;
    int64_t M, N, K, lda, ldb, ldc;
    float alpha;
    int accumulate, relu;
    int splitk;           // >1: blockIdx.z = k-slice, partial tiles go to `partial` [splitk][M][N]
    float* partial;
    int vecA, vecB;
    // grouped form (groups > 0): blockIdx.z = group.  NT/NN: group g owns rows [grow[g], grow[g+1]) of A and C and
    // its own weight matrix B + boff[g] (and bias + biasoff[g]).  TN: group g reduces over rows [grow[g], grow[g+1])
    // of A and B into its own output C + coff[g].
    int groups;
    int grow[MAXG + 1];
    int64_t boff[MAXG], biasoff[MAXG], coff[MAXG];
    // grouped TN only: asum + asumoff[g] receives sum over the group's rows of A[:, m] (the bias gradient
    // g^T 1 rides along with the weight gradient g^T x: the A tile is already in LDS)
    float* asum;
    int64_t asumoff[MAXG];
    // grouped TN split over row slices: blockIdx.z = group * gsplit + slice; partial tiles -> gpart[z][M][N] and
    // partial column sums -> gasum_part[z][M], summed in slice order by grouped_reduce_kernel (deterministic)
    int gsplit;
    float* gpart;
    float* gasum_part;
    // cross-stream signal carried by this launch (gmp::signal_on_next_gemm): the first block stores sig_value to *sig_flag as it
    // starts -- stream order has then retired, and flushed, everything enqueued before the GEMM -- which saves the separate
    // one-thread "open the gate" launch (5-6 us on a busy chip) between a producer and the GEMM that follows it anyway
    int* sig_flag;
    int sig_value;
    int vecC;             // C (+ every group's coff) 16-byte aligned and ldc % 4 == 0: the pipelined kernel stores float4 row pieces
};

thread_local int* t_sig_flag = nullptr;
thread_local int t_sig_value = 0;
inline void take_signal(GemmArgs& g) {
    g.sig_flag = t_sig_flag;
    g.sig_value = t_sig_value;
    t_sig_flag = nullptr;
}

#include "gemm_pipe.h"      // namespace g2, nested in this anonymous namespace (its kernels take GemmArgs: internal linkage throughout)

template <int BM, int BN, bool A_KMAJOR, bool B_KMAJOR, int BK, bool FAST>
__global__ __launch_bounds__(THREADS) void gemm_kernel(const GemmArgs g) {
    using LA = TileLoader<BM, A_KMAJOR, BK>;
    using LB = TileLoader<BN, B_KMAJOR, BK>;
    constexpr int TM = BM / 64, TN = BN / 64;
    // two LDS stages: the tile of K-step k+1 is written while step k is multiplied -> one barrier per step
    __shared__ __attribute__((aligned(16))) float As[2][BK * LA::LD];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK * LB::LD];
    __builtin_amdgcn_s_setprio(2);

    const int t = threadIdx.x, lane = t % 64, wv = t / 64;
    const int wm = wv / 2, wn = wv % 2, l31 = lane & 31, half = lane >> 5;
    if (g.sig_flag && t == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0)
        __hip_atomic_store(g.sig_flag, g.sig_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    int64_t m0 = (int64_t)blockIdx.y * BM;
    const int64_t n0 = (int64_t)blockIdx.x * BN;

    // NB: `g` lives in the kernarg segment; it is never written (a write would force a private copy of the
    // whole 700-byte struct into scratch and turn every field access into a scratch load)
    const float* __restrict__ Bp = g.B;
    const float* __restrict__ biasp = g.bias;
    float* Cp = g.C;
    int64_t Mrows = g.M;
    // k range of this block (split-K slices are multiples of BK)
    int64_t kbeg = 0, kend = g.K;
    if (g.groups > 0) {
        const int grp = (A_KMAJOR && B_KMAJOR) ? blockIdx.z / g.gsplit : blockIdx.z;
        if (A_KMAJOR && B_KMAJOR) {            // TN: reduction range = the group's rows (or one slice of them)
            kbeg = g.grow[grp];
            kend = g.grow[grp + 1];
            if (g.gsplit > 1) {
                const int64_t steps = (kend - kbeg + BK - 1) / BK, per = (steps + g.gsplit - 1) / g.gsplit;
                kbeg += (int64_t)(blockIdx.z % g.gsplit) * per * BK;
                kend = kbeg + per * BK < kend ? kbeg + per * BK : kend;
                Cp = g.gpart + (int64_t)blockIdx.z * g.M * g.N;
            } else {
                Cp += g.coff[grp];
            }
        } else {                                // NT / NN: row range of A and C, per-group B (and bias)
            m0 += g.grow[grp];
            Mrows = g.grow[grp + 1];
            if (m0 >= Mrows) return;
            Bp += g.boff[grp];
            if (biasp) biasp += g.biasoff[grp];
        }
    } else if (g.splitk > 1) {
        const int64_t steps = (g.K + BK - 1) / BK;
        const int64_t per = (steps + g.splitk - 1) / g.splitk;
        kbeg = (int64_t)blockIdx.z * per * BK;
        kend = kbeg + per * BK < g.K ? kbeg + per * BK : g.K;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float colacc = 0.f;
    const bool want_asum = A_KMAJOR && B_KMAJOR && g.groups > 0 && g.asum != nullptr && blockIdx.x == 0;
    LA la;
    LB lb;
    if (kbeg < kend) {
        if (FAST) { la.load_fast(g.A, g.lda, m0, Mrows, kbeg, kend, t); lb.load_fast(Bp, g.ldb, n0, g.N, kbeg, kend, t); }
        else { la.load(g.A, g.lda, m0, Mrows, kbeg, kend, g.vecA, t); lb.load(Bp, g.ldb, n0, g.N, kbeg, kend, g.vecB, t); }
        la.store(As[0], t);
        lb.store(Bs[0], t);
    }
    __syncthreads();
    int cur = 0;
    for (int64_t k0 = kbeg; k0 < kend; k0 += BK) {
        const bool more = k0 + BK < kend;
        if (more) {   // fetch the next K-step while this one is multiplied
            if (FAST) { la.load_fast(g.A, g.lda, m0, Mrows, k0 + BK, kend, t); lb.load_fast(Bp, g.ldb, n0, g.N, k0 + BK, kend, t); }
            else { la.load(g.A, g.lda, m0, Mrows, k0 + BK, kend, g.vecA, t); lb.load(Bp, g.ldb, n0, g.N, k0 + BK, kend, g.vecB, t); }
        }
        const float* __restrict__ Ac = As[cur];
        const float* __restrict__ Bc = Bs[cur];
        if (want_asum && t < BM) {
#pragma unroll 8
            for (int kk = 0; kk < BK; ++kk) colacc += Ac[kk * LA::LD + t];
        }
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = Ac[(2 * kk + half) * LA::LD + wm * (BM / 2) + i * 32 + l31];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Bc[(2 * kk + half) * LB::LD + wn * (BN / 2) + j * 32 + l31];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (more) {
            la.store(As[cur ^ 1], t);
            lb.store(Bs[cur ^ 1], t);
        }
        __syncthreads();
        cur ^= 1;
    }

    if (want_asum && t < BM && m0 + t < Mrows) {
        if (g.gsplit > 1) g.gasum_part[(int64_t)blockIdx.z * g.M + m0 + t] = colacc;
        else g.asum[g.asumoff[blockIdx.z] + m0 + t] = colacc;
    }

    // epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    float* out = Cp;
    int64_t ldo = g.ldc;
    const bool gpartial = A_KMAJOR && B_KMAJOR && g.groups > 0 && g.gsplit > 1;
    const bool partial = (g.splitk > 1 && g.groups == 0) || gpartial;
    if (g.splitk > 1 && g.groups == 0) out = g.partial + (int64_t)blockIdx.z * g.M * g.N;
    if (partial) ldo = g.N;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int64_t col = n0 + wn * (BN / 2) + j * 32 + l31;
            if (col >= g.N) continue;
            const float bv = (!partial && biasp) ? biasp[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (row >= Mrows) continue;
                float v = acc[i][j][r];
                if (!partial) {
                    v = g.alpha * v + bv;
                    if (g.accumulate) v += out[row * ldo + col];
                    if (g.relu) v = fmaxf(v, 0.f);
                }
                out[row * ldo + col] = v;
            }
        }
}

// sum the split-K slices in slice order (deterministic) and apply the epilogue
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmArgs g) {
    const int64_t total = g.M * g.N;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        float s = 0.f;
        for (int z = 0; z < g.splitk; ++z) s += g.partial[(int64_t)z * total + i];
        const int64_t row = i / g.N, col = i % g.N;
        float v = g.alpha * s + (g.bias ? g.bias[col] : 0.f);
        if (g.accumulate) v += g.C[row * g.ldc + col];
        if (g.relu) v = fmaxf(v, 0.f);
        g.C[row * g.ldc + col] = v;
    }
}

// grouped TN split: C_g = alpha * sum_slices partial, column sums likewise (slice order: deterministic)
// VEC: N, ldc and every output offset are multiples of 4 floats -- a thread owns four consecutive outputs and keeps up to eight slices'
// float4 loads in flight before it adds them IN SLICE ORDER (the scalar form issued one 4-byte load per add: 7.6 us for the 4 MB of an
// eight-slice 512 x 256 gradient, twice what the bytes take; same sums, bit for bit)
template <bool VEC>
__global__ __launch_bounds__(256) void grouped_reduce_kernel(const GemmArgs g) {
    const int grp = blockIdx.y;
    const int64_t total = g.M * g.N;
    if (VEC) {
        const int64_t quads = total / 4, mq = (g.M + 3) / 4;
        for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < quads + mq; q += (int64_t)gridDim.x * 256) {
            if (q < quads) {
                const float* base = g.gpart + (int64_t)grp * g.gsplit * total + 4 * q;
                float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int z0 = 0; z0 < g.gsplit; z0 += 8) {
                    float4 v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        v[u] = z0 + u < g.gsplit ? *reinterpret_cast<const float4*>(base + (int64_t)(z0 + u) * total) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (z0 + u < g.gsplit) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
                }
                const int64_t i = 4 * q;
                *reinterpret_cast<float4*>(g.C + g.coff[grp] + (i / g.N) * g.ldc + i % g.N) = make_float4(g.alpha * s.x, g.alpha * s.y, g.alpha * s.z, g.alpha * s.w);
            } else if (g.asum) {
                for (int64_t m = 4 * (q - quads); m < g.M && m < 4 * (q - quads) + 4; ++m) {
                    float s = 0.f;
                    for (int z = 0; z < g.gsplit; ++z) s += g.gasum_part[((int64_t)grp * g.gsplit + z) * g.M + m];
                    g.asum[g.asumoff[grp] + m] = s;
                }
            }
        }
        return;
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total + g.M; i += (int64_t)gridDim.x * 256) {
        float s = 0.f;
        if (i < total) {
            for (int z = 0; z < g.gsplit; ++z) s += g.gpart[((int64_t)grp * g.gsplit + z) * total + i];
            g.C[g.coff[grp] + (i / g.N) * g.ldc + i % g.N] = g.alpha * s;
        } else if (g.asum) {
            const int64_t m = i - total;
            for (int z = 0; z < g.gsplit; ++z) s += g.gasum_part[((int64_t)grp * g.gsplit + z) * g.M + m];
            g.asum[g.asumoff[grp] + m] = s;
        }
    }
}

// the float4 form when every group's output (and the partial buffer) is 16-byte addressable
inline void launch_grouped_reduce(const GemmArgs& g, int groups, hipStream_t st) {
    bool vec = g.N % 4 == 0 && g.ldc % 4 == 0 && (((uintptr_t)g.C | (uintptr_t)g.gpart) & 15) == 0;
    for (int i = 0; i < groups && vec; ++i) vec = g.coff[i] % 4 == 0;
    if (vec) {
        const int blocks = (int)std::min<int64_t>((g.M * g.N / 4 + (g.M + 3) / 4 + 255) / 256, 256);
        hipLaunchKernelGGL(grouped_reduce_kernel<true>, dim3(blocks, groups), dim3(256), 0, st, g);
    } else {
        const int blocks = (int)std::min<int64_t>((g.M * g.N + g.M + 255) / 256, 256);
        hipLaunchKernelGGL(grouped_reduce_kernel<false>, dim3(blocks, groups), dim3(256), 0, st, g);
    }
}

// ---- column sums (bias gradients), deterministic two-stage ---------------------
constexpr int CS_ROWS = 256;   // rows per partial block
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ A, float* __restrict__ part,
                                                             int64_t M, int64_t N, int64_t lda) {
    // block = 64 columns x 4 row lanes
    __shared__ float sh[4][64];
    const int c = threadIdx.x % 64, rl = threadIdx.x / 64;
    const int64_t col = (int64_t)blockIdx.x * 64 + c;
    const int64_t r0 = (int64_t)blockIdx.y * CS_ROWS, r1 = r0 + CS_ROWS < M ? r0 + CS_ROWS : M;
    float s = 0.f;
    if (col < N)
        for (int64_t r = r0 + rl; r < r1; r += 4) s += A[r * lda + col];
    sh[rl][c] = s;
    __syncthreads();
    if (rl == 0 && col < N) part[(int64_t)blockIdx.y * N + col] = (sh[0][c] + sh[1][c]) + (sh[2][c] + sh[3][c]);
}
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                           int64_t nparts, int64_t N, int accumulate) {
    const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (col >= N) return;
    float s = 0.f;
    for (int64_t p = 0; p < nparts; ++p) s += part[p * N + col];
    out[col] = accumulate ? out[col] + s : s;
}

inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// conditions of TileLoader::load_fast for every operand of this problem
template <int BK>
bool fast_ok(int mode, const GemmArgs& g, int64_t min_rows) {
    if (!g.vecA || !g.vecB) return false;                               // 16-B aligned bases, leading dims % 4 == 0
    if (mode == GMP_GEMM_NT) return g.K % BK == 0 && g.K > 0 && min_rows >= 1 && g.N >= 1;
    if (mode == GMP_GEMM_NN) return g.K % BK == 0 && g.K > 0 && min_rows >= 1 && g.N % 4 == 0 && g.N >= 4;
    return g.M % 4 == 0 && g.M >= 4 && g.N % 4 == 0 && g.N >= 4;       // TN: k range is select-masked, columns clamped by quads
}

template <int BM, int BN, int BK = BK_DEFAULT>
void launch_mode(int mode, const GemmArgs& g, dim3 grid, hipStream_t st, bool fast = false) {
#define GMP_GEMM_LAUNCH(AK, BKM)                                                                                    \
    do {                                                                                                           \
        if (fast) hipLaunchKernelGGL((gemm_kernel<BM, BN, AK, BKM, BK, true>), grid, dim3(THREADS), 0, st, g);      \
        else hipLaunchKernelGGL((gemm_kernel<BM, BN, AK, BKM, BK, false>), grid, dim3(THREADS), 0, st, g);          \
    } while (0)
    switch (mode) {
        case GMP_GEMM_NT: GMP_GEMM_LAUNCH(false, false); break;
        case GMP_GEMM_NN: GMP_GEMM_LAUNCH(false, true); break;
        default:          GMP_GEMM_LAUNCH(true, true); break;
    }
#undef GMP_GEMM_LAUNCH
}

// ---- the pipelined kernel (gemm_pipe.h): when it applies and with which tile ----------------------------------------
// GMP_GEMM_IMPL=old keeps every problem on gemm_kernel (A/B aid); GMP_GEMM_PIPE_TILE=0..3 forces 128x128 / 64x128 / 128x64 / 64x64.
constexpr int PIPE_TILES[4][2] = {{2, 2}, {1, 2}, {2, 1}, {1, 1}};

template <int TM, int TN, bool A_KC, bool B_KC, int STAGES, int BUFS = STAGES>
int launch_pipe_cfg(const GemmArgs& g, int tiles_m, int tiles_n, int z, hipStream_t st) {
    using C = g2::Cfg<TM, TN, A_KC, B_KC, STAGES, BUFS>;
    auto kern = g2::gemm_pipe_kernel<TM, TN, A_KC, B_KC, STAGES, BUFS>;
    // a workgroup may ask for up to 160 KiB of LDS once the function says so -- per DEVICE (the attribute belongs to the device's copy of
    // the function): one bit per device ordinal, set after the call succeeded there; racing threads at worst both make the (idempotent) call
    static std::atomic<uint64_t> attr_set{0};
    if (!gmp::lds_attr_done(attr_set)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES) != hipSuccess)
            return gmp::fail(GMP_ERR_LAUNCH, "gemm_pipe: cannot reserve %d bytes of LDS", C::LDS_BYTES);
        gmp::lds_attr_mark(attr_set);
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(tiles_m * tiles_n), 1, (unsigned)z), dim3(g2::THREADS), C::LDS_BYTES, st, g, tiles_m, tiles_n);
    return gmp::check_launch("gemm_pipe_kernel");
}

// LDS ring depth: 4 stages everywhere (128x128: 4 x 32 KB, one block per CU; 64x64: 4 x 16 KB, two blocks per CU).  A 3-stage 64x64
// ring (three blocks per CU) is 1 us faster alone and slower inside the step (1.53 against 1.45 ms): GMP_GEMM_PIPE_STAGES=3.
// Round 3, GMP_GEMM_PIPE_STAGES=32: the 3-stage schedule on TWO buffers (gemm_pipe.h "early free": 32 KB, four blocks per CU, all 928
// tiles of a layer GEMM resident at once): 23.2 against 24.6 us alone (83.3 TF/s = 0.53 of the fp32 MFMA peak, NN 82.2), and again slower
// inside the step (1.416 against 1.401 ms, two interleaved pairs): four resident GEMM blocks per CU crowd out the other streams' kernels,
// which is what the step's concurrency lives on.  Kept selectable for GEMM-only callers; the step's default stays 4 / 4.
template <bool A_KC, bool B_KC>
int launch_pipe_tile(int tile, const GemmArgs& g, int64_t rows, int z, hipStream_t st) {
    const int TMs = PIPE_TILES[tile][0], TNs = PIPE_TILES[tile][1];
    const int tiles_m = (int)((rows + 64 * TMs - 1) / (64 * TMs)), tiles_n = (int)((g.N + 64 * TNs - 1) / (64 * TNs));
    const char* e = getenv("GMP_GEMM_PIPE_STAGES");
    const int deep = e ? atoi(e) : 0;
    switch (tile) {
        case 0: return launch_pipe_cfg<2, 2, A_KC, B_KC, 4>(g, tiles_m, tiles_n, z, st);
        case 1: return deep == 3 ? launch_pipe_cfg<1, 2, A_KC, B_KC, 3>(g, tiles_m, tiles_n, z, st)
                                 : launch_pipe_cfg<1, 2, A_KC, B_KC, 4>(g, tiles_m, tiles_n, z, st);
        case 2: return deep == 3 ? launch_pipe_cfg<2, 1, A_KC, B_KC, 3>(g, tiles_m, tiles_n, z, st)
                                 : launch_pipe_cfg<2, 1, A_KC, B_KC, 4>(g, tiles_m, tiles_n, z, st);
        default: return deep == 3 ? launch_pipe_cfg<1, 1, A_KC, B_KC, 3>(g, tiles_m, tiles_n, z, st)
                      : deep == 32 ? launch_pipe_cfg<1, 1, A_KC, B_KC, 3, 2>(g, tiles_m, tiles_n, z, st)       // 32 KB: four blocks per CU
                                   : launch_pipe_cfg<1, 1, A_KC, B_KC, 4>(g, tiles_m, tiles_n, z, st);
    }
}

int launch_pipe(int mode, int tile, const GemmArgs& g_in, int64_t rows, int z, hipStream_t st) {
    const GemmArgs& g = g_in;
    if (mode == GMP_GEMM_NT) return launch_pipe_tile<true, true>(tile, g, rows, z, st);
    if (mode == GMP_GEMM_NN) return launch_pipe_tile<true, false>(tile, g, rows, z, st);
    return launch_pipe_tile<false, false>(tile, g, rows, z, st);
}

// (read per call, not cached: scripts/bench_gemm_pipe.py flips them inside one process for interleaved A/B rounds)
inline bool pipe_enabled() {
    const char* e = getenv("GMP_GEMM_IMPL");
    return !(e && !strcmp(e, "old"));
}
// smallest row count (NT / NN forms) that takes the pipelined kernel.  (Round 3: sending the task heads' small grouped GEMMs -- 8 to 340 rows
// per group -- through it as well changed nothing in the step, 1.413-1.415 against 1.400 ms: their chains are bound by launch boundaries.)
inline int64_t pipe_min_rows() { return 1024; }
inline int pipe_forced_tile() {
    const char* e = getenv("GMP_GEMM_PIPE_TILE");
    return e ? atoi(e) : -1;
}

// Tile choice.  Measured on MI355X (scripts/bench_gemm_pipe.py, profiles/README.md round 2): at the step's shapes (M ~ 7 k rows,
// 256 <-> 512, K = 256 .. 768) the 64x64 tile wins or ties everywhere -- alone (24-25 us against 25.7 for 128x128 at N = 512,
// 23 against 24-40 at N = 256, 141 against 151-167 us at the 37 k-row link-prediction shape) and inside the step, where its
// 64 KB of LDS let two blocks share a CU beside the other streams' kernels (1.42 against 1.46 ms per step).  The main loop of every
// tile is MFMA-bound at the clock the chip holds (2.0 us per 32-deep K-step of 128x128 = 4,096 MFMA cycles at ~2.05 GHz); what
// separates them is the fixed prologue / epilogue, which smaller, staggered blocks hide better.  The larger tiles stay available
// through GMP_GEMM_PIPE_TILE (0 = 128x128, 1 = 64x128, 2 = 128x64) for shapes this was not measured on.
int pipe_pick_tile(int64_t rows, int64_t N, int64_t z) {
    (void)rows; (void)N; (void)z;
    if (pipe_forced_tile() >= 0 && pipe_forced_tile() < 4) return pipe_forced_tile();
    return 3;
}

// operand conditions of the LDS-DMA loaders: 16-byte aligned bases and leading dimensions; k-contiguous operands need whole
// 32-deep K-steps; k-major operands are fetched in quads of 4 rows / columns (clamped by quads)
bool pipe_ok(int mode, const GemmArgs& g) {
    if (!pipe_enabled() || !g.vecA || !g.vecB) return false;
    if (mode == GMP_GEMM_NT) return g.K >= 64 && g.K % g2::BK == 0;
    if (mode == GMP_GEMM_NN) return g.K >= 64 && g.K % g2::BK == 0 && g.N % 4 == 0 && g.N >= 4;
    return g.M % 4 == 0 && g.M >= 4 && g.N % 4 == 0 && g.N >= 4;
}

}  // namespace

namespace gmp {
void signal_on_next_gemm(int32_t* flag, int value) {
    t_sig_flag = (int*)flag;
    t_sig_value = value;
}
bool signal_pending() { return t_sig_flag != nullptr; }
}  // namespace gmp

// C-ABI face of the above (gnnmp.h): the NEXT gmp_gemm_f32 / gmp_gemm_f32_grouped call of this thread opens the gate as its first workgroup starts
extern "C" int gmp_gate_open_by_next_gemm(int32_t* flag, int value) {
    gmp::signal_on_next_gemm(flag, value);
    return GMP_OK;
}
extern "C" int gmp_gate_open_pending(void) { return gmp::signal_pending() ? 1 : 0; }

extern "C" size_t gmp_gemm_f32_workspace_bytes(int mode, int64_t M, int64_t N, int64_t K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    // split-K is used only when the output alone cannot fill the chip and K is long
    const int64_t tiles = ((M + 63) / 64) * ((N + 63) / 64);
    if (tiles >= 256 || K < 8 * BK_DEFAULT) return 0;
    const int64_t steps = (K + BK_DEFAULT - 1) / BK_DEFAULT;
    // NT / NN (round 3, scripts/bench_gemm_slices.py): slices + their reduction launch LOSE wherever a third of the chip has a tile --
    // 172 tiles: 38.4 against 33.6 us at K = 1,440, 20.4 against 14.2 at K = 512; 144 tiles, K = 256: 18.2 against 9.1 -- and win only for
    // a handful of tiles under a long K (72 tiles, K = 1,440: 23.8 against 31.7; 40 tiles, K = 512: 12.7 against 16.4).  The weight-gradient
    // form (TN: K = thousands of rows, a few dozen output tiles) keeps the old rule.
    if (mode != GMP_GEMM_TN && !((tiles <= 96 && steps >= 32) || (tiles <= 48 && steps >= 16))) return 0;
    int64_t sk = (512 + tiles - 1) / tiles;
    if (sk > steps / 2) sk = steps / 2;
    if (sk < 2) return 0;
    return (size_t)sk * M * N * sizeof(float);
}

extern "C" int gmp_gemm_f32(int mode, const float* A, const float* B, const float* bias, float* C, int64_t M, int64_t N,
                            int64_t K, int64_t lda, int64_t ldb, int64_t ldc, float alpha, int accumulate, int relu,
                            void* workspace, size_t workspace_bytes, gmp_stream_t stream) {
    if (mode < 0 || mode > 2) return gmp::fail(GMP_ERR_ARG, "gemm: mode %d", mode);
    if (M < 0 || N < 0 || K < 0) return gmp::fail(GMP_ERR_ARG, "gemm: negative size");
    if (M == 0 || N == 0) return GMP_OK;
    if (!C || (K > 0 && (!A || !B))) return gmp::fail(GMP_ERR_ARG, "gemm: null pointer");
    const int64_t need_lda = mode == GMP_GEMM_TN ? M : K, need_ldb = mode == GMP_GEMM_NT ? K : N;
    if (lda < need_lda || ldb < need_ldb || ldc < N)
        return gmp::fail(GMP_ERR_ARG, "gemm: leading dims (%lld,%lld,%lld) too small for M=%lld N=%lld K=%lld mode %d",
                         (long long)lda, (long long)ldb, (long long)ldc, (long long)M, (long long)N, (long long)K, mode);
    hipStream_t st = (hipStream_t)stream;
    GemmArgs g{A, B, bias, C, M, N, K, lda, ldb, ldc, alpha, accumulate, relu, 1, nullptr,
               (lda % 4 == 0) && aligned16(A), (ldb % 4 == 0) && aligned16(B)};
    take_signal(g);
    g.vecC = (ldc % 4 == 0) && aligned16(C);
    const size_t want = gmp_gemm_f32_workspace_bytes(mode, M, N, K);
    if (want && workspace && workspace_bytes >= want) {
        g.splitk = (int)(want / ((size_t)M * N * sizeof(float)));
        g.partial = (float*)workspace;
    }
    // tile choice: the largest tile that still gives every CU work (bigger tiles hide the global-load latency of a
    // K-step behind 2-4x more MFMA work); GMP_GEMM_TILE=0/1/2 forces 64x64 / 128x64 / 128x128 (tuning aid)
    static const int forced = getenv("GMP_GEMM_TILE") ? atoi(getenv("GMP_GEMM_TILE")) : -1;
    const int64_t t128 = ((M + 127) / 128) * ((N + 127) / 128), t12864 = ((M + 127) / 128) * ((N + 63) / 64);
    int tile = 0;
    (void)t12864;
    if (g.splitk == 1) tile = t128 >= 4096 ? 2 : 0;     // measured on MI355X: 64x64 wins until the grid is many waves deep
    if (forced >= 0 && forced <= 2 && g.splitk == 1) tile = forced;
    // large row counts (the backbone's layer GEMMs, the link-prediction scorer): the LDS-DMA pipelined kernel
    if (mode != GMP_GEMM_TN && M >= pipe_min_rows() && N >= 64 && pipe_ok(mode, g) && (g.splitk == 1 || N % 4 == 0)) {   // (slices are stored as float4 rows of N)
        if (g.splitk == 1) return launch_pipe(mode, pipe_pick_tile(M, N, 1), g, M, 1, st);
        // few output tiles and a long K (the caller handed over a workspace): K-slices of the pipelined kernel, summed in slice order
        if (int rc = launch_pipe(mode, pipe_pick_tile(M, N, 1), g, M, g.splitk, st)) return rc;
        const int blocks = (int)std::min<int64_t>((M * N + 255) / 256, 2048);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, g);
        return gmp::check_launch("splitk_reduce_kernel");
    }
    static const bool nofast = getenv("GMP_GEMM_NOFAST") != nullptr;
    // with split-K (TN) slices start at multiples of BK inside [0,K): the fast loader's k handling covers that
    const bool fast = !nofast && fast_ok<BK_DEFAULT>(mode, g, M);
    if (tile == 2) launch_mode<128, 128>(mode, g, dim3((unsigned)((N + 127) / 128), (unsigned)((M + 127) / 128), 1), st, fast);
    else if (tile == 1) launch_mode<128, 64>(mode, g, dim3((unsigned)((N + 63) / 64), (unsigned)((M + 127) / 128), 1), st, fast);
    else if (forced == 3) launch_mode<64, 64, 64>(mode, g, dim3((unsigned)((N + 63) / 64), (unsigned)((M + 63) / 64), (unsigned)g.splitk), st, !nofast && fast_ok<64>(mode, g, M));
    else if (forced == 4) launch_mode<64, 64, 16>(mode, g, dim3((unsigned)((N + 63) / 64), (unsigned)((M + 63) / 64), (unsigned)g.splitk), st, fast);
    else launch_mode<64, 64>(mode, g, dim3((unsigned)((N + 63) / 64), (unsigned)((M + 63) / 64), (unsigned)g.splitk), st, fast);
    if (int rc = gmp::check_launch("gemm_kernel")) return rc;
    if (g.splitk > 1) {
        int blocks = (int)std::min<int64_t>((M * N + 255) / 256, 2048);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, g);
        return gmp::check_launch("splitk_reduce_kernel");
    }
    return GMP_OK;
}

extern "C" int gmp_gemm_f32_grouped(int mode, const float* A, const float* B, const float* bias, float* C, int groups,
                                    const int32_t* group_rows_host, const int64_t* b_off_host,
                                    const int64_t* bias_off_host, const int64_t* c_off_host, float* a_colsum,
                                    const int64_t* a_colsum_off_host, int64_t M_tn, int64_t N, int64_t K, int64_t lda,
                                    int64_t ldb, int64_t ldc, float alpha, int accumulate, int relu, void* workspace,
                                    size_t workspace_bytes, gmp_stream_t stream) {
    if (mode < 0 || mode > 2) return gmp::fail(GMP_ERR_ARG, "gemm_grouped: mode %d", mode);
    if (groups < 1 || groups > MAXG || !group_rows_host) return gmp::fail(GMP_ERR_ARG, "gemm_grouped: %d groups (max %d)", groups, MAXG);
    if (N <= 0 || K < 0 || !A || !B || !C) return gmp::fail(GMP_ERR_ARG, "gemm_grouped: bad argument");
    GemmArgs g{};
    g.A = A; g.B = B; g.bias = bias; g.C = C; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.alpha = alpha; g.accumulate = accumulate; g.relu = relu; g.splitk = 1; g.partial = nullptr;
    g.vecA = (lda % 4 == 0) && aligned16(A);
    g.vecB = (ldb % 4 == 0) && aligned16(B);
    g.groups = groups;
    g.gsplit = 1;
    g.vecC = (ldc % 4 == 0) && aligned16(C);
    g.asum = mode == GMP_GEMM_TN ? a_colsum : nullptr;
    int64_t max_rows = 0;
    for (int i = 0; i <= groups; ++i) {
        g.grow[i] = group_rows_host[i];
        if (i && (g.grow[i] < g.grow[i - 1] || g.grow[i - 1] < 0)) return gmp::fail(GMP_ERR_ARG, "gemm_grouped: group rows not ascending");
        if (i) max_rows = std::max<int64_t>(max_rows, g.grow[i] - g.grow[i - 1]);
    }
    for (int i = 0; i < groups; ++i) {
        g.boff[i] = b_off_host ? b_off_host[i] : 0;
        g.biasoff[i] = bias_off_host ? bias_off_host[i] : 0;
        g.coff[i] = c_off_host ? c_off_host[i] : 0;
        g.asumoff[i] = a_colsum_off_host ? a_colsum_off_host[i] : (int64_t)i * M_tn;
        if ((g.boff[i] % 4) || (g.coff[i] % 4)) g.vecB = 0;
        if (g.coff[i] % 4) g.vecC = 0;
    }
    hipStream_t st = (hipStream_t)stream;
    if (mode == GMP_GEMM_TN) {
        g.M = M_tn;          // output rows = columns of A (k-major A: lda >= M_tn); the reduction runs over the group's rows
        g.K = 0;
        if (g.M == 0) return GMP_OK;          // (nothing launched: a pending signal stays pending, the caller opens the gate itself)
        take_signal(g);
        // long reductions (the per-task weight gradients of the stacked backward): the pipelined kernel; tile and the number of
        // row slices per group are chosen together for the shortest grid (rounds of 256 CUs x K-steps per block x MFMAs per step)
        // (only for callers that hand over a workspace, i.e. opted into row slices: the grouped NT-Xent does not, and stays
        // bit-identical to its single-problem form on gemm_kernel)
        if (workspace && max_rows >= 256 && pipe_ok(mode, g)) {
            int best_tile = 0, best_split = 1;
            int64_t best_cost = -1;
            for (int c = 0; c < 4; ++c) {
                if (c != pipe_pick_tile(g.M, N, groups)) continue;
                const int64_t bm = 64 * PIPE_TILES[c][0], bn = 64 * PIPE_TILES[c][1];
                const int64_t tiles = ((g.M + bm - 1) / bm) * ((N + bn - 1) / bn) * groups;
                for (int sp = 1; sp <= 32; ++sp) {
                    if (sp > 1 && (!workspace || (size_t)groups * sp * (g.M * N + g.M) * sizeof(float) > workspace_bytes)) break;
                    const int64_t steps = (max_rows + g2::BK - 1) / g2::BK, per = (steps + sp - 1) / sp;
                    if (sp > 1 && per < 2) break;
                    // + 2 K-steps per round for the prologue / epilogue of a block, + the reduce pass over the slices
                    const int64_t cost = ((tiles * sp + 255) / 256) * (per + 2) * PIPE_TILES[c][0] * PIPE_TILES[c][1] + (sp > 1 ? sp / 2 + 2 : 0);
                    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best_tile = c; best_split = sp; }
                }
            }
            // (round 3, in the step: capping the slices at 1 / 2 gives 1.595 / 1.565 ms against 1.412, forcing 4 / 6 / 8 / 12 gives 1.437 / 1.457 /
            // 1.453 / 1.457: the cost model's choice stands)
            g.gsplit = best_split;
            if (best_split > 1) {
                g.gpart = (float*)workspace;
                g.gasum_part = g.gpart + (size_t)groups * best_split * g.M * N;
            }
            if (int rc = launch_pipe(mode, best_tile, g, g.M, groups * best_split, st)) return rc;
            if (best_split > 1) {
                launch_grouped_reduce(g, groups, st);
            }
            return gmp::check_launch("gemm_pipe_kernel (grouped)");
        }
        // few output tiles x long reductions: slice every group's rows over several blocks when a workspace is given
        const int64_t tiles = ((N + 63) / 64) * ((g.M + 63) / 64) * groups;
        int split = 1;
        if (workspace && tiles < 384 && max_rows >= 8 * BK_DEFAULT) {
            const int64_t target = 512;
            split = (int)std::min<int64_t>((target + tiles - 1) / tiles, max_rows / (4 * BK_DEFAULT));
            const size_t need = (size_t)groups * split * (g.M * N + g.M) * sizeof(float);
            if (split < 2 || need > workspace_bytes) split = 1;
        }
        g.gsplit = split;
        if (split > 1) {
            g.gpart = (float*)workspace;
            g.gasum_part = g.gpart + (size_t)groups * split * g.M * N;
        }
        static const bool nofast_g = getenv("GMP_GEMM_NOFAST") != nullptr;
        launch_mode<64, 64>(mode, g, dim3((unsigned)((N + 63) / 64), (unsigned)((g.M + 63) / 64), (unsigned)(groups * split)), st,
                            !nofast_g && fast_ok<BK_DEFAULT>(mode, g, 1));
        if (split > 1) {
            if (int rc = gmp::check_launch("gemm_kernel (grouped, split)")) return rc;
            launch_grouped_reduce(g, groups, st);
        }
    } else {
        if (max_rows == 0) return GMP_OK;
        take_signal(g);
        g.M = 0;
        static const bool nofast_g2 = getenv("GMP_GEMM_NOFAST") != nullptr;
        g.K = K;
        if (max_rows >= pipe_min_rows() && N >= 64 && pipe_ok(mode, g)) return launch_pipe(mode, pipe_pick_tile(max_rows, N, groups), g, max_rows, groups, st);
        launch_mode<64, 64>(mode, g, dim3((unsigned)((N + 63) / 64), (unsigned)((max_rows + 63) / 64), (unsigned)groups), st,
                            !nofast_g2 && fast_ok<BK_DEFAULT>(mode, g, 1));
    }
    return gmp::check_launch("gemm_kernel (grouped)");
}

extern "C" size_t gmp_colsum_workspace_bytes(int64_t M, int64_t N) {
    if (M <= 0 || N <= 0) return 0;
    return (size_t)((M + CS_ROWS - 1) / CS_ROWS) * N * sizeof(float);
}

extern "C" int gmp_colsum(const float* A, float* out, int64_t M, int64_t N, int64_t lda, int accumulate, void* ws,
                          size_t ws_bytes, gmp_stream_t stream) {
    if (M < 0 || N < 0 || lda < N) return gmp::fail(GMP_ERR_ARG, "colsum: bad size");
    if (N == 0) return GMP_OK;
    if (!out || (M > 0 && !A)) return gmp::fail(GMP_ERR_ARG, "colsum: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (M == 0) {
        if (!accumulate) (void)hipMemsetAsync(out, 0, N * sizeof(float), st);
        return GMP_OK;
    }
    if (ws_bytes < gmp_colsum_workspace_bytes(M, N) || !ws) return gmp::fail(GMP_ERR_WORKSPACE, "colsum: workspace");
    const int64_t parts = (M + CS_ROWS - 1) / CS_ROWS;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((unsigned)((N + 63) / 64), (unsigned)parts), dim3(256), 0, st, A,
                       (float*)ws, M, N, lda);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, (const float*)ws, out,
                       parts, N, accumulate);
    return gmp::check_launch("colsum kernels");
}
