// NT-Xent / InfoNCE (SimCLR) loss: normalise -> similarity GEMM (f32 MFMA) ->
// masked row softmax cross-entropy, and its backward.  The 2n x 2n similarity
// matrix lives only in the caller's workspace (it is overwritten in place by
// softmax - onehot, which is all the backward needs).
#include "gnnmp_internal.h"

namespace {

constexpr int THREADS = 256;

struct Ws {
    float* zn;     // [2n,d] normalised rows
    float* norm;   // [2n]   max(||z||, 1e-12)
    float* G;      // [2n,2n] sim, then softmax - onehot
    float* gzn;    // [2n,d]
    float* rowloss;// [2n]
};

size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

Ws carve(void* ws, int64_t n, int d) {
    char* p = (char*)ws;
    Ws w;
    const int64_t R = 2 * n;
    w.zn = (float*)p; p += al((size_t)R * d * 4);
    w.norm = (float*)p; p += al((size_t)R * 4);
    w.G = (float*)p; p += al((size_t)R * R * 4);
    w.gzn = (float*)p; p += al((size_t)R * d * 4);
    w.rowloss = (float*)p;
    return w;
}

// one wave per row: zn = z / max(||z||, eps)   (F.normalize, eps = 1e-12)
__global__ __launch_bounds__(THREADS) void normalize_kernel(const float* __restrict__ z1, const float* __restrict__ z2,
                                                            int64_t n, int d, float* __restrict__ zn, float* __restrict__ norm) {
    const int lane = threadIdx.x % 64;
    const int64_t row = ((int64_t)blockIdx.x * THREADS + threadIdx.x) / 64;
    if (row >= 2 * n) return;
    const float* src = row < n ? z1 + row * d : z2 + (row - n) * d;
    float s = 0.f;
    for (int c = lane; c < d; c += 64) s += src[c] * src[c];
    s = gmp::wave_sum(s);
    const float nr = fmaxf(sqrtf(s), 1e-12f);
    for (int c = lane; c < d; c += 64) zn[row * d + c] = src[c] / nr;
    if (lane == 0) norm[row] = nr;
}

__device__ __forceinline__ float block_reduce(float v, float* sh, bool is_max) {
    const int lane = threadIdx.x % 64, wv = threadIdx.x / 64;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float t = __shfl_xor(v, o, 64);
        v = is_max ? fmaxf(v, t) : v + t;
    }
    __syncthreads();
    if (lane == 0) sh[wv] = v;
    __syncthreads();
    float r = sh[0];
    for (int i = 1; i < THREADS / 64; ++i) r = is_max ? fmaxf(r, sh[i]) : r + sh[i];
    return r;
}

// one block per row i: loss_i = logsumexp_{j != i} sim[i,j] - sim[i,pos];  G[i,:] = softmax - onehot(pos)
__global__ __launch_bounds__(THREADS) void row_loss_kernel(float* __restrict__ G, int64_t n, float* __restrict__ rowloss) {
    __shared__ float sh[THREADS / 64];
    const int64_t R = 2 * n, i = blockIdx.x;
    float* row = G + i * R;
    const int64_t pos = i < n ? i + n : i - n;
    float m = -INFINITY;
    for (int64_t j = threadIdx.x; j < R; j += THREADS)
        if (j != i) m = fmaxf(m, row[j]);
    m = block_reduce(m, sh, true);
    float s = 0.f;
    for (int64_t j = threadIdx.x; j < R; j += THREADS)
        if (j != i) s += expf(row[j] - m);
    s = block_reduce(s, sh, false);
    const float lse = m + logf(s);
    const float sp = row[pos];
    __syncthreads();
    for (int64_t j = threadIdx.x; j < R; j += THREADS) {
        float p = j == i ? 0.f : expf(row[j] - lse);
        if (j == pos) p -= 1.f;
        row[j] = p;
    }
    if (threadIdx.x == 0) rowloss[i] = lse - sp;
}

__global__ __launch_bounds__(THREADS) void sum_rows_kernel(const float* __restrict__ v, int64_t n, float* out) {
    __shared__ float sh[THREADS];
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += THREADS) s += v[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int d = THREADS / 2; d > 0; d >>= 1) {
        if (threadIdx.x < d) sh[threadIdx.x] += sh[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sh[0];
}

// g_z = g_scale * (g_zn - zn <zn, g_zn>) / norm     (backward of x / max(||x||, eps) for ||x|| > eps;
// for ||x|| <= eps the denominator is the constant eps and the projection term vanishes)
__global__ __launch_bounds__(THREADS) void normalize_bwd_kernel(const float* __restrict__ zn, const float* __restrict__ norm,
                                                                const float* __restrict__ gzn, const float* __restrict__ g_scale,
                                                                int64_t n, int d, float* __restrict__ g1, float* __restrict__ g2) {
    const int lane = threadIdx.x % 64;
    const int64_t row = ((int64_t)blockIdx.x * THREADS + threadIdx.x) / 64;
    if (row >= 2 * n) return;
    float dot = 0.f;
    for (int c = lane; c < d; c += 64) dot += zn[row * d + c] * gzn[row * d + c];
    dot = gmp::wave_sum(dot);
    const float nr = norm[row], gs = g_scale[0];
    if (nr <= 1e-12f) dot = 0.f;
    float* dst = row < n ? g1 + row * d : g2 + (row - n) * d;
    for (int c = lane; c < d; c += 64) dst[c] = gs * (gzn[row * d + c] - zn[row * d + c] * dot) / nr;
}

int args_ok(const char* who, int64_t n, int d, float T) {
    if (n < 1 || n > 8192) return gmp::fail(GMP_ERR_ARG, "%s: n=%lld must be in [1,8192]", who, (long long)n);
    if (d < 1 || d % 4) return gmp::fail(GMP_ERR_ARG, "%s: dim %d must be a positive multiple of 4", who, d);
    if (!(T > 0.f)) return gmp::fail(GMP_ERR_ARG, "%s: temperature %f", who, T);
    return GMP_OK;
}

}  // namespace

extern "C" size_t gmp_nt_xent_workspace_bytes(int64_t n, int d) {
    if (n < 1 || d < 1) return 0;
    const size_t R = 2 * (size_t)n;
    return 2 * al(R * d * 4) + 2 * al(R * 4) + al(R * R * 4) + 256;
}

extern "C" int gmp_nt_xent_fwd(const float* z1, const float* z2, int64_t n, int d, float T, float* loss_sum, void* ws,
                               size_t ws_bytes, gmp_stream_t stream) {
    if (int rc = args_ok("nt_xent_fwd", n, d, T)) return rc;
    if (!z1 || !z2 || !loss_sum || !ws) return gmp::fail(GMP_ERR_ARG, "nt_xent_fwd: null pointer");
    if (ws_bytes < gmp_nt_xent_workspace_bytes(n, d)) return gmp::fail(GMP_ERR_WORKSPACE, "nt_xent_fwd: workspace");
    hipStream_t st = (hipStream_t)stream;
    Ws w = carve(ws, n, d);
    const int64_t R = 2 * n;
    hipLaunchKernelGGL(normalize_kernel, dim3((unsigned)((R * 64 + THREADS - 1) / THREADS)), dim3(THREADS), 0, st, z1, z2, n, d, w.zn, w.norm);
    if (int rc = gmp_gemm_f32(GMP_GEMM_NT, w.zn, w.zn, nullptr, w.G, R, R, d, d, d, R, 1.f / T, 0, 0, nullptr, 0, stream)) return rc;
    hipLaunchKernelGGL(row_loss_kernel, dim3((unsigned)R), dim3(THREADS), 0, st, w.G, n, w.rowloss);
    hipLaunchKernelGGL(sum_rows_kernel, dim3(1), dim3(THREADS), 0, st, (const float*)w.rowloss, R, loss_sum);
    return gmp::check_launch("nt_xent_fwd kernels");
}

extern "C" int gmp_nt_xent_bwd(const float* z1, const float* z2, int64_t n, int d, float T, const float* g_scale,
                               float* g_z1, float* g_z2, void* ws, size_t ws_bytes, gmp_stream_t stream) {
    (void)z1; (void)z2;
    if (int rc = args_ok("nt_xent_bwd", n, d, T)) return rc;
    if (!g_scale || !g_z1 || !g_z2 || !ws) return gmp::fail(GMP_ERR_ARG, "nt_xent_bwd: null pointer");
    if (ws_bytes < gmp_nt_xent_workspace_bytes(n, d)) return gmp::fail(GMP_ERR_WORKSPACE, "nt_xent_bwd: workspace");
    hipStream_t st = (hipStream_t)stream;
    Ws w = carve(ws, n, d);
    const int64_t R = 2 * n;
    // d loss / d zn = (G + G^T) zn / T
    if (int rc = gmp_gemm_f32(GMP_GEMM_NN, w.G, w.zn, nullptr, w.gzn, R, d, R, R, d, d, 1.f / T, 0, 0, nullptr, 0, stream)) return rc;
    if (int rc = gmp_gemm_f32(GMP_GEMM_TN, w.G, w.zn, nullptr, w.gzn, R, d, R, R, d, d, 1.f / T, 1, 0, nullptr, 0, stream)) return rc;
    hipLaunchKernelGGL(normalize_bwd_kernel, dim3((unsigned)((R * 64 + THREADS - 1) / THREADS)), dim3(THREADS), 0, st,
                       (const float*)w.zn, (const float*)w.norm, (const float*)w.gzn, g_scale, n, d, g_z1, g_z2);
    return gmp::check_launch("nt_xent_bwd kernels");
}

// ------------------------------------------------------------------------------------------------------------------
// Grouped form: the per-domain NT-Xent problems of one task (tasks.py:192-213 runs them domain by domain) in 7 launches
// instead of 7 per domain.  Every group gets a zero-padded slot of Rmax = 2 * max_n rows in the workspace, so the three
// GEMMs are uniform batched problems (gemm_f32 grouped form, blockIdx.z = group); padding rows / columns are zeros and
// add nothing, so each group's numbers are those of the single-problem entry points above.
// ------------------------------------------------------------------------------------------------------------------
namespace {

constexpr int MAXG = 8;
struct NtxGroups {
    int G, Rmax, d;
    int n[MAXG];
    int64_t off[MAXG];      // first row of the group's [z1; z2] block in z / g_z
};

struct GWs {
    float *zn, *norm, *Gm, *gzn, *gzn2, *rowloss;
};

GWs gcarve(void* ws, int G, int64_t Rmax, int d) {
    char* p = (char*)ws;
    GWs w;
    const size_t rows = (size_t)G * Rmax;
    w.zn = (float*)p; p += al(rows * d * 4);
    w.norm = (float*)p; p += al(rows * 4);
    w.Gm = (float*)p; p += al(rows * Rmax * 4);
    w.gzn = (float*)p; p += al(rows * d * 4);
    w.gzn2 = (float*)p; p += al(rows * d * 4);
    w.rowloss = (float*)p;
    return w;
}

// one wave per padded row
__global__ __launch_bounds__(THREADS) void normalize_grouped_kernel(const float* __restrict__ z, NtxGroups g, float* __restrict__ zn,
                                                                    float* __restrict__ norm) {
    const int lane = threadIdx.x % 64;
    const int64_t pr = ((int64_t)blockIdx.x * THREADS + threadIdx.x) / 64;
    if (pr >= (int64_t)g.G * g.Rmax) return;
    const int grp = (int)(pr / g.Rmax), i = (int)(pr % g.Rmax);
    float* dst = zn + pr * g.d;
    if (i >= 2 * g.n[grp]) {
        for (int c = lane; c < g.d; c += 64) dst[c] = 0.f;
        if (lane == 0) norm[pr] = 1.f;
        return;
    }
    const float* src = z + (g.off[grp] + i) * g.d;
    float s = 0.f;
    for (int c = lane; c < g.d; c += 64) s += src[c] * src[c];
    s = gmp::wave_sum(s);
    const float nr = fmaxf(sqrtf(s), 1e-12f);
    for (int c = lane; c < g.d; c += 64) dst[c] = src[c] / nr;
    if (lane == 0) norm[pr] = nr;
}

// block (i, grp): row i of group grp, exactly row_loss_kernel on the group's R x R corner; padding -> 0
__global__ __launch_bounds__(THREADS) void row_loss_grouped_kernel(float* __restrict__ Gm, NtxGroups g, float* __restrict__ rowloss) {
    __shared__ float sh[THREADS / 64];
    const int grp = blockIdx.y;
    const int64_t i = blockIdx.x, n = g.n[grp], R = 2 * n, Rm = g.Rmax;
    float* row = Gm + ((int64_t)grp * Rm + i) * Rm;
    if (i >= R) {
        for (int64_t j = threadIdx.x; j < Rm; j += THREADS) row[j] = 0.f;
        if (threadIdx.x == 0) rowloss[(int64_t)grp * Rm + i] = 0.f;
        return;
    }
    const int64_t pos = i < n ? i + n : i - n;
    float m = -INFINITY;
    for (int64_t j = threadIdx.x; j < R; j += THREADS)
        if (j != i) m = fmaxf(m, row[j]);
    m = block_reduce(m, sh, true);
    float s = 0.f;
    for (int64_t j = threadIdx.x; j < R; j += THREADS)
        if (j != i) s += expf(row[j] - m);
    s = block_reduce(s, sh, false);
    const float lse = m + logf(s);
    const float sp = row[pos];
    __syncthreads();
    for (int64_t j = threadIdx.x; j < Rm; j += THREADS) {
        float p = 0.f;
        if (j < R) {
            p = j == i ? 0.f : expf(row[j] - lse);
            if (j == pos) p -= 1.f;
        }
        row[j] = p;
    }
    if (threadIdx.x == 0) rowloss[(int64_t)grp * Rm + i] = lse - sp;
}

// one block: per-group loss sums and their total in group order.  Wave w reduces group w (butterfly over the lanes' strided partial
// sums: a fixed order), so the groups are summed side by side and one barrier is enough.
__global__ __launch_bounds__(MAXG * 64) void loss_sums_grouped_kernel(const float* __restrict__ rowloss, NtxGroups g, float* __restrict__ sums,
                                                                    float* __restrict__ total) {
    __shared__ float sh[MAXG];
    const int grp = threadIdx.x / 64, lane = threadIdx.x % 64;
    float s = 0.f;
    if (grp < g.G) {
        const int64_t R = 2 * (int64_t)g.n[grp];
        for (int64_t i = lane; i < R; i += 64) s += rowloss[(int64_t)grp * g.Rmax + i];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) sh[grp] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot = 0.f;
        for (int k = 0; k < g.G; ++k) {
            if (sums) sums[k] = sh[k];
            tot += sh[k];
        }
        if (total) total[0] = tot;
    }
}

__global__ __launch_bounds__(THREADS) void normalize_bwd_grouped_kernel(const float* __restrict__ zn, const float* __restrict__ norm,
                                                                        const float* __restrict__ gzn, const float* __restrict__ gzn2,
                                                                        const float* __restrict__ g_scale, NtxGroups g,
                                                                        float* __restrict__ g_z) {
    const int lane = threadIdx.x % 64;
    const int64_t pr = ((int64_t)blockIdx.x * THREADS + threadIdx.x) / 64;
    if (pr >= (int64_t)g.G * g.Rmax) return;
    const int grp = (int)(pr / g.Rmax), i = (int)(pr % g.Rmax);
    if (i >= 2 * g.n[grp]) return;
    const int d = g.d;
    float dot = 0.f;
    for (int c = lane; c < d; c += 64) dot += zn[pr * d + c] * (gzn[pr * d + c] + gzn2[pr * d + c]);
    dot = gmp::wave_sum(dot);
    const float nr = norm[pr], gs = g_scale[0];
    if (nr <= 1e-12f) dot = 0.f;
    float* dst = g_z + (g.off[grp] + i) * d;
    for (int c = lane; c < d; c += 64) dst[c] = gs * ((gzn[pr * d + c] + gzn2[pr * d + c]) - zn[pr * d + c] * dot) / nr;
}

}  // namespace

extern "C" size_t gmp_nt_xent_grouped_workspace_bytes(int groups, int64_t max_n, int d) {
    if (groups < 1 || max_n < 1 || d < 1) return 0;
    const size_t Rm = (2 * (size_t)max_n + 3) / 4 * 4, rows = (size_t)groups * Rm;     // slot height, a multiple of 4 (vector loads)
    return 3 * al(rows * d * 4) + 2 * al(rows * 4) + al(rows * Rm * 4) + 256;
}

extern "C" int gmp_nt_xent_grouped(const float* z, float* g_z, int groups, const int32_t* n_host, const int64_t* row_off_host, int d,
                                   float T, const float* g_scale, float* loss_sums, float* loss_total, void* ws, size_t ws_bytes,
                                   gmp_stream_t stream) {
    if (groups < 1 || groups > MAXG || !n_host || !row_off_host) return gmp::fail(GMP_ERR_ARG, "nt_xent_grouped: groups=%d (max %d)", groups, MAXG);
    if (d < 1 || d % 4) return gmp::fail(GMP_ERR_ARG, "nt_xent_grouped: dim %d must be a positive multiple of 4", d);
    if (!(T > 0.f)) return gmp::fail(GMP_ERR_ARG, "nt_xent_grouped: temperature %f", T);
    if (!z || !g_z || !g_scale || !ws || (!loss_sums && !loss_total)) return gmp::fail(GMP_ERR_ARG, "nt_xent_grouped: null pointer");
    NtxGroups g{};
    g.G = groups; g.d = d;
    int64_t max_n = 0;
    for (int i = 0; i < groups; ++i) {
        if (n_host[i] < 0 || n_host[i] > 8192 || row_off_host[i] < 0) return gmp::fail(GMP_ERR_ARG, "nt_xent_grouped: group %d: n=%d", i, n_host[i]);
        g.n[i] = n_host[i];
        g.off[i] = row_off_host[i];
        if (n_host[i] > max_n) max_n = n_host[i];
    }
    hipStream_t st = (hipStream_t)stream;
    if (max_n == 0) {                                    // nothing to contrast anywhere: zero loss, no gradient rows
        if (loss_sums && hipMemsetAsync(loss_sums, 0, groups * sizeof(float), st) != hipSuccess) return gmp::fail(GMP_ERR_LAUNCH, "nt_xent_grouped: memset");
        if (loss_total && hipMemsetAsync(loss_total, 0, sizeof(float), st) != hipSuccess) return gmp::fail(GMP_ERR_LAUNCH, "nt_xent_grouped: memset");
        return GMP_OK;
    }
    if (ws_bytes < gmp_nt_xent_grouped_workspace_bytes(groups, max_n, d)) return gmp::fail(GMP_ERR_WORKSPACE, "nt_xent_grouped: workspace");
    const int64_t Rm = (2 * max_n + 3) / 4 * 4;
    g.Rmax = (int)Rm;
    GWs w = gcarve(ws, groups, Rm, d);
    const int64_t prow = (int64_t)groups * Rm;
    int32_t rows[MAXG + 1];
    int64_t off_zn[MAXG];
    for (int i = 0; i <= groups; ++i) rows[i] = (int32_t)(i * Rm);
    for (int i = 0; i < groups; ++i) off_zn[i] = (int64_t)i * Rm * d;
    const unsigned wave_blocks = (unsigned)((prow * 64 + THREADS - 1) / THREADS);
    hipLaunchKernelGGL(normalize_grouped_kernel, dim3(wave_blocks), dim3(THREADS), 0, st, z, g, w.zn, w.norm);
    // sim_g = zn_g zn_g^T / T
    if (int rc = gmp_gemm_f32_grouped(GMP_GEMM_NT, w.zn, w.zn, nullptr, w.Gm, groups, rows, off_zn, nullptr, nullptr, nullptr, nullptr, 0, Rm, d,
                                      d, d, Rm, 1.f / T, 0, 0, nullptr, 0, stream)) return rc;
    hipLaunchKernelGGL(row_loss_grouped_kernel, dim3((unsigned)Rm, groups), dim3(THREADS), 0, st, w.Gm, g, w.rowloss);
    hipLaunchKernelGGL(loss_sums_grouped_kernel, dim3(1), dim3(MAXG * 64), 0, st, (const float*)w.rowloss, g, loss_sums, loss_total);
    // d loss / d zn = (G + G^T) zn / T, the two halves into gzn / gzn2
    if (int rc = gmp_gemm_f32_grouped(GMP_GEMM_NN, w.Gm, w.zn, nullptr, w.gzn, groups, rows, off_zn, nullptr, nullptr, nullptr, nullptr, 0, d, Rm,
                                      Rm, d, d, 1.f / T, 0, 0, nullptr, 0, stream)) return rc;
    if (int rc = gmp_gemm_f32_grouped(GMP_GEMM_TN, w.Gm, w.zn, nullptr, w.gzn2, groups, rows, nullptr, nullptr, off_zn, nullptr, nullptr, Rm, d, 0,
                                      Rm, d, d, 1.f / T, 0, 0, nullptr, 0, stream)) return rc;
    hipLaunchKernelGGL(normalize_bwd_grouped_kernel, dim3(wave_blocks), dim3(THREADS), 0, st, (const float*)w.zn, (const float*)w.norm,
                       (const float*)w.gzn, (const float*)w.gzn2, g_scale, g, g_z);
    return gmp::check_launch("nt_xent_grouped kernels");
}
