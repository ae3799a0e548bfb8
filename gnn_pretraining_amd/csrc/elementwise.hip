// Small elementwise kernels of the heads: dropout (counter-based mask, regenerated in
// the backward) and the fused ReLU+dropout backward.  float4 per thread, grid-stride.
#include "gnnmp_internal.h"

namespace {

__global__ __launch_bounds__(256) void dropout_kernel(const float4* __restrict__ x, float4* __restrict__ y, int64_t n4,
                                                      float p, uint64_t seed, uint32_t sid) {
    const float inv = 1.f / (1.f - p);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 v = x[i], d = gmp::dropout_scale4(seed, sid, (uint64_t)i, p, inv);
        y[i] = make_float4(v.x * d.x, v.y * d.y, v.z * d.z, v.w * d.w);
    }
}

// out = g * dropmask * (act > 0)   where act is the ReLU output that fed the dropout
__global__ __launch_bounds__(256) void relu_dropout_bwd_kernel(const float4* __restrict__ g, const float4* __restrict__ act,
                                                               float4* __restrict__ out, int64_t n4, float p, uint64_t seed,
                                                               uint32_t sid) {
    const float inv = p > 0.f ? 1.f / (1.f - p) : 1.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 v = g[i], a = act[i];
        float4 d = make_float4(1.f, 1.f, 1.f, 1.f);
        if (p > 0.f) d = gmp::dropout_scale4(seed, sid, (uint64_t)i, p, inv);
        out[i] = make_float4(a.x > 0.f ? v.x * d.x : 0.f, a.y > 0.f ? v.y * d.y : 0.f, a.z > 0.f ? v.z * d.z : 0.f,
                             a.w > 0.f ? v.w * d.w : 0.f);
    }
}

// ---- the 256 -> 1 layer of the link-prediction scorer (heads.py:45-52: Linear(hidden, 1) behind ReLU + dropout) without GEMM launches:
// a [M, F] x [F] product is one dot product per row, its input gradient an outer product, its weight gradient a weighted column sum --
// 12 + 11 + 54 us through the GEMM paths (N = 1 tiles, a [1 x F] grouped reduction), a few us as what they are: passes over [M, F].
// y[m] = sum_c dropout(x)[m, c] * w[c] + b; d (nullable when p = 0) receives dropout(x).  One wave per row, lanes stride float4.
__global__ __launch_bounds__(256) void dropout_rowdot_kernel(const float4* __restrict__ x, const float4* __restrict__ w, const float* __restrict__ b,
                                                             float4* __restrict__ d, float* __restrict__ y, int64_t M, int F4, float p,
                                                             uint64_t seed, uint32_t sid) {
    const int lane = threadIdx.x % 64;
    const float inv = p > 0.f ? 1.f / (1.f - p) : 1.f;
    const float bias = b ? b[0] : 0.f;
    for (int64_t m = (int64_t)blockIdx.x * 4 + threadIdx.x / 64; m < M; m += (int64_t)gridDim.x * 4) {
        float s = 0.f;
        for (int c = lane; c < F4; c += 64) {
            const int64_t i = m * F4 + c;
            float4 v = x[i];
            if (p > 0.f) {
                const float4 k = gmp::dropout_scale4(seed, sid, (uint64_t)i, p, inv);
                v = make_float4(v.x * k.x, v.y * k.y, v.z * k.z, v.w * k.w);
                d[i] = v;
            }
            const float4 ww = w[c];
            s += (v.x * ww.x + v.y * ww.y) + (v.z * ww.z + v.w * ww.w);
        }
        s = gmp::wave_sum(s);
        if (lane == 0) y[m] = s + bias;
    }
}

// out[m, c] = g[m] * w[c] * dropmask * (act[m, c] > 0): the input gradient of that layer pushed through the dropout and the ReLU in front of it
__global__ __launch_bounds__(256) void outer_relu_dropout_bwd_kernel(const float* __restrict__ g, const float4* __restrict__ w, const float4* __restrict__ act,
                                                                     float4* __restrict__ out, int64_t n4, int F4, float p, uint64_t seed, uint32_t sid) {
    const float inv = p > 0.f ? 1.f / (1.f - p) : 1.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const int64_t m = i / F4;
        const float gm = g[m];
        const float4 ww = w[i - m * F4], a = act[i];
        float4 k = make_float4(1.f, 1.f, 1.f, 1.f);
        if (p > 0.f) k = gmp::dropout_scale4(seed, sid, (uint64_t)i, p, inv);
        out[i] = make_float4(a.x > 0.f ? gm * ww.x * k.x : 0.f, a.y > 0.f ? gm * ww.y * k.y : 0.f, a.z > 0.f ? gm * ww.z * k.z : 0.f,
                             a.w > 0.f ? gm * ww.w * k.w : 0.f);
    }
}

// weight / bias gradient of that layer: part[j][c] = sum over the j-th 256-row chunk of g[m] * X[m, c] (+ the chunk's sum of g in column F),
// then the chunks in order: deterministic
constexpr int WCS_ROWS = 64;        // rows per partial block: 15 k rows -> 240 x (F / 64) blocks of 16 row steps (256-row chunks: 33 us of serial steps)
__global__ __launch_bounds__(256) void weighted_colsum_partial_kernel(const float* __restrict__ g, const float* __restrict__ X, float* __restrict__ part,
                                                                      int64_t M, int F) {
    __shared__ float sh[4][64];
    __shared__ float shg[4];
    const int c = threadIdx.x % 64, rl = threadIdx.x / 64;
    const int64_t col = (int64_t)blockIdx.x * 64 + c;
    const int64_t r0 = (int64_t)blockIdx.y * WCS_ROWS, r1 = r0 + WCS_ROWS < M ? r0 + WCS_ROWS : M;
    float s = 0.f, sg = 0.f;
    for (int64_t r = r0 + rl; r < r1; r += 4) {
        const float gr = g[r];
        if (col < F) s += gr * X[r * F + col];
        sg += gr;
    }
    sh[rl][c] = s;
    if (c == 0) shg[rl] = sg;
    __syncthreads();
    if (rl == 0 && col < F) part[(int64_t)blockIdx.y * (F + 1) + col] = (sh[0][c] + sh[1][c]) + (sh[2][c] + sh[3][c]);
    if (threadIdx.x == 0 && blockIdx.x == 0) part[(int64_t)blockIdx.y * (F + 1) + F] = (shg[0] + shg[1]) + (shg[2] + shg[3]);
}
// one wave per output column: lanes stride the chunk partials, a butterfly adds them (fixed order: deterministic)
__global__ __launch_bounds__(256) void weighted_colsum_final_kernel(const float* __restrict__ part, float* __restrict__ out_w, float* __restrict__ out_b,
                                                                    int64_t nparts, int F) {
    const int col = blockIdx.x * 4 + threadIdx.x / 64, lane = threadIdx.x % 64;
    if (col > F) return;
    float s = 0.f;
    for (int64_t q = lane; q < nparts; q += 64) s += part[q * (F + 1) + col];
    s = gmp::wave_sum(s);
    if (lane == 0) {
        if (col < F) out_w[col] = s;
        else if (out_b) out_b[0] = s;
    }
}

// ---- the same layer over MERGED link-prediction rows: one row per unordered pair through the 768 -> 256 layer and its ReLU (identical
// for (i, j) and (j, i): the features of heads.py:57-61 are symmetric), but the reference's Dropout(0.2) (heads.py:44-52) draws a mask for
// every ORDERED row of its list (tasks.py:111-120), so a merged row that stands for two ordered rows has two masks, two scores, two
// BCE terms.  pos[m] = ordered position of the row's first occurrence, pos[M + m] = of its second (-1: none); a mask is keyed by
// (seed, site, ordered position * F/4 + column quad) -- exactly the key the unmerged path (dropout_rowdot_kernel over the ordered
// list) uses, so per ordered row the two paths draw the same mask and compute the same score, bit for bit.
// y[m] = score of the first occurrence, y[M + m] = of the second (copy of the first when p = 0; unused when there is none).
__global__ __launch_bounds__(256) void lp_pair_rowdot_kernel(const float4* __restrict__ x, const float4* __restrict__ w, const float* __restrict__ b,
                                                             const int32_t* __restrict__ pos, float* __restrict__ y, int64_t M, int F4, float p,
                                                             uint64_t seed, uint32_t sid) {
    const int lane = threadIdx.x % 64;
    const float inv = p > 0.f ? 1.f / (1.f - p) : 1.f;
    const float bias = b ? b[0] : 0.f;
    for (int64_t m = (int64_t)blockIdx.x * 4 + threadIdx.x / 64; m < M; m += (int64_t)gridDim.x * 4) {
        const int64_t pa = pos[m], pb = pos[M + m];
        float sa = 0.f, sb = 0.f;
        for (int c = lane; c < F4; c += 64) {
            const float4 v = x[m * F4 + c], ww = w[c];
            if (p > 0.f) {
                const float4 k = gmp::dropout_scale4(seed, sid, (uint64_t)(pa * F4 + c), p, inv);
                sa += ((v.x * k.x) * ww.x + (v.y * k.y) * ww.y) + ((v.z * k.z) * ww.z + (v.w * k.w) * ww.w);
                if (pb >= 0) {
                    const float4 q = gmp::dropout_scale4(seed, sid, (uint64_t)(pb * F4 + c), p, inv);
                    sb += ((v.x * q.x) * ww.x + (v.y * q.y) * ww.y) + ((v.z * q.z) * ww.z + (v.w * q.w) * ww.w);
                }
            } else {
                sa += (v.x * ww.x + v.y * ww.y) + (v.z * ww.z + v.w * ww.w);
            }
        }
        sa = gmp::wave_sum(sa);
        sb = p > 0.f ? gmp::wave_sum(sb) : sa;
        if (lane == 0) {
            y[m] = sa + bias;
            y[M + m] = sb + bias;
        }
    }
}

// out[m, c] = (act[m, c] > 0) * w[c] * (g[m] * mask_a + g[M + m] * mask_b): the two ordered rows' input gradients added (they meet in the
// 768 -> 256 layer's weight and input gradients anyway); g[M + m] is zero where the row has no second occurrence
__global__ __launch_bounds__(256) void lp_pair_outer_bwd_kernel(const float* __restrict__ g, const float4* __restrict__ w, const float4* __restrict__ act,
                                                                const int32_t* __restrict__ pos, float4* __restrict__ out, int64_t M, int F4, float p,
                                                                uint64_t seed, uint32_t sid) {
    const float inv = p > 0.f ? 1.f / (1.f - p) : 1.f;
    const int64_t n4 = M * F4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const int64_t m = i / F4;
        const int c = (int)(i - m * F4);
        const float ga = g[m], gb = g[M + m];
        const float4 ww = w[c], a = act[i];
        float4 k = make_float4(ga + gb, ga + gb, ga + gb, ga + gb);
        if (p > 0.f) {
            const int64_t pa = pos[m], pb = pos[M + m];
            const float4 ka = gmp::dropout_scale4(seed, sid, (uint64_t)(pa * F4 + c), p, inv);
            k = make_float4(ga * ka.x, ga * ka.y, ga * ka.z, ga * ka.w);
            if (pb >= 0) {
                const float4 kb = gmp::dropout_scale4(seed, sid, (uint64_t)(pb * F4 + c), p, inv);
                k = make_float4(k.x + gb * kb.x, k.y + gb * kb.y, k.z + gb * kb.z, k.w + gb * kb.w);
            }
        }
        out[i] = make_float4(a.x > 0.f ? ww.x * k.x : 0.f, a.y > 0.f ? ww.y * k.y : 0.f, a.z > 0.f ? ww.z * k.z : 0.f, a.w > 0.f ? ww.w * k.w : 0.f);
    }
}

// weight / bias gradient of the layer over merged rows: part[j][c] = sum over the j-th chunk of rows of act[m, c] * (g[m] * mask_a + g[M + m] * mask_b)
// (+ the chunk's sum of g[m] + g[M + m] in column F); the dropped activations are rebuilt from the masks, never stored.  One wave per row,
// lane = column quad (F <= 256), four rows of a chunk in flight per block; weighted_colsum_final_kernel adds the chunks in order.
constexpr int LPW_ROWS = 32;
__global__ __launch_bounds__(256) void lp_pair_colsum_partial_kernel(const float* __restrict__ g, const float4* __restrict__ act, const int32_t* __restrict__ pos,
                                                                     float* __restrict__ part, int64_t M, int F4, float p, uint64_t seed, uint32_t sid) {
    __shared__ float4 sh[4][64];
    __shared__ float shg[4];
    const int lane = threadIdx.x % 64, wv = threadIdx.x / 64;
    const float inv = p > 0.f ? 1.f / (1.f - p) : 1.f;
    const int64_t r0 = (int64_t)blockIdx.x * LPW_ROWS, r1 = r0 + LPW_ROWS < M ? r0 + LPW_ROWS : M;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    float sg = 0.f;
    for (int64_t m = r0 + wv; m < r1; m += 4) {
        const float ga = g[m], gb = g[M + m];
        sg += ga + gb;
        if (lane < F4) {
            const float4 a = act[m * F4 + lane];
            float4 k = make_float4(ga + gb, ga + gb, ga + gb, ga + gb);
            if (p > 0.f) {
                const int64_t pa = pos[m], pb = pos[M + m];
                const float4 ka = gmp::dropout_scale4(seed, sid, (uint64_t)(pa * F4 + lane), p, inv);
                k = make_float4(ga * ka.x, ga * ka.y, ga * ka.z, ga * ka.w);
                if (pb >= 0) {
                    const float4 kb = gmp::dropout_scale4(seed, sid, (uint64_t)(pb * F4 + lane), p, inv);
                    k = make_float4(k.x + gb * kb.x, k.y + gb * kb.y, k.z + gb * kb.z, k.w + gb * kb.w);
                }
            }
            s = make_float4(s.x + a.x * k.x, s.y + a.y * k.y, s.z + a.z * k.z, s.w + a.w * k.w);
        }
    }
    sh[wv][lane] = s;
    if (lane == 0) shg[wv] = sg;
    __syncthreads();
    const int F = 4 * F4;
    if (wv == 0 && lane < F4) {
        const float4 a = sh[0][lane], b = sh[1][lane], c = sh[2][lane], d = sh[3][lane];
        float* o = part + (int64_t)blockIdx.x * (F + 1) + 4 * lane;
        o[0] = (a.x + b.x) + (c.x + d.x); o[1] = (a.y + b.y) + (c.y + d.y); o[2] = (a.z + b.z) + (c.z + d.z); o[3] = (a.w + b.w) + (c.w + d.w);
    }
    if (threadIdx.x == 0) part[(int64_t)blockIdx.x * (F + 1) + F] = (shg[0] + shg[1]) + (shg[2] + shg[3]);
}

int grid_for(int64_t n4) {
    int64_t b = (n4 + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

}  // namespace

extern "C" int gmp_dropout_fwd(const float* x, float* y, int64_t numel, float p, uint64_t seed, uint32_t stream_id,
                               gmp_stream_t stream) {
    if (numel < 0 || numel % 4 || p < 0.f || p >= 1.f) return gmp::fail(GMP_ERR_ARG, "dropout_fwd: numel=%lld p=%f", (long long)numel, p);
    if (numel == 0) return GMP_OK;
    if (!x || !y) return gmp::fail(GMP_ERR_ARG, "dropout_fwd: null pointer");
    hipLaunchKernelGGL(dropout_kernel, dim3(grid_for(numel / 4)), dim3(256), 0, (hipStream_t)stream, (const float4*)x,
                       (float4*)y, numel / 4, p, seed, stream_id);
    return gmp::check_launch("dropout_kernel");
}

extern "C" int gmp_relu_dropout_bwd(const float* g, const float* act, float* out, int64_t numel, float p, uint64_t seed,
                                    uint32_t stream_id, gmp_stream_t stream) {
    if (numel < 0 || numel % 4 || p < 0.f || p >= 1.f) return gmp::fail(GMP_ERR_ARG, "relu_dropout_bwd: numel=%lld p=%f", (long long)numel, p);
    if (numel == 0) return GMP_OK;
    if (!g || !act || !out) return gmp::fail(GMP_ERR_ARG, "relu_dropout_bwd: null pointer");
    hipLaunchKernelGGL(relu_dropout_bwd_kernel, dim3(grid_for(numel / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)g, (const float4*)act, (float4*)out, numel / 4, p, seed, stream_id);
    return gmp::check_launch("relu_dropout_bwd_kernel");
}

extern "C" int gmp_dropout_rowdot_fwd(const float* x, const float* w, const float* bias, float* dropped, float* y, int64_t rows, int feat, float p,
                                      uint64_t seed, uint32_t stream_id, gmp_stream_t stream) {
    if (rows < 0 || feat <= 0 || feat % 4 || p < 0.f || p >= 1.f) return gmp::fail(GMP_ERR_ARG, "dropout_rowdot_fwd: rows=%lld feat=%d p=%f", (long long)rows, feat, p);
    if (rows == 0) return GMP_OK;
    if (!x || !w || !y || (p > 0.f && !dropped)) return gmp::fail(GMP_ERR_ARG, "dropout_rowdot_fwd: null pointer");
    const int64_t blocks = (rows + 3) / 4;
    hipLaunchKernelGGL(dropout_rowdot_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, (hipStream_t)stream, (const float4*)x,
                       (const float4*)w, bias, (float4*)dropped, y, rows, feat / 4, p, seed, stream_id);
    return gmp::check_launch("dropout_rowdot_kernel");
}

extern "C" int gmp_outer_relu_dropout_bwd(const float* g, const float* w, const float* act, float* out, int64_t rows, int feat, float p,
                                          uint64_t seed, uint32_t stream_id, gmp_stream_t stream) {
    if (rows < 0 || feat <= 0 || feat % 4 || p < 0.f || p >= 1.f) return gmp::fail(GMP_ERR_ARG, "outer_relu_dropout_bwd: rows=%lld feat=%d p=%f", (long long)rows, feat, p);
    if (rows == 0) return GMP_OK;
    if (!g || !w || !act || !out) return gmp::fail(GMP_ERR_ARG, "outer_relu_dropout_bwd: null pointer");
    const int64_t n4 = rows * (feat / 4);
    hipLaunchKernelGGL(outer_relu_dropout_bwd_kernel, dim3(grid_for(n4)), dim3(256), 0, (hipStream_t)stream, g, (const float4*)w, (const float4*)act,
                       (float4*)out, n4, feat / 4, p, seed, stream_id);
    return gmp::check_launch("outer_relu_dropout_bwd_kernel");
}

extern "C" size_t gmp_weighted_colsum_workspace_bytes(int64_t rows, int feat) {
    if (rows <= 0 || feat <= 0) return 0;
    return (size_t)((rows + WCS_ROWS - 1) / WCS_ROWS) * (size_t)(feat + 1) * sizeof(float);
}

extern "C" int gmp_weighted_colsum(const float* g, const float* x, float* out_w, float* out_b, int64_t rows, int feat, void* ws, size_t ws_bytes,
                                   gmp_stream_t stream) {
    if (rows < 0 || feat <= 0) return gmp::fail(GMP_ERR_ARG, "weighted_colsum: rows=%lld feat=%d", (long long)rows, feat);
    if (!out_w) return gmp::fail(GMP_ERR_ARG, "weighted_colsum: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (rows == 0) {
        (void)hipMemsetAsync(out_w, 0, (size_t)feat * sizeof(float), st);
        if (out_b) (void)hipMemsetAsync(out_b, 0, sizeof(float), st);
        return GMP_OK;
    }
    if (!g || !x) return gmp::fail(GMP_ERR_ARG, "weighted_colsum: null pointer");
    if (!ws || ws_bytes < gmp_weighted_colsum_workspace_bytes(rows, feat)) return gmp::fail(GMP_ERR_WORKSPACE, "weighted_colsum: workspace");
    const int64_t parts = (rows + WCS_ROWS - 1) / WCS_ROWS;
    hipLaunchKernelGGL(weighted_colsum_partial_kernel, dim3((unsigned)((feat + 63) / 64), (unsigned)parts), dim3(256), 0, st, g, x, (float*)ws, rows, feat);
    hipLaunchKernelGGL(weighted_colsum_final_kernel, dim3((unsigned)((feat + 1 + 3) / 4)), dim3(256), 0, st, (const float*)ws, out_w, out_b, parts, feat);
    return gmp::check_launch("weighted_colsum kernels");
}

// ---- merged link-prediction rows, one dropout mask per ORDERED row (kernels above) ------------------------------------------------------
static int lp_pair_args(const char* what, int64_t rows, int feat, float p, const int32_t* pos) {
    if (rows < 0 || feat <= 0 || feat % 4 || feat > 256 || p < 0.f || p >= 1.f) return gmp::fail(GMP_ERR_ARG, "%s: rows=%lld feat=%d p=%f", what, (long long)rows, feat, p);
    if (rows > 0 && !pos) return gmp::fail(GMP_ERR_ARG, "%s: null positions", what);
    return GMP_OK;
}

extern "C" int gmp_lp_pair_rowdot_fwd(const float* x, const float* w, const float* bias, const int32_t* pos, float* y2, int64_t rows, int feat, float p,
                                      uint64_t seed, uint32_t stream_id, gmp_stream_t stream) {
    if (int rc = lp_pair_args("lp_pair_rowdot_fwd", rows, feat, p, pos)) return rc;
    if (rows == 0) return GMP_OK;
    if (!x || !w || !y2) return gmp::fail(GMP_ERR_ARG, "lp_pair_rowdot_fwd: null pointer");
    const int64_t blocks = (rows + 3) / 4;
    hipLaunchKernelGGL(lp_pair_rowdot_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, (hipStream_t)stream, (const float4*)x,
                       (const float4*)w, bias, pos, y2, rows, feat / 4, p, seed, stream_id);
    return gmp::check_launch("lp_pair_rowdot_kernel");
}

extern "C" int gmp_lp_pair_outer_bwd(const float* g_y2, const float* w, const float* act, const int32_t* pos, float* out, int64_t rows, int feat, float p,
                                     uint64_t seed, uint32_t stream_id, gmp_stream_t stream) {
    if (int rc = lp_pair_args("lp_pair_outer_bwd", rows, feat, p, pos)) return rc;
    if (rows == 0) return GMP_OK;
    if (!g_y2 || !w || !act || !out) return gmp::fail(GMP_ERR_ARG, "lp_pair_outer_bwd: null pointer");
    hipLaunchKernelGGL(lp_pair_outer_bwd_kernel, dim3(grid_for(rows * (feat / 4))), dim3(256), 0, (hipStream_t)stream, g_y2, (const float4*)w,
                       (const float4*)act, pos, (float4*)out, rows, feat / 4, p, seed, stream_id);
    return gmp::check_launch("lp_pair_outer_bwd_kernel");
}

extern "C" size_t gmp_lp_pair_colsum_workspace_bytes(int64_t rows, int feat) {
    if (rows <= 0 || feat <= 0) return 0;
    return (size_t)((rows + LPW_ROWS - 1) / LPW_ROWS) * (size_t)(feat + 1) * sizeof(float);
}

extern "C" int gmp_lp_pair_weighted_colsum(const float* g_y2, const float* act, const int32_t* pos, float* out_w, float* out_b, int64_t rows, int feat,
                                           float p, uint64_t seed, uint32_t stream_id, void* ws, size_t ws_bytes, gmp_stream_t stream) {
    if (int rc = lp_pair_args("lp_pair_weighted_colsum", rows, feat, p, pos)) return rc;
    if (!out_w) return gmp::fail(GMP_ERR_ARG, "lp_pair_weighted_colsum: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (rows == 0) {
        (void)hipMemsetAsync(out_w, 0, (size_t)feat * sizeof(float), st);
        if (out_b) (void)hipMemsetAsync(out_b, 0, sizeof(float), st);
        return GMP_OK;
    }
    if (!g_y2 || !act) return gmp::fail(GMP_ERR_ARG, "lp_pair_weighted_colsum: null pointer");
    if (!ws || ws_bytes < gmp_lp_pair_colsum_workspace_bytes(rows, feat)) return gmp::fail(GMP_ERR_WORKSPACE, "lp_pair_weighted_colsum: workspace");
    const int64_t parts = (rows + LPW_ROWS - 1) / LPW_ROWS;
    hipLaunchKernelGGL(lp_pair_colsum_partial_kernel, dim3((unsigned)parts), dim3(256), 0, st, g_y2, (const float4*)act, pos, (float*)ws, rows, feat / 4, p, seed,
                       stream_id);
    hipLaunchKernelGGL(weighted_colsum_final_kernel, dim3((unsigned)((feat + 1 + 3) / 4)), dim3(256), 0, st, (const float*)ws, out_w, out_b, parts, feat);
    return gmp::check_launch("lp_pair_weighted_colsum kernels");
}
