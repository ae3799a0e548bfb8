// Loss kernels of the pre-training tasks: sum-reduced MSE (tasks.py:83,305), sigmoid +
// binary cross-entropy on probabilities with torch's -100 log clamp (heads.py:67,
// tasks.py:120), row-wise cross-entropy (tasks.py:336, finetune.py:158,177), and the
// mask-token row fill of apply_node_masking (pretrain_model.py:82-85).
// Reductions are two-stage with a fixed order (deterministic).
#include <algorithm>

#include "gnnmp_internal.h"

namespace {

constexpr int T = 256;
constexpr int MAX_PARTS = 1024;

__device__ __forceinline__ void block_sum_store(float v, float* part) {
    __shared__ float sh[T / 64];
    v = gmp::wave_sum(v);
    if (threadIdx.x % 64 == 0) sh[threadIdx.x / 64] = v;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__global__ __launch_bounds__(T) void final_sum_kernel(const float* __restrict__ part, int n, float* out) {
    __shared__ float sh[T];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += T) s += part[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int d = T / 2; d > 0; d >>= 1) {
        if (threadIdx.x < d) sh[threadIdx.x] += sh[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sh[0];
}

__global__ __launch_bounds__(T) void mse_part_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t n, float* part) {
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * T + threadIdx.x; i < n; i += (int64_t)gridDim.x * T) {
        float d = a[i] - b[i];
        s += d * d;
    }
    block_sum_store(s, part);
}
__global__ __launch_bounds__(T) void mse_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ gs,
                                                    float* __restrict__ ga, int64_t n) {
    const float g = 2.f * gs[0];
    for (int64_t i = (int64_t)blockIdx.x * T + threadIdx.x; i < n; i += (int64_t)gridDim.x * T) ga[i] = g * (a[i] - b[i]);
}

__global__ __launch_bounds__(T) void sigmoid_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * T + threadIdx.x; i < n; i += (int64_t)gridDim.x * T) y[i] = 1.f / (1.f + expf(-x[i]));
}
__global__ __launch_bounds__(T) void sigmoid_bwd_kernel(const float* __restrict__ g, const float* __restrict__ y, float* __restrict__ o, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * T + threadIdx.x; i < n; i += (int64_t)gridDim.x * T) o[i] = g[i] * y[i] * (1.f - y[i]);
}

// torch.nn.functional.binary_cross_entropy: -(y*max(log p,-100) + (1-y)*max(log(1-p),-100))
__global__ __launch_bounds__(T) void bce_part_kernel(const float* __restrict__ p, const float* __restrict__ y, int64_t n, float* part) {
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * T + threadIdx.x; i < n; i += (int64_t)gridDim.x * T) {
        const float lp = fmaxf(logf(p[i]), -100.f), lq = fmaxf(log1pf(-p[i]), -100.f);
        s -= y[i] * lp + (1.f - y[i]) * lq;
    }
    block_sum_store(s, part);
}
// torch's binary_cross_entropy_backward: g * (p - y) / max((1 - p) * p, 1e-12)
__global__ __launch_bounds__(T) void bce_bwd_kernel(const float* __restrict__ p, const float* __restrict__ y, const float* __restrict__ gs,
                                                    float* __restrict__ gp, int64_t n) {
    const float g = gs[0];
    for (int64_t i = (int64_t)blockIdx.x * T + threadIdx.x; i < n; i += (int64_t)gridDim.x * T)
        gp[i] = g * (p[i] - y[i]) / fmaxf((1.f - p[i]) * p[i], 1e-12f);
}

// sigmoid -> BCE(sum) -> d/dp -> d/dx in one pass (the link-prediction scorer's tail, heads.py:67 + tasks.py:120): the
// arithmetic of sigmoid_kernel, bce_part_kernel, bce_bwd_kernel and sigmoid_bwd_kernel in that order, element by element,
// so the numbers are those of the four separate launches.  SIGNED: y holds +w for a positive pair and -w for a negative one
// (w = how many times the pair stands in the reference's list); loss term and gradient are scaled by w (exact for w = 1).
template <bool SIGNED>
__global__ __launch_bounds__(T) void sigmoid_bce_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ gs,
                                                        float* __restrict__ p_out, float* __restrict__ gx, int64_t n, float* part) {
    const float g = gs[0];
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * T + threadIdx.x; i < n; i += (int64_t)gridDim.x * T) {
        const float p = 1.f / (1.f + expf(-x[i]));
        const float lp = fmaxf(logf(p), -100.f), lq = fmaxf(log1pf(-p), -100.f);
        const float yi = SIGNED ? (y[i] > 0.f ? 1.f : 0.f) : y[i], w = SIGNED ? fabsf(y[i]) : 1.f;
        s -= w * (yi * lp + (1.f - yi) * lq);
        const float gp = g * (p - yi) / fmaxf((1.f - p) * p, 1e-12f);
        gx[i] = w * (gp * p * (1.f - p));
        if (p_out) p_out[i] = p;
    }
    block_sum_store(s, part);
}

// The same pass over MERGED link-prediction rows (elementwise.hip lp_pair_*): x = [2, M] scores (first / second ordered occurrence of every
// row), sign[m] > 0 for a positive pair, pos[M + m] < 0 where a row has no second occurrence (its entry adds nothing: loss 0, gradient 0).
// Every ordered row of the reference's list (tasks.py:111-120) gets its own term, as in the unmerged pass.
__global__ __launch_bounds__(T) void sigmoid_bce_pair_kernel(const float* __restrict__ x, const float* __restrict__ sign, const int32_t* __restrict__ pos,
                                                             const float* __restrict__ gs, float* __restrict__ p_out, float* __restrict__ gx, int64_t M,
                                                             float* part) {
    const float g = gs[0];
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * T + threadIdx.x; i < 2 * M; i += (int64_t)gridDim.x * T) {
        const int64_t m = i < M ? i : i - M;
        const bool live = i < M || pos[i] >= 0;
        const float p = 1.f / (1.f + expf(-x[i]));
        const float lp = fmaxf(logf(p), -100.f), lq = fmaxf(log1pf(-p), -100.f);
        const float yi = sign[m] > 0.f ? 1.f : 0.f;
        const float gp = g * (p - yi) / fmaxf((1.f - p) * p, 1e-12f);
        if (live) s -= yi * lp + (1.f - yi) * lq;
        gx[i] = live ? gp * p * (1.f - p) : 0.f;
        if (p_out) p_out[i] = live ? p : 0.f;
    }
    block_sum_store(s, part);
}

// one wave per row: loss_m = logsumexp(logits[m,:]) - logits[m,target]; probs optional output
__global__ __launch_bounds__(T) void ce_rows_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target, int64_t M, int C,
                                                    float* __restrict__ rowloss, float* __restrict__ gl, const float* __restrict__ gs) {
    const int lane = threadIdx.x % 64;
    const int64_t m = ((int64_t)blockIdx.x * T + threadIdx.x) / 64;
    if (m >= M) return;
    const float* row = logits + m * C;
    float mx = -INFINITY;
    for (int c = lane; c < C; c += 64) mx = fmaxf(mx, row[c]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += expf(row[c] - mx);
    s = gmp::wave_sum(s);
    const float lse = mx + logf(s);
    const int64_t t = target[m];
    const bool ok = t >= 0 && t < C;
    if (rowloss && lane == 0) rowloss[m] = ok ? lse - row[t] : 0.f;
    if (gl) {
        const float g = gs[0];
        for (int c = lane; c < C; c += 64) gl[m * C + c] = ok ? g * (expf(row[c] - lse) - (c == t ? 1.f : 0.f)) : 0.f;
    }
}

__global__ __launch_bounds__(T) void row_fill_kernel(float4* __restrict__ dst, const int64_t* __restrict__ idx, const float4* __restrict__ src,
                                                     int64_t M, int64_t ndst, int F4, int broadcast) {
    const int lane = threadIdx.x % 64;
    for (int64_t m = ((int64_t)blockIdx.x * T + threadIdx.x) / 64; m < M; m += (int64_t)gridDim.x * (T / 64)) {
        const int64_t r = idx[m];
        if (r < 0 || r >= ndst) continue;
        for (int c = lane; c < F4; c += 64) dst[r * F4 + c] = broadcast ? src[c] : src[m * F4 + c];
    }
}

int parts_for(int64_t n) {
    int64_t b = (n + T - 1) / T;
    return (int)(b < 1 ? 1 : (b > MAX_PARTS ? MAX_PARTS : b));
}

}  // namespace

extern "C" size_t gmp_loss_workspace_bytes(int64_t numel) { return (size_t)(numel > MAX_PARTS ? numel : MAX_PARTS) * sizeof(float) + 256; }

#define GMP_CHECK_N(name) \
    if (n < 0) return gmp::fail(GMP_ERR_ARG, name ": negative size"); \
    hipStream_t st = (hipStream_t)stream;

extern "C" int gmp_mse_sum_fwd(const float* a, const float* b, int64_t n, float* loss, void* ws, size_t ws_bytes, gmp_stream_t stream) {
    GMP_CHECK_N("mse_sum_fwd")
    if (!loss || !ws || ws_bytes < MAX_PARTS * sizeof(float) || (n > 0 && (!a || !b))) return gmp::fail(GMP_ERR_ARG, "mse_sum_fwd: bad argument");
    const int parts = parts_for(n);
    hipLaunchKernelGGL(mse_part_kernel, dim3(parts), dim3(T), 0, st, a, b, n, (float*)ws);
    hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(T), 0, st, (const float*)ws, parts, loss);
    return gmp::check_launch("mse kernels");
}
extern "C" int gmp_mse_sum_bwd(const float* a, const float* b, const float* g_scale, float* g_a, int64_t n, gmp_stream_t stream) {
    GMP_CHECK_N("mse_sum_bwd")
    if (n == 0) return GMP_OK;
    if (!a || !b || !g_scale || !g_a) return gmp::fail(GMP_ERR_ARG, "mse_sum_bwd: null pointer");
    hipLaunchKernelGGL(mse_bwd_kernel, dim3(parts_for(n)), dim3(T), 0, st, a, b, g_scale, g_a, n);
    return gmp::check_launch("mse_bwd_kernel");
}
extern "C" int gmp_sigmoid_fwd(const float* x, float* y, int64_t n, gmp_stream_t stream) {
    GMP_CHECK_N("sigmoid_fwd")
    if (n == 0) return GMP_OK;
    if (!x || !y) return gmp::fail(GMP_ERR_ARG, "sigmoid_fwd: null pointer");
    hipLaunchKernelGGL(sigmoid_kernel, dim3(parts_for(n)), dim3(T), 0, st, x, y, n);
    return gmp::check_launch("sigmoid_kernel");
}
extern "C" int gmp_sigmoid_bwd(const float* g, const float* y, float* out, int64_t n, gmp_stream_t stream) {
    GMP_CHECK_N("sigmoid_bwd")
    if (n == 0) return GMP_OK;
    if (!g || !y || !out) return gmp::fail(GMP_ERR_ARG, "sigmoid_bwd: null pointer");
    hipLaunchKernelGGL(sigmoid_bwd_kernel, dim3(parts_for(n)), dim3(T), 0, st, g, y, out, n);
    return gmp::check_launch("sigmoid_bwd_kernel");
}
extern "C" int gmp_bce_sum_fwd(const float* p, const float* labels, int64_t n, float* loss, void* ws, size_t ws_bytes, gmp_stream_t stream) {
    GMP_CHECK_N("bce_sum_fwd")
    if (!loss || !ws || ws_bytes < MAX_PARTS * sizeof(float) || (n > 0 && (!p || !labels))) return gmp::fail(GMP_ERR_ARG, "bce_sum_fwd: bad argument");
    const int parts = parts_for(n);
    hipLaunchKernelGGL(bce_part_kernel, dim3(parts), dim3(T), 0, st, p, labels, n, (float*)ws);
    hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(T), 0, st, (const float*)ws, parts, loss);
    return gmp::check_launch("bce kernels");
}
extern "C" int gmp_bce_sum_bwd(const float* p, const float* labels, const float* g_scale, float* g_p, int64_t n, gmp_stream_t stream) {
    GMP_CHECK_N("bce_sum_bwd")
    if (n == 0) return GMP_OK;
    if (!p || !labels || !g_scale || !g_p) return gmp::fail(GMP_ERR_ARG, "bce_sum_bwd: null pointer");
    hipLaunchKernelGGL(bce_bwd_kernel, dim3(parts_for(n)), dim3(T), 0, st, p, labels, g_scale, g_p, n);
    return gmp::check_launch("bce_bwd_kernel");
}
static int sigmoid_bce_launch(bool signed_w, const float* x, const float* labels, int64_t n, const float* g_scale, float* loss, float* p_out,
                              float* g_x, void* ws, size_t ws_bytes, gmp_stream_t stream) {
    GMP_CHECK_N("sigmoid_bce_sum_fwd_bwd")
    if (!loss || !g_scale || !ws || ws_bytes < MAX_PARTS * sizeof(float) || (n > 0 && (!x || !labels || !g_x)))
        return gmp::fail(GMP_ERR_ARG, "sigmoid_bce_sum_fwd_bwd: bad argument");
    const int parts = parts_for(n);
    if (signed_w) hipLaunchKernelGGL(sigmoid_bce_kernel<true>, dim3(parts), dim3(T), 0, st, x, labels, g_scale, p_out, g_x, n, (float*)ws);
    else hipLaunchKernelGGL(sigmoid_bce_kernel<false>, dim3(parts), dim3(T), 0, st, x, labels, g_scale, p_out, g_x, n, (float*)ws);
    hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(T), 0, st, (const float*)ws, parts, loss);
    return gmp::check_launch("sigmoid_bce kernels");
}
extern "C" int gmp_sigmoid_bce_sum_fwd_bwd(const float* x, const float* labels, int64_t n, const float* g_scale, float* loss, float* p_out,
                                          float* g_x, void* ws, size_t ws_bytes, gmp_stream_t stream) {
    return sigmoid_bce_launch(false, x, labels, n, g_scale, loss, p_out, g_x, ws, ws_bytes, stream);
}
extern "C" int gmp_sigmoid_bce_signed_sum_fwd_bwd(const float* x, const float* signed_weight, int64_t n, const float* g_scale, float* loss,
                                                 float* p_out, float* g_x, void* ws, size_t ws_bytes, gmp_stream_t stream) {
    return sigmoid_bce_launch(true, x, signed_weight, n, g_scale, loss, p_out, g_x, ws, ws_bytes, stream);
}
extern "C" int gmp_lp_pair_sigmoid_bce_fwd_bwd(const float* y2, const float* sign, const int32_t* pos, int64_t rows, const float* g_scale, float* loss,
                                               float* p_out, float* g_y2, void* ws, size_t ws_bytes, gmp_stream_t stream) {
    const int64_t n = rows;
    GMP_CHECK_N("lp_pair_sigmoid_bce_fwd_bwd")
    if (!loss || !g_scale || !ws || ws_bytes < MAX_PARTS * sizeof(float) || (n > 0 && (!y2 || !sign || !pos || !g_y2)))
        return gmp::fail(GMP_ERR_ARG, "lp_pair_sigmoid_bce_fwd_bwd: bad argument");
    const int parts = parts_for(2 * n);
    hipLaunchKernelGGL(sigmoid_bce_pair_kernel, dim3(parts), dim3(T), 0, st, y2, sign, pos, g_scale, p_out, g_y2, n, (float*)ws);
    hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(T), 0, st, (const float*)ws, parts, loss);
    return gmp::check_launch("lp_pair_sigmoid_bce kernels");
}
extern "C" int gmp_cross_entropy_sum_fwd(const float* logits, const int64_t* target, int64_t M, int C, float* loss, void* ws,
                                         size_t ws_bytes, gmp_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    if (M < 0 || C < 1) return gmp::fail(GMP_ERR_ARG, "cross_entropy_sum_fwd: M=%lld C=%d", (long long)M, C);
    if (!loss || !ws || ws_bytes < gmp_loss_workspace_bytes(M) || (M > 0 && (!logits || !target))) return gmp::fail(GMP_ERR_ARG, "cross_entropy_sum_fwd: bad argument");
    if (M > 0) hipLaunchKernelGGL(ce_rows_kernel, dim3((unsigned)((M * 64 + T - 1) / T)), dim3(T), 0, st, logits, target, M, C, (float*)ws, (float*)nullptr, (const float*)nullptr);
    hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(T), 0, st, (const float*)ws, (int)M, loss);
    return gmp::check_launch("cross entropy kernels");
}
extern "C" int gmp_cross_entropy_sum_bwd(const float* logits, const int64_t* target, int64_t M, int C, const float* g_scale,
                                         float* g_logits, gmp_stream_t stream) {
    if (M < 0 || C < 1) return gmp::fail(GMP_ERR_ARG, "cross_entropy_sum_bwd: M=%lld C=%d", (long long)M, C);
    if (M == 0) return GMP_OK;
    if (!logits || !target || !g_scale || !g_logits) return gmp::fail(GMP_ERR_ARG, "cross_entropy_sum_bwd: null pointer");
    hipLaunchKernelGGL(ce_rows_kernel, dim3((unsigned)((M * 64 + T - 1) / T)), dim3(T), 0, (hipStream_t)stream, logits, target, M, C,
                       (float*)nullptr, g_logits, g_scale);
    return gmp::check_launch("ce_rows_kernel bwd");
}
extern "C" int gmp_row_fill(float* dst, const int64_t* idx, const float* src, int64_t M, int64_t num_dst_rows, int feat, int broadcast,
                            gmp_stream_t stream) {
    if (M < 0 || num_dst_rows < 0 || feat <= 0 || feat % 4) return gmp::fail(GMP_ERR_ARG, "row_fill: bad size");
    if (M == 0) return GMP_OK;
    if (!dst || !idx || !src) return gmp::fail(GMP_ERR_ARG, "row_fill: null pointer");
    int blocks = (int)std::min<int64_t>((M + 3) / 4, 2048);
    hipLaunchKernelGGL(row_fill_kernel, dim3(blocks), dim3(T), 0, (hipStream_t)stream, (float4*)dst, idx, (const float4*)src, M,
                       num_dst_rows, feat / 4, broadcast);
    return gmp::check_launch("row_fill_kernel");
}
