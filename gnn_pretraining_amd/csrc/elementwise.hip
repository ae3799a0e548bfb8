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
