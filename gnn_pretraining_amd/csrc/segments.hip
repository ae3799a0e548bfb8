// Pack / unpack of a fixed list of slices of one buffer -- the data-parallel gradient exchange sends only the entries of the
// [tasks, params] per-task gradient matrix that carry a gradient (shared tensors once per task, every head once: 36 MB of the
// 73 MB for scheme s4) as ONE contiguous message; torch.cat + one copy_ per slice on the way back cost ~30 launches.
#include <algorithm>

#include "gnnmp_internal.h"

namespace {

constexpr int THREADS = 256;
constexpr int MAX_SEGS = 256;

// table: [n] source offsets, then [n+1] exclusive prefix of lengths in the packed buffer (device int64)
template <bool PACK>
__global__ __launch_bounds__(THREADS) void segments_kernel(float* __restrict__ base, float* __restrict__ packed,
                                                           const int64_t* __restrict__ table, int n, float scale) {
    __shared__ int64_t s_off[MAX_SEGS], s_pre[MAX_SEGS + 1];
    for (int i = threadIdx.x; i < n; i += THREADS) s_off[i] = table[i];
    for (int i = threadIdx.x; i <= n; i += THREADS) s_pre[i] = table[n + i];
    __syncthreads();
    const int64_t total = s_pre[n];
    for (int64_t q = ((int64_t)blockIdx.x * THREADS + threadIdx.x) * 4; q < total; q += (int64_t)gridDim.x * THREADS * 4) {
        int lo = 0, hi = n - 1;                       // segment holding packed position q (lengths are multiples of 4)
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (s_pre[mid] <= q) lo = mid; else hi = mid - 1;
        }
        float4* src = reinterpret_cast<float4*>(base + s_off[lo] + (q - s_pre[lo]));
        float4* pk = reinterpret_cast<float4*>(packed + q);
        if (PACK) {
            *pk = *src;
        } else {
            const float4 v = *pk;
            *src = make_float4(v.x * scale, v.y * scale, v.z * scale, v.w * scale);
        }
    }
}

int check(const char* who, const void* base, const void* packed, const int64_t* table, int n) {
    if (!base || !packed || !table || n < 1 || n > MAX_SEGS) return gmp::fail(GMP_ERR_ARG, "%s: bad argument (n=%d, max %d)", who, n, MAX_SEGS);
    return GMP_OK;
}

}  // namespace

extern "C" int gmp_segments_pack(const float* base, float* packed, const int64_t* table_dev, int n, int64_t total, gmp_stream_t stream) {
    if (int rc = check("segments_pack", base, packed, table_dev, n)) return rc;
    if (total <= 0) return GMP_OK;
    const int blocks = (int)std::min<int64_t>(2048, (total / 4 + THREADS - 1) / THREADS);
    hipLaunchKernelGGL(segments_kernel<true>, dim3(blocks), dim3(THREADS), 0, (hipStream_t)stream, (float*)base, packed, table_dev, n, 1.f);
    return gmp::check_launch("segments_pack");
}

extern "C" int gmp_segments_unpack(float* base, const float* packed, const int64_t* table_dev, int n, int64_t total, float scale,
                                   gmp_stream_t stream) {
    if (int rc = check("segments_unpack", base, packed, table_dev, n)) return rc;
    if (total <= 0) return GMP_OK;
    const int blocks = (int)std::min<int64_t>(2048, (total / 4 + THREADS - 1) / THREADS);
    hipLaunchKernelGGL(segments_kernel<false>, dim3(blocks), dim3(THREADS), 0, (hipStream_t)stream, base, (float*)packed, table_dev, n, scale);
    return gmp::check_launch("segments_unpack");
}
