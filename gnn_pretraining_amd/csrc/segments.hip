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

// Host -> device upload of a few small arrays by ONE kernel that reads the pinned host buffers directly (they are mapped into
// the device's address space): a step's index arrays are ~0.3 MB in four pieces, and four hipMemcpyAsync calls on the compute
// stream meant four hand-overs to the copy engine with the queue parked behind each (parked queues tax every running queue,
// streams.hip).  16-byte granules, 64 workgroups: ~0.25 MB of PCIe reads in flight.
constexpr int UP_MAX = 4;
struct UploadArgs {
    const uint4* src[UP_MAX];
    uint4* dst[UP_MAX];
    int64_t pre[UP_MAX + 1];     // exclusive prefix of the piece lengths, in 16-byte granules
    int n;
};
__global__ __launch_bounds__(THREADS) void upload_kernel(const UploadArgs a) {
    const int64_t total = a.pre[a.n];
    for (int64_t q = (int64_t)blockIdx.x * THREADS + threadIdx.x; q < total; q += (int64_t)gridDim.x * THREADS) {
        int s = 0;
#pragma unroll
        for (int i = 1; i < UP_MAX; ++i)
            if (i < a.n && q >= a.pre[i]) s = i;
        a.dst[s][q - a.pre[s]] = a.src[s][q - a.pre[s]];
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

extern "C" int gmp_upload(int n, const void* const* src_pinned_host, void* const* dst, const int64_t* bytes, gmp_stream_t stream) {
    if (n < 1 || n > UP_MAX || !src_pinned_host || !dst || !bytes) return gmp::fail(GMP_ERR_ARG, "upload: bad argument (n=%d, max %d)", n, UP_MAX);
    UploadArgs a;
    a.n = n;
    a.pre[0] = 0;
    for (int i = 0; i < UP_MAX; ++i) {
        a.src[i] = nullptr;
        a.dst[i] = nullptr;
        if (i < n) {
            if (bytes[i] < 0 || (bytes[i] % 16) || (bytes[i] && (!src_pinned_host[i] || !dst[i])) || ((uintptr_t)src_pinned_host[i] % 16) ||
                ((uintptr_t)dst[i] % 16))
                return gmp::fail(GMP_ERR_ARG, "upload: piece %d: %lld bytes / pointers must be multiples of 16", i, (long long)bytes[i]);
            a.src[i] = (const uint4*)src_pinned_host[i];
            a.dst[i] = (uint4*)dst[i];
            a.pre[i + 1] = a.pre[i] + bytes[i] / 16;
        } else {
            a.pre[i + 1] = a.pre[i];
        }
    }
    if (a.pre[n] == 0) return GMP_OK;
    const int blocks = (int)std::min<int64_t>(64, (a.pre[n] + THREADS - 1) / THREADS);
    hipLaunchKernelGGL(upload_kernel, dim3(blocks), dim3(THREADS), 0, (hipStream_t)stream, a);
    return gmp::check_launch("upload_kernel");
}
