// COO (int64) -> CSR (int32) in both orientations, deterministic (stable by edge id).
//
// Small graphs (the pre-training batches: N <= 16384 rows) are built by ONE
// workgroup per orientation entirely in LDS: histogram, scan, fill, per-row
// ordering -- one launch instead of seven.  Larger graphs (Cora upward, the
// roofline ladder) use the multi-kernel path.  Both give identical arrays.
#include <algorithm>

#include "gnnmp_internal.h"

namespace {

constexpr int SMALL_MAX_N = 16384;   // 2 * 4 B * N <= 128 KiB of the 160 KiB LDS
constexpr int SMALL_THREADS = 1024;
constexpr int THREAD_SORT_MAX_DEG = 32;

// rank-sort one row: out[start + rank(v)] = v  (edge ids within a row are distinct)
__device__ __forceinline__ void sort_row_serial(const int* __restrict__ tmp, int start, int deg,
                                                const int64_t* __restrict__ other, int* __restrict__ perm,
                                                int* __restrict__ col) {
    for (int i = 0; i < deg; ++i) {
        int v = tmp[start + i];
        int rank = 0;
        for (int j = 0; j < deg; ++j) rank += (tmp[start + j] < v);
        perm[start + rank] = v;
        col[start + rank] = (int)other[v];
    }
}

__device__ __forceinline__ void sort_row_wave(const int* __restrict__ tmp, int start, int deg, int lane,
                                              const int64_t* __restrict__ other, int* __restrict__ perm,
                                              int* __restrict__ col) {
    for (int i = lane; i < deg; i += GMP_WAVE) {
        int v = tmp[start + i];
        int rank = 0;
        for (int j = 0; j < deg; ++j) rank += (tmp[start + j] < v);
        perm[start + rank] = v;
        col[start + rank] = (int)other[v];
    }
}

// ------------------------------------------------------------------ small path
// TMP_LDS: the unsorted slot array also lives in LDS (2N + E + 1 ints fit), so the per-row ordering pass -- deg^2 dependent
// reads per row -- runs at LDS latency instead of L2 latency (4x faster end to end on a 6,400-row / 21k-edge step batch)
template <bool TMP_LDS>
__global__ __launch_bounds__(SMALL_THREADS) void csr_small_kernel(
    const int64_t* __restrict__ ei, int N, int E, int* rowptr0, int* col0, int* perm0, int* rowptr1, int* col1,
    int* perm1, int* status, int* tmp_all) {
    extern __shared__ int lds[];
    __shared__ int part[SMALL_THREADS];
    const int o = blockIdx.x;  // 0: group by target (row 1 of edge_index), 1: group by source
    const int64_t* key = o == 0 ? ei + E : ei;
    const int64_t* other = o == 0 ? ei : ei + E;
    int* rowptr = o == 0 ? rowptr0 : rowptr1;
    int* col = o == 0 ? col0 : col1;
    int* perm = o == 0 ? perm0 : perm1;
    int* cnt = lds;            // [N+1]
    int* cur = lds + (N + 1);  // [N]
    int* tmp = TMP_LDS ? lds + (2 * N + 1) : tmp_all + (size_t)o * E;
    const int t = threadIdx.x;

    for (int i = t; i <= N; i += SMALL_THREADS) cnt[i] = 0;
    for (int i = t; i < N; i += SMALL_THREADS) cur[i] = 0;
    __syncthreads();
    // one workgroup walks all E edges twice: keep 8 independent (key, other) loads in flight per thread per trip, otherwise
    // each trip is one exposed global-memory latency (30 trips x 2 passes at the step's 30k edges)
    constexpr int U = 8;
    int bad = 0;
    for (int e0 = t; e0 < E; e0 += SMALL_THREADS * U) {
        int64_t k[U], v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + u * SMALL_THREADS;
            k[u] = e < E ? key[e] : 0;
            v[u] = e < E ? other[e] : 0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (e0 + u * SMALL_THREADS >= E) continue;
            if (k[u] >= 0 && k[u] < N && v[u] >= 0 && v[u] < N) atomicAdd(&cnt[(int)k[u]], 1);
            else ++bad;
        }
    }
    if (o == 0 && bad) atomicAdd(status, bad);
    __syncthreads();
    // exclusive scan of cnt[0..N]: contiguous chunk per thread, then scan of chunk sums
    const int chunk = (N + 1 + SMALL_THREADS - 1) / SMALL_THREADS;
    const int lo = min(t * chunk, N + 1), hi = min(lo + chunk, N + 1);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += cnt[i];
    part[t] = s;
    __syncthreads();
    for (int d = 1; d < SMALL_THREADS; d <<= 1) {  // Hillis-Steele inclusive scan
        int v = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = t == 0 ? 0 : part[t - 1];
    for (int i = lo; i < hi; ++i) {
        int c = cnt[i];
        cnt[i] = run;
        run += c;
    }
    __syncthreads();
    for (int i = t; i <= N; i += SMALL_THREADS) rowptr[i] = cnt[i];
    for (int e0 = t; e0 < E; e0 += SMALL_THREADS * U) {
        int64_t k[U], v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + u * SMALL_THREADS;
            k[u] = e < E ? key[e] : -1;
            v[u] = e < E ? other[e] : -1;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (k[u] >= 0 && k[u] < N && v[u] >= 0 && v[u] < N) tmp[cnt[(int)k[u]] + atomicAdd(&cur[(int)k[u]], 1)] = e0 + u * SMALL_THREADS;
    }
    __threadfence_block();
    __syncthreads();
    for (int r = t; r < N; r += SMALL_THREADS) {
        int start = cnt[r], deg = cnt[r + 1] - start;
        if (deg <= THREAD_SORT_MAX_DEG) sort_row_serial(tmp, start, deg, other, perm, col);
    }
    const int wave = t / GMP_WAVE, lane = t % GMP_WAVE;
    for (int r = wave; r < N; r += SMALL_THREADS / GMP_WAVE) {
        int start = cnt[r], deg = cnt[r + 1] - start;
        if (deg > THREAD_SORT_MAX_DEG) sort_row_wave(tmp, start, deg, lane, other, perm, col);
    }
}

// ------------------------------------------------------------------ block-diagonal path
// The stacked batch of a pre-training step is block diagonal by construction: segment s owns rows [seg_row[s], seg_row[s+1]) and
// the contiguous edge range [seg_edge[s], seg_edge[s+1]), and no edge leaves its segment.  One workgroup per (segment,
// orientation) then does what csr_small_kernel does for the whole batch -- on ~265 rows / ~1,000 edges instead of 7,400 / 30,000,
// all 2 x 28 of them at once -- and writes the same arrays (within a row slots stay in ascending edge id).
constexpr int SEG_THREADS = 256;
constexpr int SEG_MAX_ROWS = 8192, SEG_MAX_EDGES = 24576;       // 2 * rows + edges + 2 ints of LDS per workgroup <= 160 KiB

__global__ __launch_bounds__(SEG_THREADS) void csr_segmented_kernel(
    const int64_t* __restrict__ ei, int N, int E, const int* __restrict__ seg_row, const int* __restrict__ seg_edge, int* rowptr0,
    int* col0, int* perm0, int* rowptr1, int* col1, int* perm1, int* status) {
    extern __shared__ int lds[];
    __shared__ int part[SEG_THREADS];
    const int sgm = blockIdx.x, o = blockIdx.y;
    const int r0 = seg_row[sgm], nr = seg_row[sgm + 1] - r0, e0 = seg_edge[sgm], ne = seg_edge[sgm + 1] - e0;
    const int64_t* key = (o == 0 ? ei + E : ei) + e0;
    const int64_t* other = (o == 0 ? ei : ei + E) + e0;
    int* rowptr = o == 0 ? rowptr0 : rowptr1;
    int* col = o == 0 ? col0 : col1;
    int* perm = o == 0 ? perm0 : perm1;
    int* cnt = lds;                 // [nr + 1]
    int* cur = lds + (nr + 1);      // [nr]
    int* tmp = lds + (2 * nr + 1);  // [ne]
    const int t = threadIdx.x;
    for (int i = t; i <= nr; i += SEG_THREADS) cnt[i] = 0;
    for (int i = t; i < nr; i += SEG_THREADS) cur[i] = 0;
    __syncthreads();
    int bad = 0;
    for (int e = t; e < ne; e += SEG_THREADS) {
        const int64_t k = key[e] - r0, v = other[e] - r0;
        if (k >= 0 && k < nr && v >= 0 && v < nr) atomicAdd(&cnt[(int)k], 1);
        else ++bad;                                            // outside the segment (or the graph): dropped and counted
    }
    if (o == 0 && bad) atomicAdd(status, bad);
    __syncthreads();
    const int chunk = (nr + 1 + SEG_THREADS - 1) / SEG_THREADS;
    const int lo = min(t * chunk, nr + 1), hi = min(lo + chunk, nr + 1);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += cnt[i];
    part[t] = s;
    __syncthreads();
    for (int d = 1; d < SEG_THREADS; d <<= 1) {
        const int v = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = t == 0 ? 0 : part[t - 1];
    for (int i = lo; i < hi; ++i) {
        const int c = cnt[i];
        cnt[i] = run;
        run += c;
    }
    __syncthreads();
    // global row pointers: the segment's slots start at e0 (edges dropped as out of range leave a gap at the segment's end, which
    // the last row pointer of the segment covers the same way the whole-batch build does only when nothing is dropped; the
    // caller treats status != 0 as an error)
    for (int i = t; i < nr; i += SEG_THREADS) rowptr[r0 + i] = e0 + cnt[i];
    if (t == 0 && sgm == gridDim.x - 1) rowptr[N] = e0 + cnt[nr];
    for (int e = t; e < ne; e += SEG_THREADS) {
        const int64_t k = key[e] - r0, v = other[e] - r0;
        if (k >= 0 && k < nr && v >= 0 && v < nr) tmp[cnt[(int)k] + atomicAdd(&cur[(int)k], 1)] = e;
    }
    __syncthreads();
    for (int r = t; r < nr; r += SEG_THREADS) {
        const int start = cnt[r], deg = cnt[r + 1] - start;
        for (int i = 0; i < deg; ++i) {
            const int v = tmp[start + i];
            int rank = 0;
            for (int j = 0; j < deg; ++j) rank += (tmp[start + j] < v);
            perm[e0 + start + rank] = e0 + v;
            col[e0 + start + rank] = (int)other[v];
        }
    }
}

// ------------------------------------------------------------------ large path
__global__ void hist_kernel(const int64_t* __restrict__ key, const int64_t* __restrict__ other, int N, int64_t E,
                            int* cnt, int* status) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int bad = 0;
    for (; e < E; e += stride) {
        int64_t k = key[e], v = other[e];
        if (k >= 0 && k < N && v >= 0 && v < N) atomicAdd(&cnt[k], 1);
        else ++bad;
    }
    if (status && bad) atomicAdd(status, bad);
}

constexpr int SCAN_T = 256, SCAN_PER = 8, SCAN_BLOCK = SCAN_T * SCAN_PER;

// in-place exclusive scan of each 2048-element block; block totals -> sums
__global__ __launch_bounds__(SCAN_T) void scan_blocks_kernel(int* data, int64_t n, int* sums) {
    __shared__ int part[SCAN_T];
    int64_t base = (int64_t)blockIdx.x * SCAN_BLOCK + (int64_t)threadIdx.x * SCAN_PER;
    int v[SCAN_PER], s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_PER; ++i) {
        v[i] = base + i < n ? data[base + i] : 0;
        s += v[i];
    }
    part[threadIdx.x] = s;
    __syncthreads();
    for (int d = 1; d < SCAN_T; d <<= 1) {
        int x = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
        __syncthreads();
        part[threadIdx.x] += x;
        __syncthreads();
    }
    int run = threadIdx.x == 0 ? 0 : part[threadIdx.x - 1];
#pragma unroll
    for (int i = 0; i < SCAN_PER; ++i) {
        if (base + i < n) data[base + i] = run;
        run += v[i];
    }
    if (threadIdx.x == SCAN_T - 1) sums[blockIdx.x] = part[SCAN_T - 1];
}

// exclusive scan of the block totals by one workgroup (sequential over chunks)
__global__ __launch_bounds__(SCAN_T) void scan_sums_kernel(int* sums, int nb) {
    __shared__ int part[SCAN_T];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nb; base += SCAN_T) {
        int i = base + threadIdx.x;
        int v = i < nb ? sums[i] : 0;
        part[threadIdx.x] = v;
        __syncthreads();
        for (int d = 1; d < SCAN_T; d <<= 1) {
            int x = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
            __syncthreads();
            part[threadIdx.x] += x;
            __syncthreads();
        }
        int c = carry;
        if (i < nb) sums[i] = c + part[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 0) carry = c + part[SCAN_T - 1];
        __syncthreads();
    }
}

__global__ void scan_add_kernel(int* data, int64_t n, const int* sums) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) data[i] += sums[i / SCAN_BLOCK];
}

__global__ void fill_kernel(const int64_t* __restrict__ key, const int64_t* __restrict__ other, int N, int64_t E,
                            const int* __restrict__ rowptr, int* cur, int* tmp) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; e < E; e += stride) {
        int64_t k = key[e], v = other[e];
        if (k >= 0 && k < N && v >= 0 && v < N) tmp[rowptr[k] + atomicAdd(&cur[k], 1)] = (int)e;
    }
}

__global__ void sort_rows_thread_kernel(const int* __restrict__ rowptr, const int* __restrict__ tmp,
                                        const int64_t* __restrict__ other, int N, int* perm, int* col) {
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= N) return;
    int start = rowptr[r], deg = rowptr[r + 1] - start;
    if (deg <= THREAD_SORT_MAX_DEG) sort_row_serial(tmp, start, deg, other, perm, col);
}

__global__ void sort_rows_wave_kernel(const int* __restrict__ rowptr, const int* __restrict__ tmp,
                                      const int64_t* __restrict__ other, int N, int* perm, int* col) {
    int wave = (blockIdx.x * blockDim.x + threadIdx.x) / GMP_WAVE, lane = threadIdx.x % GMP_WAVE;
    int nw = gridDim.x * blockDim.x / GMP_WAVE;
    for (int r = wave; r < N; r += nw) {
        int start = rowptr[r], deg = rowptr[r + 1] - start;
        if (deg > THREAD_SORT_MAX_DEG) sort_row_wave(tmp, start, deg, lane, other, perm, col);
    }
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

extern "C" size_t gmp_csr_build_workspace_bytes(int64_t N, int64_t E) {
    if (N < 0 || E < 0) return 0;
    // tmp perm for two orientations + cursor + scan block sums
    return align256(2 * (size_t)E * 4) + align256((size_t)N * 4 + 4) +
           align256(((size_t)(N + 1) / SCAN_BLOCK + 2) * 4) + 256;
}

extern "C" int gmp_csr_build(const int64_t* ei, int64_t N, int64_t E, int32_t* rowptr, int32_t* col, int32_t* perm,
                             int32_t* rowptr_t, int32_t* col_t, int32_t* perm_t, int32_t* status, void* ws,
                             size_t ws_bytes, gmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (N < 0 || E < 0 || N > INT32_MAX - 1 || E > INT32_MAX)
        return gmp::fail(GMP_ERR_ARG, "csr_build: N=%lld E=%lld out of int32 range", (long long)N, (long long)E);
    if (!rowptr || !status || (E > 0 && (!ei || !col || !perm)))
        return gmp::fail(GMP_ERR_ARG, "csr_build: null pointer");
    const bool both = rowptr_t != nullptr;
    if (both && E > 0 && (!col_t || !perm_t)) return gmp::fail(GMP_ERR_ARG, "csr_build: partial transposed outputs");
    if (ws_bytes < gmp_csr_build_workspace_bytes(N, E))
        return gmp::fail(GMP_ERR_WORKSPACE, "csr_build: workspace %zu < %zu", ws_bytes,
                         gmp_csr_build_workspace_bytes(N, E));
    hipError_t herr = hipMemsetAsync(status, 0, sizeof(int), stream);
    if (herr != hipSuccess) return gmp::fail(GMP_ERR_LAUNCH, "csr_build: memset: %s", hipGetErrorString(herr));
    char* w = (char*)ws;
    int* tmp = (int*)w;
    w += align256(2 * (size_t)E * 4);
    int* cur = (int*)w;
    w += align256((size_t)N * 4 + 4);
    int* sums = (int*)w;

    if (N <= SMALL_MAX_N) {
        constexpr int LDS_INTS = (160 * 1024 - SMALL_THREADS * 4 - 256) / 4;    // 160 KiB minus the static scan array
        static std::atomic<uint64_t> attr_set{0};   // > 64 KiB of dynamic LDS needs the opt-in once per device (gnnmp_internal.h)
        if (!gmp::lds_attr_done(attr_set)) {
            (void)hipFuncSetAttribute((const void*)csr_small_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_INTS * 4);
            (void)hipFuncSetAttribute((const void*)csr_small_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_INTS * 4);
            gmp::lds_attr_mark(attr_set);
        }
        if (2 * N + 1 + E <= LDS_INTS) {
            hipLaunchKernelGGL(csr_small_kernel<true>, dim3(both ? 2 : 1), dim3(SMALL_THREADS), (size_t)(2 * N + 1 + E) * sizeof(int), stream,
                               ei, (int)N, (int)E, rowptr, col, perm, rowptr_t, col_t, perm_t, status, tmp);
        } else {
            hipLaunchKernelGGL(csr_small_kernel<false>, dim3(both ? 2 : 1), dim3(SMALL_THREADS), (size_t)(2 * N + 1) * sizeof(int), stream,
                               ei, (int)N, (int)E, rowptr, col, perm, rowptr_t, col_t, perm_t, status, tmp);
        }
        return gmp::check_launch("csr_small_kernel");
    }
    for (int o = 0; o < (both ? 2 : 1); ++o) {
        const int64_t* key = o == 0 ? ei + E : ei;
        const int64_t* other = o == 0 ? ei : ei + E;
        int* rp = o == 0 ? rowptr : rowptr_t;
        int* cl = o == 0 ? col : col_t;
        int* pm = o == 0 ? perm : perm_t;
        int* tp = tmp + (size_t)o * E;
        if (hipMemsetAsync(rp, 0, (size_t)(N + 1) * 4, stream) != hipSuccess || hipMemsetAsync(cur, 0, (size_t)N * 4, stream) != hipSuccess)
            return gmp::fail(GMP_ERR_LAUNCH, "csr_build: memset");
        int gb = (int)std::min<int64_t>(gmp::cdiv(E, 256), 4096);
        if (E > 0) hipLaunchKernelGGL(hist_kernel, dim3(gb), dim3(256), 0, stream, key, other, (int)N, E, rp,
                                      o == 0 ? status : (int*)nullptr);
        int nb = gmp::cdiv(N + 1, SCAN_BLOCK);
        hipLaunchKernelGGL(scan_blocks_kernel, dim3(nb), dim3(SCAN_T), 0, stream, rp, N + 1, sums);
        hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(SCAN_T), 0, stream, sums, nb);
        hipLaunchKernelGGL(scan_add_kernel, dim3(gmp::cdiv(N + 1, 256)), dim3(256), 0, stream, rp, N + 1, sums);
        if (E > 0) {
            hipLaunchKernelGGL(fill_kernel, dim3(gb), dim3(256), 0, stream, key, other, (int)N, E, rp, cur, tp);
            hipLaunchKernelGGL(sort_rows_thread_kernel, dim3(gmp::cdiv(N, 256)), dim3(256), 0, stream, rp, tp, other,
                               (int)N, pm, cl);
            int wb = (int)std::min<int64_t>(gmp::cdiv(N, 4), 2048);
            hipLaunchKernelGGL(sort_rows_wave_kernel, dim3(wb), dim3(256), 0, stream, rp, tp, other, (int)N, pm, cl);
        }
        int rc = gmp::check_launch("csr_build large path");
        if (rc) return rc;
    }
    return GMP_OK;
}

extern "C" int gmp_csr_build_segmented(const int64_t* edge_index, int64_t N, int64_t E, const int32_t* seg_row_ptr, const int32_t* seg_edge_ptr,
                                       int num_segments, int64_t max_seg_rows, int64_t max_seg_edges, int32_t* rowptr, int32_t* col,
                                       int32_t* perm, int32_t* rowptr_t, int32_t* col_t, int32_t* perm_t, int32_t* status,
                                       gmp_stream_t stream) {
    if (N < 0 || E < 0 || num_segments < 1 || max_seg_rows < 0 || max_seg_edges < 0)
        return gmp::fail(GMP_ERR_ARG, "csr_build_segmented: bad sizes");
    if (!rowptr || !col || !perm || !status || !seg_row_ptr || !seg_edge_ptr || (E > 0 && !edge_index))
        return gmp::fail(GMP_ERR_ARG, "csr_build_segmented: null pointer");
    const bool both = rowptr_t && col_t && perm_t;
    if (!both && (rowptr_t || col_t || perm_t)) return gmp::fail(GMP_ERR_ARG, "csr_build_segmented: transposed outputs must be all set or all NULL");
    if (max_seg_rows > SEG_MAX_ROWS || max_seg_edges > SEG_MAX_EDGES || 2 * max_seg_rows + max_seg_edges + 2 > (160 * 1024 - SEG_THREADS * 4 - 256) / 4)
        return gmp::fail(GMP_ERR_UNSUPPORTED, "csr_build_segmented: a segment of %lld rows / %lld edges does not fit one workgroup's LDS",
                         (long long)max_seg_rows, (long long)max_seg_edges);
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(status, 0, sizeof(int32_t), st) != hipSuccess) return gmp::fail(GMP_ERR_LAUNCH, "csr_build_segmented: memset");
    const size_t lds = (size_t)(2 * max_seg_rows + max_seg_edges + 2) * sizeof(int);
    static std::atomic<uint64_t> attr_set{0};            // once per device, for the largest segment the check above admits
    if (!gmp::lds_attr_done(attr_set)) {
        if (hipFuncSetAttribute((const void*)csr_segmented_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - SEG_THREADS * 4 - 256) != hipSuccess)
            return gmp::fail(GMP_ERR_LAUNCH, "csr_build_segmented: LDS attribute");
        gmp::lds_attr_mark(attr_set);
    }
    hipLaunchKernelGGL(csr_segmented_kernel, dim3((unsigned)num_segments, both ? 2 : 1), dim3(SEG_THREADS), lds, st, edge_index, (int)N, (int)E,
                       seg_row_ptr, seg_edge_ptr, rowptr, col, perm, rowptr_t, col_t, perm_t, status);
    return gmp::check_launch("csr_segmented_kernel");
}
