// Hard-negative mining for link-prediction fine-tuning (reference src/finetune/finetune.py:45-75:
// LinkPredictionHardNegativeMiner.mine_hard_negatives_for_edges, the top-k half).
//
// The reference builds, per training batch, the dense n x n cosine-similarity matrix of the node embeddings, an n x n
// boolean mask of existing edges (both directions) + the diagonal, gathers the unmasked scores (7.3 M for Cora) with
// torch.where, and calls torch.topk on them.  Here:
//   1. rows are normalised (F.normalize, eps 1e-12)                                    one wave per row
//   2. S = Zn Zn^T through the fp32 MFMA GEMM of gemm_f32.hip                           29 MB for Cora: stays in L2/MALL
//   3. masked pairs are overwritten with -inf (one thread per edge / diagonal entry)
//   4. exact top-k by RADIX SELECT over a 64-bit key (orderable score bits << 32 | ~flat index), so ties are broken
//      by the lower flat index (i * n + j) and every key is distinct: 6 histogram passes (11-bit digits, LDS
//      histograms) find the k-th key, one pass gathers the k winners, a rank-by-counting kernel orders them.
// Every pass streams S once with float4 loads; nothing of size n^2 besides S itself is ever written.
#include <algorithm>

#include "gnnmp_internal.h"

namespace {

constexpr int THREADS = 256;
constexpr int BINS = 2048;        // 11-bit digits
constexpr int MAX_K = 4096;

struct SelectState {
    unsigned long long prefix;    // decided high bits of the k-th largest key
    unsigned long long k_rem;     // rank still to be resolved inside the current prefix bucket
    unsigned int hist[BINS];
    unsigned int out_count;
};

__device__ __forceinline__ uint32_t orderable(float v) {
    uint32_t u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float from_orderable(uint32_t u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}
__device__ __forceinline__ unsigned long long make_key(float v, uint32_t flat) {
    return ((unsigned long long)orderable(v) << 32) | (unsigned long long)(0xffffffffu - flat);
}

__global__ __launch_bounds__(THREADS) void hn_normalize_kernel(const float* __restrict__ z, int64_t n, int d, float* __restrict__ zn) {
    const int lane = threadIdx.x % 64;
    const int64_t row = ((int64_t)blockIdx.x * THREADS + threadIdx.x) / 64;
    if (row >= n) return;
    const float* src = z + row * d;
    float s = 0.f;
    for (int c = lane; c < d; c += 64) s += src[c] * src[c];
    s = gmp::wave_sum(s);
    const float nr = fmaxf(sqrtf(s), 1e-12f);
    for (int c = lane; c < d; c += 64) zn[row * d + c] = src[c] / nr;
}

__global__ __launch_bounds__(THREADS) void hn_mask_kernel(float* __restrict__ S, int64_t n, const int64_t* __restrict__ edges, int64_t E) {
    const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    const float ninf = -__builtin_inff();
    if (i < E) {
        const int64_t s = edges[i], d = edges[E + i];
        if (s >= 0 && s < n && d >= 0 && d < n) {
            S[s * n + d] = ninf;
            S[d * n + s] = ninf;
        }
    } else if (i < E + n) {
        const int64_t v = i - E;
        S[v * n + v] = ninf;
    }
}

__global__ void hn_init_kernel(SelectState* st, unsigned long long k) {
    for (int i = threadIdx.x; i < BINS; i += blockDim.x) st->hist[i] = 0;
    if (threadIdx.x == 0) {
        st->prefix = 0;
        st->k_rem = k;
        st->out_count = 0;
    }
}

// histogram of digit [shift, shift+bits) over the keys whose higher bits equal the decided prefix
template <bool FIRST>
__global__ __launch_bounds__(THREADS) void hn_hist_kernel(const float* __restrict__ S, int64_t total, SelectState* __restrict__ st,
                                                          int shift, int bits) {
    __shared__ unsigned int h[BINS];
    for (int i = threadIdx.x; i < BINS; i += THREADS) h[i] = 0;
    __syncthreads();
    const unsigned long long prefix = st->prefix;
    const int hi = shift + bits;
    const unsigned int mask = (1u << bits) - 1u;
    const int64_t quads = total >> 2;
    for (int64_t q = (int64_t)blockIdx.x * THREADS + threadIdx.x; q < quads; q += (int64_t)gridDim.x * THREADS) {
        const float4 v = reinterpret_cast<const float4*>(S)[q];
        const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned long long key = make_key(e[j], (uint32_t)(q * 4 + j));
            if (FIRST || (key >> hi) == (prefix >> hi)) atomicAdd(&h[(unsigned int)(key >> shift) & mask], 1u);
        }
    }
    if (blockIdx.x == 0) {                                   // tail (total % 4 elements)
        for (int64_t i = (quads << 2) + threadIdx.x; i < total; i += THREADS) {
            const unsigned long long key = make_key(S[i], (uint32_t)i);
            if (FIRST || (key >> hi) == (prefix >> hi)) atomicAdd(&h[(unsigned int)(key >> shift) & mask], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < BINS; i += THREADS)
        if (h[i]) atomicAdd(&st->hist[i], h[i]);
}

// one block: walk the digit histogram from the top, find the bucket holding the k_rem-th largest key
__global__ __launch_bounds__(THREADS) void hn_pick_kernel(SelectState* st, int shift, int bits) {
    __shared__ unsigned int part[THREADS];
    const int nb = 1 << bits;
    const int per = (nb + THREADS - 1) / THREADS;
    unsigned int s = 0;
    for (int j = 0; j < per; ++j) {
        const int b = threadIdx.x * per + j;
        if (b < nb) s += st->hist[b];
    }
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long rem = st->k_rem;
        int t = THREADS - 1;
        for (; t > 0; --t) {
            if (part[t] >= rem) break;
            rem -= part[t];
        }
        int b = min(t * per + per - 1, nb - 1);
        for (; b > t * per; --b) {
            const unsigned int c = st->hist[b];
            if (c >= rem) break;
            rem -= c;
        }
        st->prefix |= (unsigned long long)b << shift;
        st->k_rem = rem;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < BINS; i += THREADS) st->hist[i] = 0;
}

// gather the keys >= threshold (exactly k of them: keys are distinct)
__global__ __launch_bounds__(THREADS) void hn_gather_kernel(const float* __restrict__ S, int64_t total, SelectState* __restrict__ st,
                                                            unsigned long long* __restrict__ keys, unsigned int cap) {
    const unsigned long long thr = st->prefix;
    const int64_t quads = total >> 2;
    for (int64_t q = (int64_t)blockIdx.x * THREADS + threadIdx.x; q < quads; q += (int64_t)gridDim.x * THREADS) {
        const float4 v = reinterpret_cast<const float4*>(S)[q];
        const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned long long key = make_key(e[j], (uint32_t)(q * 4 + j));
            if (key >= thr) {
                const unsigned int p = atomicAdd(&st->out_count, 1u);
                if (p < cap) keys[p] = key;
            }
        }
    }
    if (blockIdx.x == 0) {
        for (int64_t i = (quads << 2) + threadIdx.x; i < total; i += THREADS) {
            const unsigned long long key = make_key(S[i], (uint32_t)i);
            if (key >= thr) {
                const unsigned int p = atomicAdd(&st->out_count, 1u);
                if (p < cap) keys[p] = key;
            }
        }
    }
}

// order the k winners: rank = number of winners with a larger key (all distinct), then scatter
__global__ __launch_bounds__(THREADS) void hn_rank_kernel(const unsigned long long* __restrict__ keys, int k, int64_t n,
                                                          int64_t* __restrict__ out_edges, float* __restrict__ out_scores) {
    __shared__ unsigned long long tile[THREADS];
    const int i = blockIdx.x * THREADS + threadIdx.x;
    const unsigned long long mine = i < k ? keys[i] : 0ull;
    int rank = 0;
    for (int base = 0; base < k; base += THREADS) {
        __syncthreads();
        tile[threadIdx.x] = base + threadIdx.x < k ? keys[base + threadIdx.x] : 0ull;
        __syncthreads();
        const int lim = min(THREADS, k - base);
        for (int j = 0; j < lim; ++j) rank += tile[j] > mine;
    }
    if (i < k) {
        const uint32_t flat = 0xffffffffu - (uint32_t)(mine & 0xffffffffull);
        out_edges[rank] = (int64_t)(flat / (uint32_t)n);
        out_edges[k + rank] = (int64_t)(flat % (uint32_t)n);
        if (out_scores) out_scores[rank] = from_orderable((uint32_t)(mine >> 32));
    }
}

inline size_t al(size_t b) { return (b + 255) & ~(size_t)255; }

}  // namespace

extern "C" size_t gmp_hard_negative_workspace_bytes(int64_t n, int64_t d) {
    return al((size_t)n * d * 4) + al((size_t)n * n * 4) + al(sizeof(SelectState)) + al((size_t)MAX_K * 8) +
           gmp_gemm_f32_workspace_bytes(GMP_GEMM_NT, n, n, d) + 256;
}

extern "C" int gmp_hard_negative_topk(const float* emb, int64_t n, int64_t d, const int64_t* existing_edges, int64_t E,
                                      int64_t k, int64_t* out_edges, float* out_scores, float* scores_out,
                                      void* workspace, size_t workspace_bytes, gmp_stream_t stream) {
    if (n <= 0 || d <= 0 || E < 0 || k < 0) return gmp::fail(GMP_ERR_ARG, "hard_negative_topk: bad sizes n=%lld d=%lld E=%lld k=%lld", (long long)n, (long long)d, (long long)E, (long long)k);
    if (n > 65535) return gmp::fail(GMP_ERR_ARG, "hard_negative_topk: n=%lld > 65535 (flat pair index must fit 32 bits)", (long long)n);
    if (k > MAX_K) return gmp::fail(GMP_ERR_ARG, "hard_negative_topk: k=%lld > %d", (long long)k, MAX_K);
    if (!emb || (E && !existing_edges) || (k && !out_edges) || !workspace) return gmp::fail(GMP_ERR_ARG, "hard_negative_topk: null pointer");
    if (workspace_bytes < gmp_hard_negative_workspace_bytes(n, d)) return gmp::fail(GMP_ERR_ARG, "hard_negative_topk: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    char* p = (char*)workspace;
    float* zn = (float*)p; p += al((size_t)n * d * 4);
    float* S = (float*)p; p += al((size_t)n * n * 4);
    SelectState* sel = (SelectState*)p; p += al(sizeof(SelectState));
    unsigned long long* keys = (unsigned long long*)p; p += al((size_t)MAX_K * 8);
    void* gws = p;
    const size_t gws_bytes = gmp_gemm_f32_workspace_bytes(GMP_GEMM_NT, n, n, d);
    const int64_t total = n * n;

    hipLaunchKernelGGL(hn_normalize_kernel, dim3(gmp::cdiv(n * 64, THREADS)), dim3(THREADS), 0, st, emb, n, (int)d, zn);
    int rc = gmp_gemm_f32(GMP_GEMM_NT, zn, zn, nullptr, S, n, n, d, d, d, n, 1.f, 0, 0, gws, gws_bytes, stream);
    if (rc != GMP_OK) return rc;
    hipLaunchKernelGGL(hn_mask_kernel, dim3(gmp::cdiv(E + n, THREADS)), dim3(THREADS), 0, st, S, n, existing_edges, E);
    if (scores_out) {                                        // the masked similarity matrix, for tests / diagnostics
        hipError_t e = hipMemcpyAsync(scores_out, S, (size_t)total * 4, hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) return gmp::fail(GMP_ERR_LAUNCH, "hard_negative_topk: copy: %s", hipGetErrorString(e));
    }
    if (k == 0) return gmp::check_launch("hard_negative_topk");

    const int blocks = (int)std::min<int64_t>(2048, std::max<int64_t>(1, (total / 4 + THREADS - 1) / THREADS));
    hipLaunchKernelGGL(hn_init_kernel, dim3(1), dim3(THREADS), 0, st, sel, (unsigned long long)k);
    const int shifts[6] = {53, 42, 31, 20, 9, 0};
    const int nbits[6] = {11, 11, 11, 11, 11, 9};
    for (int pass = 0; pass < 6; ++pass) {
        if (pass == 0)
            hipLaunchKernelGGL(hn_hist_kernel<true>, dim3(blocks), dim3(THREADS), 0, st, S, total, sel, shifts[pass], nbits[pass]);
        else
            hipLaunchKernelGGL(hn_hist_kernel<false>, dim3(blocks), dim3(THREADS), 0, st, S, total, sel, shifts[pass], nbits[pass]);
        hipLaunchKernelGGL(hn_pick_kernel, dim3(1), dim3(THREADS), 0, st, sel, shifts[pass], nbits[pass]);
    }
    hipLaunchKernelGGL(hn_gather_kernel, dim3(blocks), dim3(THREADS), 0, st, S, total, sel, keys, (unsigned int)MAX_K);
    hipLaunchKernelGGL(hn_rank_kernel, dim3(gmp::cdiv(k, THREADS)), dim3(THREADS), 0, st, keys, (int)k, n, out_edges, out_scores);
    return gmp::check_launch("hard_negative_topk");
}
