// Row gathers, max read-out pooling and the link-prediction edge-feature builder.
// All are HBM/L2-bound row movers: one wave per output row, float4 per lane, so a
// 256-wide row is one 1-KiB coalesced wave instruction.
#include "gnnmp_internal.h"

namespace {

constexpr int BLOCK = 256;
constexpr int WPB = BLOCK / GMP_WAVE;

__device__ __forceinline__ int64_t wave_id() { return ((int64_t)blockIdx.x * BLOCK + threadIdx.x) / GMP_WAVE; }
__device__ __forceinline__ int64_t wave_count() { return (int64_t)gridDim.x * WPB; }

__global__ __launch_bounds__(BLOCK) void row_gather_kernel(const float4* __restrict__ src, const int64_t* __restrict__ idx,
                                                           const int* __restrict__ seg_ptr, float4* __restrict__ out,
                                                           int64_t M, int64_t nsrc, int F4) {
    const int lane = threadIdx.x % GMP_WAVE;
    for (int64_t m = wave_id(); m < M; m += wave_count()) {
        const int64_t r = idx[m];
        const bool ok = r >= 0 && r < nsrc;   // out-of-range index -> zero row, never a fault
        float s = 1.f;
        if (seg_ptr && ok) {
            int c = seg_ptr[r + 1] - seg_ptr[r];
            s = 1.f / (float)(c > 1 ? c : 1);
        }
        for (int c = lane; c < F4; c += GMP_WAVE) {
            float4 v = ok ? src[r * F4 + c] : make_float4(0.f, 0.f, 0.f, 0.f);
            out[m * F4 + c] = make_float4(v.x * s, v.y * s, v.z * s, v.w * s);
        }
    }
}

// any feature width (e.g. 7 class logits): one thread per element
__global__ __launch_bounds__(BLOCK) void row_gather_scalar_kernel(const float* __restrict__ src, const int64_t* __restrict__ idx,
                                                                  const int* __restrict__ seg_ptr, float* __restrict__ out,
                                                                  int64_t M, int64_t nsrc, int F) {
    const int64_t total = M * F;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * BLOCK) {
        const int64_t m = i / F, r = idx[m];
        float v = 0.f;
        if (r >= 0 && r < nsrc) {
            v = src[r * F + i % F];
            if (seg_ptr) {
                int c = seg_ptr[r + 1] - seg_ptr[r];
                v /= (float)(c > 1 ? c : 1);
            }
        }
        out[i] = v;
    }
}

// One workgroup per segment: its four waves take rows s + w, s + w + 4, ... and keep four row loads in flight each (a wave per
// segment walking its ~30 rows one dependent load at a time was latency-bound: 12 us forward, 33 us backward for 64 graphs).
// max is order-independent, so the result is bit-identical to the sequential sweep.
__device__ __forceinline__ float4 max4(float4 a, float4 b) {
    return make_float4(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w));
}

__global__ __launch_bounds__(BLOCK) void seg_max_fwd_kernel(const float4* __restrict__ x, const int* __restrict__ ptr,
                                                            float4* __restrict__ out, int64_t B, int F4) {
    __shared__ float4 sh[WPB][GMP_WAVE];
    const int lane = threadIdx.x % GMP_WAVE, w = threadIdx.x / GMP_WAVE;
    for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
        const int s = ptr[b], e = ptr[b + 1];
        for (int c0 = 0; c0 < F4; c0 += GMP_WAVE) {
            const int c = c0 + lane;
            const float ninf = -__builtin_huge_valf();
            float4 m = make_float4(ninf, ninf, ninf, ninf);
            if (c < F4) {
                int r = s + w;
                for (; r + 3 * WPB < e; r += 4 * WPB) {
                    const float4 v0 = x[(int64_t)r * F4 + c], v1 = x[(int64_t)(r + WPB) * F4 + c], v2 = x[(int64_t)(r + 2 * WPB) * F4 + c],
                                 v3 = x[(int64_t)(r + 3 * WPB) * F4 + c];
                    m = max4(max4(m, v0), max4(max4(v1, v2), v3));
                }
                for (; r < e; r += WPB) m = max4(m, x[(int64_t)r * F4 + c]);
            }
            sh[w][lane] = m;
            __syncthreads();
            if (w == 0 && c < F4) {
                m = max4(max4(sh[0][lane], sh[1][lane]), max4(sh[2][lane], sh[3][lane]));
                if (e <= s) m = make_float4(0.f, 0.f, 0.f, 0.f);   // empty segment -> 0 (PyG new_zeros, include_self=False)
                out[b * F4 + c] = m;
            }
            __syncthreads();
        }
    }
}

// torch scatter_reduce('amax') backward: gradient split evenly between tied maxima.
// Quirk reproduced on purpose: PyG calls it on a zero-initialised output with
// include_self=False, and torch still counts that initial 0 as one more tie when the
// segment maximum equals 0 (N_to_distribute = (self == result) + #ties).  After ReLU
// whole columns are 0, so this matters: such a column gets g / (n + 1), not g / n.
__device__ __forceinline__ float4 eq4(float4 v, float4 m) { return make_float4(v.x == m.x, v.y == m.y, v.z == m.z, v.w == m.w); }
__device__ __forceinline__ float4 add4f(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

__global__ __launch_bounds__(BLOCK) void seg_max_bwd_kernel(const float4* __restrict__ g, const float4* __restrict__ x,
                                                            const float4* __restrict__ mx, const int* __restrict__ ptr,
                                                            float4* __restrict__ gx, int64_t B, int F4, int accumulate) {
    __shared__ float4 sh[WPB][GMP_WAVE];
    const int lane = threadIdx.x % GMP_WAVE, w = threadIdx.x / GMP_WAVE;
    for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
        const int s = ptr[b], e = ptr[b + 1];
        for (int c0 = 0; c0 < F4; c0 += GMP_WAVE) {
            const int c = c0 + lane;
            float4 m = make_float4(0.f, 0.f, 0.f, 0.f), gg = m, n = m;    // tie counts are small integers: exact in any order
            if (c < F4) {
                m = mx[b * F4 + c];
                gg = g[b * F4 + c];
                int r = s + w;
                for (; r + 3 * WPB < e; r += 4 * WPB) {
                    const float4 v0 = x[(int64_t)r * F4 + c], v1 = x[(int64_t)(r + WPB) * F4 + c], v2 = x[(int64_t)(r + 2 * WPB) * F4 + c],
                                 v3 = x[(int64_t)(r + 3 * WPB) * F4 + c];
                    n = add4f(add4f(n, eq4(v0, m)), add4f(add4f(eq4(v1, m), eq4(v2, m)), eq4(v3, m)));
                }
                for (; r < e; r += WPB) n = add4f(n, eq4(x[(int64_t)r * F4 + c], m));
            }
            sh[w][lane] = n;
            __syncthreads();
            if (c < F4) {
                n = add4f(add4f(sh[0][lane], sh[1][lane]), add4f(sh[2][lane], sh[3][lane]));
                n = add4f(n, eq4(m, make_float4(0.f, 0.f, 0.f, 0.f)));     // the zero-initialised output counts as one more tie
                const float4 q = make_float4(gg.x / n.x, gg.y / n.y, gg.z / n.z, gg.w / n.w);
                for (int r = s + w; r < e; r += 4 * WPB) {
                    float4 v[4], p[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (r + i * WPB < e) {
                            v[i] = x[(int64_t)(r + i * WPB) * F4 + c];
                            if (accumulate) p[i] = gx[(int64_t)(r + i * WPB) * F4 + c];
                        }
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (r + i * WPB < e) {
                            float4 o = make_float4(v[i].x == m.x ? q.x : 0.f, v[i].y == m.y ? q.y : 0.f, v[i].z == m.z ? q.z : 0.f,
                                                   v[i].w == m.w ? q.w : 0.f);
                            if (accumulate) o = add4f(o, p[i]);
                            gx[(int64_t)(r + i * WPB) * F4 + c] = o;
                        }
                }
            }
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(BLOCK) void lp_feat_fwd_kernel(const float4* __restrict__ h, const int64_t* __restrict__ edges,
                                                            float4* __restrict__ feat, int64_t N, int64_t K, int F4) {
    const int lane = threadIdx.x % GMP_WAVE;
    for (int64_t k = wave_id(); k < K; k += wave_count()) {
        const int64_t a = edges[k], b = edges[K + k];
        const bool ok = a >= 0 && a < N && b >= 0 && b < N;
        for (int c = lane; c < F4; c += GMP_WAVE) {
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f), d = s;
            if (ok) { s = h[a * F4 + c]; d = h[b * F4 + c]; }
            float4* o = feat + k * 3 * F4;
            o[c] = make_float4(s.x + d.x, s.y + d.y, s.z + d.z, s.w + d.w);
            o[F4 + c] = make_float4(s.x * d.x, s.y * d.y, s.z * d.z, s.w * d.w);
            o[2 * F4 + c] = make_float4(fabsf(s.x - d.x), fabsf(s.y - d.y), fabsf(s.z - d.z), fabsf(s.w - d.w));
        }
    }
}

__device__ __forceinline__ float sgn(float v) { return (float)((v > 0.f) - (v < 0.f)); }   // torch.abs' subgradient: 0 at 0

__global__ __launch_bounds__(BLOCK) void lp_feat_bwd_kernel(const float4* __restrict__ gf, const float4* __restrict__ h,
                                                            const int64_t* __restrict__ edges, float4* __restrict__ ghs,
                                                            float4* __restrict__ ghd, int64_t N, int64_t K, int F4) {
    const int lane = threadIdx.x % GMP_WAVE;
    for (int64_t k = wave_id(); k < K; k += wave_count()) {
        const int64_t a = edges[k], b = edges[K + k];
        const bool ok = a >= 0 && a < N && b >= 0 && b < N;
        for (int c = lane; c < F4; c += GMP_WAVE) {
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f), d = s;
            if (ok) { s = h[a * F4 + c]; d = h[b * F4 + c]; }
            const float4* g = gf + k * 3 * F4;
            const float4 gs = g[c], gp = g[F4 + c], ga = g[2 * F4 + c];
            float4 t = make_float4(ga.x * sgn(s.x - d.x), ga.y * sgn(s.y - d.y), ga.z * sgn(s.z - d.z), ga.w * sgn(s.w - d.w));
            ghs[k * F4 + c] = make_float4(gs.x + gp.x * d.x + t.x, gs.y + gp.y * d.y + t.y, gs.z + gp.z * d.z + t.z, gs.w + gp.w * d.w + t.w);
            ghd[k * F4 + c] = make_float4(gs.x + gp.x * s.x - t.x, gs.y + gp.y * s.y - t.y, gs.z + gp.z * s.z - t.z, gs.w + gp.w * s.w - t.w);
        }
    }
}

int grid_for(int64_t rows) {
    int64_t b = (rows + WPB - 1) / WPB;
    if (b < 1) b = 1;
    if (b > 8192) b = 8192;
    return (int)b;
}

int feat_ok(const char* who, int feat) {
    if (feat <= 0 || feat % 4) return gmp::fail(GMP_ERR_ARG, "%s: feature width %d must be a positive multiple of 4", who, feat);
    return GMP_OK;
}

}  // namespace

extern "C" int gmp_row_gather(const float* src, const int64_t* idx, const int32_t* seg_ptr, float* out, int64_t M,
                              int64_t num_src_rows, int feat, gmp_stream_t stream) {
    if (feat <= 0) return gmp::fail(GMP_ERR_ARG, "row_gather: feature width %d", feat);
    if (M < 0 || num_src_rows < 0 || (M > 0 && (!src || !idx || !out))) return gmp::fail(GMP_ERR_ARG, "row_gather: bad argument");
    if (M == 0) return GMP_OK;
    if (feat % 4) {
        int64_t b = (M * feat + BLOCK - 1) / BLOCK;
        hipLaunchKernelGGL(row_gather_scalar_kernel, dim3((unsigned)(b > 4096 ? 4096 : b)), dim3(BLOCK), 0, (hipStream_t)stream, src,
                           idx, seg_ptr, out, M, num_src_rows, feat);
        return gmp::check_launch("row_gather_scalar_kernel");
    }
    hipLaunchKernelGGL(row_gather_kernel, dim3(grid_for(M)), dim3(BLOCK), 0, (hipStream_t)stream, (const float4*)src, idx,
                       seg_ptr, (float4*)out, M, num_src_rows, feat / 4);
    return gmp::check_launch("row_gather_kernel");
}

extern "C" int gmp_segment_max_fwd(const float* x, const int32_t* ptr, float* out, int64_t B, int feat, gmp_stream_t stream) {
    if (int rc = feat_ok("segment_max_fwd", feat)) return rc;
    if (B < 0 || (B > 0 && (!x || !ptr || !out))) return gmp::fail(GMP_ERR_ARG, "segment_max_fwd: bad argument");
    if (B == 0) return GMP_OK;
    hipLaunchKernelGGL(seg_max_fwd_kernel, dim3((unsigned)(B < 65536 ? B : 65536)), dim3(BLOCK), 0, (hipStream_t)stream, (const float4*)x, ptr,
                       (float4*)out, B, feat / 4);
    return gmp::check_launch("seg_max_fwd_kernel");
}

extern "C" int gmp_segment_max_bwd(const float* g_out, const float* x, const float* out, const int32_t* ptr, float* g_x,
                                   int64_t B, int feat, int accumulate, gmp_stream_t stream) {
    if (int rc = feat_ok("segment_max_bwd", feat)) return rc;
    if (B < 0 || (B > 0 && (!g_out || !x || !out || !ptr || !g_x))) return gmp::fail(GMP_ERR_ARG, "segment_max_bwd: bad argument");
    if (B == 0) return GMP_OK;
    hipLaunchKernelGGL(seg_max_bwd_kernel, dim3((unsigned)(B < 65536 ? B : 65536)), dim3(BLOCK), 0, (hipStream_t)stream, (const float4*)g_out,
                       (const float4*)x, (const float4*)out, ptr, (float4*)g_x, B, feat / 4, accumulate);
    return gmp::check_launch("seg_max_bwd_kernel");
}

extern "C" int gmp_lp_edge_features_fwd(const float* h, const int64_t* edges, float* feat, int64_t N, int64_t K, int F,
                                        gmp_stream_t stream) {
    if (int rc = feat_ok("lp_edge_features_fwd", F)) return rc;
    if (N < 0 || K < 0 || (K > 0 && (!h || !edges || !feat))) return gmp::fail(GMP_ERR_ARG, "lp_edge_features_fwd: bad argument");
    if (K == 0) return GMP_OK;
    hipLaunchKernelGGL(lp_feat_fwd_kernel, dim3(grid_for(K)), dim3(BLOCK), 0, (hipStream_t)stream, (const float4*)h, edges,
                       (float4*)feat, N, K, F / 4);
    return gmp::check_launch("lp_feat_fwd_kernel");
}

extern "C" int gmp_lp_edge_features_bwd(const float* g_feat, const float* h, const int64_t* edges, float* g_hs, float* g_hd,
                                        int64_t N, int64_t K, int F, gmp_stream_t stream) {
    if (int rc = feat_ok("lp_edge_features_bwd", F)) return rc;
    if (N < 0 || K < 0 || (K > 0 && (!g_feat || !h || !edges || !g_hs || !g_hd)))
        return gmp::fail(GMP_ERR_ARG, "lp_edge_features_bwd: bad argument");
    if (K == 0) return GMP_OK;
    hipLaunchKernelGGL(lp_feat_bwd_kernel, dim3(grid_for(K)), dim3(BLOCK), 0, (hipStream_t)stream, (const float4*)g_feat,
                       (const float4*)h, edges, (float4*)g_hs, (float4*)g_hd, N, K, F / 4);
    return gmp::check_launch("lp_feat_bwd_kernel");
}
