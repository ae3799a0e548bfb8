// Multi-tensor PCGrad + gradient clipping + AdamW over flat buffers: the tail of every optimisation
// step (reference src/pretrain/gradient_surgery.py:41-103, pretrain.py:152-153, optimizers.py:8-75).
//
// The reference walks ~62 shared tensors x 10 task pairs with three host syncs each (norm()==0,
// dot<0) -- about 1,900 device->host round trips per s4 step.  Here the whole thing is five launches:
//   gram     32 blocks per tensor + a finish pass: every <g_i, g_j> of that tensor (per-tensor, not whole-model,
//   solve    one thread per tensor: PCGrad's sequential projections solved in    exactly as the reference)
//            5x5 Gram space -> per-task mixing weights, plus the reference's
//            "which tensors get a gradient at all" rule (its _set_gradients quirk)
//   combine  final = sum_t w[k][t] * g_t, and the per-block sums of squares for the clip norm
//   norm     one block: total squared norm (fixed order)
//   adamw    clip coefficient + decoupled weight decay + Adam, per-tensor lr / step count
// Everything is deterministic (no atomics).
#include "gnnmp_internal.h"

namespace {

constexpr int MAXT = 8;        // tasks
constexpr int CH = 32;         // chunks per tensor in the elementwise kernels (the 131k-element weights dominate)
constexpr int GCH = 32;        // chunks per tensor in the Gram pass (the 131k-element weights need many blocks)
constexpr int TB = 256;

struct MtArgs {
    const float* tg;           // [T][stride] per-task gradients
    int64_t stride;
    int T, K;
    int k0, k1;                // the tensors this launch works on (gram / solve / combine); norm and AdamW always sweep all K
    const int64_t* off;        // [K]
    const int* len;            // [K]
    const unsigned char* has;  // [K][MAXT]
    double* gram;              // [K][MAXT][MAXT]
    double* gram_part;         // [K][GCH][MAXT][MAXT] chunk partials
    float* weights;            // [K][MAXT]
    int* flags;                // [K]
    float* steps;              // [K]
    int* metrics;              // [2] conflicts, projections
    int* block_metrics;        // [2 * solve blocks]
    int order[MAXT];           // shuffled task order
    int n_order;               // tasks taking part in PCGrad
    int last_task;             // last task in dict order (keeps its raw .grad where PCGrad emits nothing)
    int extra_task;            // -1, or a task whose gradient is ADDED afterwards (domain_adv in s5)
    float* final_grad;         // [P]
    float* partial;            // [K*CH]
    float* normsq;             // [1]
    float* params;
    float* exp_avg;
    float* exp_avg_sq;
    const float* lr;           // [K]
    const float* wd;           // [K]
    float beta1, beta2, eps, max_norm;
    const int* abort;          // nullable: nonzero at run time = something upstream went wrong (a cross-stream gate timed out): no update
};

// ---- which (tensor, chunk) a workgroup works on ------------------------------------------------------------------------------
// A tensor of len4 float4 is cut into count = ceil(len4 / per4) <= NCH chunks of per4 = max(ceil(len4 / NCH), MINP4) float4: the 131k-element
// weights into 32 chunks of 4,096 elements, the ~100 biases / BatchNorm vectors / small encoders into ONE.  The grids are one-dimensional
// and dense over those chunks (workgroup b -> the b-th chunk in tensor order, found by wave 0 with a prefix sum over the tensors' counts);
// round 3: the first form launched NCH workgroups for EVERY tensor -- 4,160 for the s4 model, ~3,700 of them with nothing or eight
// elements to do -- and the launches' time followed the workgroup count, not the bytes (Gram pass 25 us for 27 MB; 61 us with 64 chunks).
constexpr int MINP4 = 1024;
__device__ __forceinline__ int chunk_per4(int len4, int nch) { return max((len4 + nch - 1) / nch, MINP4); }
__device__ __forceinline__ int chunk_count(int len, int nch) {
    const int len4 = (len + 3) / 4;
    return len4 > 0 ? (len4 + chunk_per4(len4, nch) - 1) / chunk_per4(len4, nch) : 1;
}
// workgroup b of a launch over tensors [k0, k1): its tensor and chunk; false = b lies beyond the last chunk (the grid is an upper bound)
__device__ __forceinline__ bool locate_chunk(const MtArgs& a, int k0, int k1, int nch, int b, int* k_out, int* j_out) {
    __shared__ int s_kj[2];
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        if (lane == 0) s_kj[0] = -1;
        int base = 0;
        for (int t0 = k0; t0 < k1; t0 += 64) {
            const int k = t0 + lane;
            const int cnt = k < k1 ? chunk_count(a.len[k], nch) : 0;
            int incl = cnt;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int v = __shfl_up(incl, o, 64);
                if (lane >= o) incl += v;
            }
            const int excl = base + incl - cnt;
            if (cnt > 0 && b >= excl && b < excl + cnt) {
                s_kj[0] = k;
                s_kj[1] = b - excl;
            }
            base += __shfl(incl, 63, 64);
            if (base > b) break;                             // wave-uniform
        }
    }
    __syncthreads();
    *k_out = s_kj[0];
    *j_out = s_kj[1];
    return s_kj[0] >= 0;
}

// NT = number of tasks, a compile-time constant: the products form a fixed triangle with no run-time masks (a task that does not own the
// tensor loads zeros), the (up to) four float4 of every task are requested together -- one memory latency per block instead of four -- and a
// product's 64 lane partials (<= 16 elements each) fold by fp32 shuffles; waves and chunks are added in double, in a fixed order.  (Round 3:
// 25 us for the 27 MB of an s4 step before; profiles/README.md has the new figure.  The first form masked 36 products at run time inside a four-trip load loop and
// folded each product by six double-precision shuffle steps: nine and ten microseconds of the 25.)
template <int NT>
__global__ __launch_bounds__(TB) void gram_kernel(const MtArgs a) {
    int k, j;
    if (!locate_chunk(a, a.k0, a.k1, GCH, blockIdx.x, &k, &j)) return;
    const int len = a.len[k], per = 4 * chunk_per4((len + 3) / 4, GCH);
    bool has[NT];
    int nh = 0;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        has[t] = t < a.T && a.has[k * MAXT + t];
        nh += has[t];
    }
    if (nh < 2) return;                         // nothing to project against
    const int lo = j * per, hi = min(lo + per, (len + 3) / 4 * 4);     // slots are zero-padded to a multiple of 4 floats
    double* out = a.gram_part + ((int64_t)k * GCH + j) * (MAXT * MAXT);
    constexpr int NP = NT * (NT + 1) / 2, NW = TB / 64, IT = 4;
    float acc[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) acc[p] = 0.f;
    const int64_t off = a.off[k];
    for (int i0 = lo + 4 * threadIdx.x; i0 < hi; i0 += IT * 4 * TB) {     // one trip for the shared tensors (<= 4,096 elements per chunk)
        float4 g[IT][NT];
#pragma unroll
        for (int u = 0; u < IT; ++u) {
            const int i = i0 + u * 4 * TB;
#pragma unroll
            for (int t = 0; t < NT; ++t)
                g[u][t] = (has[t] && i < hi) ? *reinterpret_cast<const float4*>(a.tg + (int64_t)t * a.stride + off + i) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < IT; ++u) {
            int p = 0;
#pragma unroll
            for (int x = 0; x < NT; ++x)
#pragma unroll
                for (int y = x; y < NT; ++y, ++p) {
                    acc[p] += g[u][x].x * g[u][y].x;
                    acc[p] += g[u][x].y * g[u][y].y;
                    acc[p] += g[u][x].z * g[u][y].z;
                    acc[p] += g[u][x].w * g[u][y].w;
                }
        }
    }
    __shared__ double sh[NW][NP];
    const int lane = threadIdx.x % 64, wv = threadIdx.x / 64;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        float v = acc[p];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) sh[wv][p] = (double)v;
    }
    __syncthreads();
    if (threadIdx.x < NP) {
        int x = 0, rem = threadIdx.x;                      // (x, y) of upper-triangle slot threadIdx.x
        while (rem >= NT - x) { rem -= NT - x; ++x; }
        const int y = x + rem;
        if (a.has[k * MAXT + x] && a.has[k * MAXT + y]) {
            double v = sh[0][threadIdx.x];
#pragma unroll
            for (int w = 1; w < NW; ++w) v += sh[w][threadIdx.x];
            out[x * MAXT + y] = v;
        }
    }
}

// sum the chunk partials in chunk order into the symmetric Gram matrix of every tensor
__global__ __launch_bounds__(64) void gram_finish_kernel(MtArgs a) {
    const int k = a.k0 + blockIdx.x, x = threadIdx.x / MAXT, y = threadIdx.x % MAXT;
    if (x >= a.T || y >= a.T || x > y) return;
    if (!a.has[k * MAXT + x] || !a.has[k * MAXT + y]) return;
    double s = 0.0;
    const int cnt = chunk_count(a.len[k], GCH);
    double v[GCH];                              // all chunk partials requested together (slots past the tensor's chunk count are never written)
#pragma unroll
    for (int j = 0; j < GCH; ++j) v[j] = j < cnt ? a.gram_part[((int64_t)k * GCH + j) * (MAXT * MAXT) + x * MAXT + y] : 0.0;
#pragma unroll
    for (int j = 0; j < GCH; ++j) s += v[j];
    a.gram[((int64_t)k * MAXT + x) * MAXT + y] = s;
    a.gram[((int64_t)k * MAXT + y) * MAXT + x] = s;
}

__global__ __launch_bounds__(64) void solve_kernel(const MtArgs a) {
    __shared__ int s_conf[64], s_proj[64];
    int conf = 0, proj = 0;
    const int k = a.k0 + blockIdx.x * 64 + threadIdx.x;
    if (k < a.k1) {
        const unsigned char* has = a.has + k * MAXT;
        const double* G = a.gram + (int64_t)k * MAXT * MAXT;
        int flag = 0;
        float wout[MAXT];                      // indexed by ORDER POSITION (static after unrolling), scattered at the end
#pragma unroll
        for (int i = 0; i < MAXT; ++i) wout[i] = 0.f;
        // everything below works in order space: position i <-> task a.order[i]; arrays are indexed only by unrolled
        // loop counters, so they live in registers (task-indexed arrays would be runtime-indexed -> scratch)
        bool hp[MAXT];
        int nh = 0;
#pragma unroll
        for (int i = 0; i < MAXT; ++i) {
            hp[i] = i < a.n_order && has[a.order[i]];
            nh += hp[i];
        }
        if (hp[0]) {
            double Gp[MAXT][MAXT];
#pragma unroll
            for (int i = 0; i < MAXT; ++i)
#pragma unroll
                for (int j = 0; j < MAXT; ++j) Gp[i][j] = (hp[i] && hp[j]) ? G[a.order[i] * MAXT + a.order[j]] : 0.0;
            double alpha[MAXT][MAXT];          // g_i' = sum_j alpha[i][j] g_j   (positions)
#pragma unroll
            for (int i = 0; i < MAXT; ++i)
#pragma unroll
                for (int j = 0; j < MAXT; ++j) alpha[i][j] = i == j ? 1.0 : 0.0;
#pragma unroll
            for (int i = 1; i < MAXT; ++i) {
                if (!hp[i]) continue;
#pragma unroll
                for (int j = 0; j < i; ++j) {
                    if (!hp[j]) continue;
                    double ni = 0.0, dot = 0.0;
#pragma unroll
                    for (int p = 0; p <= i; ++p) {          // alpha[i][p] is zero beyond p = i
                        double row = 0.0;
#pragma unroll
                        for (int q = 0; q <= i; ++q) row += alpha[i][q] * Gp[p][q];
                        ni += alpha[i][p] * row;
                        dot += alpha[i][p] * Gp[p][j];
                    }
                    const double nj = Gp[j][j];
                    if (ni <= 0.0 || nj <= 0.0) continue;   // reference: norm() == 0 -> skip the pair
                    ++proj;
                    if (dot < 0.0) {
                        ++conf;
                        alpha[i][j] -= dot / nj;            // project off task j's ORIGINAL gradient
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < MAXT; ++i)
                if (hp[i]) {
#pragma unroll
                    for (int b = 0; b < MAXT; ++b) wout[b] += (float)(alpha[i][b] / nh);
                }
            flag = 1;
        }
        float* w = a.weights + k * MAXT;
        for (int t = 0; t < MAXT; ++t) w[t] = 0.f;
        if (flag) {
#pragma unroll
            for (int i = 0; i < MAXT; ++i)
                if (i < a.n_order) w[a.order[i]] = wout[i];
        } else if (a.last_task >= 0 && has[a.last_task]) {
            w[a.last_task] = 1.f;      // untouched by _set_gradients: keeps the last backward's .grad
            flag = 1;
        }
        if (a.extra_task >= 0 && has[a.extra_task]) {
            w[a.extra_task] += 1.f;    // domain_adv_loss.backward() accumulates on top (pretrain.py:149-150)
            flag = 1;
        }
        a.flags[k] = flag;
        if (a.steps && !(a.abort && *a.abort)) a.steps[k] += (float)flag;     // torch.optim keeps a per-parameter step that only advances with a gradient
    }
    s_conf[threadIdx.x] = conf;
    s_proj[threadIdx.x] = proj;
    __syncthreads();
    for (int d = 32; d > 0; d >>= 1) {
        if (threadIdx.x < d) {
            s_conf[threadIdx.x] += s_conf[threadIdx.x + d];
            s_proj[threadIdx.x] += s_proj[threadIdx.x + d];
        }
        __syncthreads();
    }
    // per-block counts, summed by norm_kernel: a range [k0, k1) owns the slots k0 .. k1-1 (its blocks fill the first ones, the
    // rest are zeroed), so ranges launched separately never share a slot and every slot is rewritten every step
    if (threadIdx.x == 0) {
        a.block_metrics[2 * (a.k0 + blockIdx.x)] = s_conf[0];
        a.block_metrics[2 * (a.k0 + blockIdx.x) + 1] = s_proj[0];
    }
    if (blockIdx.x == 0)
        for (int i = (int)gridDim.x + threadIdx.x; i < a.k1 - a.k0; i += 64) {
            a.block_metrics[2 * (a.k0 + i)] = 0;
            a.block_metrics[2 * (a.k0 + i) + 1] = 0;
        }
}

// Every tensor starts on a 16-byte boundary of the flat buffers and is padded to a multiple of 4 floats (zeros in params, in the
// gradients and in the optimizer state, and they stay zeros), so both sweeps below move float4.
__global__ __launch_bounds__(TB) void combine_kernel(MtArgs a) {
    __shared__ float sh[TB / 64];
    int k, j;
    if (!locate_chunk(a, a.k0, a.k1, CH, blockIdx.x, &k, &j)) return;
    float ss = 0.f;
    if (a.flags[k]) {
        float w[MAXT];
        for (int t = 0; t < MAXT; ++t) w[t] = a.weights[k * MAXT + t];
        const int len4 = (a.len[k] + 3) / 4, per = chunk_per4(len4, CH);
        const int lo = j * per, hi = min(lo + per, len4);
        const int64_t off = a.off[k];
        for (int i = lo + threadIdx.x; i < hi; i += TB) {
            float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int t = 0; t < a.T; ++t)
                if (w[t] != 0.f) {
                    const float4 v = *reinterpret_cast<const float4*>(a.tg + (int64_t)t * a.stride + off + 4 * (int64_t)i);
                    g = make_float4(fmaf(w[t], v.x, g.x), fmaf(w[t], v.y, g.y), fmaf(w[t], v.z, g.z), fmaf(w[t], v.w, g.w));
                }
            *reinterpret_cast<float4*>(a.final_grad + off + 4 * (int64_t)i) = g;
            ss = fmaf(g.x, g.x, ss); ss = fmaf(g.y, g.y, ss); ss = fmaf(g.z, g.z, ss); ss = fmaf(g.w, g.w, ss);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    if (threadIdx.x % 64 == 0) sh[threadIdx.x / 64] = ss;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = sh[0];
        for (int w = 1; w < TB / 64; ++w) s += sh[w];
        a.partial[k * CH + j] = s;
    }
}

// Data-parallel runs that SHARD PCGrad over the ranks (dist.ShardedGradSync): a rank runs gram / solve / combine only for the tensors it owns
// and receives every other tensor's combined gradient from its owner (all-gather into final_grad).  For those FOREIGN tensors [k0, k1) this
// pass leaves behind exactly what the owner's solve / combine left on the owner: the "has a gradient" flag and the step count (functions
// of the availability table and the task order alone, no Gram matrix needed), zeroed conflict / projection slots (the owner counted them),
// and the per-chunk sums of squares of the gradient now in place -- combine_kernel's own arithmetic, so the clip norm that follows is
// bit for bit the unsharded one.
__global__ __launch_bounds__(TB) void foreign_kernel(MtArgs a) {
    __shared__ float sh[TB / 64];
    int k, j;
    if (!locate_chunk(a, a.k0, a.k1, CH, blockIdx.x, &k, &j)) return;
    const unsigned char* has = a.has + k * MAXT;
    int flag = a.n_order > 0 && has[a.order[0]] ? 1 : 0;                    // PCGrad emits a gradient iff the first-shuffled task has the tensor
    if (!flag && a.last_task >= 0 && has[a.last_task]) flag = 1;            // ... else the last task's raw .grad stays (gradient_surgery.py:61)
    if (a.extra_task >= 0 && has[a.extra_task]) flag = 1;
    if (j == 0 && threadIdx.x == 0) {
        a.flags[k] = flag;
        if (a.steps && !(a.abort && *a.abort)) a.steps[k] += (float)flag;
        a.block_metrics[2 * k] = 0;
        a.block_metrics[2 * k + 1] = 0;
    }
    float ss = 0.f;
    if (flag) {
        const int len4 = (a.len[k] + 3) / 4, per = chunk_per4(len4, CH);
        const int lo = j * per, hi = min(lo + per, len4);
        const int64_t off = a.off[k];
        for (int i = lo + threadIdx.x; i < hi; i += TB) {
            const float4 g = *reinterpret_cast<const float4*>(a.final_grad + off + 4 * (int64_t)i);
            ss = fmaf(g.x, g.x, ss); ss = fmaf(g.y, g.y, ss); ss = fmaf(g.z, g.z, ss); ss = fmaf(g.w, g.w, ss);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    if (threadIdx.x % 64 == 0) sh[threadIdx.x / 64] = ss;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s2 = sh[0];
        for (int w = 1; w < TB / 64; ++w) s2 += sh[w];
        a.partial[k * CH + j] = s2;
    }
}

__global__ __launch_bounds__(TB) void norm_kernel(MtArgs a) {
    __shared__ double sh[TB];
    __shared__ int shc[TB], shp[TB];
    double s = 0.0;
    // the chunk sums of every tensor, a thread per tensor (only the chunks the tensor has: the other slots of its row are never written); the row's
    // CH slots are requested together (one dependent load after the other cost 8 of this launch's 12.6 us, r03a trace)
    for (int k = threadIdx.x; k < a.K; k += TB) {
        const int cnt = chunk_count(a.len[k], CH);
        const float4* row = reinterpret_cast<const float4*>(a.partial + (int64_t)k * CH);      // CH is a multiple of 4 and the buffer 16-byte aligned
        float4 v[CH / 4];
#pragma unroll
        for (int q = 0; q < CH / 4; ++q) v[q] = 4 * q < cnt ? row[q] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int q = 0; q < CH / 4; ++q) {
            if (4 * q + 0 < cnt) s += (double)v[q].x;
            if (4 * q + 1 < cnt) s += (double)v[q].y;
            if (4 * q + 2 < cnt) s += (double)v[q].z;
            if (4 * q + 3 < cnt) s += (double)v[q].w;
        }
    }
    // the conflict / projection counts ride the same tree (one thread walking the K slots one dependent load after the other was
    // 8 of this launch's 12 us, on the serial tail of the step)
    int c = 0, pr = 0;
    for (int b = threadIdx.x; b < a.K; b += TB) {
        c += a.block_metrics[2 * b];
        pr += a.block_metrics[2 * b + 1];
    }
    sh[threadIdx.x] = s;
    shc[threadIdx.x] = c;
    shp[threadIdx.x] = pr;
    __syncthreads();
    for (int d = TB / 2; d > 0; d >>= 1) {
        if (threadIdx.x < d) {
            sh[threadIdx.x] += sh[threadIdx.x + d];
            shc[threadIdx.x] += shc[threadIdx.x + d];
            shp[threadIdx.x] += shp[threadIdx.x + d];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        a.normsq[0] = (float)sh[0];
        a.metrics[0] = shc[0];
        a.metrics[1] = shp[0];
    }
}

__global__ __launch_bounds__(TB) void adamw_kernel(MtArgs a) {
    int k, j;
    if (!locate_chunk(a, 0, a.K, CH, blockIdx.x, &k, &j)) return;
    if (!a.flags[k]) return;                                   // grad is None: torch skips the parameter entirely
    if (a.abort && *a.abort) return;                           // the step's inputs are not to be trusted: leave parameters and moments alone
    // clip_grad_norm_: coef = min(1, max_norm / (total_norm + 1e-6)); max_norm <= 0 disables clipping
    float coef = 1.f;
    if (a.max_norm > 0.f) coef = fminf(1.f, a.max_norm / (sqrtf(a.normsq[0]) + 1e-6f));
    const float lr = a.lr[k], wd = a.wd[k], step = a.steps[k];
    const float bc1 = 1.f - powf(a.beta1, step), bc2 = 1.f - powf(a.beta2, step);
    const float step_size = lr / bc1, bc2s = sqrtf(bc2);
    const int len4 = (a.len[k] + 3) / 4, per = chunk_per4(len4, CH);
    const int lo = j * per, hi = min(lo + per, len4);
    const int64_t off = a.off[k];
    auto one = [&](float g, float p, float m0, float v0, float* mo, float* vo) -> float {
        g *= coef;
        p *= (1.f - lr * wd);
        const float m = m0 + (g - m0) * (1.f - a.beta1);       // lerp_
        const float v = v0 * a.beta2 + g * g * (1.f - a.beta2);
        *mo = m;
        *vo = v;
        return p - step_size * (m / (sqrtf(v) / bc2s + a.eps));
    };
    for (int i = lo + threadIdx.x; i < hi; i += TB) {
        const int64_t o = off + 4 * (int64_t)i;
        const float4 g = *reinterpret_cast<const float4*>(a.final_grad + o), p = *reinterpret_cast<const float4*>(a.params + o);
        const float4 m0 = *reinterpret_cast<const float4*>(a.exp_avg + o), v0 = *reinterpret_cast<const float4*>(a.exp_avg_sq + o);
        float4 m, v, q;
        q.x = one(g.x, p.x, m0.x, v0.x, &m.x, &v.x);
        q.y = one(g.y, p.y, m0.y, v0.y, &m.y, &v.y);
        q.z = one(g.z, p.z, m0.z, v0.z, &m.z, &v.z);
        q.w = one(g.w, p.w, m0.w, v0.w, &m.w, &v.w);
        *reinterpret_cast<float4*>(a.exp_avg + o) = m;
        *reinterpret_cast<float4*>(a.exp_avg_sq + o) = v;
        *reinterpret_cast<float4*>(a.params + o) = q;
    }
}

}  // namespace

extern "C" size_t gmp_mt_workspace_bytes(int num_tensors) {
    const size_t K = num_tensors > 0 ? num_tensors : 0;
    return K * MAXT * MAXT * sizeof(double) * (1 + GCH) + K * MAXT * sizeof(float) + K * sizeof(int) + K * CH * sizeof(float) +
           (K + 2) * 2 * sizeof(int) + 1024;
}

extern "C" int gmp_mt_pcgrad_clip_adamw_ex(const float* task_grads, int64_t task_stride, int num_tasks, int num_tensors,
                                           const int64_t* tensor_off, const int32_t* tensor_len, const uint8_t* has,
                                           const int32_t* order_host, int n_order, int last_task, int extra_task,
                                           float* params, float* exp_avg, float* exp_avg_sq, float* steps,
                                           const float* lr, const float* wd, float beta1, float beta2, float eps,
                                           float max_norm, float* final_grad, float* normsq_out, int32_t* metrics_out,
                                           int32_t* flags_out, void* ws, size_t ws_bytes, int apply_update,
                                           int k_begin, int k_end, int phases, const int32_t* abort_flag, gmp_stream_t stream) {
    if (k_begin < 0 || k_end > num_tensors || k_begin > k_end || !(phases & 7))
        return gmp::fail(GMP_ERR_ARG, "mt_pcgrad: tensors [%d, %d) of %d, phases %d", k_begin, k_end, num_tensors, phases);
    if (num_tasks < 1 || num_tasks > MAXT || num_tensors < 1 || n_order < 1 || n_order > num_tasks)
        return gmp::fail(GMP_ERR_ARG, "mt_pcgrad: tasks=%d tensors=%d n_order=%d", num_tasks, num_tensors, n_order);
    if (!task_grads || !tensor_off || !tensor_len || !has || !order_host || !final_grad || !normsq_out || !metrics_out || !flags_out || !ws)
        return gmp::fail(GMP_ERR_ARG, "mt_pcgrad: null pointer");
    if (apply_update && (!params || !exp_avg || !exp_avg_sq || !steps || !lr || !wd)) return gmp::fail(GMP_ERR_ARG, "mt_pcgrad: optimizer state");
    if (ws_bytes < gmp_mt_workspace_bytes(num_tensors)) return gmp::fail(GMP_ERR_WORKSPACE, "mt_pcgrad: workspace");
    if (last_task >= num_tasks || extra_task >= num_tasks) return gmp::fail(GMP_ERR_ARG, "mt_pcgrad: task index");
    MtArgs a{};
    a.tg = task_grads; a.stride = task_stride; a.T = num_tasks; a.K = num_tensors; a.off = tensor_off; a.len = tensor_len;
    a.k0 = k_begin; a.k1 = k_end;
    a.has = has;
    char* w = (char*)ws;
    a.gram = (double*)w; w += (size_t)num_tensors * MAXT * MAXT * sizeof(double);
    a.gram_part = (double*)w; w += (size_t)num_tensors * GCH * MAXT * MAXT * sizeof(double);
    a.weights = (float*)w; w += (size_t)num_tensors * MAXT * sizeof(float);
    a.partial = (float*)w; w += (size_t)num_tensors * CH * sizeof(float);
    a.block_metrics = (int*)w;
    a.flags = flags_out; a.steps = steps; a.metrics = metrics_out;
    for (int i = 0; i < n_order; ++i) {
        if (order_host[i] < 0 || order_host[i] >= num_tasks) return gmp::fail(GMP_ERR_ARG, "mt_pcgrad: order[%d]=%d", i, order_host[i]);
        a.order[i] = order_host[i];
    }
    a.n_order = n_order; a.last_task = last_task; a.extra_task = extra_task;
    a.final_grad = final_grad; a.normsq = normsq_out; a.params = params; a.exp_avg = exp_avg; a.exp_avg_sq = exp_avg_sq;
    a.lr = lr; a.wd = wd; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.max_norm = max_norm;
    a.abort = abort_flag;
    hipStream_t st = (hipStream_t)stream;
    const int nk = k_end - k_begin;
    // upper bound of the chunk count of n tensors inside a buffer of task_stride floats: every tensor has at least one chunk, and at most one
    // more than its float4 / MINP4 (the kernels find their chunk themselves and the surplus workgroups leave at once)
    auto nblk = [&](int n) { return (unsigned)(task_stride / (4 * MINP4) + 2 * (int64_t)n + 1); };
    if ((phases & 1) && nk > 0) {
        if (n_order > 1) {
            switch (num_tasks) {          // (tasks beyond a.T never own a tensor: the generic form covers any count)
                case 2: hipLaunchKernelGGL(gram_kernel<2>, dim3(nblk(nk)), dim3(TB), 0, st, a); break;
                case 3: hipLaunchKernelGGL(gram_kernel<3>, dim3(nblk(nk)), dim3(TB), 0, st, a); break;
                case 4: hipLaunchKernelGGL(gram_kernel<4>, dim3(nblk(nk)), dim3(TB), 0, st, a); break;
                case 5: hipLaunchKernelGGL(gram_kernel<5>, dim3(nblk(nk)), dim3(TB), 0, st, a); break;
                case 6: hipLaunchKernelGGL(gram_kernel<6>, dim3(nblk(nk)), dim3(TB), 0, st, a); break;
                default: hipLaunchKernelGGL(gram_kernel<MAXT>, dim3(nblk(nk)), dim3(TB), 0, st, a); break;
            }
            hipLaunchKernelGGL(gram_finish_kernel, dim3(nk), dim3(64), 0, st, a);
        }
        hipLaunchKernelGGL(solve_kernel, dim3((nk + 63) / 64), dim3(64), 0, st, a);
        hipLaunchKernelGGL(combine_kernel, dim3(nblk(nk)), dim3(TB), 0, st, a);
    }
    if ((phases & 4) && nk > 0) hipLaunchKernelGGL(foreign_kernel, dim3(nblk(nk)), dim3(TB), 0, st, a);
    if (phases & 2) {
        hipLaunchKernelGGL(norm_kernel, dim3(1), dim3(TB), 0, st, a);
        if (apply_update) hipLaunchKernelGGL(adamw_kernel, dim3(nblk(num_tensors)), dim3(TB), 0, st, a);
    }
    return gmp::check_launch("mt_pcgrad_clip_adamw kernels");
}

extern "C" int gmp_mt_pcgrad_clip_adamw(const float* task_grads, int64_t task_stride, int num_tasks, int num_tensors,
                                        const int64_t* tensor_off, const int32_t* tensor_len, const uint8_t* has,
                                        const int32_t* order_host, int n_order, int last_task, int extra_task,
                                        float* params, float* exp_avg, float* exp_avg_sq, float* steps,
                                        const float* lr, const float* wd, float beta1, float beta2, float eps,
                                        float max_norm, float* final_grad, float* normsq_out, int32_t* metrics_out,
                                        int32_t* flags_out, void* ws, size_t ws_bytes, int apply_update,
                                        gmp_stream_t stream) {
    return gmp_mt_pcgrad_clip_adamw_ex(task_grads, task_stride, num_tasks, num_tensors, tensor_off, tensor_len, has, order_host, n_order,
                                       last_task, extra_task, params, exp_avg, exp_avg_sq, steps, lr, wd, beta1, beta2, eps, max_norm,
                                       final_grad, normsq_out, metrics_out, flags_out, ws, ws_bytes, apply_update, 0, num_tensors, 3, nullptr, stream);
}
