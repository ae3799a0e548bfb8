// Segmented neighbour gather + sum over CSR rows (GIN aggregation, mean pooling,
// scatter-add backward of row gathers).
//
// Mapping for CDNA4: one 64-lane wave owns a destination row; lane l holds float4
// #l of the row, so a 256-wide fp32 row is exactly ONE global_load_dwordx4 wave
// instruction (1 KiB, fully coalesced).  Neighbour ids are fetched 64 at a time
// with one coalesced load and handed out with v_readlane (wave-uniform row base in
// SGPRs).  Each wave walks a contiguous chunk of rows and the block->chunk map
// gives every XCD one contiguous span of the row space, so the neighbour rows of a
// graph (which sit next to each other in the batch) are re-read from that XCD's
// own L2 instead of crossing to HBM again.  HBM-bound: compulsory traffic is
// read x once + write out once.
#include <cstdlib>

#include "gnnmp_internal.h"

namespace {

constexpr int BLOCK = 256;                     // 4 waves
constexpr int WAVES_PER_BLOCK = BLOCK / GMP_WAVE;
constexpr int NUM_XCD = 8;

__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4fma(float s, float4 a, float4 b) {
    return make_float4(fmaf(s, a.x, b.x), fmaf(s, a.y, b.y), fmaf(s, a.z, b.z), fmaf(s, a.w, b.w));
}
__device__ __forceinline__ float f4dot(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

// A row piece out of LDS through an address-space-3 pointer.  `cond ? s_x[i] : x[j]` with both sides reached through generic pointers is
// compiled to ONE flat_load of a selected address -- and a flat load is waited for with s_waitcnt vmcnt(0) lgkmcnt(0), which also waits for
// every global prefetch in flight behind it: in the tile kernel below that serialised "prefetch the next tile" and "reduce this one"
// (round 3: found in the ISA).  A load through a typed LDS pointer cannot be merged with a global one: ds_read_b128, lgkmcnt only.
typedef float lds_f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) lds_f32x4* lds_row_ptr;
typedef const __attribute__((address_space(3))) int* lds_int_ptr;
// Pins a value where it is: an LDS-only branch ends with this so that the optimiser cannot sink its additions into a block shared with
// the branch that reads global memory -- the shared block would wait for BOTH counters (vmcnt(0) lgkmcnt(0)) on every path.
__device__ __forceinline__ void pin4(float4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }
__device__ __forceinline__ float4 lds_f4(lds_row_ptr p, int i) {
    const lds_f32x4 v = p[i];
    return make_float4(v.x, v.y, v.z, v.w);
}

// NV = float4 per lane (F <= 256*NV).  SELF: out = scale*self + sum.  DOT: also write
// rowdot[r] = <self[r], dotx[r]> (eps gradient, reduced per task afterwards).  ADDEND: out += addend[r]
// (the residual branch's gradient joins the aggregation's in the same pass).
template <int NV, bool SELF, bool IDX, bool MEAN, bool ACCUM, bool DOT, bool ADDEND = false>
__global__ __launch_bounds__(BLOCK) void seg_sum_kernel(const float4* __restrict__ src, const int* __restrict__ ptr,
                                                        const int* __restrict__ idx, const float4* __restrict__ self,
                                                        const float* __restrict__ eps, const float4* __restrict__ dotx,
                                                        float4* __restrict__ out, float* __restrict__ rowdot,
                                                        int64_t nrows, int F4, int rows_per_wave,
                                                        const float4* __restrict__ addend = nullptr) {
    // XCD-aware: hardware deals blocks round-robin over 8 XCDs; give XCD k the k-th
    // contiguous eighth of the chunk space (gridDim.x is a multiple of 8).
    __builtin_amdgcn_s_setprio(3);
    const int per_xcd = gridDim.x / NUM_XCD;
    const int lb = (blockIdx.x % NUM_XCD) * per_xcd + blockIdx.x / NUM_XCD;
    const int lane = threadIdx.x % GMP_WAVE, wv = threadIdx.x / GMP_WAVE;
    const int64_t gw = (int64_t)lb * WAVES_PER_BLOCK + wv;
    int64_t r0 = gw * rows_per_wave;
    int64_t r1 = r0 + rows_per_wave < nrows ? r0 + rows_per_wave : nrows;
    float scale = 1.f;
    if (SELF) scale = 1.f + (eps ? eps[0] : 0.f);
    bool act[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) act[v] = lane + v * GMP_WAVE < F4;

    for (int64_t r = r0; r < r1; ++r) {
        // Every load that does not depend on the row's neighbour list is ISSUED here, ahead of the ptr -> idx -> rows chain, and first used
        // behind it: the row's own operand (SELF), the eps-gradient operand (DOT) and the residual gradient (ADDEND).  The backward form
        // used to wait for dotx in front of the chain and fetch addend behind it -- five dependent memory latencies per row where the
        // forward has three (19.5 against 8.4 us in the step, profiles r02f) -- now three for every form.  Sums keep their order.
        float4 sv[NV], dv[NV], av[NV], ov[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            sv[v] = dv[v] = av[v] = ov[v] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (act[v]) {
                const int64_t o = r * F4 + lane + v * GMP_WAVE;
                if (SELF) sv[v] = self[o];
                if (SELF && DOT) dv[v] = dotx[o];
                if (ADDEND) av[v] = addend[o];
                if (ACCUM) ov[v] = out[o];
            }
        }
        const int start = __builtin_amdgcn_readfirstlane(ptr[r]);
        const int end = __builtin_amdgcn_readfirstlane(ptr[r + 1]);
        float4 acc[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[v] = make_float4(scale * sv[v].x, scale * sv[v].y, scale * sv[v].z, scale * sv[v].w);
        for (int e = start; e < end; e += GMP_WAVE) {
            const int cnt = end - e < GMP_WAVE ? end - e : GMP_WAVE;
            int mine = e + lane;
            if (IDX) mine = lane < cnt ? idx[e + lane] : 0;
            int j = 0;
            for (; j + 4 <= cnt; j += 4) {   // 4 independent row loads in flight per wave
                const int64_t c0 = __builtin_amdgcn_readlane(mine, j), c1 = __builtin_amdgcn_readlane(mine, j + 1);
                const int64_t c2 = __builtin_amdgcn_readlane(mine, j + 2), c3 = __builtin_amdgcn_readlane(mine, j + 3);
#pragma unroll
                for (int v = 0; v < NV; ++v)
                    if (act[v]) {
                        const int o = lane + v * GMP_WAVE;
                        float4 a = src[c0 * F4 + o], b = src[c1 * F4 + o], c = src[c2 * F4 + o], d = src[c3 * F4 + o];
                        acc[v] = f4add(f4add(f4add(f4add(acc[v], a), b), c), d);
                    }
            }
            for (; j < cnt; ++j) {
                const int64_t c0 = __builtin_amdgcn_readlane(mine, j);
#pragma unroll
                for (int v = 0; v < NV; ++v)
                    if (act[v]) acc[v] = f4add(acc[v], src[c0 * F4 + lane + v * GMP_WAVE]);
            }
        }
        float m = 1.f;
        if (MEAN) {
            int c = end - start;
            m = 1.f / (float)(c > 1 ? c : 1);
        }
        float dot = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v)
            if (act[v]) {
                float4 a = acc[v];
                if (MEAN) a = make_float4(a.x * m, a.y * m, a.z * m, a.w * m);
                const int64_t o = r * F4 + lane + v * GMP_WAVE;
                if (ADDEND) a = f4add(a, av[v]);
                if (ACCUM) a = f4add(ov[v], a);
                out[o] = a;
                if (SELF && DOT) dot += f4dot(sv[v], dv[v]);
            }
        if (DOT) {
            dot = gmp::wave_sum(dot);
            if (lane == 0) rowdot[r] = dot;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Inputs beyond the caches (the 65,536-graph rung: x = 2.2 GB): the LDS-resident tile form below.  (Its cache-resident predecessor --
// 1,024-thread workgroups over adjacent rows, rowptr / col staged in LDS, neighbour rows through L1 / L2 -- topped out at 4.0 TB/s on
// ~11.5 TB/s of L2 -> CU traffic and was removed in round 3 together with its GMP_AGG_VARIANT switch; profiles/README.md round 1 has its numbers.)
// LDS-resident tile form.  Leaving every neighbour read to L1/L2 moves ~4.8 KiB of x rows per output row across the L2->CU fabric
// (10.4 GB reads + 2.2 GB writes in 1.1 ms ~ 11.5 TB/s), which is what bounds such a kernel, not HBM.  Here a workgroup copies its tile of TILE consecutive rows (TILE KiB) into LDS once -- fully coalesced, one
// pass over x -- and neighbour rows that fall inside the tile (almost all: a tile spans ~4 whole graphs, and edges never
// leave a graph) are read from LDS; only the graphs cut by a tile border reach into global memory.  The next tile's
// rows / rowptr / col are prefetched into registers while the current tile is being reduced, so HBM stays busy during
// the LDS phase; the output leaves with non-temporal stores.
// DOT (the backward with the eps gradient, gmp_gin_aggregate_bwd: x = upstream gradient on the TRANSPOSED CSR, dotx = the layer's
// forward input): also rowdot[r] = <x[r], dotx[r]>.  The rows a wave reduces are exactly the rows whose float4 its threads would
// hold of a tile laid out thread-linear (thread i <-> row i / 64 = wave + 16 k), so the dotx tile never goes through LDS: each
// thread loads its own nine float4 of it when the tile is entered -- in flight beside the next tile's prefetch while the first
// rows are reduced -- and multiplies them with the row it is reducing anyway.  One extra KiB per row read, nothing re-read.
// [pmc-stamp-begin] bench.py stamps roofline.traffic with the hash of the lines between these two markers (the kernel the PMC passes measured and its launcher)
template <int SB, int TILE, int CCAP, bool NT_STORE, bool DOT = false>
__global__ __launch_bounds__(SB) void gin_aggregate_ldstile_kernel(const float4* __restrict__ x, const int* __restrict__ rowptr,
                                                                   const int* __restrict__ col, const float* __restrict__ eps,
                                                                   float4* __restrict__ out, int64_t nrows, int tiles_per_block,
                                                                   const float4* __restrict__ dotx = nullptr, float* __restrict__ rowdot = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* s_x = reinterpret_cast<float4*>(smem);
    int* s_col = reinterpret_cast<int*>(smem + (size_t)TILE * 1024);
    int* s_ptr = s_col + CCAP;
    constexpr int SWAVES = SB / GMP_WAVE;
    constexpr int XPT = TILE * 64 / SB;          // float4 of the tile per thread
    constexpr int CPT = CCAP / SB;
    static_assert(TILE * 64 % SB == 0 && CCAP % SB == 0 && TILE + 1 <= SB, "tile shape");
    const int per_xcd = gridDim.x / NUM_XCD;
    const int lb = (blockIdx.x % NUM_XCD) * per_xcd + blockIdx.x / NUM_XCD;
    const int lane = threadIdx.x % GMP_WAVE, wv = threadIdx.x / GMP_WAVE;
    const float scale = 1.f + (eps ? eps[0] : 0.f);
    const int64_t ntiles = (nrows + TILE - 1) / TILE;
    // Which tiles a workgroup takes.  Round 3: INTERLEAVED -- in its i-th step the workgroup on XCD slot (xcd, cu) takes tile
    // (i * 8 + xcd) * per_xcd + cu, so at any moment the chip streams ONE window of 8 x per_xcd consecutive tiles (36 MB) instead of
    // 256 streams 8.6 MB apart (DRAM pages), while a tile's neighbours t - 1 / t + 1 are still worked on by the same XCD at the same
    // time (the rows a graph cut by the tile border reaches for sit in that XCD's L2).  The contiguous span per workgroup (one eighth
    // of the rows per XCD) was the cache-resident kernel's mapping.
    (void)lb;
    const int xcd = blockIdx.x % NUM_XCD, cu = blockIdx.x / NUM_XCD;
    // (one window per XCD instead -- eight streams -- measured the same: 5.23 against 5.20-5.25 TB/s)
    auto tile_of = [&](int64_t i) -> int64_t { return (i * NUM_XCD + xcd) * per_xcd + cu; };
    int64_t nsteps = 0;
    while (nsteps < tiles_per_block && tile_of(nsteps) < ntiles) ++nsteps;
    if (nsteps == 0) return;

    float4 px[XPT];
    int pc[CPT];
    int pp = 0, pbase = 0, pcnt = 0;
#define GMP_PREFETCH_TILE(T, BASE, END)                                                        \
    do {                                                                                       \
        const int64_t pr0 = (T) * TILE;                                                        \
        const int pnr = (int)(nrows - pr0 < TILE ? nrows - pr0 : TILE);                        \
        _Pragma("unroll") for (int k = 0; k < XPT; ++k) {                                      \
            const int i = threadIdx.x + k * SB;                                                \
            px[k] = i < pnr * 64 ? x[pr0 * 64 + i] : make_float4(0.f, 0.f, 0.f, 0.f);          \
        }                                                                                      \
        pp = (int)threadIdx.x <= pnr ? rowptr[pr0 + threadIdx.x] : 0;                          \
        pbase = (BASE);                                                                        \
        pcnt = (END) - pbase;                                                                  \
        _Pragma("unroll") for (int k = 0; k < CPT; ++k) {                                      \
            const int i = threadIdx.x + k * SB;                                                \
            pc[k] = (pcnt <= CCAP && i < pcnt) ? col[pbase + i] : 0;                           \
        }                                                                                      \
    } while (0)
    // tile borders in col, read two tiles ahead so the (scalar, uncached) loads never stall the col prefetch
    auto border = [&](int64_t k) -> int { return rowptr[k * TILE < nrows ? k * TILE : nrows]; };
    // col range [nb0, nb1) of the tile of step i + 1, read one step ahead of its prefetch (scalar loads off the critical path)
    int nb0 = border(tile_of(0)), nb1 = border(tile_of(0) + 1);
    GMP_PREFETCH_TILE(tile_of(0), nb0, nb1);
    {
        const int64_t tn = nsteps > 1 ? tile_of(1) : tile_of(0);
        nb0 = border(tn); nb1 = border(tn + 1);
    }
    for (int64_t i = 0; i < nsteps; ++i) {
        const int64_t t = tile_of(i);
        const int64_t tn2 = i + 2 < nsteps ? tile_of(i + 2) : t;
        const int fb0 = border(tn2), fb1 = border(tn2 + 1);
        const int64_t r0 = t * TILE;
        const int nr = (int)(nrows - r0 < TILE ? nrows - r0 : TILE);
        const int base = pbase, cnt = pcnt;
        const bool staged = cnt <= CCAP;
        __syncthreads();                                   // the previous tile's LDS is no longer read
#pragma unroll
        for (int k = 0; k < XPT; ++k) {
            const int i = threadIdx.x + k * SB;
            if (i < nr * 64) s_x[i] = px[k];
        }
        if ((int)threadIdx.x <= nr) s_ptr[threadIdx.x] = pp;
        if (staged) {
#pragma unroll
            for (int k = 0; k < CPT; ++k) {
                const int i = threadIdx.x + k * SB;
                if (i < cnt) s_col[i] = pc[k];
            }
        }
        __syncthreads();
        // this tile's rows of dotx, thread-linear: qx[k] belongs to row wv + k * SWAVES.  Requested AHEAD rows before their use (a row
        // of the reduction lasts about one HBM latency at the target rate), not all at once: nine live float4 beside the next tile's
        // nine spilled 30 registers of the 128 a 1024-thread block may hold
        constexpr int AHEAD = 3;
        float4 qx[DOT ? XPT : 1];
        auto load_q = [&](int k) -> float4 {
            const int i = threadIdx.x + k * SB;
            return i < nr * 64 ? dotx[r0 * 64 + i] : make_float4(0.f, 0.f, 0.f, 0.f);
        };
        if (DOT) {
#pragma unroll
            for (int k = 0; k < AHEAD && k < XPT; ++k) qx[k] = load_q(k);
        }
        if (i + 1 < nsteps) GMP_PREFETCH_TILE(tile_of(i + 1), nb0, nb1);   // in flight while this tile is reduced out of LDS
        const int r0i = (int)r0;
        const lds_row_ptr sx3 = (lds_row_ptr)smem;
        auto row_of = [&](int u) -> float4 {              // u is wave-uniform: a scalar branch, two differently typed loads (never one flat load)
            const unsigned loc = (unsigned)(u - r0i);
            if (loc < (unsigned)nr) return lds_f4(sx3, loc * 64 + lane);
            return x[(int64_t)u * 64 + lane];
        };
        static_assert(!DOT || SB / GMP_WAVE * XPT >= TILE, "DOT: every row of the tile has a register slot");
        const lds_int_ptr sc3 = (lds_int_ptr)s_col, sp3 = (lds_int_ptr)s_ptr;
        auto reduce_row = [&](int rr) {
            const int start = sp3[rr], end = sp3[rr + 1];
            float4 acc = lds_f4(sx3, rr * 64 + lane);
            acc = make_float4(scale * acc.x, scale * acc.y, scale * acc.z, scale * acc.w);
            int e = start;
            for (; e + 4 <= end; e += 4) {
                int c0, c1, c2, c3;
                if (staged) { c0 = sc3[e - base]; c1 = sc3[e - base + 1]; c2 = sc3[e - base + 2]; c3 = sc3[e - base + 3]; }
                else { c0 = col[e]; c1 = col[e + 1]; c2 = col[e + 2]; c3 = col[e + 3]; }
                const int u0 = __builtin_amdgcn_readfirstlane(c0), u1 = __builtin_amdgcn_readfirstlane(c1);
                const int u2 = __builtin_amdgcn_readfirstlane(c2), u3 = __builtin_amdgcn_readfirstlane(c3);
                const unsigned l0 = (unsigned)(u0 - r0i), l1 = (unsigned)(u1 - r0i), l2 = (unsigned)(u2 - r0i), l3 = (unsigned)(u3 - r0i);
                if (staged && l0 < (unsigned)nr && l1 < (unsigned)nr && l2 < (unsigned)nr && l3 < (unsigned)nr) {
                    // the common case (whole graphs inside the tile): LDS only -- no vector-memory wait on this path, the next tile's
                    // prefetch stays in flight behind it
                    const float4 a = lds_f4(sx3, l0 * 64 + lane), b = lds_f4(sx3, l1 * 64 + lane);
                    const float4 c = lds_f4(sx3, l2 * 64 + lane), d = lds_f4(sx3, l3 * 64 + lane);
                    acc = f4add(f4add(f4add(f4add(acc, a), b), c), d);
                    pin4(acc);
                } else {
                    const float4 a = row_of(u0), b = row_of(u1), c = row_of(u2), d = row_of(u3);
                    acc = f4add(f4add(f4add(f4add(acc, a), b), c), d);
                }
            }
            for (; e < end; ++e) {
                const int cc = staged ? sc3[e - base] : col[e];
                const int u = __builtin_amdgcn_readfirstlane(cc);
                const unsigned l = (unsigned)(u - r0i);
                if (l < (unsigned)nr) { acc = f4add(acc, lds_f4(sx3, l * 64 + lane)); pin4(acc); }
                else acc = f4add(acc, x[(int64_t)u * 64 + lane]);
            }
            if (NT_STORE) {
                float* o = reinterpret_cast<float*>(out + (r0 + rr) * 64 + lane);
                __builtin_nontemporal_store(acc.x, o);
                __builtin_nontemporal_store(acc.y, o + 1);
                __builtin_nontemporal_store(acc.z, o + 2);
                __builtin_nontemporal_store(acc.w, o + 3);
            } else {
                out[(r0 + rr) * 64 + lane] = acc;
            }
        };
        if (DOT) {            // unrolled: the register slot of a row's dotx piece is a compile-time index
#pragma unroll
            for (int kq = 0; kq < XPT; ++kq) {
                const int rr = wv + kq * SWAVES;
                if (rr >= nr) break;
                if (kq + AHEAD < XPT) qx[kq + AHEAD] = load_q(kq + AHEAD);
                const float4 g = lds_f4(sx3, rr * 64 + lane);
                const float d = gmp::wave_sum((g.x * qx[kq].x + g.y * qx[kq].y) + (g.z * qx[kq].z + g.w * qx[kq].w));
                if (lane == 0) rowdot[r0 + rr] = d;
                reduce_row(rr);
            }
        } else {
            for (int rr = wv; rr < nr; rr += SWAVES) reduce_row(rr);
        }
        nb0 = fb0;
        nb1 = fb1;
    }
}

#undef GMP_PREFETCH_TILE

// any feature width: one thread per output element (class logits, 12 graph properties ...)
__global__ __launch_bounds__(BLOCK) void seg_sum_scalar_kernel(const float* __restrict__ src, const int* __restrict__ ptr,
                                                               const int* __restrict__ idx, float* __restrict__ out,
                                                               int64_t nrows, int F, int mean, int accumulate) {
    const int64_t total = nrows * F;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * BLOCK) {
        const int64_t r = i / F;
        const int f = (int)(i % F), s = ptr[r], e = ptr[r + 1];
        float acc = 0.f;
        for (int k = s; k < e; ++k) acc += src[(int64_t)(idx ? idx[k] : k) * F + f];
        if (mean) acc /= (float)(e - s > 1 ? e - s : 1);
        out[i] = accumulate ? out[i] + acc : acc;
    }
}

__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ p, int n, float* out) {
    __shared__ float sh[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += p[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if (threadIdx.x < d) sh[threadIdx.x] += sh[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sh[0];
}

struct Plan {
    int grid, rows_per_wave;
};

// Enough waves to fill 256 CUs x 8 waves/SIMD (8192 waves), chunks of >= 1 row;
// larger inputs get longer chunks instead of more blocks (grid-stride by chunk).
Plan make_plan(int64_t nrows) {
    const int64_t target_waves = 256 * 32;
    int64_t rpw = (nrows + target_waves - 1) / target_waves;
    if (rpw < 1) rpw = 1;
    int64_t waves = (nrows + rpw - 1) / rpw;
    int64_t blocks = (waves + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
    blocks = (blocks + NUM_XCD - 1) / NUM_XCD * NUM_XCD;
    if (blocks < NUM_XCD) blocks = NUM_XCD;
    return Plan{(int)blocks, (int)rpw};
}

template <bool SELF, bool IDX, bool MEAN, bool ACCUM, bool DOT, bool ADDEND = false>
int launch_nv(int nv, const Plan& p, hipStream_t st, const float* src, const int* ptr, const int* idx,
              const float* self, const float* eps, const float* dotx, float* out, float* partials, int64_t nrows,
              int F4, const float* addend = nullptr) {
#define GMP_LAUNCH_NV(NV)                                                                                       \
    hipLaunchKernelGGL((seg_sum_kernel<NV, SELF, IDX, MEAN, ACCUM, DOT, ADDEND>), dim3(p.grid), dim3(BLOCK), 0, st, \
                       (const float4*)src, ptr, idx, (const float4*)self, eps, (const float4*)dotx, (float4*)out, \
                       partials, nrows, F4, p.rows_per_wave, (const float4*)addend)
    switch (nv) {
        case 1: GMP_LAUNCH_NV(1); break;
        case 2: GMP_LAUNCH_NV(2); break;
        case 3: GMP_LAUNCH_NV(3); break;
        case 4: GMP_LAUNCH_NV(4); break;
        default: return gmp::fail(GMP_ERR_UNSUPPORTED, "feature width %d > 1024", F4 * 4);
    }
#undef GMP_LAUNCH_NV
    return gmp::check_launch("seg_sum_kernel");
}

// the default streaming form (144-row tiles = 153 KB of the CU's 160 KB LDS, 1024 threads, one block per CU), forward or -- on the
// transposed CSR, with the optional <g, x> row products for the eps gradient -- backward
template <bool DOT>
int launch_ldstile144(const float* x, const int* rowptr, const int* col, const float* eps, float* out, int64_t N, const float* dotx,
                      float* rowdot, hipStream_t st) {
    constexpr int SBV = 1024, TILEV = 144, CCAPV = 2048;
    auto kern = gin_aggregate_ldstile_kernel<SBV, TILEV, CCAPV, true, DOT>;
    const size_t lds = (size_t)TILEV * 1024 + (size_t)CCAPV * 4 + (size_t)(TILEV + 16) * 4;
    static std::atomic<uint64_t> attr_set{0};        // one bit per device (gnnmp_internal.h)
    if (!gmp::lds_attr_done(attr_set)) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return gmp::fail(GMP_ERR_LAUNCH, "gin_aggregate_ldstile: LDS attribute: %s", hipGetErrorString(e));
        gmp::lds_attr_mark(attr_set);
    }
    const int64_t ntiles = (N + TILEV - 1) / TILEV;
    const int blocks = 256;
    const int tpb = (int)((ntiles + blocks - 1) / blocks);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(SBV), lds, st, (const float4*)x, rowptr, col, eps, (float4*)out, N, tpb,
                       (const float4*)dotx, rowdot);
    return gmp::check_launch("gin_aggregate_ldstile_kernel");
}
// [pmc-stamp-end]

// sum of n floats in a fixed order: `nb` block partials (grid-stride by block, tree inside), then reduce_partials_kernel over them
__global__ __launch_bounds__(256) void block_partials_kernel(const float* __restrict__ p, int64_t n, float* __restrict__ part) {
    __shared__ float sh[256];
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) s += p[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if (threadIdx.x < d) sh[threadIdx.x] += sh[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = sh[0];
}

int check_feat(const char* who, int feat) {
    if (feat <= 0 || feat % 4 != 0 || feat > 1024)
        return gmp::fail(GMP_ERR_ARG, "%s: feature width %d must be a multiple of 4 in [4,1024]", who, feat);
    return GMP_OK;
}

}  // namespace

extern "C" int gmp_gin_aggregate_fwd(const float* x, const int32_t* rowptr, const int32_t* col, const float* eps,
                                     float* out, int64_t N, int feat, gmp_stream_t stream) {
    if (int rc = check_feat("gin_aggregate_fwd", feat)) return rc;
    if (N < 0 || (N > 0 && (!x || !rowptr || !out))) return gmp::fail(GMP_ERR_ARG, "gin_aggregate_fwd: null pointer");
    if (N == 0) return GMP_OK;
    const int F4 = feat / 4;
    if (feat == 256 && N >= 65536)            // working set beyond the caches: LDS-resident tiles, 1,024 threads, one block per CU
        return launch_ldstile144<false>(x, rowptr, col, eps, out, N, nullptr, nullptr, (hipStream_t)stream);
    Plan p = make_plan(N);
    return launch_nv<true, true, false, false, false>((F4 + 63) / 64, p, (hipStream_t)stream, x, rowptr, col, x, eps,
                                                      nullptr, out, nullptr, N, F4);
}

// Rows [row0, row1) of the same aggregation (cache-resident sizes): the stacked step runs its forward in two halves on two streams.
// x and col keep the whole batch's numbering; only the output rows are restricted.
extern "C" int gmp_gin_aggregate_fwd_rows(const float* x, const int32_t* rowptr, const int32_t* col, const float* eps, float* out,
                                          int64_t row0, int64_t row1, int feat, gmp_stream_t stream) {
    if (int rc = check_feat("gin_aggregate_fwd_rows", feat)) return rc;
    if (row0 < 0 || row1 < row0) return gmp::fail(GMP_ERR_ARG, "gin_aggregate_fwd_rows: bad row range");
    if (row1 == row0) return GMP_OK;
    if (!x || !rowptr || !out) return gmp::fail(GMP_ERR_ARG, "gin_aggregate_fwd_rows: null pointer");
    const int F4 = feat / 4;
    Plan p = make_plan(row1 - row0);
    return launch_nv<true, true, false, false, false>((F4 + 63) / 64, p, (hipStream_t)stream, x, rowptr + row0, col, x + row0 * feat, eps, nullptr,
                                                      out + row0 * feat, nullptr, row1 - row0, F4);
}

extern "C" size_t gmp_gin_aggregate_bwd_workspace_bytes(int64_t N, int feat) {
    (void)feat;
    return (size_t)(N > 0 ? N : 1) * sizeof(float) + 256 + 1024 * sizeof(float);   // one <g, x> per row (+ block partials), summed in a second pass
}

extern "C" int gmp_gin_aggregate_bwd(const float* g_out, const int32_t* rowptr_t, const int32_t* col_t,
                                     const float* eps, const float* x, float* g_x, float* g_eps, int64_t N, int feat,
                                     void* ws, size_t ws_bytes, gmp_stream_t stream) {
    if (int rc = check_feat("gin_aggregate_bwd", feat)) return rc;
    if (N < 0 || (N > 0 && (!g_out || !rowptr_t || !g_x)))
        return gmp::fail(GMP_ERR_ARG, "gin_aggregate_bwd: null pointer");
    if (g_eps && !x) return gmp::fail(GMP_ERR_ARG, "gin_aggregate_bwd: g_eps needs x");
    const int F4 = feat / 4;
    hipStream_t st = (hipStream_t)stream;
    if (N == 0) {
        if (g_eps && hipMemsetAsync(g_eps, 0, sizeof(float), st) != hipSuccess) return gmp::fail(GMP_ERR_LAUNCH, "gin_aggregate_bwd: memset");
        return GMP_OK;
    }
    // beyond the caches (the roofline rung): the forward's LDS-resident-tile kernel on the transposed CSR
    if (feat == 256 && N >= 65536 && col_t) {
        if (!g_eps) return launch_ldstile144<false>(g_out, rowptr_t, col_t, eps, g_x, N, nullptr, nullptr, st);
        if (ws_bytes < gmp_gin_aggregate_bwd_workspace_bytes(N, feat)) return gmp::fail(GMP_ERR_WORKSPACE, "gin_aggregate_bwd: workspace");
        float* rowdot = (float*)ws;
        float* part = rowdot + ((N + 63) / 64) * 64;
        if (int rc = launch_ldstile144<true>(g_out, rowptr_t, col_t, eps, g_x, N, x, rowdot, st)) return rc;
        hipLaunchKernelGGL(block_partials_kernel, dim3(1024), dim3(256), 0, st, (const float*)rowdot, N, part);
        hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(256), 0, st, (const float*)part, 1024, g_eps);
        return gmp::check_launch("gin_aggregate_bwd reduce");
    }
    Plan p = make_plan(N);
    if (!g_eps)
        return launch_nv<true, true, false, false, false>((F4 + 63) / 64, p, st, g_out, rowptr_t, col_t, g_out, eps,
                                                          nullptr, g_x, nullptr, N, F4);
    if (ws_bytes < (size_t)N * sizeof(float)) return gmp::fail(GMP_ERR_WORKSPACE, "gin_aggregate_bwd: workspace");
    int rc = launch_nv<true, true, false, false, true>((F4 + 63) / 64, p, st, g_out, rowptr_t, col_t, g_out, eps, x, g_x,
                                                       (float*)ws, N, F4);
    if (rc) return rc;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(256), 0, st, (const float*)ws, (int)N, g_eps);
    return gmp::check_launch("reduce_partials_kernel");
}

// Stacked-pass form: g_x = (1+eps) g + sum over the transposed CSR of g (+ addend), and rowdot[r] = <g[r], x[r]>
// so the caller can reduce the eps gradient per task (gmp_group_sum_1d).
extern "C" int gmp_gin_aggregate_bwd_ex(const float* g_out, const int32_t* rowptr_t, const int32_t* col_t, const float* eps,
                                        const float* x, const float* addend, float* g_x, float* rowdot, int64_t N,
                                        int feat, gmp_stream_t stream) {
    if (int rc = check_feat("gin_aggregate_bwd_ex", feat)) return rc;
    if (N < 0 || (N > 0 && (!g_out || !rowptr_t || !g_x))) return gmp::fail(GMP_ERR_ARG, "gin_aggregate_bwd_ex: null pointer");
    if (rowdot && !x) return gmp::fail(GMP_ERR_ARG, "gin_aggregate_bwd_ex: rowdot needs x");
    if (N == 0) return GMP_OK;
    const int F4 = feat / 4, nv = (F4 + 63) / 64;
    hipStream_t st = (hipStream_t)stream;
    Plan p = make_plan(N);
    if (rowdot) {
        if (addend) return launch_nv<true, true, false, false, true, true>(nv, p, st, g_out, rowptr_t, col_t, g_out, eps, x, g_x, rowdot, N, F4, addend);
        return launch_nv<true, true, false, false, true, false>(nv, p, st, g_out, rowptr_t, col_t, g_out, eps, x, g_x, rowdot, N, F4);
    }
    if (addend) return launch_nv<true, true, false, false, false, true>(nv, p, st, g_out, rowptr_t, col_t, g_out, eps, nullptr, g_x, nullptr, N, F4, addend);
    return launch_nv<true, true, false, false, false, false>(nv, p, st, g_out, rowptr_t, col_t, g_out, eps, nullptr, g_x, nullptr, N, F4);
}

namespace {
struct Groups1d {
    int n;
    int row[GMP_MAX_GROUPS + 1];
    int64_t off[GMP_MAX_GROUPS];
};
__global__ __launch_bounds__(256) void group_sum_1d_kernel(const float* __restrict__ v, Groups1d g, float* out) {
    __shared__ float sh[256];
    const int grp = blockIdx.x;
    float s = 0.f;
    for (int i = g.row[grp] + threadIdx.x; i < g.row[grp + 1]; i += 256) s += v[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if (threadIdx.x < d) sh[threadIdx.x] += sh[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[g.off[grp]] = sh[0];
}
}  // namespace

// out[off[g]] = sum of vals[rows[g] .. rows[g+1])  (fixed tree order: deterministic)
extern "C" int gmp_group_sum_1d(const float* vals, int groups, const int32_t* group_rows_host, const int64_t* out_off_host,
                                float* out, gmp_stream_t stream) {
    if (groups < 1 || groups > GMP_MAX_GROUPS || !group_rows_host || !vals || !out)
        return gmp::fail(GMP_ERR_ARG, "group_sum_1d: bad argument (groups=%d)", groups);
    Groups1d g{};
    g.n = groups;
    for (int i = 0; i <= groups; ++i) g.row[i] = group_rows_host[i];
    for (int i = 0; i < groups; ++i) g.off[i] = out_off_host ? out_off_host[i] : i;
    hipLaunchKernelGGL(group_sum_1d_kernel, dim3(groups), dim3(256), 0, (hipStream_t)stream, vals, g, out);
    return gmp::check_launch("group_sum_1d_kernel");
}

extern "C" int gmp_segment_sum(const float* src, const int32_t* ptr, const int32_t* idx, float* out, int64_t nseg,
                               int feat, int mean, int accumulate, gmp_stream_t stream) {
    if (nseg < 0 || (nseg > 0 && (!src || !ptr || !out))) return gmp::fail(GMP_ERR_ARG, "segment_sum: null pointer");
    if (nseg == 0) return GMP_OK;
    if (feat > 0 && feat % 4) {
        int64_t b = (nseg * feat + BLOCK - 1) / BLOCK;
        hipLaunchKernelGGL(seg_sum_scalar_kernel, dim3((unsigned)(b > 4096 ? 4096 : b)), dim3(BLOCK), 0, (hipStream_t)stream, src, ptr,
                           idx, out, nseg, feat, mean, accumulate);
        return gmp::check_launch("seg_sum_scalar_kernel");
    }
    if (int rc = check_feat("segment_sum", feat)) return rc;
    const int F4 = feat / 4, nv = (F4 + 63) / 64;
    Plan p = make_plan(nseg);
    hipStream_t st = (hipStream_t)stream;
#define GMP_SS(IDX, MEAN, ACC) \
    return launch_nv<false, IDX, MEAN, ACC, false>(nv, p, st, src, ptr, idx, nullptr, nullptr, nullptr, out, nullptr, nseg, F4)
    if (idx) {
        if (mean) { if (accumulate) GMP_SS(true, true, true); else GMP_SS(true, true, false); }
        else      { if (accumulate) GMP_SS(true, false, true); else GMP_SS(true, false, false); }
    } else {
        if (mean) { if (accumulate) GMP_SS(false, true, true); else GMP_SS(false, true, false); }
        else      { if (accumulate) GMP_SS(false, false, true); else GMP_SS(false, false, false); }
    }
#undef GMP_SS
}
