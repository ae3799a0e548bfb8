// Device-side augmentation and masking: the per-graph Python loops of
// src/pretrain/augmentations.py:17-111 (node drop -> subgraph relabel -> edge drop -> attribute mask, two views per graph,
// common-node masks) and src/models/pretrain_model.py:67-88 (node-feature-masking indices) as kernels over a whole domain batch.
//
// What the reference draws with torch.randperm(n)[:k] from its CPU generator is, as a distribution, "a uniformly random k-subset":
// here every node / edge / feature column gets a Philox4x32-10 key (seed, stream, global element id) and the k smallest keys of a
// graph are the subset (ties broken by index).  The draws therefore do not replay the reference's mt19937 stream -- nothing on a
// device can -- and serve the engine's "device" RNG mode; the reference-order CPU path stays for bit-exact parity.  The STRUCTURE
// is the reference's, and tests/test_gpu_augment.py holds it to that: counts (n - max(1, int(.2 n)) kept nodes for n >= 3,
// max(1, int(.15 n)) masked nodes, E' - max(1, int(.2 E')) kept edges when the coin falls and E' >= 3, max(1, int(.2 F)) zeroed
// columns), kept nodes sorted, edges relabelled exactly as PyG subgraph(relabel_nodes=True) does, common-node sets -- and to the
// oracle itself, by replaying the device's decisions through oracle.augment.create_two_views as injected permutations.
//
// Layout: one workgroup per graph draws BOTH views (the common-node sets need both).  Kept-node counts are a function of the
// graph sizes alone, so `rows` is written at its final place; edge and common-node counts are random, so the first kernel leaves
// every graph's survivors compacted at the start of the graph's own slot (its original edge / node range) and a second kernel
// moves the slots to their final offsets (exclusive scan of the counts) in view-local numbering.
#include "gnnmp_internal.h"

namespace {

constexpr int AB = 256;                 // threads per graph
constexpr int MAX_NODES = 4096;         // per graph: keep flags / new ids / keys of both views live in LDS
constexpr int EDGES_LDS = 4096;
constexpr uint32_t S_NODE = 0x6e6f6465u, S_EDGE = 0x65646765u, S_ECOIN = 0x65636f69u, S_ACOIN = 0x61636f69u, S_ATTR = 0x61747472u,
                   S_NFM = 0x6e666d6bu;

__device__ __forceinline__ uint32_t key32(uint64_t seed, uint32_t stream, uint32_t purpose, uint64_t idx) {
    return gmp::philox4x32(make_uint4((uint32_t)idx, (uint32_t)(idx >> 32), stream, purpose), make_uint2((uint32_t)seed, (uint32_t)(seed >> 32))).x;
}
__device__ __forceinline__ float unif(uint64_t seed, uint32_t stream, uint32_t purpose, uint64_t idx) {
    return (float)key32(seed, stream, purpose, idx) * 2.3283064365386963e-10f;        // [0, 1)
}
__device__ __forceinline__ int drop_count(int n, double rate) {                      // max(1, int(n * rate)), Python double arithmetic
    const int k = (int)((double)n * rate);
    return k > 1 ? k : 1;
}

// ---- node-feature-masking indices: per graph with n >= 3 the max(1, int(.15 n)) nodes of smallest key, ascending -------------------
__device__ __forceinline__ void nfm_masks_body(const int64_t* __restrict__ ptr, const int64_t* __restrict__ out_ptr, uint64_t seed,
                                               uint32_t stream, int64_t* __restrict__ out, const int g) {
    __shared__ uint32_t keys[MAX_NODES];
    __shared__ uint8_t chosen[MAX_NODES];
    const int64_t s = ptr[g];
    const int n = (int)(ptr[g + 1] - s);
    if (n < 3) return;
    const int k = drop_count(n, 0.15);
    for (int i = threadIdx.x; i < n; i += AB) keys[i] = key32(seed, stream, S_NFM, (uint64_t)(s + i));
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += AB) {
        const uint32_t ki = keys[i];
        int rank = 0;
        for (int j = 0; j < n; ++j) rank += (keys[j] < ki) || (keys[j] == ki && j < i);
        chosen[i] = rank < k;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += AB) {
        if (!chosen[i]) continue;
        int before = 0;
        for (int j = 0; j < i; ++j) before += chosen[j];
        out[out_ptr[g] + before] = s + i;
    }
}
__global__ __launch_bounds__(AB) void nfm_masks_kernel(const int64_t* __restrict__ ptr, const int64_t* __restrict__ out_ptr, uint64_t seed,
                                                       uint32_t stream, int64_t* __restrict__ out) {
    nfm_masks_body(ptr, out_ptr, seed, stream, out, blockIdx.x);
}

// Batched forms: ONE launch for all the (task, domain) jobs of a step -- the jobs ride in the kernel arguments, a workgroup finds its job
// from the running graph counts.  Same arithmetic, same keys: results identical to the per-job launches.
constexpr int AUG_MAXJ = 8;
struct MaskJob { const int64_t* ptr; const int64_t* out_ptr; int64_t* out; uint32_t stream; };
struct MaskBatch { MaskJob job[AUG_MAXJ]; int first[AUG_MAXJ + 1]; int count; uint64_t seed; };
__device__ __forceinline__ int job_of(const int* first, int count, int b) {
    int j = 0;
    while (j + 1 < count && b >= first[j + 1]) ++j;
    return j;
}
__global__ __launch_bounds__(AB) void nfm_masks_batch_kernel(const MaskBatch b) {
    const int j = job_of(b.first, b.count, blockIdx.x);
    nfm_masks_body(b.job[j].ptr, b.job[j].out_ptr, b.seed, b.job[j].stream, b.job[j].out, blockIdx.x - b.first[j]);
}

struct ViewArgs {
    const int64_t* ptr;          // [G + 1] node offsets of the batch (domain-local row ids)
    const int64_t* eptr;         // [G + 1] edge offsets
    const int64_t* src;          // [E] edge_index[0]
    const int64_t* dst;          // [E] edge_index[1]
    const int64_t* vptr;         // [G + 1] node offsets of a VIEW (both views keep the same number of nodes per graph)
    int G, F;
    uint64_t seed;
    uint32_t stream;             // views use stream and stream + 1
    int64_t* rows[2];            // [vptr[G]] kept nodes, batch numbering, ascending            (final)
    uint64_t* rowmask[2];        // [vptr[G]] bit c set = feature column c zeroed               (final)
    int64_t* slot_src[2];        // [E] surviving edges of graph g compacted at eptr[g], graph-LOCAL new ids
    int64_t* slot_dst[2];
    int64_t* slot_common[2];     // [N] common nodes of graph g compacted at ptr[g], graph-local new ids of that view
    int32_t* edge_count[2];      // [G]
    int32_t* common_count;       // [G] (the same for both views)
    int32_t* mask_flag[2];       // [G] this graph drew an attribute mask in view v
    uint32_t* ekey;              // [E] scratch: key of a surviving edge (a graph's block only touches its own range)
    uint8_t* eflag;              // [E] scratch: survives node drop / survives edge drop
};

// LDS is sized per launch from the batch's largest graph (ncap nodes, ecap edges; a few KB for TUDataset graphs): these blocks run
// beside the step's GEMMs, whose 64 KB blocks leave little LDS free on a CU -- a 56 KB static footprint waited for a GEMM block to
// retire before it could start (58 us per launch in the step against ~10 alone)
__device__ __forceinline__ void two_views_body(const ViewArgs& a, const int ncap, const int ecap, const int g, unsigned char* dyn) {
    unsigned long long* const s_bits_p = reinterpret_cast<unsigned long long*>(dyn);
    int* const s_cnt = reinterpret_cast<int*>(dyn + 8);
    uint32_t* const keys = reinterpret_cast<uint32_t*>(dyn + 32);
    uint32_t* const s_ekey = keys + ncap;           // edge keys / flags of the graph: in LDS when they fit (every TUDataset graph does),
    uint16_t* const newid0 = reinterpret_cast<uint16_t*>(s_ekey + ecap);      // in the workspace otherwise: the rank loops are latency-bound
    uint16_t* const newid1 = newid0 + ncap;                                   // new ids per view; 0xffff = dropped
    uint8_t* const s_eflag = reinterpret_cast<uint8_t*>(newid1 + ncap);
    uint16_t* const newid[2] = {newid0, newid1};
#define s_bits (*s_bits_p)
    const int t = threadIdx.x;
    const int64_t s = a.ptr[g], es = a.eptr[g];
    const int n = (int)(a.ptr[g + 1] - s), ne = (int)(a.eptr[g + 1] - es);
    const int keep_n = n >= 3 ? n - drop_count(n, 0.2) : n;
    for (int v = 0; v < 2; ++v) {
        // ---- node drop: the keep_n smallest keys stay; new id = number of kept nodes in front (sorted kept list, subgraph's relabelling)
        __syncthreads();
        for (int i = t; i < n; i += AB) keys[i] = key32(a.seed, a.stream + v, S_NODE, (uint64_t)(s + i));
        __syncthreads();
        for (int i = t; i < n; i += AB) {
            const uint32_t ki = keys[i];
            int rank = 0;
            for (int j = 0; j < n; ++j) rank += (keys[j] < ki) || (keys[j] == ki && j < i);
            newid[v][i] = rank < keep_n ? 1 : 0xffff;            // provisional: kept flag
        }
        __syncthreads();
        for (int i = t; i < n; i += AB) {
            if (newid[v][i] == 0xffff) continue;
            int before = 0;
            for (int j = 0; j < i; ++j) before += newid[v][j] != 0xffff;
            keys[i] = (uint32_t)before;                           // (keys are spent: reuse the array for the prefix counts)
        }
        __syncthreads();
        // attribute mask of this (view, graph): coin, then the max(1, int(.2 F)) smallest of F column keys (thread c ranks column c)
        if (t == 0) s_bits = 0ull;
        __syncthreads();
        if (a.F >= 3 && unif(a.seed, a.stream + v, S_ACOIN, (uint64_t)(s)) < 0.2f) {
            const int m = drop_count(a.F, 0.2);
            if (t < a.F) {
                const uint32_t kc = key32(a.seed, a.stream + v, S_ATTR, (uint64_t)s * 64 + t);
                int rank = 0;
                for (int q = 0; q < a.F; ++q) {
                    const uint32_t kq = key32(a.seed, a.stream + v, S_ATTR, (uint64_t)s * 64 + q);
                    rank += (kq < kc) || (kq == kc && q < t);
                }
                if (rank < m) atomicOr(&s_bits, 1ull << t);
            }
        }
        __syncthreads();
        const uint64_t bits = s_bits;
        if (t == 0) a.mask_flag[v][g] = bits != 0ull;
        for (int i = t; i < n; i += AB) {
            if (newid[v][i] == 0xffff) continue;
            const int id = (int)keys[i];
            newid[v][i] = (uint16_t)id;
            a.rows[v][a.vptr[g] + id] = s + i;
            a.rowmask[v][a.vptr[g] + id] = bits;
        }
    }
    __syncthreads();
    // ---- common nodes (augmentations.py:77-85): kept in both views; listed per view in that view's local numbering, ascending
    if (t == 0) {
        int c = 0;
        for (int i = 0; i < n; ++i)
            if (newid[0][i] != 0xffff && newid[1][i] != 0xffff) {
                a.slot_common[0][s + c] = newid[0][i];
                a.slot_common[1][s + c] = newid[1][i];
                ++c;
            }
        a.common_count[g] = c;
    }
    // ---- edges: subgraph() keeps an edge when both endpoints stay (order preserved); then, if the coin falls and at least 3 are
    // left, the E' - max(1, int(.2 E')) smallest edge keys stay (kept in their original order)
    uint32_t* const ekey = ne <= ecap ? s_ekey : a.ekey + es;          // (generic pointers: LDS or global)
    uint8_t* const eflag = ne <= ecap ? s_eflag : a.eflag + es;
    for (int v = 0; v < 2; ++v) {
        __syncthreads();
        if (t == 0) s_cnt[0] = 0;
        __syncthreads();
        int alive_mine = 0;
        for (int e = t; e < ne; e += AB) {
            const int i = (int)(a.src[es + e] - s), j = (int)(a.dst[es + e] - s);
            const bool al = newid[v][i] != 0xffff && newid[v][j] != 0xffff;
            eflag[e] = al;
            ekey[e] = al ? key32(a.seed, a.stream + v, S_EDGE, (uint64_t)(es + e)) : 0u;
            alive_mine += al;
        }
        atomicAdd(&s_cnt[0], alive_mine);
        __threadfence_block();
        __syncthreads();
        const int alive = s_cnt[0];
        const bool drop = alive >= 3 && unif(a.seed, a.stream + v, S_ECOIN, (uint64_t)s) < 0.2f;
        const int keep_e = drop ? alive - drop_count(alive, 0.2) : alive;
        if (drop) {                                  // rank among the survivors; the keep_e smallest keys stay
            uint8_t mine[(8192 + AB - 1) / AB];      // this thread's verdicts (ne <= 8192 is checked by the host wrapper)
            int m = 0;
            for (int e = t; e < ne; e += AB, ++m) {
                bool kept = eflag[e];
                if (kept) {
                    const uint32_t ke = ekey[e];
                    int rank = 0;
                    for (int q = 0; q < ne; ++q) rank += eflag[q] && ((ekey[q] < ke) || (ekey[q] == ke && q < e));
                    kept = rank < keep_e;
                }
                mine[m] = kept;
            }
            __syncthreads();                          // every rank is taken before any flag changes
            m = 0;
            for (int e = t; e < ne; e += AB, ++m) eflag[e] = mine[m];
            __threadfence_block();
            __syncthreads();
        }
        for (int e = t; e < ne; e += AB) {           // position among the kept edges = kept edges in front (original order preserved)
            if (!eflag[e]) continue;
            int pos = 0;
            for (int q = 0; q < e; ++q) pos += eflag[q];
            a.slot_src[v][es + pos] = newid[v][(int)(a.src[es + e] - s)];
            a.slot_dst[v][es + pos] = newid[v][(int)(a.dst[es + e] - s)];
        }
        if (t == 0) a.edge_count[v][g] = keep_e;
    }
#undef s_bits
}

struct EmitArgs {
    const int64_t *ptr, *eptr, *vptr;
    int G;
    const int64_t* slot_src[2];
    const int64_t* slot_dst[2];
    const int64_t* slot_common[2];
    const int32_t* edge_count[2];
    const int32_t* common_count;
    const int32_t* mask_flag[2];
    int64_t* edges[2];          // [2, ecap]: row 0 at edges[v], row 1 at edges[v] + ecap
    int64_t ecap;
    int64_t* common[2];
    int32_t* totals;            // [5]: edges of view 0, edges of view 1, common nodes, any attribute mask in view 0 / view 1
};

// slots -> final offsets (exclusive scan of the per-graph counts), graph-local new ids -> view numbering (+ vptr[g])
__device__ __forceinline__ void emit_views_body(const EmitArgs& a, const int g) {
    const int t = threadIdx.x;
    for (int v = 0; v < 2; ++v) {
        int64_t off = 0;
        for (int q = 0; q < g; ++q) off += a.edge_count[v][q];
        const int cnt = a.edge_count[v][g];
        for (int e = t; e < cnt; e += AB) {
            a.edges[v][off + e] = a.slot_src[v][a.eptr[g] + e] + a.vptr[g];
            a.edges[v][a.ecap + off + e] = a.slot_dst[v][a.eptr[g] + e] + a.vptr[g];
        }
        if (g == a.G - 1 && t == 0) a.totals[v] = (int32_t)(off + cnt);
    }
    int64_t coff = 0;
    for (int q = 0; q < g; ++q) coff += a.common_count[q];
    const int cc = a.common_count[g];
    for (int c = t; c < cc; c += AB) {
        a.common[0][coff + c] = a.slot_common[0][a.ptr[g] + c] + a.vptr[g];
        a.common[1][coff + c] = a.slot_common[1][a.ptr[g] + c] + a.vptr[g];
    }
    if (g == a.G - 1 && t == 0) {
        a.totals[2] = (int32_t)(coff + cc);
        for (int v = 0; v < 2; ++v) {           // (written by every launch: no memset in front of the kernels)
            int any = 0;
            for (int q = 0; q < a.G; ++q) any |= a.mask_flag[v][q];
            a.totals[3 + v] = any;
        }
    }
}
__global__ __launch_bounds__(AB) void emit_views_kernel(const EmitArgs a) { emit_views_body(a, blockIdx.x); }
__global__ __launch_bounds__(AB) void two_views_kernel(const ViewArgs a, int ncap, int ecap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    two_views_body(a, ncap, ecap, blockIdx.x, dyn);
}
struct ViewBatch { ViewArgs job[AUG_MAXJ]; int first[AUG_MAXJ + 1]; int count; };
struct EmitBatch { EmitArgs job[AUG_MAXJ]; int first[AUG_MAXJ + 1]; int count; };
__global__ __launch_bounds__(AB) void two_views_batch_kernel(const ViewBatch b, int ncap, int ecap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    const int j = job_of(b.first, b.count, blockIdx.x);
    two_views_body(b.job[j], ncap, ecap, blockIdx.x - b.first[j], dyn);
}
__global__ __launch_bounds__(AB) void emit_views_batch_kernel(const EmitBatch b) {
    const int j = job_of(b.first, b.count, blockIdx.x);
    emit_views_body(b.job[j], blockIdx.x - b.first[j]);
}

}  // namespace

extern "C" size_t gmp_aug_workspace_bytes(int64_t num_nodes, int64_t num_edges, int num_graphs) {
    if (num_nodes < 0 || num_edges < 0 || num_graphs < 0) return 0;
    // slot_src / slot_dst x 2 views, slot_common x 2, edge_count x 2 + common_count
    return (size_t)(4 * num_edges + 2 * num_nodes) * sizeof(int64_t) + (size_t)num_edges * 5 + 512;      // + edge keys (u32) and flags (u8)
}

extern "C" int gmp_aug_node_masks(const int64_t* ptr, const int64_t* out_ptr, int num_graphs, int64_t max_graph_nodes, uint64_t seed,
                                  uint32_t stream_id, int64_t* out_idx, gmp_stream_t stream) {
    if (num_graphs < 0 || (num_graphs > 0 && (!ptr || !out_ptr || !out_idx))) return gmp::fail(GMP_ERR_ARG, "aug_node_masks: bad argument");
    if (max_graph_nodes > MAX_NODES) return gmp::fail(GMP_ERR_UNSUPPORTED, "aug_node_masks: a graph of %lld nodes (limit %d)", (long long)max_graph_nodes, MAX_NODES);
    if (num_graphs == 0) return GMP_OK;
    hipLaunchKernelGGL(nfm_masks_kernel, dim3(num_graphs), dim3(AB), 0, (hipStream_t)stream, ptr, out_ptr, seed, stream_id, out_idx);
    return gmp::check_launch("nfm_masks_kernel");
}

namespace {
struct ViewsCall {            // plain-argument form of one two-views job (what both entry points validate and turn into kernel arguments)
    const int64_t *ptr, *eptr, *edge_index;
    int64_t num_nodes, num_edges;
    const int64_t* view_ptr;
    int num_graphs, num_features;
    uint32_t stream_id;
    int64_t *rows1, *rows2;
    uint64_t *rowmask1, *rowmask2;
    int64_t *edges1, *edges2;
    int64_t edge_capacity;
    int64_t *common1, *common2;
    int32_t *counts, *totals_and_flags;
    void* workspace;
    size_t workspace_bytes;
};
int views_check(const ViewsCall& c, int64_t max_graph_nodes, int64_t max_graph_edges) {
    if (c.num_graphs < 0 || c.num_edges < 0 || c.num_features < 0 || c.num_features > 64)
        return gmp::fail(GMP_ERR_ARG, "aug_two_views: bad sizes (attribute masks are 64-bit column sets)");
    if (c.num_graphs == 0) return GMP_OK;
    if (!c.ptr || !c.eptr || !c.view_ptr || !c.rows1 || !c.rows2 || !c.rowmask1 || !c.rowmask2 || !c.edges1 || !c.edges2 || !c.common1 || !c.common2 ||
        !c.counts || !c.totals_and_flags || (c.num_edges > 0 && !c.edge_index))
        return gmp::fail(GMP_ERR_ARG, "aug_two_views: null pointer");
    if (max_graph_nodes > MAX_NODES) return gmp::fail(GMP_ERR_UNSUPPORTED, "aug_two_views: a graph of %lld nodes (limit %d)", (long long)max_graph_nodes, MAX_NODES);
    if (c.edge_capacity < c.num_edges) return gmp::fail(GMP_ERR_ARG, "aug_two_views: edge capacity below the batch's edge count");
    if (max_graph_edges > 8192) return gmp::fail(GMP_ERR_UNSUPPORTED, "aug_two_views: a graph of %lld edges (limit 8192)", (long long)max_graph_edges);
    if (!c.workspace || c.workspace_bytes < gmp_aug_workspace_bytes(c.num_nodes, c.num_edges, c.num_graphs)) return gmp::fail(GMP_ERR_WORKSPACE, "aug_two_views: workspace");
    return GMP_OK;
}
void views_args(const ViewsCall& c, uint64_t seed, ViewArgs* ap, EmitArgs* ep) {
    int64_t* w = (int64_t*)c.workspace;
    ViewArgs a{};
    a.ptr = c.ptr; a.eptr = c.eptr; a.src = c.edge_index; a.dst = c.edge_index ? c.edge_index + c.num_edges : nullptr; a.vptr = c.view_ptr;
    a.G = c.num_graphs; a.F = c.num_features; a.seed = seed; a.stream = c.stream_id;
    a.rows[0] = c.rows1; a.rows[1] = c.rows2; a.rowmask[0] = c.rowmask1; a.rowmask[1] = c.rowmask2;
    a.slot_src[0] = w; a.slot_dst[0] = w + c.num_edges; a.slot_src[1] = w + 2 * c.num_edges; a.slot_dst[1] = w + 3 * c.num_edges;
    a.slot_common[0] = w + 4 * c.num_edges; a.slot_common[1] = a.slot_common[0] + c.num_nodes;
    a.ekey = (uint32_t*)(a.slot_common[1] + c.num_nodes);
    a.eflag = (uint8_t*)(a.ekey + c.num_edges);
    a.edge_count[0] = c.counts; a.edge_count[1] = c.counts + c.num_graphs; a.common_count = c.counts + 2 * c.num_graphs;
    a.mask_flag[0] = c.counts + 3 * c.num_graphs; a.mask_flag[1] = c.counts + 4 * c.num_graphs;
    EmitArgs e{};
    e.ptr = c.ptr; e.eptr = c.eptr; e.vptr = c.view_ptr; e.G = c.num_graphs;
    for (int v = 0; v < 2; ++v) { e.slot_src[v] = a.slot_src[v]; e.slot_dst[v] = a.slot_dst[v]; e.slot_common[v] = a.slot_common[v]; e.edge_count[v] = a.edge_count[v]; }
    e.common_count = a.common_count;
    e.mask_flag[0] = a.mask_flag[0]; e.mask_flag[1] = a.mask_flag[1];
    e.edges[0] = c.edges1; e.edges[1] = c.edges2; e.ecap = c.edge_capacity; e.common[0] = c.common1; e.common[1] = c.common2; e.totals = c.totals_and_flags;
    *ap = a; *ep = e;
}
size_t views_lds(int64_t max_graph_nodes, int64_t max_graph_edges, int* ncap, int* ecap) {
    *ncap = (int)((max_graph_nodes + 63) / 64 * 64);
    *ecap = max_graph_edges <= EDGES_LDS ? (int)((max_graph_edges + 63) / 64 * 64) : 0;
    return 32 + (size_t)*ncap * 4 + (size_t)*ecap * 4 + (size_t)*ncap * 4 + (size_t)*ecap + 64;
}
}  // namespace

extern "C" int gmp_aug_two_views(const int64_t* ptr, const int64_t* eptr, const int64_t* edge_index, int64_t num_nodes, int64_t num_edges,
                                 const int64_t* view_ptr, int num_graphs, int64_t max_graph_nodes, int64_t max_graph_edges, int num_features, uint64_t seed, uint32_t stream_id,
                                 int64_t* rows1, int64_t* rows2, uint64_t* rowmask1, uint64_t* rowmask2, int64_t* edges1, int64_t* edges2,
                                 int64_t edge_capacity, int64_t* common1, int64_t* common2, int32_t* counts, int32_t* totals_and_flags,
                                 void* workspace, size_t workspace_bytes, gmp_stream_t stream) {
    const ViewsCall c{ptr, eptr, edge_index, num_nodes, num_edges, view_ptr, num_graphs, num_features, stream_id, rows1, rows2, rowmask1, rowmask2,
                      edges1, edges2, edge_capacity, common1, common2, counts, totals_and_flags, workspace, workspace_bytes};
    if (int rc = views_check(c, max_graph_nodes, max_graph_edges)) return rc;
    if (num_graphs == 0) return GMP_OK;
    ViewArgs a;
    EmitArgs e;
    views_args(c, seed, &a, &e);
    int ncap, ecap;
    const size_t lds = views_lds(max_graph_nodes, max_graph_edges, &ncap, &ecap);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(two_views_kernel, dim3(num_graphs), dim3(AB), lds, st, a, ncap, ecap);
    hipLaunchKernelGGL(emit_views_kernel, dim3(num_graphs), dim3(AB), 0, st, e);
    return gmp::check_launch("aug_two_views kernels");
}

// All the two-views jobs of a step -- one per contrastive (task, domain) pair -- in TWO launches (build, emit) instead of two per
// job: the draws ride the aux stream beside a running step, where sixteen 10-60 us launches per step cost it 0.3 ms.
extern "C" int gmp_aug_two_views_batch(const gmp_aug_views_job* jobs, int count, int64_t max_graph_nodes, int64_t max_graph_edges, uint64_t seed,
                                       gmp_stream_t stream) {
    if (count < 0 || (count > 0 && !jobs)) return gmp::fail(GMP_ERR_ARG, "aug_two_views_batch: bad job list");
    int ncap, ecap;
    const size_t lds = views_lds(max_graph_nodes, max_graph_edges, &ncap, &ecap);
    hipStream_t st = (hipStream_t)stream;
    for (int j0 = 0; j0 < count;) {
        ViewBatch vb{};
        EmitBatch eb{};
        int n = 0, blocks = 0;
        for (; j0 < count && n < AUG_MAXJ; ++j0) {
            const gmp_aug_views_job& q = jobs[j0];
            const ViewsCall c{q.ptr, q.eptr, q.edge_index, q.num_nodes, q.num_edges, q.view_ptr, q.num_graphs, q.num_features, q.stream_id, q.rows1, q.rows2,
                              q.rowmask1, q.rowmask2, q.edges1, q.edges2, q.edge_capacity, q.common1, q.common2, q.counts, q.totals_and_flags, q.workspace,
                              q.workspace_bytes};
            if (int rc = views_check(c, max_graph_nodes, max_graph_edges)) return rc;
            if (c.num_graphs == 0) continue;
            views_args(c, seed, &vb.job[n], &eb.job[n]);
            vb.first[n] = eb.first[n] = blocks;
            blocks += c.num_graphs;
            ++n;
        }
        if (n == 0) continue;
        vb.first[n] = eb.first[n] = blocks;
        vb.count = eb.count = n;
        hipLaunchKernelGGL(two_views_batch_kernel, dim3(blocks), dim3(AB), lds, st, vb, ncap, ecap);
        hipLaunchKernelGGL(emit_views_batch_kernel, dim3(blocks), dim3(AB), 0, st, eb);
    }
    return gmp::check_launch("aug_two_views_batch kernels");
}

extern "C" int gmp_aug_node_masks_batch(const gmp_aug_masks_job* jobs, int count, int64_t max_graph_nodes, uint64_t seed, gmp_stream_t stream) {
    if (count < 0 || (count > 0 && !jobs)) return gmp::fail(GMP_ERR_ARG, "aug_node_masks_batch: bad job list");
    if (max_graph_nodes > MAX_NODES) return gmp::fail(GMP_ERR_UNSUPPORTED, "aug_node_masks_batch: a graph of %lld nodes (limit %d)", (long long)max_graph_nodes, MAX_NODES);
    for (int j0 = 0; j0 < count;) {
        MaskBatch mb{};
        int n = 0, blocks = 0;
        for (; j0 < count && n < AUG_MAXJ; ++j0) {
            const gmp_aug_masks_job& q = jobs[j0];
            if (q.num_graphs < 0 || (q.num_graphs > 0 && (!q.ptr || !q.out_ptr || !q.out_idx))) return gmp::fail(GMP_ERR_ARG, "aug_node_masks_batch: bad job");
            if (q.num_graphs == 0) continue;
            mb.job[n] = MaskJob{q.ptr, q.out_ptr, q.out_idx, q.stream_id};
            mb.first[n] = blocks;
            blocks += q.num_graphs;
            ++n;
        }
        if (n == 0) continue;
        mb.first[n] = blocks;
        mb.count = n;
        mb.seed = seed;
        hipLaunchKernelGGL(nfm_masks_batch_kernel, dim3(blocks), dim3(AB), 0, (hipStream_t)stream, mb);
    }
    return gmp::check_launch("nfm_masks_batch_kernel");
}
