// Stacked input encoders: the Linear of InputEncoder (src/models/gnn.py:14,19) for every segment of a
// pre-training step in one launch.  Each segment belongs to a domain with its own weight matrix and
// its own (tiny) input width d_in in {7, 4, 37, 21}; features of all domains live padded to DPAD
// columns in one resident matrix and a segment's rows are gathered through `src_row`, with the
// attribute-mask augmentation (zeroed feature columns, augmentations.py:17-29) applied as a per-segment
// column bitmask per stacked row -- so augmented views never materialise their own feature matrices.
// K <= 64 is far too small for MFMA tiles to pay: one thread per output column keeps its weight row in
// registers and streams the rows of a 32-row tile from LDS.
#include "gnnmp_internal.h"

namespace {

constexpr int H = 256;          // hidden width (one thread per output column)
constexpr int TR = 32;          // rows per tile
constexpr int MAXD = 8;         // domains
constexpr int DPAD_MAX = 64;

struct EncArgs {
    const float* x_all;         // [R, dpad]
    int num_x_rows, num_rows, num_segs;   // bounds: a bad index reads as zero instead of faulting
    const int* src_row;         // [N] row of x_all feeding stacked row r
    const int* seg_ptr;         // [S+1]
    const int* seg_dom;         // [S]
    const unsigned long long* row_colmask;   // [N] bit k set -> feature k of that stacked row zeroed (nullable)
    const int* tiles;           // [T][2] = (segment, first row)
    const float* params;        // flat parameter buffer
    int64_t w_off[MAXD], b_off[MAXD];
    int d_in[MAXD];
    int dpad;
    float* z;                   // [N, 256]
};

template <int DP>
__global__ __launch_bounds__(H) void encoder_fwd_kernel(EncArgs a) {
    __shared__ float xs[TR][DP + 1];
    const int seg = a.tiles[2 * blockIdx.x], r0 = a.tiles[2 * blockIdx.x + 1];
    if (seg < 0 || seg >= a.num_segs || r0 < 0 || r0 >= a.num_rows) return;
    const int r1 = min(min(r0 + TR, a.seg_ptr[seg + 1]), a.num_rows);
    const int dom = a.seg_dom[seg], din = a.d_in[dom];
    const int c = threadIdx.x;
    for (int i = c; i < TR * DP; i += H) {
        const int rr = i / DP, k = i % DP;
        float v = 0.f;
        if (r0 + rr < r1 && k < din) {
            const unsigned long long mask = a.row_colmask ? a.row_colmask[r0 + rr] : 0ull;
            const int sr = a.src_row[r0 + rr];
            if (!((mask >> k) & 1ull) && sr >= 0 && sr < a.num_x_rows) v = a.x_all[(int64_t)sr * a.dpad + k];
        }
        xs[rr][k] = v;
    }
    float w[DP];
    const float* wp = a.params + a.w_off[dom] + (int64_t)c * din;
#pragma unroll
    for (int k = 0; k < DP; ++k) w[k] = k < din ? wp[k] : 0.f;
    const float b = a.params[a.b_off[dom] + c];
    __syncthreads();
    for (int rr = 0; rr < r1 - r0; ++rr) {
        float acc = b;
#pragma unroll
        for (int k = 0; k < DP; ++k) acc = fmaf(xs[rr][k], w[k], acc);
        a.z[(int64_t)(r0 + rr) * H + c] = acc;
    }
}

struct EncBwdArgs {
    const float* x_all;
    const int* src_row;
    const int* seg_ptr;
    const int* seg_dom;
    const unsigned long long* row_colmask;
    int num_x_rows, num_rows, num_segs;
    const float* gz;            // [N, 256]
    int dpad;
    int d_in[MAXD];
    int groups;                 // gradient groups = (task, domain) pairs; group g covers segments gseg[g]..gseg[g+1]
    int gseg[GMP_MAX_GROUPS + 1];
    int64_t off_w[GMP_MAX_GROUPS], off_b[GMP_MAX_GROUPS];
    float* out;                 // per-task gradient buffer
    float* part;                // nullable: [groups][RSPLIT][DP+1][256] row-slice partials (channel fastest: coalesced both ways)
};
constexpr int RSPLIT = 16;

// block (group, slice): dW[c][k] = sum_r gz[r][c] x[r][k], db[c] = sum_r gz[r][c] over every RSPLIT-th 32-row tile of the
// group (all of them without a workspace); partials are summed in slice order by encoder_bwd_reduce_kernel
template <int DP>
__global__ __launch_bounds__(H) void encoder_bwd_kernel(const EncBwdArgs a) {
    __shared__ float xs[TR][DP + 1];
    const int g = blockIdx.x, c = threadIdx.x;
    const int nsl = a.part ? RSPLIT : 1, sl = blockIdx.y;
    int tile = 0;
    float acc[DP], accb = 0.f;
#pragma unroll
    for (int k = 0; k < DP; ++k) acc[k] = 0.f;
    int din = 0;
    for (int seg = max(a.gseg[g], 0); seg < min(a.gseg[g + 1], a.num_segs); ++seg) {
        const int dom = a.seg_dom[seg];
        din = a.d_in[dom];
        const int s1 = min(a.seg_ptr[seg + 1], a.num_rows);
        for (int r0 = max(a.seg_ptr[seg], 0); r0 < s1; r0 += TR, ++tile) {
            if (tile % nsl != sl) continue;            // block-uniform
            const int r1 = min(r0 + TR, s1);
            __syncthreads();
            for (int i = c; i < TR * DP; i += H) {
                const int rr = i / DP, k = i % DP;
                float v = 0.f;
                if (r0 + rr < r1 && k < din) {
                    const unsigned long long mask = a.row_colmask ? a.row_colmask[r0 + rr] : 0ull;
                    const int sr = a.src_row[r0 + rr];
                    if (!((mask >> k) & 1ull) && sr >= 0 && sr < a.num_x_rows) v = a.x_all[(int64_t)sr * a.dpad + k];
                }
                xs[rr][k] = v;
            }
            __syncthreads();
            // eight gradient rows in flight at a time (one dependent load per row made this kernel the longest of the step's tail);
            // rows past the tile's end contribute gv = 0 against zeroed xs rows: the sums are unchanged bit for bit
            for (int rr0 = 0; rr0 < r1 - r0; rr0 += 8) {
                float gv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) gv[j] = (r0 + rr0 + j < r1) ? a.gz[(int64_t)(r0 + rr0 + j) * H + c] : 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (r0 + rr0 + j < r1) accb += gv[j];
#pragma unroll
                    for (int k = 0; k < DP; ++k) acc[k] = fmaf(gv[j], xs[rr0 + j][k], acc[k]);
                }
            }
        }
    }
    if (a.part) {
        float* po = a.part + ((int64_t)g * RSPLIT + sl) * (DP + 1) * H + c;
#pragma unroll
        for (int k = 0; k < DP; ++k) po[(int64_t)k * H] = acc[k];
        po[(int64_t)DP * H] = accb;
    } else if (a.gseg[g + 1] > a.gseg[g]) {
        float* wo = a.out + a.off_w[g] + (int64_t)c * din;
#pragma unroll
        for (int k = 0; k < DP; ++k)
            if (k < din) wo[k] = acc[k];
        a.out[a.off_b[g] + c] = accb;
    }
}

// block (group, k): thread c adds the RSPLIT slices of dW[c][k] (k = DP: db[c]) in slice order, all loads in flight together.  (The first
// form -- one block per group, a thread walking 16 x 41 partials laid out [c][k] -- read 10 MB at a 164-byte stride: 21 us in the step's tail.)
template <int DP>
__global__ __launch_bounds__(H) void encoder_bwd_reduce_kernel(const EncBwdArgs a) {
    const int g = blockIdx.x, k = blockIdx.y, c = threadIdx.x;
    if (a.gseg[g + 1] <= a.gseg[g]) return;
    const int din = a.d_in[a.seg_dom[a.gseg[g]]];
    if (k < DP && k >= din) return;
    float v[RSPLIT];
#pragma unroll
    for (int sl = 0; sl < RSPLIT; ++sl) v[sl] = a.part[(((int64_t)g * RSPLIT + sl) * (DP + 1) + k) * H + c];
    float acc = 0.f;
#pragma unroll
    for (int sl = 0; sl < RSPLIT; ++sl) acc += v[sl];
    if (k < DP) a.out[a.off_w[g] + (int64_t)c * din + k] = acc;
    else a.out[a.off_b[g] + c] = acc;
}

}  // namespace

extern "C" int gmp_encoder_fwd(const float* x_all, int64_t num_x_rows, int64_t num_rows, int num_segments, const int32_t* src_row, const int32_t* seg_ptr, const int32_t* seg_dom,
                               const uint64_t* row_colmask, const int32_t* tiles, int num_tiles, const float* params,
                               int num_domains, const int64_t* w_off_host, const int64_t* b_off_host,
                               const int32_t* d_in_host, int dpad, float* z, gmp_stream_t stream) {
    if (num_tiles < 0 || num_domains < 1 || num_domains > MAXD || dpad < 1 || dpad > DPAD_MAX)
        return gmp::fail(GMP_ERR_ARG, "encoder_fwd: tiles=%d domains=%d dpad=%d", num_tiles, num_domains, dpad);
    if (num_tiles == 0) return GMP_OK;
    if (!x_all || !src_row || !seg_ptr || !seg_dom || !tiles || !params || !w_off_host || !b_off_host || !d_in_host || !z)
        return gmp::fail(GMP_ERR_ARG, "encoder_fwd: null pointer");
    EncArgs a{};
    a.num_x_rows = (int)num_x_rows; a.num_rows = (int)num_rows; a.num_segs = num_segments;
    a.x_all = x_all; a.src_row = src_row; a.seg_ptr = seg_ptr; a.seg_dom = seg_dom;
    a.row_colmask = (const unsigned long long*)row_colmask; a.tiles = tiles; a.params = params; a.dpad = dpad; a.z = z;
    for (int d = 0; d < num_domains; ++d) {
        a.w_off[d] = w_off_host[d]; a.b_off[d] = b_off_host[d]; a.d_in[d] = d_in_host[d];
        if (a.d_in[d] < 1 || a.d_in[d] > dpad) return gmp::fail(GMP_ERR_ARG, "encoder_fwd: d_in[%d]=%d exceeds dpad %d", d, a.d_in[d], dpad);
    }
    hipStream_t st = (hipStream_t)stream;
    if (dpad <= 8) hipLaunchKernelGGL(encoder_fwd_kernel<8>, dim3(num_tiles), dim3(H), 0, st, a);
    else if (dpad <= 24) hipLaunchKernelGGL(encoder_fwd_kernel<24>, dim3(num_tiles), dim3(H), 0, st, a);
    else if (dpad <= 40) hipLaunchKernelGGL(encoder_fwd_kernel<40>, dim3(num_tiles), dim3(H), 0, st, a);
    else hipLaunchKernelGGL(encoder_fwd_kernel<64>, dim3(num_tiles), dim3(H), 0, st, a);
    return gmp::check_launch("encoder_fwd_kernel");
}

extern "C" int gmp_encoder_bwd(const float* x_all, int64_t num_x_rows, int64_t num_rows, int num_segments, const int32_t* src_row, const int32_t* seg_ptr, const int32_t* seg_dom,
                               const uint64_t* row_colmask, const float* g_z, int num_domains, const int32_t* d_in_host,
                               int dpad, int groups, const int32_t* group_seg_host, const int64_t* off_w_host,
                               const int64_t* off_b_host, float* grad_out, void* workspace, size_t workspace_bytes,
                               gmp_stream_t stream) {
    if (groups < 0 || groups > GMP_MAX_GROUPS || num_domains < 1 || num_domains > MAXD || dpad < 1 || dpad > DPAD_MAX)
        return gmp::fail(GMP_ERR_ARG, "encoder_bwd: groups=%d domains=%d dpad=%d", groups, num_domains, dpad);
    if (groups == 0) return GMP_OK;
    if (!x_all || !src_row || !seg_ptr || !seg_dom || !g_z || !d_in_host || !group_seg_host || !off_w_host || !off_b_host || !grad_out)
        return gmp::fail(GMP_ERR_ARG, "encoder_bwd: null pointer");
    EncBwdArgs a{};
    a.num_x_rows = (int)num_x_rows; a.num_rows = (int)num_rows; a.num_segs = num_segments;
    a.x_all = x_all; a.src_row = src_row; a.seg_ptr = seg_ptr; a.seg_dom = seg_dom;
    a.row_colmask = (const unsigned long long*)row_colmask; a.gz = g_z; a.dpad = dpad; a.groups = groups; a.out = grad_out;
    for (int d = 0; d < num_domains; ++d) a.d_in[d] = d_in_host[d];
    for (int g = 0; g <= groups; ++g) a.gseg[g] = group_seg_host[g];
    for (int g = 0; g < groups; ++g) { a.off_w[g] = off_w_host[g]; a.off_b[g] = off_b_host[g]; }
    hipStream_t st = (hipStream_t)stream;
    const int dp = dpad <= 8 ? 8 : (dpad <= 24 ? 24 : (dpad <= 40 ? 40 : 64));
    const size_t need = (size_t)groups * RSPLIT * H * (dp + 1) * sizeof(float);
    a.part = (workspace && workspace_bytes >= need) ? (float*)workspace : nullptr;
    const dim3 grid(groups, a.part ? RSPLIT : 1);
#define GMP_ENC_BWD(DPV)                                                                  \
    do {                                                                                  \
        hipLaunchKernelGGL(encoder_bwd_kernel<DPV>, grid, dim3(H), 0, st, a);             \
        if (a.part) hipLaunchKernelGGL(encoder_bwd_reduce_kernel<DPV>, dim3(groups, DPV + 1), dim3(H), 0, st, a); \
    } while (0)
    if (dp == 8) GMP_ENC_BWD(8);
    else if (dp == 24) GMP_ENC_BWD(24);
    else if (dp == 40) GMP_ENC_BWD(40);
    else GMP_ENC_BWD(64);
#undef GMP_ENC_BWD
    return gmp::check_launch("encoder_bwd_kernel");
}
