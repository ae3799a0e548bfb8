/*
 * libgnnmp -- MI355X (gfx950) message-passing kernels behind a flat C ABI.
 *
 * This is the drop-in boundary of the build.  The reference
 * (alonbebchuk/GNN-Pretraining) has no FFI of its own: its hot path calls
 * torch-geometric / torch operators from Python.  Each entry point below names the
 * reference call site (file:line under /root/reference) whose operator it serves.
 *
 * Conventions
 *   - every function returns GMP_OK (0) or a negative GMP_ERR_* code; the text of
 *     the last error on the calling thread is at gmp_last_error_string();
 *   - no exceptions cross the boundary, nothing here allocates or frees device
 *     memory: the caller owns every buffer and passes a workspace where one is
 *     needed (size from the matching *_workspace_bytes function);
 *   - all pointers are DEVICE pointers unless a parameter says "host";
 *   - all work is enqueued asynchronously on `stream` (a hipStream_t passed as
 *     void*); no call synchronises;
 *   - feature matrices are row-major fp32; graph indices handed over by the caller
 *     are int64 (the reference's dtype), CSR arrays produced here are int32;
 *   - F (feature width) must be a multiple of 4 and rows must be 16-byte aligned.
 */
#ifndef GNNMP_H
#define GNNMP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GMP_OK 0
#define GMP_ERR_ARG (-1)       /* bad argument (null pointer, negative size, bad alignment) */
#define GMP_ERR_LAUNCH (-2)    /* HIP reported a launch/runtime error */
#define GMP_ERR_WORKSPACE (-3) /* workspace too small */
#define GMP_ERR_UNSUPPORTED (-4)

typedef void* gmp_stream_t; /* hipStream_t */

int gmp_version(void);
const char* gmp_last_error_string(void);

/* ------------------------------------------------------------------------- *
 * Index build: COO int64 -> CSR int32, both orientations.
 * Serves PyG MessagePassing.propagate's gather/scatter indexing for
 * GINConv(h, edge_index) (src/models/gnn.py:41) and its backward.
 *   edge_index  [2,E] int64, row 0 = source j, row 1 = target i (PyG flow).
 *   rowptr/col/perm      grouped by TARGET: row i lists the sources of edges j->i
 *   rowptr_t/col_t/perm_t grouped by SOURCE (transposed graph, for the backward);
 *                         the three *_t pointers may all be NULL.
 *   perm[k] = COO edge id stored at CSR slot k; within a row slots are in ascending
 *   edge id (stable counting sort) -- bit-exact against oracle.graph_ops.coo_to_csr.
 *   status      device int32[1]: number of endpoints outside [0,N) (those edges are
 *               dropped instead of faulting); 0 on well-formed input.
 * ------------------------------------------------------------------------- */
size_t gmp_csr_build_workspace_bytes(int64_t num_nodes, int64_t num_edges);
int gmp_csr_build(const int64_t* edge_index, int64_t num_nodes, int64_t num_edges,
                  int32_t* rowptr, int32_t* col, int32_t* perm,
                  int32_t* rowptr_t, int32_t* col_t, int32_t* perm_t,
                  int32_t* status, void* workspace, size_t workspace_bytes, gmp_stream_t stream);

/* The same arrays for a BLOCK-DIAGONAL batch whose blocks ("segments": one forward() call's graphs) own contiguous row ranges
 * seg_row_ptr[s]..[s+1] and contiguous edge ranges seg_edge_ptr[s]..[s+1] (device int32 [S+1]; no edge leaves its segment):
 * one workgroup per (segment, orientation) instead of one per orientation -- the stacked batch of a pre-training step.
 * max_seg_rows / max_seg_edges: host-side maxima (LDS sizing; GMP_ERR_UNSUPPORTED when a segment does not fit).  An edge with
 * an endpoint outside its segment is dropped and counted in status. */
int gmp_csr_build_segmented(const int64_t* edge_index, int64_t num_nodes, int64_t num_edges, const int32_t* seg_row_ptr,
                            const int32_t* seg_edge_ptr, int num_segments, int64_t max_seg_rows, int64_t max_seg_edges,
                            int32_t* rowptr, int32_t* col, int32_t* perm, int32_t* rowptr_t, int32_t* col_t, int32_t* perm_t,
                            int32_t* status, gmp_stream_t stream);

/* ------------------------------------------------------------------------- *
 * GIN neighbourhood aggregation (the SpMM-style kernel of BASELINE.json).
 *   fwd: out[i,:] = (1 + eps) * x[i,:] + sum_{k in rowptr[i]..rowptr[i+1]} x[col[k],:]
 *        == PyG GINConv before its nn (src/models/gnn.py:29-41).
 *   bwd: g_x[i,:] = (1 + eps) * g_out[i,:] + sum over the TRANSPOSED CSR of g_out;
 *        g_eps[0] = sum_i <g_out[i,:], x[i,:]>  (deterministic two-stage reduction).
 *   eps is a device float[1] (the learnable GINConv.eps, shape [1]).
 * Algorithmic bytes per launch: 2*4*F*N + 4*(N+1) + 4*E.
 * ------------------------------------------------------------------------- */
int gmp_gin_aggregate_fwd(const float* x, const int32_t* rowptr, const int32_t* col, const float* eps,
                          float* out, int64_t num_nodes, int feat, gmp_stream_t stream);
/* rows [row0, row1) of gmp_gin_aggregate_fwd (x, rowptr, col in the whole batch's numbering; cache-resident sizes) */
int gmp_gin_aggregate_fwd_rows(const float* x, const int32_t* rowptr, const int32_t* col, const float* eps, float* out,
                               int64_t row0, int64_t row1, int feat, gmp_stream_t stream);
size_t gmp_gin_aggregate_bwd_workspace_bytes(int64_t num_nodes, int feat);
int gmp_gin_aggregate_bwd(const float* g_out, const int32_t* rowptr_t, const int32_t* col_t, const float* eps,
                          const float* x, float* g_x, float* g_eps, int64_t num_nodes, int feat,
                          void* workspace, size_t workspace_bytes, gmp_stream_t stream);

/* Stacked-pass backward: g_x = (1+eps) g + sum over the transposed CSR of g (+ addend: the residual
 * branch's gradient), rowdot[r] = <g[r,:], x[r,:]> (nullable) so that the eps gradient can be reduced
 * per task with gmp_group_sum_1d: out[out_off_host[g]] = sum vals[rows[g] .. rows[g+1]). */
int gmp_gin_aggregate_bwd_ex(const float* g_out, const int32_t* rowptr_t, const int32_t* col_t, const float* eps,
                             const float* x, const float* addend, float* g_x, float* rowdot, int64_t num_nodes,
                             int feat, gmp_stream_t stream);
int gmp_group_sum_1d(const float* vals, int groups, const int32_t* group_rows_host, const int64_t* out_off_host,
                     float* out, gmp_stream_t stream);

/* ------------------------------------------------------------------------- *
 * Generic segmented row sum:  out[r,:] (+)= scale_r * sum_{k in ptr[r]..ptr[r+1]} src[idx ? idx[k] : k, :]
 *   mean != 0  -> scale_r = 1 / max(ptr[r+1]-ptr[r], 1)   (PyG scatter 'mean')
 * Serves global_mean_pool (src/pretrain/tasks.py:241,244,299,331; finetune_model.py:75)
 * and the scatter-add backward of row gathers h[idx] (tasks.py:80, heads.py:59-60).
 * ------------------------------------------------------------------------- */
int gmp_segment_sum(const float* src, const int32_t* ptr, const int32_t* idx, float* out,
                    int64_t num_segments, int feat, int mean, int accumulate, gmp_stream_t stream);

/* out[m,:] = scale_m * src[idx[m],:];  scale_m = 1, or 1/max(count[idx[m]],1) when
 * seg_ptr != NULL (backward of global_mean_pool: idx = batch vector). idx int64. */
int gmp_row_gather(const float* src, const int64_t* idx, const int32_t* seg_ptr, float* out,
                   int64_t num_out_rows, int64_t num_src_rows, int feat, gmp_stream_t stream);

/* ------------------------------------------------------------------------- *
 * global_max_pool (tasks.py:242,245): out[b,f] = max over rows ptr[b]..ptr[b+1];
 * empty segment -> 0 (PyG: new_zeros + scatter_reduce amax, include_self=False).
 * bwd splits g evenly between tied maxima, as torch's scatter_reduce('amax') does
 * (including its count of the zero-initialised output as one more tie when max == 0).
 * ------------------------------------------------------------------------- */
int gmp_segment_max_fwd(const float* x, const int32_t* ptr, float* out, int64_t num_segments, int feat,
                        gmp_stream_t stream);
int gmp_segment_max_bwd(const float* g_out, const float* x, const float* out, const int32_t* ptr,
                        float* g_x, int64_t num_segments, int feat, int accumulate, gmp_stream_t stream);

/* ------------------------------------------------------------------------- *
 * Dense fp32 GEMM on the f32 MFMA (v_mfma_f32_32x32x2_f32), C = alpha * op(A) op(B) (+ bias)(+ C).
 * nn.Linear forward / input-grad / weight-grad of the GIN MLP and the heads
 * (gnn.py:14,31,34; heads.py:42).
 *   GMP_GEMM_NT: C[M,N] = A[M,K] * B[N,K]^T      (Linear forward: x W^T)
 *   GMP_GEMM_NN: C[M,N] = A[M,K] * B[K,N]        (input grad: g W)
 *   GMP_GEMM_TN: C[M,N] = A[K,M]^T * B[K,N]      (weight grad: g^T x), reduction over rows
 *   bias: NULL or [N], added to every row.  accumulate != 0 -> C += result.
 *   relu != 0 -> C = max(C, 0) after bias (MLPHead hidden layers).
 * ------------------------------------------------------------------------- */
#define GMP_GEMM_NT 0
#define GMP_GEMM_NN 1
#define GMP_GEMM_TN 2
/* workspace: optional; when the output tile count cannot fill the chip and K is long
 * (weight gradients) the kernel splits K over blockIdx.z into this buffer and sums the
 * slices in order (deterministic).  0 bytes / NULL -> no split-K. */
size_t gmp_gemm_f32_workspace_bytes(int mode, int64_t M, int64_t N, int64_t K);
int gmp_gemm_f32(int mode, const float* A, const float* B, const float* bias, float* C,
                 int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc,
                 float alpha, int accumulate, int relu, void* workspace, size_t workspace_bytes,
                 gmp_stream_t stream);
/* Grouped form: `groups` (<= 24) independent problems in ONE launch (blockIdx.z = group), row ranges
 * group_rows_host[g] .. group_rows_host[g+1] (HOST int32, ascending):
 *   NT / NN: group g multiplies ITS rows of A by ITS OWN weight matrix B + b_off_host[g] (+ bias +
 *            bias_off_host[g]) into the same rows of C  -- the per-domain heads of one task
 *            (pretrain_model.py:41-63) in one launch;
 *   TN:      group g reduces over ITS rows of A [rows, M_tn] and B [rows, N] into C + c_off_host[g]
 *            -- per-task weight gradients of the stacked backbone pass (one group per task, the
 *            offsets point into the per-task gradient buffer PCGrad reads).
 *            a_colsum (nullable): a_colsum + a_colsum_off_host[g] + m receives sum over the group's rows of
 *            A[:, m] -- the bias gradient rides along with the weight gradient for free.
 *            workspace (nullable): lets TN slice each group's rows over several workgroups (partials summed
 *            in slice order by a second kernel) when the output alone cannot fill the chip.
 * Offsets are in floats; NULL offset arrays mean 0. */
int gmp_gemm_f32_grouped(int mode, const float* A, const float* B, const float* bias, float* C, int groups,
                         const int32_t* group_rows_host, const int64_t* b_off_host, const int64_t* bias_off_host,
                         const int64_t* c_off_host, float* a_colsum, const int64_t* a_colsum_off_host,
                         int64_t M_tn, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, float alpha,
                         int accumulate, int relu, void* workspace, size_t workspace_bytes, gmp_stream_t stream);
/* column sums: out[n] (+)= sum_m A[m,n]  (bias gradient), rows [0,M) */
size_t gmp_colsum_workspace_bytes(int64_t M, int64_t N);
int gmp_colsum(const float* A, float* out, int64_t M, int64_t N, int64_t lda, int accumulate,
               void* workspace, size_t workspace_bytes, gmp_stream_t stream);

/* ------------------------------------------------------------------------- *
 * Segment-wise BatchNorm1d (+ residual, ReLU, dropout) -- gnn.py:15,19-22,32,38,42-43.
 * A "segment" is one reference forward() call's worth of rows: statistics are
 * taken per segment so several independent forwards can share one launch.
 *   seg_ptr     device int32 [S+1] row offsets; max_seg_rows = host-side max length.
 *   training:   mean/biased var over the segment (eps 1e-5), saved to save_mean /
 *               save_rstd [S,C]; running_mean/var (nullable) are updated segment by
 *               segment in order with momentum 0.1 and the unbiased variance,
 *               num_batches_tracked is the caller's business.
 *   eval:       uses running_mean / running_var.
 *   y = dropout(relu?( gamma * (x [+ residual] - mean) * rstd + beta ))
 *   dropout keep-mask comes from Philox(seed, stream_id, element) and is NOT stored:
 *   the backward regenerates it.  p = 0 disables dropout.
 * ------------------------------------------------------------------------- */
typedef struct {
    int training;        /* batch statistics vs running statistics */
    int relu;            /* apply ReLU after the affine transform */
    float eps;           /* 1e-5 */
    float momentum;      /* 0.1 */
    float dropout_p;     /* 0 = none */
    uint64_t seed;       /* dropout seed */
    uint32_t stream_id;  /* distinct per dropout site */
    const uint64_t* seed_dev;  /* NULL, or a DEVICE word added to `seed` when the kernel runs: a launch captured in a hipGraph draws fresh
                                  masks on every replay (the caller bumps the word between replays, gmp_counter_add) */
    int32_t* sync;       /* NULL, or a DEVICE buffer of gmp_bn_sync_bytes, zero-filled by the caller ONCE and private to the stream the
                            call runs on: segments of 1,025..4,096 rows (the Cora graph) are then cut into 128-row slabs over the whole chip
                            that exchange their partial statistics through this buffer inside ONE launch, instead of one workgroup per
                            (segment, column strip).  sync[0] != 0 afterwards = a wait timed out (buffer not zeroed, or shared by streams) */
    uint32_t sync_words; /* int32 words behind `sync`; fewer than gmp_bn_sync_bytes asks for = the slab form is not used */
} gmp_bn_config;

size_t gmp_bn_workspace_bytes(int64_t rows, int channels, int num_segments, int64_t max_seg_rows);
size_t gmp_bn_sync_bytes(int channels, int num_segments);
/* seg_group: NULL, or device int32 [S] giving each segment's PARAMETER GROUP: gamma, beta and the running
 * statistics are then [groups][C] arrays (the four per-domain input encoders stacked in one launch). */
int gmp_bn_fwd(const float* x, const float* residual, const int32_t* seg_ptr, const int32_t* seg_group,
               int num_segments, int64_t max_seg_rows, int64_t rows, int channels,
               const float* gamma, const float* beta, float* running_mean, float* running_var,
               float* save_mean, float* save_rstd, float* y,
               const gmp_bn_config* cfg, void* workspace, size_t workspace_bytes, gmp_stream_t stream);
/* Training-mode gmp_bn_fwd with running_mean == NULL leaves the running statistics alone; this applies that update later
 * from the saved batch statistics (segments in order, as S successive forward() calls would: bit-identical). */
int gmp_bn_running_update(const int32_t* seg_ptr, const int32_t* seg_group, int num_segments, int channels,
                          float* running_mean, float* running_var, const float* save_mean, const float* save_rstd,
                          const gmp_bn_config* cfg, gmp_stream_t stream);
/* The same for `count` (<= 16) BatchNorms that share seg_ptr, in one launch: HOST arrays of device pointers / channel counts
 * (seg_group: NULL, or per entry NULL / device int32 [S]). */
int gmp_bn_running_update_batch(int count, const int32_t* seg_ptr, int num_segments, const int32_t* const* seg_group,
                                const int32_t* channels, float* const* running_mean, float* const* running_var,
                                const float* const* save_mean, const float* const* save_rstd, const gmp_bn_config* cfg,
                                gmp_stream_t stream);
/* backward: given g_y, the BN input u = x (+ residual, recomputed on the fly), saved stats.
 *   g_u [rows,C]   gradient w.r.t. the BN input (also the residual's gradient)
 *   g_gamma/g_beta: per GRADIENT group sums (<= 24 groups, one launch); group g covers segments
 *   grp_seg_ptr_host[g]..[g+1] (host int32 [G+1]) and is written at g_gamma + grp_off_gamma_host[g]
 *   (float offsets, host int64 [G]; NULL -> g*C).  G=1 -> ordinary gradients; G=#tasks with offsets
 *   into the per-task gradient buffer -> the per-task gradients PCGrad needs, from ONE backward. */
int gmp_bn_bwd(const float* g_y, const float* x, const float* residual, const int32_t* seg_ptr,
               const int32_t* seg_group, int num_segments, int64_t max_seg_rows, int64_t rows, int channels,
               const float* gamma, const float* beta, const float* running_mean, const float* running_var,
               const float* save_mean, const float* save_rstd,
               float* g_u, float* g_gamma, float* g_beta, const int32_t* grp_seg_ptr_host,
               const int64_t* grp_off_gamma_host, const int64_t* grp_off_beta_host, int num_groups,
               const gmp_bn_config* cfg, void* workspace, size_t workspace_bytes, gmp_stream_t stream);

/* gmp_bn_bwd with num_groups = 0 leaves the per-segment sums at the start of its workspace; this turns them into the per-group
 * g_gamma / g_beta later, on any stream (bit-identical to passing the groups to gmp_bn_bwd). */
int gmp_bn_param_grads(const void* bwd_workspace, int num_segments, int channels, float* g_gamma, float* g_beta,
                       const int32_t* grp_seg_ptr_host, const int64_t* grp_off_gamma_host, const int64_t* grp_off_beta_host,
                       int num_groups, gmp_stream_t stream);

/* ------------------------------------------------------------------------- *
 * Link-prediction edge features (the per-edge MLP input, heads.py:58-66):
 *   feat[k,:] = [ hs+hd | hs*hd | |hs-hd| ],  hs = h[edges[0,k]], hd = h[edges[1,k]]
 * bwd produces per-edge gradients g_hs, g_hd [K,F]; the caller reduces them onto
 * nodes with gmp_segment_sum over a CSR of the decoder edges.
 * ------------------------------------------------------------------------- */
int gmp_lp_edge_features_fwd(const float* h, const int64_t* edges, float* feat, int64_t num_nodes,
                             int64_t num_edges, int feat_dim, gmp_stream_t stream);
int gmp_lp_edge_features_bwd(const float* g_feat, const float* h, const int64_t* edges, float* g_hs,
                             float* g_hd, int64_t num_nodes, int64_t num_edges, int feat_dim,
                             gmp_stream_t stream);

/* ------------------------------------------------------------------------- *
 * NT-Xent / InfoNCE (tasks.py:192-213, 265-287).
 *   z = [normalize(z1); normalize(z2)]  (eps 1e-12), sim = z z^T / T, diag = -inf,
 *   loss_sum = sum_i CE(sim[i,:], pos_i),  pos_i = (i + n) mod 2n.
 * fwd writes loss_sum (device float[1]); bwd returns d loss_sum / d z1, d z2 scaled
 * by `g_scale` (device float[1], the upstream gradient).  n <= 8192.
 * An index outside [0, num_src_rows) in gmp_row_gather / gmp_lp_edge_features_* reads
 * as a zero row instead of faulting.
 * ------------------------------------------------------------------------- */
size_t gmp_nt_xent_workspace_bytes(int64_t n, int dim);
int gmp_nt_xent_fwd(const float* z1, const float* z2, int64_t n, int dim, float temperature,
                    float* loss_sum, void* workspace, size_t workspace_bytes, gmp_stream_t stream);
int gmp_nt_xent_bwd(const float* z1, const float* z2, int64_t n, int dim, float temperature,
                    const float* g_scale, float* g_z1, float* g_z2,
                    void* workspace /* the one fwd filled */, size_t workspace_bytes, gmp_stream_t stream);

/* ------------------------------------------------------------------------- *
 * MLPHead pieces (heads.py:42-45): Linear -> ReLU -> Dropout.  The ReLU rides in the
 * GEMM epilogue; these two handle the dropout and the fused backward
 *   out = g * dropmask * (act > 0),  act = the ReLU output.  numel % 4 == 0.
 * ------------------------------------------------------------------------- */
int gmp_dropout_fwd(const float* x, float* y, int64_t numel, float p, uint64_t seed, uint32_t stream_id,
                    gmp_stream_t stream);
int gmp_relu_dropout_bwd(const float* g, const float* act, float* out, int64_t numel, float p, uint64_t seed,
                         uint32_t stream_id, gmp_stream_t stream);
/* The Linear(hidden, 1) that ends the link-prediction scorer (src/models/heads.py:45-52, behind ReLU + dropout) as what it is -- a dot
 * product per row, an outer product, a weighted column sum -- instead of three GEMM-path launches with N = 1:
 *   gmp_dropout_rowdot_fwd:      dropped = dropout(x) (written when p > 0; may be NULL otherwise), y[m] = <dropped[m, :], w> + bias[0]
 *   gmp_outer_relu_dropout_bwd:  out[m, c] = g[m] * w[c] * dropout_mask * (act[m, c] > 0)   (same site / seed as the forward)
 *   gmp_weighted_colsum:         out_w[c] = sum_m g[m] * x[m, c], out_b[0] = sum_m g[m] (nullable); two ordered stages, deterministic */
int gmp_dropout_rowdot_fwd(const float* x, const float* w, const float* bias, float* dropped, float* y, int64_t rows, int feat, float p,
                           uint64_t seed, uint32_t stream_id, gmp_stream_t stream);
int gmp_outer_relu_dropout_bwd(const float* g, const float* w, const float* act, float* out, int64_t rows, int feat, float p,
                               uint64_t seed, uint32_t stream_id, gmp_stream_t stream);
size_t gmp_weighted_colsum_workspace_bytes(int64_t rows, int feat);
int gmp_weighted_colsum(const float* g, const float* x, float* out_w, float* out_b, int64_t rows, int feat, void* workspace,
                        size_t workspace_bytes, gmp_stream_t stream);

/* The same layer over MERGED link-prediction rows.  The scorer's features [hs + hd, hs * hd, |hs - hd|] (src/models/heads.py:57-61) are
 * symmetric in (src, dst), so the engine sends one row per unordered pair through Linear(768, 256) + ReLU; but the reference scores an
 * ORDERED list (src/pretrain/tasks.py:111-120: positives in both directions, then the negatives) and its Dropout(0.2) (heads.py:44-52) draws
 * an independent mask for every row of that list.  A merged row therefore carries the one or two ordered rows it stands for:
 * pos[m] = position of its first occurrence in the reference's list, pos[rows + m] = of its second, or -1.  Masks are keyed by
 * (seed, stream_id, ordered position * feat / 4 + column quad) -- the key gmp_dropout_rowdot_fwd uses when it is run over the ordered list --
 * so every ordered row gets the mask, the score and the loss term the unmerged path gives it (scores bit for bit).
 *   gmp_lp_pair_rowdot_fwd:       y2[m] / y2[rows + m] = <dropout_{first / second}(x[m, :]), w> + bias[0]   (second = first when p == 0)
 *   gmp_lp_pair_sigmoid_bce_fwd_bwd (losses): loss = sum over ordered rows of BCE(sigmoid(y2), sign[m] > 0) with torch's -100 clamp,
 *                                 g_y2 = d (g_scale * loss) / d y2 (zero where there is no second occurrence), p_out nullable [2 rows]
 *   gmp_lp_pair_outer_bwd:        out[m, c] = (act[m, c] > 0) * w[c] * (g_y2[m] * mask_first + g_y2[rows + m] * mask_second)
 *   gmp_lp_pair_weighted_colsum:  out_w[c] = sum_m act[m, c] * (g_y2[m] * mask_first + g_y2[rows + m] * mask_second),
 *                                 out_b[0] = sum_m g_y2[m] + g_y2[rows + m]  (nullable); two ordered stages, deterministic
 * feat % 4 == 0, feat <= 256.  workspace >= gmp_lp_pair_colsum_workspace_bytes(rows, feat). */
int gmp_lp_pair_rowdot_fwd(const float* x, const float* w, const float* bias, const int32_t* pos, float* y2, int64_t rows, int feat, float p,
                           uint64_t seed, uint32_t stream_id, gmp_stream_t stream);
int gmp_lp_pair_sigmoid_bce_fwd_bwd(const float* y2, const float* sign, const int32_t* pos, int64_t rows, const float* g_scale, float* loss,
                                    float* p_out, float* g_y2, void* workspace, size_t workspace_bytes, gmp_stream_t stream);
int gmp_lp_pair_outer_bwd(const float* g_y2, const float* w, const float* act, const int32_t* pos, float* out, int64_t rows, int feat, float p,
                          uint64_t seed, uint32_t stream_id, gmp_stream_t stream);
size_t gmp_lp_pair_colsum_workspace_bytes(int64_t rows, int feat);
int gmp_lp_pair_weighted_colsum(const float* g_y2, const float* act, const int32_t* pos, float* out_w, float* out_b, int64_t rows, int feat,
                                float p, uint64_t seed, uint32_t stream_id, void* workspace, size_t workspace_bytes, gmp_stream_t stream);

/* Grouped form: the per-domain problems of one contrastive task (tasks.py:192-213, 265-287 loop over domains) in 7 launches
 * instead of 7 per domain; forward and backward in one call.  Group g's 2*n_host[g] rows [z1; z2] start at row
 * row_off_host[g] of z [*, dim] and its gradient goes to the same rows of g_z; loss_sums[g] (nullable) receives the group's
 * loss sum, loss_total (nullable) their sum in group order.  groups <= 8; n may be 0 for a group.  Per group the gradients
 * are bit for bit those of gmp_nt_xent_fwd / _bwd (zero padding adds nothing); the loss sums add the same row losses in another order. */
size_t gmp_nt_xent_grouped_workspace_bytes(int groups, int64_t max_n, int dim);
int gmp_nt_xent_grouped(const float* z, float* g_z, int groups, const int32_t* n_host, const int64_t* row_off_host, int dim,
                        float temperature, const float* g_scale, float* loss_sums, float* loss_total, void* workspace,
                        size_t workspace_bytes, gmp_stream_t stream);

/* ------------------------------------------------------------------------- *
 * Task losses (sum reductions; the caller divides by the pooled size, tasks.py:84-93).
 *   mse_sum          F.mse_loss(a, b, reduction='sum')               tasks.py:83,305
 *   sigmoid          torch.sigmoid of the edge scores                heads.py:67
 *   bce_sum          F.binary_cross_entropy(p, y, 'sum'), log terms clamped at -100,
 *                    backward g*(p-y)/max(p(1-p),1e-12) as torch does  tasks.py:120
 *   cross_entropy_sum F.cross_entropy(logits, target, 'sum')         tasks.py:336, finetune.py:158,177
 *   *_bwd take the upstream scalar gradient as device float[1] g_scale.
 *   workspace >= gmp_loss_workspace_bytes(numel or rows).
 * gmp_row_fill: dst[idx[m],:] = broadcast ? src[0,:] : src[m,:]  (mask-token rows,
 *   pretrain_model.py:82-85); out-of-range idx entries are skipped.
 * ------------------------------------------------------------------------- */
size_t gmp_loss_workspace_bytes(int64_t numel);
int gmp_mse_sum_fwd(const float* a, const float* b, int64_t numel, float* loss, void* workspace,
                    size_t workspace_bytes, gmp_stream_t stream);
int gmp_mse_sum_bwd(const float* a, const float* b, const float* g_scale, float* g_a, int64_t numel,
                    gmp_stream_t stream);
int gmp_sigmoid_fwd(const float* x, float* y, int64_t numel, gmp_stream_t stream);
int gmp_sigmoid_bwd(const float* g, const float* y, float* out, int64_t numel, gmp_stream_t stream);
int gmp_bce_sum_fwd(const float* p, const float* labels, int64_t numel, float* loss, void* workspace,
                    size_t workspace_bytes, gmp_stream_t stream);
int gmp_bce_sum_bwd(const float* p, const float* labels, const float* g_scale, float* g_p, int64_t numel,
                    gmp_stream_t stream);
/* sigmoid_fwd + bce_sum_fwd + bce_sum_bwd + sigmoid_bwd in one pass over the scores x (same arithmetic, element by element):
 * loss = BCE(sigmoid(x), labels) summed, g_x = d (g_scale * loss) / d x; p_out (nullable) receives sigmoid(x). */
int gmp_sigmoid_bce_sum_fwd_bwd(const float* x, const float* labels, int64_t numel, const float* g_scale, float* loss,
                                float* p_out, float* g_x, void* workspace, size_t workspace_bytes, gmp_stream_t stream);
/* The same pass over pairs that carry a multiplicity: signed_weight[i] = +w for a positive pair, -w for a negative pair, w >= 1 the
 * number of times the pair stands in the reference's list (the scorer of src/models/heads.py:57-67 is symmetric in (src, dst), so
 * (i, j) and (j, i) need scoring once); loss term and g_x[i] are scaled by w.  With w = 1 everywhere: the call above, bit for bit. */
int gmp_sigmoid_bce_signed_sum_fwd_bwd(const float* x, const float* signed_weight, int64_t numel, const float* g_scale, float* loss,
                                       float* p_out, float* g_x, void* workspace, size_t workspace_bytes, gmp_stream_t stream);
int gmp_cross_entropy_sum_fwd(const float* logits, const int64_t* target, int64_t rows, int classes,
                              float* loss, void* workspace, size_t workspace_bytes, gmp_stream_t stream);
int gmp_cross_entropy_sum_bwd(const float* logits, const int64_t* target, int64_t rows, int classes,
                              const float* g_scale, float* g_logits, gmp_stream_t stream);
int gmp_row_fill(float* dst, const int64_t* idx, const float* src, int64_t num_idx, int64_t num_dst_rows,
                 int feat, int broadcast, gmp_stream_t stream);

/* ------------------------------------------------------------------------- *
 * Hard-negative mining for link-prediction fine-tuning
 * (finetune.py:45-75, LinkPredictionHardNegativeMiner.mine_hard_negatives_for_edges; called per
 * training batch from finetune.py:190-196).
 *   zn = F.normalize(emb, dim=1) (eps 1e-12); S = zn zn^T; pairs (s,d) and (d,s) of
 *   existing_edges ([2,E] int64, row 0 = sources) and the diagonal are excluded;
 *   out_edges [2,k] int64 = the k highest-scoring remaining ordered pairs (i,j), sorted by score
 *   descending -- what torch.topk(potential_scores, k) + potential_indices gives.  Ties (S is
 *   symmetric, so every score appears twice) are broken by the LOWER flat index i*n+j; torch.topk
 *   leaves that order unspecified.
 *   out_scores: NULL or [k] float, the selected scores.  scores_out: NULL or [n,n] float, receives the
 *   masked similarity matrix (masked entries = -inf) -- test / diagnostic hook.
 *   The caller guarantees k <= number of unmasked pairs (finetune.py:66-67 clamps it); n <= 65535,
 *   k <= 4096.  workspace >= gmp_hard_negative_workspace_bytes(n, dim).
 * ------------------------------------------------------------------------- */
size_t gmp_hard_negative_workspace_bytes(int64_t n, int64_t dim);
int gmp_hard_negative_topk(const float* emb, int64_t n, int64_t dim, const int64_t* existing_edges, int64_t E,
                           int64_t k, int64_t* out_edges, float* out_scores, float* scores_out,
                           void* workspace, size_t workspace_bytes, gmp_stream_t stream);

/* ------------------------------------------------------------------------- *
 * Pack / unpack a fixed list of slices of one buffer into one contiguous message (the data-parallel exchange of per-task
 * gradients: only the (task, tensor) pairs that carry a gradient travel).  table_dev: device int64 [2n+1] = n source offsets
 * (floats into base) followed by the n+1 exclusive prefix sums of the slice lengths; offsets and lengths are multiples of 4;
 * n <= 256; total = prefix[n].  unpack multiplies by `scale` (1 / world size).
 * ------------------------------------------------------------------------- */
int gmp_segments_pack(const float* base, float* packed, const int64_t* table_dev, int n, int64_t total, gmp_stream_t stream);
int gmp_segments_unpack(float* base, const float* packed, const int64_t* table_dev, int n, int64_t total, float scale,
                        gmp_stream_t stream);

/* ------------------------------------------------------------------------- *
 * Host -> device upload of up to 4 small arrays by one kernel (the per-step index arrays: the reference builds them on the
 * CPU, masks / negatives / views at pretrain_model.py:72-80, tasks.py:107-111, augmentations.py:17-111, and moves them with
 * .to(device)).  src_pinned_host[i]: PINNED host memory (hipHostMalloc / torch pin_memory: mapped into the device's address
 * space), dst[i]: device memory, bytes[i]: multiples of 16, all pointers 16-byte aligned; the three arrays are host arrays.
 * The caller keeps a source buffer untouched until the stream has passed the call.
 * ------------------------------------------------------------------------- */
int gmp_upload(int n, const void* const* src_pinned_host, void* const* dst, const int64_t* bytes, gmp_stream_t stream);

/* ------------------------------------------------------------------------- *
 * Streams and hardware queues (no reference counterpart: the reference runs everything on one stream).  The runtime maps a
 * process's HIP streams onto a few hardware queues (4 by default) and packets of one queue run in order, so two streams on
 * the same queue never overlap.  gmp_streams_share_queue MEASURES it: a 400 us spin kernel on `a`, an empty kernel on `b`;
 * *share_host = 1 when b's kernel only finished with a's (same queue, or a == b), 0 when it ran beside it.  Synchronises both
 * streams (start-up calibration, not a per-step call).  gmp_spin_us enqueues the busy-wait kernel (diagnostics).
 * ------------------------------------------------------------------------- */
int gmp_streams_share_queue(gmp_stream_t a, gmp_stream_t b, int* share_host);
int gmp_spin_us(int microseconds, gmp_stream_t stream);
/* Device-side gates: cross-stream dependencies without barrier packets.  Measured (scripts/diag_blocked_queues.py): every
 * hardware queue parked on a hipStreamWaitEvent adds ~2 us to every kernel boundary of the queues that are running; a gate --
 * one wave sleeping on flag words -- costs them nothing.  gmp_gate_wait parks one wave on `stream` until flags[i] >= want for
 * every bit i of mask (flags: device int32[64]); after two minutes it gives up and ORs 1 into *err (may be NULL).
 * gmp_gate_open stores `value` to *flag from a kernel on its stream (after everything enqueued there before).  ONLY between
 * streams on different hardware queues (gmp_streams_share_queue): in one in-order queue a gate ahead of its opener never opens. */
int gmp_gate_wait(const int32_t* flags, uint64_t mask, int want, int32_t* err, gmp_stream_t stream);
int gmp_gate_open(int32_t* flag, int value, gmp_stream_t stream);
/* The same signal without a launch of its own: the NEXT gmp_gemm_f32 / gmp_gemm_f32_grouped call made by this host thread stores `value` to
 * *flag when its first workgroup starts -- stream order has then retired everything enqueued on that GEMM's stream before it -- which takes
 * the one-thread launch (4 us + a kernel boundary) off a critical chain (flag NULL: cancel).  gmp_gate_open_pending: 1 while a signal waits
 * for its GEMM (a GEMM call that launched nothing leaves it pending: open the gate with gmp_gate_open then). */
int gmp_gate_open_by_next_gemm(int32_t* flag, int value);
int gmp_gate_open_pending(void);
/* Time-out of the gates enqueued from now on (default: GMP_GATE_TIMEOUT_S seconds, else 120; fractions allowed).  A gate that
 * times out sets *err and lets its stream go on; the engine's data-parallel start-up check (StepEngine.verify_gates) runs
 * its probe steps with a short one so that a stream layout that cannot carry gates shows within seconds. */
int gmp_gate_set_timeout(double seconds);
/* *word += inc on the device, in stream order (one thread).  With gmp_bn_config.seed_dev: a step captured in a hipGraph ends with this
 * launch, so every replay draws the dropout masks of the next step number -- the masks an eager run of that step would draw. */
int gmp_counter_add(uint64_t* word, uint64_t inc, gmp_stream_t stream);

/* ------------------------------------------------------------------------- *
 * Stacked input encoders (InputEncoder.linear, gnn.py:14,19) for every segment of a step in one
 * launch.  Features of all domains sit padded to `dpad` (<= 64) columns in x_all [R, dpad]; stacked row
 * r reads x_all[src_row[r]]; segment s belongs to domain seg_dom[s] (weights at params + w_off_host[d],
 * shape [256, d_in_host[d]], bias at params + b_off_host[d]); row_colmask[r] (nullable) bit k = feature k of stacked row r zeroed
 * (attribute-mask augmentation, augmentations.py:17-29).  tiles [num_tiles][2] = (segment, first row),
 * 32 rows each; indices outside [0,num_x_rows) / [0,num_rows) / [0,num_segments) read as zero rows.
 * bwd: one gradient group per (task, domain) pair covering segments
 * group_seg_host[g]..[g+1]; dW lands at grad_out + off_w_host[g] ([256, d_in]), db at + off_b_host[g].
 * ------------------------------------------------------------------------- */
int gmp_encoder_fwd(const float* x_all, int64_t num_x_rows, int64_t num_rows, int num_segments, const int32_t* src_row, const int32_t* seg_ptr, const int32_t* seg_dom,
                    const uint64_t* row_colmask, const int32_t* tiles, int num_tiles, const float* params,
                    int num_domains, const int64_t* w_off_host, const int64_t* b_off_host,
                    const int32_t* d_in_host, int dpad, float* z, gmp_stream_t stream);
int gmp_encoder_bwd(const float* x_all, int64_t num_x_rows, int64_t num_rows, int num_segments, const int32_t* src_row, const int32_t* seg_ptr, const int32_t* seg_dom,
                    const uint64_t* row_colmask, const float* g_z, int num_domains, const int32_t* d_in_host,
                    int dpad, int groups, const int32_t* group_seg_host, const int64_t* off_w_host,
                    const int64_t* off_b_host, float* grad_out, void* workspace, size_t workspace_bytes,
                    gmp_stream_t stream);

/* ------------------------------------------------------------------------- *
 * Multi-tensor PCGrad + clip_grad_norm_ + AdamW over flat buffers (gradient_surgery.py:41-103,
 * pretrain.py:152-153, optimizers.py:8-75), five launches, deterministic.
 *   task_grads [T][task_stride]: per-task gradients laid out like `params`; tensor k occupies
 *   [tensor_off[k], +tensor_len[k]) (device arrays); has[k*8 + t] != 0 iff task t's backward produced a
 *   gradient for tensor k (device uint8, row stride 8).
 *   order_host[0..n_order): the shuffled task order of PCGrad (gradient_surgery.py:43).
 *   last_task: last task in dict order -- where PCGrad emits nothing (tensor absent from the FIRST
 *   shuffled task) the reference leaves that task's raw .grad in place (gradient_surgery.py:61 quirk);
 *   extra_task: -1 or a task whose gradient is added on top (domain_adv, pretrain.py:149-150).
 *   Tensors that end up without a gradient (flags_out[k] == 0) are skipped by AdamW entirely, including
 *   their step count, as torch.optim does for grad=None.
 *   apply_update == 0: stop after final_grad / normsq_out (unclipped) -- used by parity tests.
 *   Layout contract (the sweeps move float4): every buffer 16-byte aligned, task_stride and every tensor_off multiples of 4,
 *   and each tensor's slot padded to a multiple of 4 floats with zeros in params, task_grads and the optimizer state (the
 *   padding is swept along and stays zero).
 * ------------------------------------------------------------------------- */
size_t gmp_mt_workspace_bytes(int num_tensors);
int gmp_mt_pcgrad_clip_adamw(const float* task_grads, int64_t task_stride, int num_tasks, int num_tensors,
                             const int64_t* tensor_off, const int32_t* tensor_len, const uint8_t* has,
                             const int32_t* order_host, int n_order, int last_task, int extra_task,
                             float* params, float* exp_avg, float* exp_avg_sq, float* steps,
                             const float* lr, const float* wd, float beta1, float beta2, float eps,
                             float max_norm, float* final_grad, float* normsq_out, int32_t* metrics_out,
                             int32_t* flags_out, void* workspace, size_t workspace_bytes, int apply_update,
                             gmp_stream_t stream);
/* The same in pieces, so PCGrad can follow the backward part by part: phases bit 0 = Gram / solve / combine for the tensors
 * [k_begin, k_end) only (their final_grad, flags, step counts, norm partials), bit 1 = total norm + clip + AdamW over ALL tensors
 * (call it once, after bit 0 has covered every tensor; same stream order or an explicit dependency in between).  Results are
 * bitwise those of the one-shot call: every tensor's arithmetic is independent of the others up to the norm.
 * bit 2 (value 4), for data-parallel runs that shard PCGrad over the ranks: the tensors [k_begin, k_end) are FOREIGN -- their combined
 * gradient was computed by another rank and already stands in final_grad (all-gathered); this leaves their flag, step count and norm
 * partials as the owner's bit 0 pass left them there, without touching task_grads (gnn_pretraining_amd/dist.py ShardedGradSync). */
int gmp_mt_pcgrad_clip_adamw_ex(const float* task_grads, int64_t task_stride, int num_tasks, int num_tensors,
                                const int64_t* tensor_off, const int32_t* tensor_len, const uint8_t* has,
                                const int32_t* order_host, int n_order, int last_task, int extra_task,
                                float* params, float* exp_avg, float* exp_avg_sq, float* steps,
                                const float* lr, const float* wd, float beta1, float beta2, float eps,
                                float max_norm, float* final_grad, float* normsq_out, int32_t* metrics_out,
                                int32_t* flags_out, void* ws, size_t ws_bytes, int apply_update,
                                int k_begin, int k_end, int phases, const int32_t* abort_flag, gmp_stream_t stream);
/* abort_flag (nullable, device): when *abort_flag != 0 at run time neither parameters, moments nor step counters are touched -- the
 * engine passes the error word of its cross-stream gates (sync_flags[63]), so a gate that timed out cannot turn into an update
 * computed from incomplete gradients; the host sees the flag at its next check_gates(). */


/* ---- device-side augmentation and masking ------------------------------------------------------------
 * The per-graph loops of src/pretrain/augmentations.py:17-111 (GraphAugmentor.create_two_views: node drop, subgraph relabel,
 * edge drop, attribute mask, common-node masks) and of src/models/pretrain_model.py:71-80 (mask indices) over a whole domain batch.
 * Random subsets are "the k smallest Philox keys of the graph" (seed, stream_id, element id): the reference's distributions, not its
 * mt19937 stream -- the bit-exact replay of that stream stays on the host (csrc_host/hostdraw.cpp).  All index arrays are int64
 * like the reference's; graphs are given as offset arrays (ptr / eptr: [G + 1]) over a batch in batch-local numbering.
 *
 * gmp_aug_node_masks: out_idx[out_ptr[g] ...] = the max(1, int(.15 n_g)) masked nodes of graph g (n_g >= 3), ascending; out_ptr is the
 *   caller's exclusive scan of those counts (they depend on the graph sizes only).
 * gmp_aug_two_views: both views of every graph.  view_ptr = exclusive scan of the kept-node counts n_g - max(1, int(.2 n_g)) (n_g >= 3,
 *   else n_g), the same for both views.  Outputs per view: rows [view_ptr[G]] (kept nodes, batch numbering, ascending), rowmask
 *   (bit c = column c zeroed, per row), edges [2, edge_capacity] (view numbering; PyG subgraph(relabel_nodes=True) order, minus the
 *   dropped ones), common (view-local ids of nodes kept in BOTH views); counts [5 G] = edges of view 1 / view 2 / common nodes /
 *   "drew an attribute mask" in view 1 / view 2, per graph; totals_and_flags [5] = total edges of view 1, of view 2, total common nodes, "some graph of view 1 / 2 drew an attribute
 *   mask".  Limits: 4,096 nodes and 8,192 edges per graph, 64 feature columns. */
size_t gmp_aug_workspace_bytes(int64_t num_nodes, int64_t num_edges, int num_graphs);
int gmp_aug_node_masks(const int64_t* ptr, const int64_t* out_ptr, int num_graphs, int64_t max_graph_nodes, uint64_t seed,
                       uint32_t stream_id, int64_t* out_idx, gmp_stream_t stream);
int gmp_aug_two_views(const int64_t* ptr, const int64_t* eptr, const int64_t* edge_index, int64_t num_nodes, int64_t num_edges,
                      const int64_t* view_ptr, int num_graphs, int64_t max_graph_nodes, int64_t max_graph_edges, int num_features,
                      uint64_t seed, uint32_t stream_id, int64_t* rows1, int64_t* rows2, uint64_t* rowmask1, uint64_t* rowmask2,
                      int64_t* edges1, int64_t* edges2, int64_t edge_capacity, int64_t* common1, int64_t* common2, int32_t* counts,
                      int32_t* totals_and_flags, void* workspace, size_t workspace_bytes, gmp_stream_t stream);
/* The same for all the jobs of a step -- one per (task, domain) pair -- in one launch (masks) / two launches (views): host arrays of
 * plain-data jobs whose fields are the arguments of the calls above (seed shared, stream_id per job); results identical to them. */
typedef struct {
    const int64_t *ptr, *out_ptr;
    int32_t num_graphs;
    uint32_t stream_id;
    int64_t* out_idx;
} gmp_aug_masks_job;
typedef struct {
    const int64_t *ptr, *eptr, *edge_index;
    int64_t num_nodes, num_edges;
    const int64_t* view_ptr;
    int32_t num_graphs, num_features;
    uint32_t stream_id;
    int64_t *rows1, *rows2;
    uint64_t *rowmask1, *rowmask2;
    int64_t *edges1, *edges2;
    int64_t edge_capacity;
    int64_t *common1, *common2;
    int32_t *counts, *totals_and_flags;
    void* workspace;
    size_t workspace_bytes;
} gmp_aug_views_job;
int gmp_aug_node_masks_batch(const gmp_aug_masks_job* jobs, int count, int64_t max_graph_nodes, uint64_t seed, gmp_stream_t stream);
int gmp_aug_two_views_batch(const gmp_aug_views_job* jobs, int count, int64_t max_graph_nodes, int64_t max_graph_edges, uint64_t seed,
                            gmp_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GNNMP_H */
