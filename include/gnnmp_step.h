/*
 * libgnnmp -- whole-step entry point: the stacked forward + task heads + stacked backward of one
 * pre-training step (reference src/pretrain/pretrain.py:113-150 over src/pretrain/tasks.py) enqueued by ONE C
 * call, so the host pays one FFI crossing instead of ~300.  It issues exactly the kernel sequence
 * gnn_pretraining_amd/engine.py issues through the per-operator entry points of gnnmp.h (the Python sequence stays
 * as the tested reference: tests/test_gpu_engine.py checks the two give bitwise-identical parameters).
 *
 * All pointers are device pointers unless the field says "host".  Offsets "off_*" are in floats into `flat`
 * (parameters) and "tg_*" in floats into `task_grads` ([tasks][P] per-task gradients).  The descriptor is plain
 * data: fill it, call gmp_pretrain_step_fwd_bwd, then all-reduce task_grads if data-parallel, then
 * gmp_mt_pcgrad_clip_adamw.
 */
#ifndef GNNMP_STEP_H
#define GNNMP_STEP_H

#include "gnnmp.h"

#ifdef __cplusplus
extern "C" {
#endif

#define GMP_STEP_MAX_DOMAINS 8
#define GMP_STEP_MAX_TASKS 8
#define GMP_STEP_LAYERS 5
#define GMP_STEP_MAX_ENC_GROUPS 24

enum { GMP_TASK_NFM = 0, GMP_TASK_LP = 1, GMP_TASK_NC = 2, GMP_TASK_GC = 3, GMP_TASK_GP = 4, GMP_TASK_DA = 5 };

/* per-domain two-layer MLPHead (Linear-ReLU-Dropout-Linear), one row group per domain */
typedef struct {
    int32_t k_in, k_hid, k_out, site;
    int32_t rows[GMP_STEP_MAX_DOMAINS + 1];
    int64_t off_w0[GMP_STEP_MAX_DOMAINS], off_b0[GMP_STEP_MAX_DOMAINS], off_w3[GMP_STEP_MAX_DOMAINS], off_b3[GMP_STEP_MAX_DOMAINS];
    int64_t tg_w0[GMP_STEP_MAX_DOMAINS], tg_b0[GMP_STEP_MAX_DOMAINS], tg_w3[GMP_STEP_MAX_DOMAINS], tg_b3[GMP_STEP_MAX_DOMAINS];
    float *x, *y1, *d1, *y2, *g_out, *g_hid, *g_in;
} gmp_mlp2;

typedef struct {
    int32_t kind;                 /* GMP_TASK_* */
    int32_t row0, row1;           /* this task's rows of the stacked batch */
    float* g_scale;               /* device float: 1 / pooled size (d total / d loss_sum) */
    float* loss_sum;              /* device float: receives the task's loss SUM */
    void* gemm_ws; size_t gemm_ws_bytes;
    void* loss_ws; size_t loss_ws_bytes;
    gmp_mlp2 mlp;                 /* NFM, NC, GC, GP */
    /* gathers: NFM masked rows / NC common rows */
    const int64_t* idx; int64_t num_idx;
    float* nfm_target;
    /* NT-Xent (NC, GC): per-domain pair counts and workspaces; sums land in ntx_sums[d] */
    int32_t ntx_n[GMP_STEP_MAX_DOMAINS];
    void* ntx_ws[GMP_STEP_MAX_DOMAINS]; size_t ntx_ws_bytes[GMP_STEP_MAX_DOMAINS];
    float* ntx_sums;
    float temperature;
    /* read-out pooling (GC, GP) */
    const int32_t* pool_ptr; const int64_t* pool_gid; int32_t pool_B, pool_r0, pool_M;
    float *pool_mean, *pool_max, *g_mean, *g_max;
    const float* labels;          /* GP: [B, 12] */
    /* link prediction */
    int64_t lp_K; const int64_t* lp_edges; const float* lp_labels;
    /* lp_pos: NULL = lp_edges is the reference's ordered list (lp_labels +1 / -1).  Otherwise lp_edges holds one row per unordered pair and
       lp_pos [2, lp_K] names the one or two ordered rows each stands for (gnnmp.h gmp_lp_pair_*: a dropout mask, a score and a loss term per
       ordered row); lp_y2 / lp_p / lp_gy2 then hold 2 lp_K floats and lp_d1 is unused. */
    const int32_t* lp_pos;
    float *lp_feat, *lp_y1, *lp_d1, *lp_y2, *lp_p, *lp_gp, *lp_gy2, *lp_gy1, *lp_gfeat, *lp_ghs, *lp_ghd;
    /* offsets of a head shared by all domains: the LP scorer, and the domain classifier of GMP_TASK_DA */
    int64_t lp_off_w0, lp_off_b0, lp_off_w3, lp_off_b3, lp_tg_w0, lp_tg_b0, lp_tg_w3, lp_tg_b3;
    int32_t lp_site;
    /* domain-adversarial task (scheme s5; tasks.py:315-343, heads.py:16-32,70-82): mean read-out -> gradient reversal
     * (backward scaled by -da_lambda) -> Linear 256->128, ReLU, Dropout(da_dropout), Linear 128->da_classes -> CE(sum)
     * against da_labels (the graph's domain index).  Buffers: mlp.x = pooled, mlp.y1/d1 hidden, mlp.y2 logits,
     * mlp.g_out = d loss / d logits, mlp.g_hid, mlp.g_in = d loss / d pooled. */
    const int64_t* da_labels; int32_t da_classes, da_hidden; float da_lambda, da_dropout;
} gmp_task_desc;

typedef struct {
    int64_t off_eps, off_w1, off_b1, off_g1, off_be1, off_w2, off_b2, off_g2, off_be2;
    int64_t tg_eps[GMP_STEP_MAX_TASKS], tg_w1[GMP_STEP_MAX_TASKS], tg_b1[GMP_STEP_MAX_TASKS], tg_g1[GMP_STEP_MAX_TASKS],
        tg_be1[GMP_STEP_MAX_TASKS], tg_w2[GMP_STEP_MAX_TASKS], tg_b2[GMP_STEP_MAX_TASKS], tg_g2[GMP_STEP_MAX_TASKS],
        tg_be2[GMP_STEP_MAX_TASKS];
    float *rm1, *rv1, *rm2, *rv2;                 /* BatchNorm running statistics */
    float *a, *z1, *r1, *z2, *m1, *s1, *m2, *s2;  /* saved activations / batch statistics */
} gmp_layer_desc;

typedef struct {
    /* sizes of this step */
    int32_t N, E, S, max_seg, num_tiles, num_tasks, num_domains, dpad, training, hidden;
    int32_t max_seg_edges;        /* largest number of edges of one segment (the batch is block diagonal: CSR is built per segment) */
    float dropout_p;
    int32_t dp_exchange;          /* nonzero: record the events gmp_step_wait_grads needs (data-parallel run) */
    int32_t epoch;                /* with sync_flags: strictly increasing from call to call (>= 1) */
    int32_t upload_on_aux;        /* nonzero: the caller enqueued this step's uploads (gmp_upload) on the AUX stream, not on main: main
                                     waits for aux at the start instead of aux for main */
    uint64_t seed;
    int32_t* sync_flags;          /* device int32[64], zeroed once by the caller, or NULL.  Non-NULL = the caller has MEASURED that main,
                                     aux and the task streams sit on different hardware queues (gmp_streams_share_queue): cross-stream
                                     dependencies are then carried by gates (gmp_gate_wait / gmp_gate_open) instead of events.
                                     sync_flags[63] becomes nonzero if a gate ever timed out. */
    /* uploaded index arrays */
    const int32_t *seg_ptr, *seg_dom, *src_row, *tiles;
    const int32_t* seg_eptr;      /* [S+1] first edge of each segment in edge_index */
    const int64_t *edge_index, *rowmask;
    int32_t task_row[GMP_STEP_MAX_TASKS + 1];   /* host: first stacked row of each task */
    int32_t task_seg[GMP_STEP_MAX_TASKS + 1];   /* host: first segment of each task */
    /* graph structure scratch */
    int32_t* csr[6]; int32_t* csr_status; void* csr_ws; size_t csr_ws_bytes;
    int32_t* lp_csr[6]; int32_t* lp_csr_status; void* lp_csr_ws; size_t lp_csr_ws_bytes;
    /* the link-prediction decoder's edge list is block diagonal too (one block per domain: positives + negatives of that domain's
       batch, rows of the task's own segments): lp_seg_ptr / lp_seg_eptr (device int32 [lp_S + 1]) let its CSR be built by one
       workgroup per (domain, orientation) -- gmp_csr_build_segmented over rows [0, lp_rows_end) -- instead of one workgroup for all
       ~30 k edges (0.49 ms).  lp_S = 0: whole-batch build. */
    const int32_t *lp_seg_ptr, *lp_seg_eptr; int32_t lp_S; int64_t lp_max_seg_rows, lp_max_seg_edges, lp_rows_end;
    /* Stacked forward in up to three row ranges: cut k (k = 0, 1) ends a range at segment fwd_cut_seg[k] = row fwd_cut_row[k]
       (ascending; 0 = no cut).  The first range runs on main, the others on task streams (idle until the heads), so one range's
       latency-bound BatchNorm / aggregation launches run beside the others' GEMMs.  Segments are independent through the backbone
       (per-segment statistics, block-diagonal adjacency) and every kernel is element-wise identical under a row split: the result is
       bit-identical to the single pass.  Ignored (one pass on main) unless every range keeps at least 1,024 rows. */
    int32_t fwd_cut_seg[2], fwd_cut_row[2];
    /* parameters and per-task gradients */
    float* flat; int64_t P; float* task_grads;
    /* encoders */
    const float* x_all; int64_t x_rows;
    int64_t enc_off_w[GMP_STEP_MAX_DOMAINS], enc_off_b[GMP_STEP_MAX_DOMAINS]; int32_t enc_d_in[GMP_STEP_MAX_DOMAINS];
    int64_t enc_off_gamma0, enc_off_beta0;
    float *enc_rm, *enc_rv, *enc_mean, *enc_rstd, *z0;
    int32_t enc_groups; int32_t enc_gseg[GMP_STEP_MAX_ENC_GROUPS + 1];
    int64_t enc_tg_w[GMP_STEP_MAX_ENC_GROUPS], enc_tg_b[GMP_STEP_MAX_ENC_GROUPS], enc_tg_gamma[GMP_STEP_MAX_ENC_GROUPS],
        enc_tg_beta[GMP_STEP_MAX_ENC_GROUPS];
    int64_t off_mask_token, tg_mask_token; int32_t nfm_task;      /* nfm_task = -1 when the scheme has no NFM */
    /* backbone */
    float* h[GMP_STEP_LAYERS + 1];
    gmp_layer_desc layer[GMP_STEP_LAYERS];
    float *gA, *gB, *gW, *gW2, *rowdot;
    float* ga;                    /* [max_rows, hidden] gradient w.r.t. a layer's aggregated input (scratch of the backward; a buffer of its
                                     own, so that every forward activation h[0..L], r1, z1, z2 is still intact after a step) */
    float *gB2, *gW3;             /* second copies of gB / gW2: weight-gradient GEMMs of layer l read them on the aux stream
                                     while layer l-1 already writes the other copy */
    float* gu_l[GMP_STEP_LAYERS];   /* [N,256] per layer, or all NULL: g_u of every backward layer in a buffer of its own ... */
    float* gz1_l[GMP_STEP_LAYERS];  /* [N,512] per layer: ... and g_z1, so the aux stream may lag main by any number of layers.
                                       With them `rowdot` must hold GMP_STEP_LAYERS * N floats (one slice per layer). */
    void* bn_ws; size_t bn_ws_bytes;
    void* gemm_ws; size_t gemm_ws_bytes;
    void* loss_ws; size_t loss_ws_bytes;
    gmp_task_desc task[GMP_STEP_MAX_TASKS];
} gmp_step_desc;

size_t gmp_step_desc_size(void);
/* main: stream of the stacked pass; task_streams[t]: one stream per task head (may all equal main);
 * aux: stream for the CSR builds (may equal main).  NOT re-entrant: the call keeps its events, and what gmp_step_wait_grads
 * needs, in process-wide state (one engine steps at a time in a process; engines may alternate between calls). */
int gmp_pretrain_step_fwd_bwd(const gmp_step_desc* desc, gmp_stream_t main, const gmp_stream_t* task_streams,
                              gmp_stream_t aux);
/* Data-parallel exchange beside the backward: make `stream` wait until the per-task gradients of one part of
 * the model are final in task_grads for the most recent gmp_pretrain_step_fwd_bwd of this process.  part 0: the task heads
 * (final before the stacked backward starts); part 1 + k: backbone layer GMP_STEP_LAYERS-1-k; part GMP_STEP_LAYERS: layer 0,
 * mask token and input encoders (end of the backward).  The caller then packs / all-reduces that part on `stream` while
 * the backward of the layers below is still running. */
int gmp_step_wait_grads(int part, gmp_stream_t stream);
/* Diagnostic: with GMP_STEP_TIMING=1 in the environment the call above records events on `main` at step start, forward
 * done, heads joined and backward done; this waits for the last one and returns the three phase durations in ms
 * (forward, heads, backward) of the most recent step. */
int gmp_step_phase_ms(float* out3);
/* The same in GMP_STEP_PHASES intervals: encoders (incl. the wait for the CSR build), forward layers 0..4, heads, backward
 * layers 4..0, below-backbone tail (mask token + encoder backward). */
#define GMP_STEP_PHASES (2 * GMP_STEP_LAYERS + 3)
int gmp_step_phase_detail_ms(float* out);
/* (diagnostic) per task: ms from "stacked forward done" to the head's start / end of its input-gradient half / end of its
 * weight-gradient half: out[3 * task + k]; needs GMP_STEP_TIMING=1 */
int gmp_step_head_ms(float* out, int max_tasks);

#ifdef __cplusplus
}
#endif
#endif /* GNNMP_STEP_H */
