// Whole-step executor: stacked forward, the five task heads (each on its own stream) and the stacked
// backward of one pre-training step, enqueued from C++ so the host crosses the FFI once.  It is a
// transcription of gnn_pretraining_amd/engine.py (_forward, _task_head, _backbone_backward): same entry
// points, same order, same buffers -> bitwise the same result (tests/test_gpu_engine.py).
#include "../../include/gnnmp_step.h"
#include <stdlib.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>

#include "gnnmp_internal.h"

namespace {

#define GMP_TRY(expr)                \
    do {                             \
        int _rc = (expr);            \
        if (_rc != GMP_OK) return _rc; \
    } while (0)

constexpr int H = 256;
constexpr int NEV = 4 + 2 * GMP_STEP_MAX_TASKS + 4 * GMP_STEP_LAYERS + 9;
constexpr int EV_LAYER0_DONE = NEV - 4, EV_HEADS_DONE = NEV - 3, EV_BWD_DONE = NEV - 2;
constexpr int EV_HEAD_PARAMS = 4 + GMP_STEP_MAX_TASKS + 4 * GMP_STEP_LAYERS;   // [task], [MAX_TASKS] = the heads of the main stream: weight-gradient GEMMs done
constexpr int EV_MAIN_HEADS = EV_HEAD_PARAMS + GMP_STEP_MAX_TASKS + 1;           // main: input halves of its own heads done
constexpr int EV_FWD_FORK = EV_MAIN_HEADS + 1, EV_FWD_JOIN = EV_MAIN_HEADS + 2;  // split forward: main -> the other streams, and back [2]
static_assert(EV_FWD_JOIN + 1 < NEV - 4, "event pool too small");
// ev[NEV - 1]: running statistics done (aux)

// Gate flags (d.sync_flags, int32[64], all compared against the step's epoch): the same dependencies as the events above, carried
// by sleeping waves instead of barrier packets when the caller vouches that the streams sit on different hardware queues.
// A parked barrier packet costs every RUNNING queue ~2 us per kernel boundary (streams.hip); with the host several steps ahead
// two or three of the four queues were parked most of the time.
enum {
    F_START = 0,                       // main -> aux (or aux -> main, upload_on_aux): the step's uploads are done
    F_FWD = 1,                         // main -> head streams: stacked forward done
    F_HEAD_IN = 2,                     // [task] head stream -> main: input-gradient half done
    F_MAIN_HEADS = 10,                 // main -> helper: input halves of main's own heads done
    F_BWD_MA = 11,                     // [2 * layer + k] main -> aux: gu (k = 0) / g_z1 (k = 1) of the layer ready
    F_AUX_L = 21,                      // [layer] aux -> exchange stream: the layer's weight gradients are final (data-parallel runs)
    F_HEAD_PARAMS = 26,                // [task], [MAX_TASKS] = main's heads: weight-gradient GEMMs done (data-parallel runs)
    F_L0 = 35,                         // main -> exchange stream: past layer 0's eps sum
    F_BWD_DONE = 36,                   // main -> exchange stream: backward done
    F_AUX_DONE = 37,                   // aux -> main: everything aux did for this step is done
    F_EXCHANGE_DONE = 38,              // exchange stream -> main: set by the caller after its last unpack (dist.OverlappedGradSync)
    F_FWD_FORK = 39,                   // main -> the second forward stream: encoders done (split forward)
    F_FWD_JOIN = 40,                   // [2] the other forward streams -> main: their row range of the stacked forward is done
    F_WG1_DONE = 42,
    F_CSR = 43,                        // aux -> main: the batch's CSR / CSC are built (two-lane enqueue: an event recorded by one host thread cannot be waited on by the other)
    F_LPCSR = 44,                      // aux -> the link-prediction head's stream: the decoder pairs' CSR is built (likewise)                   // second weight-gradient stream -> main: everything it did for this step is done
    F_ERR = 63                         // a gate timed out
};
struct SyncState {                     // what gmp_step_wait_grads needs from the most recent step
    int32_t* flags = nullptr;
    int epoch = 0;
    uint64_t head_params_mask = 0;
};
SyncState g_sync;
}  // namespace
static void g_sync_publish(const SyncState& s) { g_sync = s; }
namespace {

bool wg1_enabled() {
    static const bool on = !(getenv("GMP_STEP_WG1") && atoi(getenv("GMP_STEP_WG1")) == 0);
    return on;
}

hipEvent_t* events() {   // one process drives one engine: a small static pool of timing-free events
    static hipEvent_t ev[NEV];
    static bool made = false;
    if (!made) {
        for (int i = 0; i < NEV; ++i) (void)hipEventCreateWithFlags(&ev[i], hipEventDisableTiming);
        made = true;
    }
    return ev;
}

// phase timing (diagnostic, GMP_STEP_TIMING=1): events on the main stream at step start / forward done / heads joined /
// backward done, read back by gmp_step_phase_ms
hipEvent_t* phase_events() {
    static hipEvent_t ev[GMP_STEP_PHASES + 1];
    static bool made = false;
    if (!made) {
        for (int i = 0; i <= GMP_STEP_PHASES; ++i) (void)hipEventCreate(&ev[i]);
        made = true;
    }
    return ev;
}
// (diagnostic) per head: its stream at the start of the head, after the input-gradient half, after the weight-gradient half; all against
// the main stream's "forward done" event
hipEvent_t* head_events() {
    static hipEvent_t ev[3 * GMP_STEP_MAX_TASKS];
    static bool made = false;
    if (!made) {
        for (int i = 0; i < 3 * GMP_STEP_MAX_TASKS; ++i) (void)hipEventCreate(&ev[i]);
        made = true;
    }
    return ev;
}
int g_head_tasks = 0;
bool g_head_recorded[3 * GMP_STEP_MAX_TASKS] = {false};
bool phase_timing() {
    static const bool on = getenv("GMP_STEP_TIMING") != nullptr;
    return on;
}

gmp_bn_config bn_cfg(const gmp_step_desc& d, bool relu, bool dropout, uint32_t site) {
    gmp_bn_config c{};
    c.training = d.training;
    c.relu = relu;
    c.eps = 1e-5f;
    c.momentum = 0.1f;
    c.dropout_p = (dropout && d.training) ? d.dropout_p : 0.f;
    c.seed = d.seed;
    c.stream_id = site;
    c.seed_dev = nullptr;
    return c;
}

int gemm(int mode, const float* A, const float* B, const float* bias, float* C, int64_t M, int64_t N, int64_t K, int64_t lda,
         int64_t ldb, int64_t ldc, bool relu, gmp_stream_t st) {
    return gmp_gemm_f32(mode, A, B, bias, C, M, N, K, lda, ldb, ldc, 1.f, 0, relu ? 1 : 0, nullptr, 0, st);
}

// dropout(src) -> dst, or alias src when dropout is off; returns the buffer holding the result
float* drop(const gmp_step_desc& d, float* src, float* dst, int64_t numel, uint32_t site, gmp_stream_t st, int* rc) {
    *rc = GMP_OK;
    if (!d.training || d.dropout_p <= 0.f) return src;
    *rc = gmp_dropout_fwd(src, dst, numel, d.dropout_p, d.seed, site, st);
    return dst;
}

int mlp2_fwd(const gmp_step_desc& d, const gmp_task_desc& t, float** d1_out, gmp_stream_t st) {
    const gmp_mlp2& m = t.mlp;
    const int G = d.num_domains;
    GMP_TRY(gmp_gemm_f32_grouped(GMP_GEMM_NT, m.x, d.flat, d.flat, m.y1, G, m.rows, m.off_w0, m.off_b0, nullptr, nullptr, nullptr, 0,
                                 m.k_hid, m.k_in, m.k_in, m.k_in, m.k_hid, 1.f, 0, 1, nullptr, 0, st));
    int rc;
    float* d1 = drop(d, m.y1, m.d1, (int64_t)m.rows[G] * m.k_hid, m.site, st, &rc);
    GMP_TRY(rc);
    GMP_TRY(gmp_gemm_f32_grouped(GMP_GEMM_NT, d1, d.flat, d.flat, m.y2, G, m.rows, m.off_w3, m.off_b3, nullptr, nullptr, nullptr, 0,
                                 m.k_out, m.k_hid, m.k_hid, m.k_hid, m.k_out, 1.f, 0, 0, nullptr, 0, st));
    *d1_out = d1;
    return GMP_OK;
}

// Backward of the two-layer head, in two halves: the input-gradient chain (what the stacked backward waits for) and the two
// weight-gradient GEMMs (they only feed task_grads).  mlp2_bwd_params reads g_out, d1, g_hid (after the ReLU/dropout gate) and x,
// none of which the input chain overwrites, so it may run after it.
int mlp2_bwd_inputs(const gmp_step_desc& d, const gmp_task_desc& t, gmp_stream_t st) {
    const gmp_mlp2& m = t.mlp;
    const int G = d.num_domains;
    GMP_TRY(gmp_gemm_f32_grouped(GMP_GEMM_NN, m.g_out, d.flat, nullptr, m.g_hid, G, m.rows, m.off_w3, nullptr, nullptr, nullptr, nullptr, 0,
                                 m.k_hid, m.k_out, m.k_out, m.k_hid, m.k_hid, 1.f, 0, 0, nullptr, 0, st));
    GMP_TRY(gmp_relu_dropout_bwd(m.g_hid, m.y1, m.g_hid, (int64_t)m.rows[G] * m.k_hid, d.training ? d.dropout_p : 0.f, d.seed, m.site, st));
    GMP_TRY(gmp_gemm_f32_grouped(GMP_GEMM_NN, m.g_hid, d.flat, nullptr, m.g_in, G, m.rows, m.off_w0, nullptr, nullptr, nullptr, nullptr, 0,
                                 m.k_in, m.k_hid, m.k_hid, m.k_in, m.k_in, 1.f, 0, 0, nullptr, 0, st));
    return GMP_OK;
}

int mlp2_bwd_params(const gmp_step_desc& d, const gmp_task_desc& t, float* d1, gmp_stream_t st) {
    const gmp_mlp2& m = t.mlp;
    const int G = d.num_domains;
    float* tg = d.task_grads;
    GMP_TRY(gmp_gemm_f32_grouped(GMP_GEMM_TN, m.g_out, d1, nullptr, tg, G, m.rows, nullptr, nullptr, m.tg_w3, tg, m.tg_b3, m.k_out, m.k_hid, 0,
                                 m.k_out, m.k_hid, m.k_hid, 1.f, 0, 0, t.gemm_ws, t.gemm_ws_bytes, st));
    GMP_TRY(gmp_gemm_f32_grouped(GMP_GEMM_TN, m.g_hid, m.x, nullptr, tg, G, m.rows, nullptr, nullptr, m.tg_w0, tg, m.tg_b0, m.k_hid, m.k_in, 0,
                                 m.k_hid, m.k_in, m.k_in, 1.f, 0, 0, t.gemm_ws, t.gemm_ws_bytes, st));
    return GMP_OK;
}

int nt_xent_domains(const gmp_step_desc& d, const gmp_task_desc& t, float* z, float* gz, gmp_stream_t st) {
    int64_t off[GMP_STEP_MAX_DOMAINS];
    for (int di = 0; di < d.num_domains; ++di) off[di] = t.mlp.rows[di];
    return gmp_nt_xent_grouped(z, gz, d.num_domains, t.ntx_n, off, 128, t.temperature, t.g_scale, t.ntx_sums, t.loss_sum, t.ntx_ws[0],
                               t.ntx_ws_bytes[0], st);
}

// A task head in two halves.  task_head_inputs: forward, loss, and the gradient with respect to the backbone output (added into
// gA) -- what the stacked backward waits for.  task_head_params: the head's own weight-gradient GEMMs, which only feed
// task_grads: on a stream of its own they run beside the first layers of the stacked backward instead of in front of it.
int task_head_inputs(const gmp_step_desc& d, int ti, gmp_stream_t st, float** d1_out) {
    const gmp_task_desc& t = d.task[ti];
    const gmp_mlp2& m = t.mlp;
    const int64_t N = d.N;
    float* hL = d.h[GMP_STEP_LAYERS];
    float* gH = d.gA;
    float* d1 = nullptr;
    switch (t.kind) {
        case GMP_TASK_NFM: {
            const int64_t M = t.num_idx;
            if (M == 0) return GMP_OK;
            GMP_TRY(gmp_row_gather(hL, t.idx, nullptr, m.x, M, N, H, st));
            GMP_TRY(mlp2_fwd(d, t, &d1, st));
            GMP_TRY(gmp_mse_sum_bwd(m.y2, t.nfm_target, t.g_scale, m.g_out, M * H, st));
            GMP_TRY(mlp2_bwd_inputs(d, t, st));
            *d1_out = d1;
            return gmp_row_fill(gH, t.idx, m.g_in, M, N, H, 0, st);
        }
        case GMP_TASK_LP: {
            const int64_t K = t.lp_K;
            const float *w0 = d.flat + t.lp_off_w0, *b0 = d.flat + t.lp_off_b0, *w3 = d.flat + t.lp_off_w3, *b3 = d.flat + t.lp_off_b3;
            GMP_TRY(gmp_lp_edge_features_fwd(hL, t.lp_edges, t.lp_feat, N, K, H, st));
            GMP_TRY(gemm(GMP_GEMM_NT, t.lp_feat, w0, b0, t.lp_y1, K, H, 3 * H, 3 * H, 3 * H, H, true, st));
            // the 256 -> 1 layer: a dot product per row with the dropout in the same pass, and its input gradient as an outer product
            // pushed through the dropout and the ReLU (gmp_dropout_rowdot_fwd / gmp_outer_relu_dropout_bwd: no N = 1 GEMM launches)
            const float pdrop = d.training && d.dropout_p > 0.f ? d.dropout_p : 0.f;
            float* ld1 = pdrop > 0.f ? t.lp_d1 : t.lp_y1;
            if (t.lp_pos) {       // merged rows (one per unordered pair): a dropout mask, a score and a BCE term per ORDERED row (gnnmp.h gmp_lp_pair_*)
                GMP_TRY(gmp_lp_pair_rowdot_fwd(t.lp_y1, w3, b3, t.lp_pos, t.lp_y2, K, H, pdrop, d.seed, t.lp_site, st));
                GMP_TRY(gmp_lp_pair_sigmoid_bce_fwd_bwd(t.lp_y2, t.lp_labels, t.lp_pos, K, t.g_scale, t.loss_sum, t.lp_p, t.lp_gy2, t.loss_ws, t.loss_ws_bytes, st));
                *d1_out = t.lp_y1;
                GMP_TRY(gmp_lp_pair_outer_bwd(t.lp_gy2, w3, t.lp_y1, t.lp_pos, t.lp_gy1, K, H, pdrop, d.seed, t.lp_site, st));
            } else {
                GMP_TRY(gmp_dropout_rowdot_fwd(t.lp_y1, w3, b3, t.lp_d1, t.lp_y2, K, H, pdrop, d.seed, t.lp_site, st));
                GMP_TRY(gmp_sigmoid_bce_signed_sum_fwd_bwd(t.lp_y2, t.lp_labels, K, t.g_scale, t.loss_sum, t.lp_p, t.lp_gy2, t.loss_ws, t.loss_ws_bytes, st));
                *d1_out = ld1;
                GMP_TRY(gmp_outer_relu_dropout_bwd(t.lp_gy2, w3, t.lp_y1, t.lp_gy1, K, H, pdrop, d.seed, t.lp_site, st));
            }
            GMP_TRY(gemm(GMP_GEMM_NN, t.lp_gy1, w0, nullptr, t.lp_gfeat, K, 3 * H, H, H, 3 * H, 3 * H, false, st));
            GMP_TRY(gmp_lp_edge_features_bwd(t.lp_gfeat, hL, t.lp_edges, t.lp_ghs, t.lp_ghd, N, K, H, st));
            float* g_rows = gH + (int64_t)H * t.row0;
            GMP_TRY(gmp_segment_sum(t.lp_ghs, d.lp_csr[3] + t.row0, d.lp_csr[5], g_rows, t.row1 - t.row0, H, 0, 1, st));
            return gmp_segment_sum(t.lp_ghd, d.lp_csr[0] + t.row0, d.lp_csr[2], g_rows, t.row1 - t.row0, H, 0, 1, st);
        }
        case GMP_TASK_NC: {
            const int64_t M = t.num_idx;
            if (M == 0) return GMP_OK;
            GMP_TRY(gmp_row_gather(hL, t.idx, nullptr, m.x, M, N, H, st));
            GMP_TRY(mlp2_fwd(d, t, &d1, st));
            GMP_TRY(nt_xent_domains(d, t, m.y2, m.g_out, st));
            GMP_TRY(mlp2_bwd_inputs(d, t, st));
            *d1_out = d1;
            return gmp_row_fill(gH, t.idx, m.g_in, M, N, H, 0, st);
        }
        case GMP_TASK_GC: {
            const int B = t.pool_B;
            if (B == 0) return GMP_OK;
            GMP_TRY(gmp_segment_sum(hL, t.pool_ptr, nullptr, t.pool_mean, B, H, 1, 0, st));
            GMP_TRY(gmp_segment_max_fwd(hL, t.pool_ptr, t.pool_max, B, H, st));
            // [mean | max] -> x [B, 512]
            if (hipMemcpy2DAsync(m.x, 2 * H * sizeof(float), t.pool_mean, H * sizeof(float), H * sizeof(float), B, hipMemcpyDeviceToDevice,
                                 (hipStream_t)st) != hipSuccess ||
                hipMemcpy2DAsync(m.x + H, 2 * H * sizeof(float), t.pool_max, H * sizeof(float), H * sizeof(float), B, hipMemcpyDeviceToDevice,
                                 (hipStream_t)st) != hipSuccess)
                return gmp::fail(GMP_ERR_LAUNCH, "step: read-out concat copy failed");
            GMP_TRY(mlp2_fwd(d, t, &d1, st));
            GMP_TRY(nt_xent_domains(d, t, m.y2, m.g_out, st));
            GMP_TRY(mlp2_bwd_inputs(d, t, st));
            *d1_out = d1;
            if (hipMemcpy2DAsync(t.g_mean, H * sizeof(float), m.g_in, 2 * H * sizeof(float), H * sizeof(float), B, hipMemcpyDeviceToDevice,
                                 (hipStream_t)st) != hipSuccess ||
                hipMemcpy2DAsync(t.g_max, H * sizeof(float), m.g_in + H, 2 * H * sizeof(float), H * sizeof(float), B, hipMemcpyDeviceToDevice,
                                 (hipStream_t)st) != hipSuccess)
                return gmp::fail(GMP_ERR_LAUNCH, "step: read-out split copy failed");
            GMP_TRY(gmp_row_gather(t.g_mean, t.pool_gid, t.pool_ptr, gH + (int64_t)H * t.pool_r0, t.pool_M, B, H, st));
            return gmp_segment_max_bwd(t.g_max, hL, t.pool_max, t.pool_ptr, gH, B, H, 1, st);
        }
        case GMP_TASK_GP: {
            const int B = t.pool_B;
            GMP_TRY(gmp_segment_sum(hL, t.pool_ptr, nullptr, m.x, B, H, 1, 0, st));
            GMP_TRY(mlp2_fwd(d, t, &d1, st));
            GMP_TRY(gmp_mse_sum_bwd(m.y2, t.labels, t.g_scale, m.g_out, (int64_t)B * m.k_out, st));
            GMP_TRY(mlp2_bwd_inputs(d, t, st));
            *d1_out = d1;
            return gmp_row_gather(m.g_in, t.pool_gid, t.pool_ptr, gH + (int64_t)H * t.pool_r0, t.pool_M, B, H, st);
        }
        case GMP_TASK_DA: {
            const int B = t.pool_B, Hd = t.da_hidden, Cc = t.da_classes;
            const float *w0 = d.flat + t.lp_off_w0, *b0 = d.flat + t.lp_off_b0, *w3 = d.flat + t.lp_off_w3, *b3 = d.flat + t.lp_off_b3;
            const float pdrop = d.training ? t.da_dropout : 0.f;
            GMP_TRY(gmp_segment_sum(hL, t.pool_ptr, nullptr, m.x, B, H, 1, 0, st));
            GMP_TRY(gemm(GMP_GEMM_NT, m.x, w0, b0, m.y1, B, Hd, H, H, H, Hd, true, st));
            float* dd1 = m.y1;
            if (pdrop > 0.f) {
                GMP_TRY(gmp_dropout_fwd(m.y1, m.d1, (int64_t)B * Hd, pdrop, d.seed, t.lp_site, st));
                dd1 = m.d1;
            }
            GMP_TRY(gemm(GMP_GEMM_NT, dd1, w3, b3, m.y2, B, Cc, Hd, Hd, Hd, Cc, false, st));
            GMP_TRY(gmp_cross_entropy_sum_fwd(m.y2, t.da_labels, B, Cc, t.loss_sum, t.loss_ws, t.loss_ws_bytes, st));
            GMP_TRY(gmp_cross_entropy_sum_bwd(m.y2, t.da_labels, B, Cc, t.g_scale, m.g_out, st));
            *d1_out = dd1;
            GMP_TRY(gemm(GMP_GEMM_NN, m.g_out, w3, nullptr, m.g_hid, B, Hd, Cc, Cc, Hd, Hd, false, st));
            GMP_TRY(gmp_relu_dropout_bwd(m.g_hid, m.y1, m.g_hid, (int64_t)B * Hd, pdrop, d.seed, t.lp_site, st));
            // gradient reversal: d/d pooled = -lambda * (g_hid W0)
            GMP_TRY(gmp_gemm_f32(GMP_GEMM_NN, m.g_hid, w0, nullptr, m.g_in, B, H, Hd, Hd, H, H, -t.da_lambda, 0, 0, nullptr, 0, st));
            return gmp_row_gather(m.g_in, t.pool_gid, t.pool_ptr, gH + (int64_t)H * t.pool_r0, t.pool_M, B, H, st);
        }
        default:
            return gmp::fail(GMP_ERR_ARG, "step: unknown task kind %d", t.kind);
    }
}

int task_head_params(const gmp_step_desc& d, int ti, gmp_stream_t st, float* d1) {
    const gmp_task_desc& t = d.task[ti];
    const gmp_mlp2& m = t.mlp;
    float* tg = d.task_grads;
    switch (t.kind) {
        case GMP_TASK_NFM:      // the loss VALUE (two launches, reporting only) is not on the chain the backward waits for
            if (t.num_idx == 0) return GMP_OK;
            GMP_TRY(mlp2_bwd_params(d, t, d1, st));
            return gmp_mse_sum_fwd(m.y2, t.nfm_target, t.num_idx * H, t.loss_sum, t.loss_ws, t.loss_ws_bytes, st);
        case GMP_TASK_NC:
            if (t.num_idx == 0) return GMP_OK;
            return mlp2_bwd_params(d, t, d1, st);
        case GMP_TASK_GC:
            if (t.pool_B == 0) return GMP_OK;
            return mlp2_bwd_params(d, t, d1, st);
        case GMP_TASK_GP:
            GMP_TRY(mlp2_bwd_params(d, t, d1, st));
            return gmp_mse_sum_fwd(m.y2, t.labels, (int64_t)t.pool_B * m.k_out, t.loss_sum, t.loss_ws, t.loss_ws_bytes, st);
        case GMP_TASK_LP: {
            const int64_t K = t.lp_K;
            const int32_t one[2] = {0, (int32_t)K};
            const int64_t cw0[1] = {t.lp_tg_w0}, cb0[1] = {t.lp_tg_b0};
            // dW0 with db0 riding along (column sums of the A tile already in LDS); first: this GEMM carries the "input half done" signal
            GMP_TRY(gmp_gemm_f32_grouped(GMP_GEMM_TN, t.lp_gy1, t.lp_feat, nullptr, tg, 1, one, nullptr, nullptr, cw0, tg, cb0, H, 3 * H, 0, H, 3 * H, 3 * H,
                                         1.f, 0, 0, t.gemm_ws, t.gemm_ws_bytes, st));
            // dW3 [1, 256] and db3: a weighted column sum (the grouped GEMM path took 54 us for these 257 numbers)
            if (t.lp_pos)
                return gmp_lp_pair_weighted_colsum(t.lp_gy2, t.lp_y1, t.lp_pos, tg + t.lp_tg_w3, tg + t.lp_tg_b3, K, H,
                                                   d.training && d.dropout_p > 0.f ? d.dropout_p : 0.f, d.seed, t.lp_site, t.gemm_ws, t.gemm_ws_bytes, st);
            return gmp_weighted_colsum(t.lp_gy2, d1, tg + t.lp_tg_w3, tg + t.lp_tg_b3, K, H, t.gemm_ws, t.gemm_ws_bytes, st);
        }
        case GMP_TASK_DA: {
            const int B = t.pool_B, Hd = t.da_hidden, Cc = t.da_classes;
            const int32_t one[2] = {0, B};
            const int64_t cw3[1] = {t.lp_tg_w3}, cb3[1] = {t.lp_tg_b3}, cw0[1] = {t.lp_tg_w0}, cb0[1] = {t.lp_tg_b0};
            GMP_TRY(gmp_gemm_f32_grouped(GMP_GEMM_TN, m.g_out, d1, nullptr, tg, 1, one, nullptr, nullptr, cw3, tg, cb3, Cc, Hd, 0, Cc, Hd, Hd, 1.f, 0, 0,
                                         nullptr, 0, st));
            return gmp_gemm_f32_grouped(GMP_GEMM_TN, m.g_hid, m.x, nullptr, tg, 1, one, nullptr, nullptr, cw0, tg, cb0, Hd, H, 0, Hd, H, H, 1.f, 0, 0,
                                        nullptr, 0, st);
        }
        default:
            return gmp::fail(GMP_ERR_ARG, "step: unknown task kind %d", t.kind);
    }
}

}  // namespace

extern "C" size_t gmp_step_desc_size(void) { return sizeof(gmp_step_desc); }

namespace {

// The launch sequence of one step.  `lanes`: the sequence is being walked by two host threads at once (gnnmp_internal.h lane filter: each
// skips the launches on streams it does not take); `primary`: this thread also leaves the step's flag state behind for gmp_step_wait_grads.
int step_body(const gmp_step_desc* dp, gmp_stream_t main_, const gmp_stream_t* task_streams, gmp_stream_t aux_, bool lanes, bool primary) {
    const gmp_step_desc& d = *dp;
    SyncState g_sync;              // this walk's copy (shadows the global: both lanes compute the same state, the primary publishes it)
    if (d.hidden != H || d.num_tasks < 1 || d.num_tasks > GMP_STEP_MAX_TASKS || d.num_domains < 1 || d.num_domains > GMP_STEP_MAX_DOMAINS ||
        d.N <= 0 || d.S <= 0 || d.enc_groups < 0 || d.enc_groups > GMP_STEP_MAX_ENC_GROUPS)
        return gmp::fail(GMP_ERR_ARG, "step: bad sizes (N=%d S=%d tasks=%d domains=%d hidden=%d)", d.N, d.S, d.num_tasks, d.num_domains, d.hidden);
    hipStream_t main = (hipStream_t)main_, aux = (hipStream_t)aux_;
    hipEvent_t* ev = events();
    const int64_t N = d.N;
    const int T = d.num_tasks;
    float* tg = d.task_grads;

    const bool timing = phase_timing();
    if (timing) (void)hipEventRecord(phase_events()[0], main);
    // cross-stream dependencies: gates when the caller passed flags (streams on different hardware queues), events otherwise
    const bool gates = d.sync_flags != nullptr;
    const bool per_layer = d.gu_l[0] != nullptr && d.gz1_l[0] != nullptr;     // per-layer g_u / g_z1 buffers (see the backward below)
    const bool lean = gates && per_layer;        // no event waits on main after the forward, no records nobody waits for
    gmp::signal_on_next_gemm(nullptr, 0);           // nothing left pending by an earlier call that failed half-way
    g_sync.flags = gates ? d.sync_flags : nullptr;
    g_sync.epoch = d.epoch;
    g_sync.head_params_mask = 0;
    auto signal = [&](int flag, hipEvent_t e, hipStream_t s) -> int {
        if (gates) return gmp_gate_open(d.sync_flags + flag, d.epoch, (gmp_stream_t)s);
        (void)hipEventRecord(e, s);
        return GMP_OK;
    };
    // signal carried by the NEXT GEMM launched from this thread (it must be on stream s and follow immediately in host order):
    // saves the one-thread launch on a critical chain; signal_flush opens the gate itself if that GEMM turned out to be empty
    auto signal_by_gemm = [&](int flag, hipEvent_t e, hipStream_t s) {
        if (gates) {
            if (gmp::lane_takes(s)) gmp::signal_on_next_gemm(d.sync_flags + flag, d.epoch);      // (the GEMM that carries it is this thread's to launch)
        } else (void)hipEventRecord(e, s);
    };
    auto signal_flush = [&](int flag, hipStream_t s) -> int {
        if (!gates || !gmp::lane_takes(s) || !gmp::signal_pending()) return GMP_OK;
        gmp::signal_on_next_gemm(nullptr, 0);
        return gmp_gate_open(d.sync_flags + flag, d.epoch, (gmp_stream_t)s);
    };
    auto await = [&](int flag, hipEvent_t e, hipStream_t s) -> int {
        if (gates) return gmp_gate_wait(d.sync_flags, 1ull << flag, d.epoch, d.sync_flags + F_ERR, (gmp_stream_t)s);
        (void)hipStreamWaitEvent(s, e, 0);
        return GMP_OK;
    };
    // ---- CSR builds beside the encoders (they only need the uploaded indices)
    if (d.upload_on_aux && aux != main) {      // the uploads came up on aux (in order there): main is the one that waits
        GMP_TRY(signal(F_START, ev[0], aux));
        GMP_TRY(await(F_START, ev[0], main));
    } else {
        GMP_TRY(signal(F_START, ev[0], main));
        GMP_TRY(await(F_START, ev[0], aux));
    }
    if (d.max_seg <= 8192 && d.max_seg_edges <= 24576)      // block diagonal: one workgroup per (segment, orientation)
        GMP_TRY(gmp_csr_build_segmented(d.edge_index, N, d.E, d.seg_ptr, d.seg_eptr, d.S, d.max_seg, d.max_seg_edges, d.csr[0], d.csr[1], d.csr[2], d.csr[3],
                                        d.csr[4], d.csr[5], d.csr_status, aux_));
    else
        GMP_TRY(gmp_csr_build(d.edge_index, N, d.E, d.csr[0], d.csr[1], d.csr[2], d.csr[3], d.csr[4], d.csr[5], d.csr_status, d.csr_ws, d.csr_ws_bytes, aux_));
    if (lanes) GMP_TRY(signal(F_CSR, ev[1], aux));
    else (void)hipEventRecord(ev[1], aux);
    int lp_task = -1;
    for (int ti = 0; ti < T; ++ti)
        if (d.task[ti].kind == GMP_TASK_LP) lp_task = ti;
    if (lp_task >= 0) {
        const gmp_task_desc& t = d.task[lp_task];
        if (d.lp_S > 0)
            GMP_TRY(gmp_csr_build_segmented(t.lp_edges, d.lp_rows_end, t.lp_K, d.lp_seg_ptr, d.lp_seg_eptr, d.lp_S, d.lp_max_seg_rows, d.lp_max_seg_edges,
                                            d.lp_csr[0], d.lp_csr[1], d.lp_csr[2], d.lp_csr[3], d.lp_csr[4], d.lp_csr[5], d.lp_csr_status, aux_));
        else
            GMP_TRY(gmp_csr_build(t.lp_edges, N, t.lp_K, d.lp_csr[0], d.lp_csr[1], d.lp_csr[2], d.lp_csr[3], d.lp_csr[4], d.lp_csr[5], d.lp_csr_status,
                                  d.lp_csr_ws, d.lp_csr_ws_bytes, aux_));
    }
    if (lanes) GMP_TRY(signal(F_LPCSR, ev[2], aux));
    else (void)hipEventRecord(ev[2], aux);

    // ---- encoders
    GMP_TRY(gmp_encoder_fwd(d.x_all, d.x_rows, N, d.S, d.src_row, d.seg_ptr, d.seg_dom, (const uint64_t*)d.rowmask, d.tiles, d.num_tiles, d.flat,
                            d.num_domains, d.enc_off_w, d.enc_off_b, d.enc_d_in, d.dpad, d.z0, main_));
    gmp_bn_config c = bn_cfg(d, true, true, 1);
    // training: the running statistics are brought up to date at the end of the step on the aux stream (see below)
    const bool defer = d.training != 0;
    GMP_TRY(gmp_bn_fwd(d.z0, nullptr, d.seg_ptr, d.seg_dom, d.S, d.max_seg, N, H, d.flat + d.enc_off_gamma0, d.flat + d.enc_off_beta0,
                       defer ? nullptr : d.enc_rm, defer ? nullptr : d.enc_rv, d.enc_mean, d.enc_rstd, d.h[0], &c, d.bn_ws, d.bn_ws_bytes, main_));
    if (d.nfm_task >= 0 && d.task[d.nfm_task].num_idx > 0) {
        const gmp_task_desc& t = d.task[d.nfm_task];
        GMP_TRY(gmp_row_gather(d.h[0], t.idx, nullptr, t.nfm_target, t.num_idx, N, H, main_));
        GMP_TRY(gmp_row_fill(d.h[0], t.idx, d.flat + d.off_mask_token, t.num_idx, N, H, 1, main_));
    }
    if (lanes) GMP_TRY(await(F_CSR, ev[1], main));
    else (void)hipStreamWaitEvent(main, ev[1], 0);
    if (timing) (void)hipEventRecord(phase_events()[1], main);

    // ---- stacked backbone forward: one pass on main, or up to three row ranges on as many streams (gnnmp_step.h fwd_cut_*)
    hipStream_t fwd_streams[3] = {main, main, main};
    int extra = 0;
    for (int ti = 0; ti < T && extra < 2; ++ti) {
        hipStream_t ts = (hipStream_t)task_streams[ti];
        if (ts != main && ts != aux && ts != fwd_streams[1]) fwd_streams[++extra] = ts;
    }
    const size_t bn_fwd_slice = gmp_bn_workspace_bytes(N, 2 * H, d.S, d.max_seg);
    struct Range { int s0, s1; int64_t r0, r1; hipStream_t st; void* ws; };
    Range ranges[3];
    int nranges = 1;
    ranges[0] = Range{0, d.S, 0, N, main, d.bn_ws};
    {
        // (every range at least 1,024 rows: below that gmp_gemm_f32 runs its first kernel, whose K order differs from the pipelined one the
        // single pass uses -- the split must not change a bit -- and a step that small has nothing to hide behind anyway)
        int cuts = 0;
        int64_t prev_row = 0;
        int prev_seg = 0;
        bool ok = N < 65536;
        for (int k = 0; k < 2 && ok; ++k) {
            if (d.fwd_cut_seg[k] <= 0) break;
            ok = d.fwd_cut_seg[k] > prev_seg && d.fwd_cut_seg[k] < d.S && d.fwd_cut_row[k] - prev_row >= 1024 && N - d.fwd_cut_row[k] >= 1024;
            prev_seg = d.fwd_cut_seg[k]; prev_row = d.fwd_cut_row[k];
            ++cuts;
        }
        if (ok && cuts > 0 && cuts <= extra && d.bn_ws_bytes >= (size_t)(cuts + 1) * bn_fwd_slice) {
            nranges = cuts + 1;
            for (int k = 0; k < nranges; ++k) {
                ranges[k].s0 = k ? d.fwd_cut_seg[k - 1] : 0;
                ranges[k].s1 = k + 1 < nranges ? d.fwd_cut_seg[k] : d.S;
                ranges[k].r0 = k ? d.fwd_cut_row[k - 1] : 0;
                ranges[k].r1 = k + 1 < nranges ? d.fwd_cut_row[k] : N;
                ranges[k].st = fwd_streams[k];
                ranges[k].ws = (char*)d.bn_ws + (size_t)k * bn_fwd_slice;
            }
        }
    }
    const bool split_fwd = nranges > 1;
    if (split_fwd) {
        GMP_TRY(signal(F_FWD_FORK, ev[EV_FWD_FORK], main));
        for (int k = 1; k < nranges; ++k) GMP_TRY(await(F_FWD_FORK, ev[EV_FWD_FORK], ranges[k].st));
    }
    for (int l = 0; l < GMP_STEP_LAYERS; ++l) {
        const gmp_layer_desc& L = d.layer[l];
        for (int k = 0; k < nranges; ++k) {
            const Range& R = ranges[k];
            gmp_stream_t st = (gmp_stream_t)R.st;
            const int64_t M = R.r1 - R.r0;
            const int Sk = R.s1 - R.s0;
            const int32_t* sp = d.seg_ptr + R.s0;
            if (split_fwd) GMP_TRY(gmp_gin_aggregate_fwd_rows(d.h[l], d.csr[0], d.csr[1], d.flat + L.off_eps, L.a, R.r0, R.r1, H, st));
            else GMP_TRY(gmp_gin_aggregate_fwd(d.h[l], d.csr[0], d.csr[1], d.flat + L.off_eps, L.a, N, H, st));
            GMP_TRY(gemm(GMP_GEMM_NT, L.a + R.r0 * H, d.flat + L.off_w1, d.flat + L.off_b1, L.z1 + R.r0 * 2 * H, M, 2 * H, H, H, H, 2 * H, false, st));
            c = bn_cfg(d, true, false, 0);
            // (absolute row numbers in seg_ptr: the BatchNorm takes whole-batch base pointers and this range's segments)
            GMP_TRY(gmp_bn_fwd(L.z1, nullptr, sp, nullptr, Sk, d.max_seg, N, 2 * H, d.flat + L.off_g1, d.flat + L.off_be1, defer ? nullptr : L.rm1,
                               defer ? nullptr : L.rv1, L.m1 + (size_t)R.s0 * 2 * H, L.s1 + (size_t)R.s0 * 2 * H, L.r1, &c, R.ws, split_fwd ? bn_fwd_slice : d.bn_ws_bytes, st));
            GMP_TRY(gemm(GMP_GEMM_NT, L.r1 + R.r0 * 2 * H, d.flat + L.off_w2, d.flat + L.off_b2, L.z2 + R.r0 * H, M, H, 2 * H, 2 * H, 2 * H, H, false, st));
            c = bn_cfg(d, true, true, 10 + l);
            GMP_TRY(gmp_bn_fwd(L.z2, d.h[l], sp, nullptr, Sk, d.max_seg, N, H, d.flat + L.off_g2, d.flat + L.off_be2, defer ? nullptr : L.rm2,
                               defer ? nullptr : L.rv2, L.m2 + (size_t)R.s0 * H, L.s2 + (size_t)R.s0 * H, d.h[l + 1], &c, R.ws, split_fwd ? bn_fwd_slice : d.bn_ws_bytes, st));
        }
        if (timing && l + 1 < GMP_STEP_LAYERS) (void)hipEventRecord(phase_events()[2 + l], main);
    }
    if (split_fwd) {
        for (int k = 1; k < nranges; ++k) GMP_TRY(signal(F_FWD_JOIN + k - 1, ev[EV_FWD_JOIN + k - 1], ranges[k].st));
        if (gates) {
            GMP_TRY(gmp_gate_wait(d.sync_flags, (nranges > 2 ? 3ull : 1ull) << F_FWD_JOIN, d.epoch, d.sync_flags + F_ERR, main_));
        } else {
            for (int k = 1; k < nranges; ++k) (void)hipStreamWaitEvent(main, ev[EV_FWD_JOIN + k - 1], 0);
        }
    }

    // ---- task heads, each on its own stream
    if (hipMemsetAsync(d.gA, 0, (size_t)N * H * sizeof(float), main) != hipSuccess) return gmp::fail(GMP_ERR_LAUNCH, "step: memset");
    GMP_TRY(signal(F_FWD, ev[3], main));
    if (timing) (void)hipEventRecord(phase_events()[1 + GMP_STEP_LAYERS], main);
    // heads on other streams first (their chains are the long ones: they start while main is still being fed), the ones packed
    // onto main last; main joins the other streams only after everything is enqueued.  Every head comes in two halves: main
    // waits for the input-gradient half only; the weight-gradient GEMMs (they only feed task_grads) follow on the head's own
    // stream and run beside the first layers of the stacked backward.  The heads that live on main hand theirs to `helper`.
    float* head_d1[GMP_STEP_MAX_TASKS] = {nullptr};
    g_head_tasks = T;
    hipStream_t helper = main;
    for (int ti = 0; ti < T; ++ti) {
        hipStream_t ts = (hipStream_t)task_streams[ti];
        if (ts != main && (helper == main || ts != aux)) helper = ts;     // prefer a stream the backward does not use
    }
    bool main_heads = false;
    for (int pass = 0; pass < 2; ++pass)
        for (int ti = 0; ti < T; ++ti) {
            hipStream_t ts = (hipStream_t)task_streams[ti];
            // (round 3: enqueueing the main stream's heads first moves THEIR start from ~100 us to 8 us after the forward and the others'
            // back by as much -- the step takes the same 1.413 ms either way)
            if ((ts == main) != (pass == 1)) continue;
            if (ts != main) GMP_TRY(await(F_FWD, ev[3], ts));
            if (d.task[ti].kind == GMP_TASK_LP) {
                if (lanes) GMP_TRY(await(F_LPCSR, ev[2], ts));
                else (void)hipStreamWaitEvent(ts, ev[2], 0);
            }
            if (timing) { (void)hipEventRecord(head_events()[3 * ti], ts); g_head_recorded[3 * ti] = true; g_head_recorded[3 * ti + 2] = false; }
            GMP_TRY(task_head_inputs(d, ti, task_streams[ti], &head_d1[ti]));
            if (timing) { (void)hipEventRecord(head_events()[3 * ti + 1], ts); g_head_recorded[3 * ti + 1] = true; }
            if (ts != main) {
                signal_by_gemm(F_HEAD_IN + ti, ev[4 + ti], ts);          // the head's first weight-gradient GEMM opens main's gate
                GMP_TRY(task_head_params(d, ti, task_streams[ti], head_d1[ti]));
                GMP_TRY(signal_flush(F_HEAD_IN + ti, ts));
                if (timing) { (void)hipEventRecord(head_events()[3 * ti + 2], ts); g_head_recorded[3 * ti + 2] = true; }
                if (!lean) (void)hipEventRecord(ev[EV_HEAD_PARAMS + ti], ts);       // main joins these before the tail: long complete by then
                if (gates) {
                    GMP_TRY(gmp_gate_open(d.sync_flags + F_HEAD_PARAMS + ti, d.epoch, task_streams[ti]));
                    g_sync.head_params_mask |= 1ull << (F_HEAD_PARAMS + ti);
                }
            } else {
                main_heads = true;
            }
        }
    if (main_heads) {
        if (helper != main) {
            GMP_TRY(signal(F_MAIN_HEADS, ev[EV_MAIN_HEADS], main));
            GMP_TRY(await(F_MAIN_HEADS, ev[EV_MAIN_HEADS], helper));
        }
        for (int ti = 0; ti < T; ++ti)
            if ((hipStream_t)task_streams[ti] == main) GMP_TRY(task_head_params(d, ti, (gmp_stream_t)helper, head_d1[ti]));
        if (helper != main) {
            if (!lean) (void)hipEventRecord(ev[EV_HEAD_PARAMS + GMP_STEP_MAX_TASKS], helper);
            if (gates) {
                GMP_TRY(gmp_gate_open(d.sync_flags + F_HEAD_PARAMS + GMP_STEP_MAX_TASKS, d.epoch, (gmp_stream_t)helper));
                g_sync.head_params_mask |= 1ull << (F_HEAD_PARAMS + GMP_STEP_MAX_TASKS);
            }
        }
    }
    if (gates) {          // one sleeping wave on main for all the heads' flags
        uint64_t mask = 0;
        for (int ti = 0; ti < T; ++ti)
            if ((hipStream_t)task_streams[ti] != main) mask |= 1ull << (F_HEAD_IN + ti);
        GMP_TRY(gmp_gate_wait(d.sync_flags, mask, d.epoch, d.sync_flags + F_ERR, main_));
    } else {
        for (int ti = 0; ti < T; ++ti)
            if ((hipStream_t)task_streams[ti] != main) (void)hipStreamWaitEvent(main, ev[4 + ti], 0);
    }
    if (!gates) (void)hipStreamWaitEvent(main, ev[2], 0);     // (with an LP head its join implies it; without one ev[1] does)
    if (d.dp_exchange && !gates) (void)hipEventRecord(ev[EV_HEADS_DONE], main);   // every head's input half is done (gmp_step_wait_grads adds the params events)
    if (timing) (void)hipEventRecord(phase_events()[2 + GMP_STEP_LAYERS], main);

    // ---- stacked backbone backward: per-task parameter gradients from ONE pass.
    // The input-gradient chain (BN bwd -> dgrad GEMM -> BN bwd -> dgrad GEMM -> aggregation bwd) is the critical path;
    // the weight-gradient GEMMs only feed task_grads, so they run beside it on the aux stream.  gB/gB2 and gW2/gW3
    // alternate per layer so a layer's weight-gradient GEMM can still read its operand while the next layer writes.
    hipEvent_t* evl = ev + 4 + GMP_STEP_MAX_TASKS;       // per layer: [0] g_u ready, [1] dW2 done, [2] g_z1 ready, [3] dW1 done
    // BatchNorm gamma/beta gradients also only feed task_grads: each BN backward leaves its per-segment sums in its own slice
    // of bn_ws and the reduction per task runs on aux next to the weight-gradient GEMM (falls back to inline when the
    // workspace cannot hold a slice per BatchNorm)
    const size_t bn_slice = gmp_bn_workspace_bytes(N, 2 * H, d.S, d.max_seg);
    const bool split_pg = d.bn_ws_bytes >= bn_slice * (2 * GMP_STEP_LAYERS + 1);
    auto slice = [&](int i) { return (void*)((char*)d.bn_ws + (split_pg ? bn_slice * (size_t)i : 0)); };
    const size_t slice_bytes = split_pg ? bn_slice : d.bn_ws_bytes;
    // Per-layer g_u / g_z1 buffers (gu_l / gz1_l): aux may lag main by any number of layers and main never waits for it inside
    // the backward.  (A wait on an event that was not complete when it was ENQUEUED costs the waiting stream 3-4 us even when
    // the event has long fired by the time the packet is reached -- scripts/diag_blocked_queues.py -- and the host runs ahead.)
    // Without them: two alternating copies, guarded by event waits two layers later.
    // main's encoder backward and aux's grouped weight-gradient GEMMs each get half of gemm_ws (no join between them)
    // Two weight-gradient streams: per layer aux carries dW2 (+ the BatchNorm-2 sums, the eps sum), and a head stream -- idle once its head's
    // weight gradients are out -- carries dW1 (+ the BatchNorm-1 sums).  On ONE stream the two GEMMs, their slice reductions and the sums
    // add up to ~105 us per layer, as long as main's input-gradient chain (110 us).  Worth 0.8 % (1.409 -> 1.397, 1.417 -> 1.401 ms in two
    // A/B pairs): the backward is bound by the chip's throughput, not by either chain ("two row ranges").
    // Needs gates, per-layer buffers and a third of gemm_ws that still holds the slices (GMP_STEP_WG1=0: everything on aux).  Not in
    // data-parallel runs: the exchange lives on a head stream too (engine.py comm_stream) and follows the backward layer by layer through aux's
    // flags; behind a whole weight-gradient chain in the same in-order queue it would start when the backward ends.
    hipStream_t wg1 = aux;
    if (lean && aux != main && !d.dp_exchange && d.gemm_ws_bytes >= ((size_t)36 << 20) && wg1_enabled())
        for (int pass = 0; pass < 2 && wg1 == aux; ++pass)       // not the link-prediction head's stream if there is another: its weight gradients take longest
            for (int ti = 0; ti < T && wg1 == aux; ++ti) {
                hipStream_t ts = (hipStream_t)task_streams[ti];
                if (ts != main && ts != aux && (pass == 1 || d.task[ti].kind != GMP_TASK_LP)) wg1 = ts;
            }
    gmp_stream_t wg1_ = (gmp_stream_t)wg1;
    const bool two_wg = wg1 != aux;
    const size_t ws_part = per_layer ? (d.gemm_ws_bytes / (two_wg ? 3 : 2)) & ~(size_t)255 : 0;
    void* const aux_ws = d.gemm_ws;
    const size_t aux_ws_bytes = per_layer ? ws_part : d.gemm_ws_bytes;
    void* const wg1_ws = two_wg ? (void*)((char*)d.gemm_ws + ws_part) : aux_ws;
    const size_t wg1_ws_bytes = aux_ws_bytes;
    void* const enc_ws = per_layer ? (void*)((char*)d.gemm_ws + (two_wg ? 2 : 1) * ws_part) : d.gemm_ws;
    const size_t enc_ws_bytes = per_layer ? d.gemm_ws_bytes - (two_wg ? 2 : 1) * ws_part : d.gemm_ws_bytes;
    // Training with gates and per-layer buffers: the eps gradient of layer l (a 5 us sum over rowdot that only feeds task_grads)
    // runs on aux at the start of aux's layer l-1 work -- the flag aux waits for there is set after main's aggregation backward
    // of layer l -- from a rowdot buffer per layer; layer 0's goes to aux's tail.
    const bool eps_on_aux = lean && defer && aux != main;
    float *gcur = d.gA, *ga = d.ga;
    for (int l = GMP_STEP_LAYERS - 1; l >= 0; --l) {
        const gmp_layer_desc& L = d.layer[l];
        float* gu = per_layer ? d.gu_l[l] : ((l & 1) ? d.gB2 : d.gB);
        float* gz1 = per_layer ? d.gz1_l[l] : ((l & 1) ? d.gW3 : d.gW2);
        hipEvent_t* e = evl + 4 * l;
        if (!per_layer && l + 2 < GMP_STEP_LAYERS) (void)hipStreamWaitEvent(main, evl[4 * (l + 2) + 1], 0);   // dW2 of layer l+2 has read this gu copy
        c = bn_cfg(d, true, true, 10 + l);
        GMP_TRY(gmp_bn_bwd(gcur, L.z2, d.h[l], d.seg_ptr, nullptr, d.S, d.max_seg, N, H, d.flat + L.off_g2, d.flat + L.off_be2, L.rm2, L.rv2, L.m2, L.s2, gu,
                           tg, tg, d.task_seg, L.tg_g2, L.tg_be2, split_pg ? 0 : T, &c, slice(1 + 2 * l), slice_bytes, main_));
        signal_by_gemm(F_BWD_MA + 2 * l, e[0], main);            // g_u ready: the input-gradient GEMM below tells aux as it starts
        GMP_TRY(gemm(GMP_GEMM_NN, gu, d.flat + L.off_w2, nullptr, d.gW, N, 2 * H, H, H, 2 * H, 2 * H, false, main_));
        GMP_TRY(signal_flush(F_BWD_MA + 2 * l, main));
        GMP_TRY(await(F_BWD_MA + 2 * l, e[0], aux));
        if (eps_on_aux && l + 1 < GMP_STEP_LAYERS) {
            GMP_TRY(gmp_group_sum_1d(d.rowdot + (size_t)(l + 1) * N, T, d.task_row, d.layer[l + 1].tg_eps, tg, aux_));
            if (d.dp_exchange) GMP_TRY(gmp_gate_open(d.sync_flags + F_AUX_L + l + 1, d.epoch, aux_));     // layer l+1 is final now
        }
        if (split_pg) GMP_TRY(gmp_bn_param_grads(slice(1 + 2 * l), d.S, H, tg, tg, d.task_seg, L.tg_g2, L.tg_be2, T, aux_));
        GMP_TRY(gmp_gemm_f32_grouped(GMP_GEMM_TN, gu, L.r1, nullptr, tg, T, d.task_row, nullptr, nullptr, L.tg_w2, tg, L.tg_b2, H, 2 * H, 0, H, 2 * H, 2 * H,
                                     1.f, 0, 0, aux_ws, aux_ws_bytes, aux_));
        if (!lean) (void)hipEventRecord(e[1], aux);      // (a record costs its stream ~3 us: only where somebody waits for it)
        if (!per_layer && l + 2 < GMP_STEP_LAYERS) (void)hipStreamWaitEvent(main, evl[4 * (l + 2) + 3], 0);   // dW1 of layer l+2 has read this g_z1 copy
        c = bn_cfg(d, true, false, 0);
        GMP_TRY(gmp_bn_bwd(d.gW, L.z1, nullptr, d.seg_ptr, nullptr, d.S, d.max_seg, N, 2 * H, d.flat + L.off_g1, d.flat + L.off_be1, L.rm1, L.rv1, L.m1,
                           L.s1, gz1, tg, tg, d.task_seg, L.tg_g1, L.tg_be1, split_pg ? 0 : T, &c, slice(2 + 2 * l), slice_bytes, main_));
        signal_by_gemm(F_BWD_MA + 2 * l + 1, e[2], main);        // g_z1 ready
        GMP_TRY(gemm(GMP_GEMM_NN, gz1, d.flat + L.off_w1, nullptr, ga, N, H, 2 * H, 2 * H, H, H, false, main_));
        GMP_TRY(signal_flush(F_BWD_MA + 2 * l + 1, main));
        GMP_TRY(await(F_BWD_MA + 2 * l + 1, e[2], wg1));
        if (split_pg) GMP_TRY(gmp_bn_param_grads(slice(2 + 2 * l), d.S, 2 * H, tg, tg, d.task_seg, L.tg_g1, L.tg_be1, T, wg1_));
        GMP_TRY(gmp_gemm_f32_grouped(GMP_GEMM_TN, gz1, L.a, nullptr, tg, T, d.task_row, nullptr, nullptr, L.tg_w1, tg, L.tg_b1, 2 * H, H, 0, 2 * H, H, H,
                                     1.f, 0, 0, wg1_ws, wg1_ws_bytes, wg1_));
        if (!lean) (void)hipEventRecord(e[3], aux);
        if (gates && d.dp_exchange && !eps_on_aux) GMP_TRY(gmp_gate_open(d.sync_flags + F_AUX_L + l, d.epoch, aux_));
        float* rowdot = eps_on_aux ? d.rowdot + (size_t)l * N : d.rowdot;
        GMP_TRY(gmp_gin_aggregate_bwd_ex(ga, d.csr[3], d.csr[4], d.flat + L.off_eps, d.h[l], gu, gcur, rowdot, N, H, main_));
        if (!eps_on_aux) GMP_TRY(gmp_group_sum_1d(rowdot, T, d.task_row, L.tg_eps, tg, main_));
        if (timing) (void)hipEventRecord(phase_events()[3 + GMP_STEP_LAYERS + (GMP_STEP_LAYERS - 1 - l)], main);
    }
    float* gu = d.gB;     // scratch for the encoder backward below (without per-layer buffers layer 0 used gB: its dW2 GEMM is awaited first)
    if (!lean) {
        for (int l = 0; l < GMP_STEP_LAYERS && l < 2; ++l) {
            (void)hipStreamWaitEvent(main, evl[4 * l + 1], 0);
            (void)hipStreamWaitEvent(main, evl[4 * l + 3], 0);
        }
        // join the heads' weight-gradient GEMMs, which ran beside the backward (long done; the mask-token sum below reuses the
        // NFM head's input buffer, which its dW0 GEMM reads)
        for (int ti = 0; ti < T; ++ti)
            if ((hipStream_t)task_streams[ti] != main) (void)hipStreamWaitEvent(main, ev[EV_HEAD_PARAMS + ti], 0);
        if (main_heads && helper != main) (void)hipStreamWaitEvent(main, ev[EV_HEAD_PARAMS + GMP_STEP_MAX_TASKS], 0);
    }
    // ---- below the backbone: mask token (NFM) and the encoders (every task but NFM).  Training: the mask-token sum (3 launches
    // that only feed task_grads) goes to aux, in front of the running statistics, beside the encoder backward on main.
    const bool nfm_tail = d.nfm_task >= 0 && d.task[d.nfm_task].num_idx > 0;
    const bool tail_on_aux = defer && aux != main;
    if (d.dp_exchange || (tail_on_aux && nfm_tail) || eps_on_aux) GMP_TRY(signal(F_L0, ev[EV_LAYER0_DONE], main));   // main is past layer 0
    gmp_stream_t tail_st = tail_on_aux ? aux_ : main_;
    if (lean) {               // one gate in front of the tail work: main past layer 0 (when the work is on aux), and the NFM head's
                              // dW0 GEMM done (it reads the buffer the mask-token sum reuses)
        uint64_t mask = 0;
        if (nfm_tail) {
            const bool nfm_on_main = (hipStream_t)task_streams[d.nfm_task] == main;
            mask = (1ull << (F_HEAD_PARAMS + (nfm_on_main ? GMP_STEP_MAX_TASKS : d.nfm_task))) & g_sync.head_params_mask;
        }
        if (tail_on_aux && (nfm_tail || eps_on_aux)) mask |= 1ull << F_L0;
        GMP_TRY(gmp_gate_wait(d.sync_flags, mask, d.epoch, d.sync_flags + F_ERR, tail_st));
    } else if (tail_on_aux && nfm_tail) {
        GMP_TRY(await(F_L0, ev[EV_LAYER0_DONE], aux));
    }
    if (eps_on_aux) {
        GMP_TRY(gmp_group_sum_1d(d.rowdot, T, d.task_row, d.layer[0].tg_eps, tg, aux_));
        if (d.dp_exchange) GMP_TRY(gmp_gate_open(d.sync_flags + F_AUX_L + 0, d.epoch, aux_));
    }
    if (nfm_tail) {
        const gmp_task_desc& t = d.task[d.nfm_task];
        GMP_TRY(gmp_row_gather(gcur, t.idx, nullptr, t.mlp.x, t.num_idx, N, H, tail_st));
        GMP_TRY(gmp_colsum(t.mlp.x, tg + d.tg_mask_token, t.num_idx, H, H, 0, t.loss_ws, t.loss_ws_bytes, tail_st));
    }
    if (defer) {          // running statistics of the 11 BatchNorms (training never reads them): on aux behind its last weight-gradient GEMM,
                          // beside the encoder backward and PCGrad (in front of the aux stream's head it delayed that head by 45 us)
        c = bn_cfg(d, true, false, 0);
        constexpr int NB = 2 * GMP_STEP_LAYERS + 1;
        const int32_t* sg[NB];
        int32_t ch[NB];
        float *rm[NB], *rv[NB];
        const float *sm[NB], *sr[NB];
        sg[0] = d.seg_dom; ch[0] = H; rm[0] = d.enc_rm; rv[0] = d.enc_rv; sm[0] = d.enc_mean; sr[0] = d.enc_rstd;
        for (int l = 0; l < GMP_STEP_LAYERS; ++l) {
            const gmp_layer_desc& L = d.layer[l];
            const int a = 1 + 2 * l, b = 2 + 2 * l;
            sg[a] = nullptr; ch[a] = 2 * H; rm[a] = L.rm1; rv[a] = L.rv1; sm[a] = L.m1; sr[a] = L.s1;
            sg[b] = nullptr; ch[b] = H; rm[b] = L.rm2; rv[b] = L.rv2; sm[b] = L.m2; sr[b] = L.s2;
        }
        GMP_TRY(gmp_bn_running_update_batch(NB, d.seg_ptr, d.S, sg, ch, rm, rv, sm, sr, &c, aux_));
        if (!lean) (void)hipEventRecord(ev[NEV - 1], aux);
    }
    if (lean && aux != main) GMP_TRY(gmp_gate_open(d.sync_flags + F_AUX_DONE, d.epoch, aux_));
    if (two_wg) GMP_TRY(gmp_gate_open(d.sync_flags + F_WG1_DONE, d.epoch, wg1_));
    if (d.enc_groups > 0) {
        c = bn_cfg(d, true, true, 1);
        // (its own slice of the BatchNorm scratch: aux may still be reducing layer 0's slices)
        GMP_TRY(gmp_bn_bwd(gcur, d.z0, nullptr, d.seg_ptr, d.seg_dom, d.S, d.max_seg, N, H, d.flat + d.enc_off_gamma0, d.flat + d.enc_off_beta0, d.enc_rm,
                           d.enc_rv, d.enc_mean, d.enc_rstd, gu, tg, tg, d.enc_gseg, d.enc_tg_gamma, d.enc_tg_beta, d.enc_groups, &c,
                           lean ? slice(0) : d.bn_ws, lean ? slice_bytes : d.bn_ws_bytes, main_));
        GMP_TRY(gmp_encoder_bwd(d.x_all, d.x_rows, N, d.S, d.src_row, d.seg_ptr, d.seg_dom, (const uint64_t*)d.rowmask, gu, d.num_domains, d.enc_d_in,
                                d.dpad, d.enc_groups, d.enc_gseg, d.enc_tg_w, d.enc_tg_b, tg, enc_ws, enc_ws_bytes, main_));
    }
    if (lean) {           // ONE sleeping wave joins everything that ran beside main: aux (weight gradients, mask token, running
                          // statistics) and the heads' weight-gradient GEMMs -- instead of ten event waits at 3-4 us each
        uint64_t mask = g_sync.head_params_mask;
        if (aux != main) mask |= 1ull << F_AUX_DONE;
        if (two_wg) mask |= 1ull << F_WG1_DONE;
        GMP_TRY(gmp_gate_wait(d.sync_flags, mask, d.epoch, d.sync_flags + F_ERR, main_));
    } else if (defer) {
        (void)hipStreamWaitEvent(main, ev[NEV - 1], 0);   // the next step's forward overwrites the saved batch statistics they read
    }
    if (d.dp_exchange) GMP_TRY(signal(F_BWD_DONE, ev[EV_BWD_DONE], main));
    if (timing) (void)hipEventRecord(phase_events()[GMP_STEP_PHASES], main);
    if (primary) ::g_sync_publish(g_sync);
    return GMP_OK;
}

// ---- the second enqueue thread ---------------------------------------------------------------------------------------------------
// One worker per process, started at the first two-lane step and never joined (it sleeps on a condition variable between jobs after a short
// spin: a step arrives every ~1.4 ms).  A job is one walk of step_body with the lane filter "everything but the main stream".
struct LaneJob {
    const gmp_step_desc* d;
    gmp_stream_t main_, aux_;
    const gmp_stream_t* task_streams;
    int device;
};
class LaneWorker {
  public:
    void post(const LaneJob& j) {
        job_ = j;
        done_.store(false, std::memory_order_relaxed);
        {
            std::lock_guard<std::mutex> lk(m_);
            seq_.fetch_add(1, std::memory_order_release);
        }
        cv_.notify_one();
    }
    int wait(char* err, size_t n) {          // spin: the worker finishes within tens of microseconds of the caller's own walk
        while (!done_.load(std::memory_order_acquire)) std::this_thread::yield();
        if (rc_ && err) snprintf(err, n, "%s", err_);
        return rc_;
    }
    static LaneWorker& get() {
        static LaneWorker* w = new LaneWorker();     // leaked on purpose: the thread may outlive static destruction
        return *w;
    }

  private:
    LaneWorker() { std::thread([this] { run(); }).detach(); }
    void run() {
        unsigned seen = 0;
        for (;;) {
            // a step every ~1.4 ms: spin for a while, then sleep
            const auto t0 = std::chrono::steady_clock::now();
            while (seq_.load(std::memory_order_acquire) == seen) {
                if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(400)) {
                    std::unique_lock<std::mutex> lk(m_);
                    cv_.wait(lk, [&] { return seq_.load(std::memory_order_acquire) != seen; });
                    break;
                }
            }
            seen = seq_.load(std::memory_order_acquire);
            (void)hipSetDevice(job_.device);
            gmp::lane_mode() = gmp::LANE_ALL_BUT;
            gmp::lane_stream() = (hipStream_t)job_.main_;
            rc_ = step_body(job_.d, job_.main_, job_.task_streams, job_.aux_, true, false);
            if (rc_) snprintf(err_, sizeof(err_), "%s", gmp::err_buf());
            gmp::lane_mode() = gmp::LANE_ALL;
            done_.store(true, std::memory_order_release);
        }
    }
    std::mutex m_;
    std::condition_variable cv_;
    std::atomic<unsigned> seq_{0};
    std::atomic<bool> done_{true};
    LaneJob job_{};
    int rc_ = 0;
    char err_[512] = "";
};

bool lanes_enabled() {
    static const bool on = !(getenv("GMP_STEP_LANES") && atoi(getenv("GMP_STEP_LANES")) == 0);
    return on;
}

}  // namespace

extern "C" int gmp_pretrain_step_fwd_bwd(const gmp_step_desc* dp, gmp_stream_t main_, const gmp_stream_t* task_streams, gmp_stream_t aux_) {
    if (!dp || !task_streams) return gmp::fail(GMP_ERR_ARG, "step: null descriptor");
    const gmp_step_desc& d = *dp;
    if (d.num_tasks < 1 || d.num_tasks > GMP_STEP_MAX_TASKS) return gmp::fail(GMP_ERR_ARG, "step: %d tasks", d.num_tasks);
    // Two enqueue threads when every cross-stream dependency of the step is a gate (nothing a host thread records for the other to wait on),
    // no phase timing, and there is a stream besides main to hand over.  GMP_STEP_LANES=0: one thread.
    bool other = false;
    for (int ti = 0; ti < d.num_tasks; ++ti) other = other || task_streams[ti] != main_;
    const bool two = lanes_enabled() && d.sync_flags != nullptr && d.gu_l[0] != nullptr && d.gz1_l[0] != nullptr && !phase_timing() &&
                     aux_ != main_ && other && gmp::lane_mode() == gmp::LANE_ALL;
    (void)events();                // (the static event pool: made before two threads could race for it)
    if (!two) return step_body(dp, main_, task_streams, aux_, false, true);
    LaneWorker& w = LaneWorker::get();
    w.post(LaneJob{dp, main_, aux_, task_streams, gmp::cur_device()});
    gmp::lane_mode() = gmp::LANE_ONLY;
    gmp::lane_stream() = (hipStream_t)main_;
    const int rc0 = step_body(dp, main_, task_streams, aux_, true, true);
    gmp::lane_mode() = gmp::LANE_ALL;
    char err[512];
    const int rc1 = w.wait(err, sizeof(err));
    if (!rc0 && rc1) return gmp::fail(rc1, "%s", err);
    return rc0;
}

// Data-parallel exchange beside the backward: `st` waits until the per-task gradients of one part of the model are final in
// task_grads -- part 0: the task heads (written before the stacked backward starts), part 1 + k: backbone layer L-1-k (its
// weight-gradient GEMMs and BatchNorm parameter sums on aux, its eps sum on main), last part: the mask token and the encoders
// (end of the backward).  Uses the events of the most recent gmp_pretrain_step_fwd_bwd of this process (dp_exchange set).
extern "C" int gmp_step_wait_grads(int part, gmp_stream_t st_) {
    if (part < 0 || part > GMP_STEP_LAYERS + 1) return gmp::fail(GMP_ERR_ARG, "step_wait_grads: part %d not in [0, %d]", part, GMP_STEP_LAYERS + 1);
    if (g_sync.flags) {       // the step ran with gates: one sleeping wave on `st` for the flags of this part
        uint64_t mask;
        if (part == 0) mask = g_sync.head_params_mask;
        else if (part == GMP_STEP_LAYERS + 1) mask = 1ull << F_BWD_DONE;
        else {
            const int l = GMP_STEP_LAYERS - part;
            mask = (1ull << (F_AUX_L + l)) | (1ull << (l > 0 ? F_BWD_MA + 2 * (l - 1) : F_L0));
        }
        return gmp_gate_wait(g_sync.flags, mask, g_sync.epoch, g_sync.flags + F_ERR, st_);
    }
    hipStream_t st = (hipStream_t)st_;
    hipEvent_t* ev = events();
    hipEvent_t* evl = ev + 4 + GMP_STEP_MAX_TASKS;
    hipError_t e = hipSuccess;
    if (part == 0) {      // input halves joined on main + every head's weight-gradient GEMMs (events of absent tasks: never recorded = no wait)
        e = hipStreamWaitEvent(st, ev[EV_HEADS_DONE], 0);
        for (int i = 0; i <= GMP_STEP_MAX_TASKS && e == hipSuccess; ++i) e = hipStreamWaitEvent(st, ev[EV_HEAD_PARAMS + i], 0);
    } else if (part == GMP_STEP_LAYERS + 1) {
        e = hipStreamWaitEvent(st, ev[EV_BWD_DONE], 0);
    } else {
        const int l = GMP_STEP_LAYERS - part;                    // 4 .. 0
        e = hipStreamWaitEvent(st, evl[4 * l + 3], 0);           // aux: dW2, dW1 and both BatchNorm parameter sums of layer l
        if (e == hipSuccess)                                     // main: past layer l's eps sum
            e = hipStreamWaitEvent(st, l > 0 ? evl[4 * (l - 1)] : ev[EV_LAYER0_DONE], 0);
    }
    return e == hipSuccess ? GMP_OK : gmp::fail(GMP_ERR_LAUNCH, "step_wait_grads: hipStreamWaitEvent failed");
}

extern "C" int gmp_step_phase_ms(float* out3) {
    if (!out3) return gmp::fail(GMP_ERR_ARG, "step_phase_ms: null pointer");
    float all[GMP_STEP_PHASES];
    if (int rc = gmp_step_phase_detail_ms(all)) return rc;
    out3[0] = out3[1] = out3[2] = 0.f;
    for (int i = 0; i < GMP_STEP_PHASES; ++i) out3[i <= GMP_STEP_LAYERS ? 0 : (i == GMP_STEP_LAYERS + 1 ? 1 : 2)] += all[i];
    return GMP_OK;
}

extern "C" int gmp_step_phase_detail_ms(float* out) {
    if (!out) return gmp::fail(GMP_ERR_ARG, "step_phase_detail_ms: null pointer");
    if (!phase_timing()) return gmp::fail(GMP_ERR_UNSUPPORTED, "step_phase_detail_ms: set GMP_STEP_TIMING=1 before the first step");
    hipEvent_t* e = phase_events();
    if (hipEventSynchronize(e[GMP_STEP_PHASES]) != hipSuccess) return gmp::fail(GMP_ERR_LAUNCH, "step_phase_detail_ms: no step recorded");
    for (int i = 0; i < GMP_STEP_PHASES; ++i)
        if (hipEventElapsedTime(&out[i], e[i], e[i + 1]) != hipSuccess) return gmp::fail(GMP_ERR_LAUNCH, "step_phase_detail_ms: elapsed");
    return GMP_OK;
}

// (diagnostic, GMP_STEP_TIMING=1) per task: ms from "stacked forward done" to the head's start, to the end of its input-gradient half and
// to the end of its weight-gradient half (0 where the head runs on the main stream and has no separate second half) -- out[3 * task + k]
extern "C" int gmp_step_head_ms(float* out, int max_tasks) {
    if (!out || max_tasks < g_head_tasks) return gmp::fail(GMP_ERR_ARG, "step_head_ms: bad argument");
    if (!phase_timing()) return gmp::fail(GMP_ERR_UNSUPPORTED, "step_head_ms: set GMP_STEP_TIMING=1 before the first step");
    hipEvent_t fwd_done = phase_events()[1 + GMP_STEP_LAYERS];
    (void)hipEventSynchronize(phase_events()[GMP_STEP_PHASES]);
    for (int i = 0; i < 3 * g_head_tasks; ++i) {
        out[i] = 0.f;
        if (g_head_recorded[i] && hipEventSynchronize(head_events()[i]) == hipSuccess) (void)hipEventElapsedTime(&out[i], fwd_done, head_events()[i]);
    }
    return GMP_OK;
}
