"""ctypes mirror of include/gnnmp_step.h (field order and types must match the C structs exactly;
gmp_step_desc_size() is compared against ctypes.sizeof at load time)."""
import ctypes as C

MAXD, MAXT, LAYERS, MAXG = 8, 8, 5, 24
TASK_KIND = {"node_feat_mask": 0, "link_pred": 1, "node_contrast": 2, "graph_contrast": 3, "graph_prop": 4, "domain_adv": 5}
i32, i64, u64, f32, p, sz = C.c_int32, C.c_int64, C.c_uint64, C.c_float, C.c_void_p, C.c_size_t


class Mlp2(C.Structure):
    _fields_ = [("k_in", i32), ("k_hid", i32), ("k_out", i32), ("site", i32), ("rows", i32 * (MAXD + 1)),
                ("off_w0", i64 * MAXD), ("off_b0", i64 * MAXD), ("off_w3", i64 * MAXD), ("off_b3", i64 * MAXD),
                ("tg_w0", i64 * MAXD), ("tg_b0", i64 * MAXD), ("tg_w3", i64 * MAXD), ("tg_b3", i64 * MAXD),
                ("x", p), ("y1", p), ("d1", p), ("y2", p), ("g_out", p), ("g_hid", p), ("g_in", p)]


class TaskDesc(C.Structure):
    _fields_ = [("kind", i32), ("row0", i32), ("row1", i32), ("g_scale", p), ("loss_sum", p),
                ("gemm_ws", p), ("gemm_ws_bytes", sz), ("loss_ws", p), ("loss_ws_bytes", sz),
                ("mlp", Mlp2), ("idx", p), ("num_idx", i64), ("nfm_target", p),
                ("ntx_n", i32 * MAXD), ("ntx_ws", p * MAXD), ("ntx_ws_bytes", sz * MAXD), ("ntx_sums", p), ("temperature", f32),
                ("pool_ptr", p), ("pool_gid", p), ("pool_B", i32), ("pool_r0", i32), ("pool_M", i32),
                ("pool_mean", p), ("pool_max", p), ("g_mean", p), ("g_max", p), ("labels", p),
                ("lp_K", i64), ("lp_edges", p), ("lp_labels", p), ("lp_pos", p),
                ("lp_feat", p), ("lp_y1", p), ("lp_d1", p), ("lp_y2", p), ("lp_p", p), ("lp_gp", p), ("lp_gy2", p), ("lp_gy1", p),
                ("lp_gfeat", p), ("lp_ghs", p), ("lp_ghd", p),
                ("lp_off_w0", i64), ("lp_off_b0", i64), ("lp_off_w3", i64), ("lp_off_b3", i64),
                ("lp_tg_w0", i64), ("lp_tg_b0", i64), ("lp_tg_w3", i64), ("lp_tg_b3", i64), ("lp_site", i32),
                ("da_labels", p), ("da_classes", i32), ("da_hidden", i32), ("da_lambda", f32), ("da_dropout", f32)]


class LayerDesc(C.Structure):
    _fields_ = [(n, i64) for n in ("off_eps", "off_w1", "off_b1", "off_g1", "off_be1", "off_w2", "off_b2", "off_g2", "off_be2")] + \
               [(n, i64 * MAXT) for n in ("tg_eps", "tg_w1", "tg_b1", "tg_g1", "tg_be1", "tg_w2", "tg_b2", "tg_g2", "tg_be2")] + \
               [(n, p) for n in ("rm1", "rv1", "rm2", "rv2", "a", "z1", "r1", "z2", "m1", "s1", "m2", "s2")]


class StepDesc(C.Structure):
    _fields_ = [("N", i32), ("E", i32), ("S", i32), ("max_seg", i32), ("num_tiles", i32), ("num_tasks", i32), ("num_domains", i32),
                ("dpad", i32), ("training", i32), ("hidden", i32), ("max_seg_edges", i32), ("dropout_p", f32), ("dp_exchange", i32), ("epoch", i32), ("upload_on_aux", i32), ("seed", u64), ("sync_flags", p),
                ("seg_ptr", p), ("seg_dom", p), ("src_row", p), ("tiles", p), ("seg_eptr", p), ("edge_index", p), ("rowmask", p),
                ("task_row", i32 * (MAXT + 1)), ("task_seg", i32 * (MAXT + 1)),
                ("csr", p * 6), ("csr_status", p), ("csr_ws", p), ("csr_ws_bytes", sz),
                ("lp_csr", p * 6), ("lp_csr_status", p), ("lp_csr_ws", p), ("lp_csr_ws_bytes", sz),
                ("lp_seg_ptr", p), ("lp_seg_eptr", p), ("lp_S", i32), ("lp_max_seg_rows", i64), ("lp_max_seg_edges", i64), ("lp_rows_end", i64),
                ("fwd_cut_seg", i32 * 2), ("fwd_cut_row", i32 * 2),
                ("flat", p), ("P", i64), ("task_grads", p),
                ("x_all", p), ("x_rows", i64),
                ("enc_off_w", i64 * MAXD), ("enc_off_b", i64 * MAXD), ("enc_d_in", i32 * MAXD),
                ("enc_off_gamma0", i64), ("enc_off_beta0", i64),
                ("enc_rm", p), ("enc_rv", p), ("enc_mean", p), ("enc_rstd", p), ("z0", p),
                ("enc_groups", i32), ("enc_gseg", i32 * (MAXG + 1)),
                ("enc_tg_w", i64 * MAXG), ("enc_tg_b", i64 * MAXG), ("enc_tg_gamma", i64 * MAXG), ("enc_tg_beta", i64 * MAXG),
                ("off_mask_token", i64), ("tg_mask_token", i64), ("nfm_task", i32),
                ("h", p * (LAYERS + 1)), ("layer", LayerDesc * LAYERS),
                ("gA", p), ("gB", p), ("gW", p), ("gW2", p), ("rowdot", p), ("ga", p), ("gB2", p), ("gW3", p), ("gu_l", p * LAYERS), ("gz1_l", p * LAYERS),
                ("bn_ws", p), ("bn_ws_bytes", sz), ("gemm_ws", p), ("gemm_ws_bytes", sz), ("loss_ws", p), ("loss_ws_bytes", sz),
                ("task", TaskDesc * MAXT)]
