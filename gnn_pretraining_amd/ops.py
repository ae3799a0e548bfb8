"""Raw (non-autograd) wrappers: torch tensors in, one C-ABI call each.

These check shapes/dtypes/devices on the host before the launch (a wrong shape
must raise here, not fault on the GPU) and allocate outputs/workspaces with
torch.  Autograd lives in operators.py.
"""
from __future__ import annotations

import ctypes as C
from typing import NamedTuple, Optional, Tuple

import torch
from torch import Tensor

from . import _lib as L


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)      # the handle without building a Stream object (5 us -> 0.3 us per op)


def _stream(t: Tensor) -> C.c_void_p:
    if _raw_stream is not None:
        idx = t.device.index
        return C.c_void_p(_raw_stream(idx if idx is not None else torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _ptr(t: Optional[Tensor]) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


def _need(t: Tensor, dtype, name: str, ndim: Optional[int] = None) -> None:
    if not t.is_cuda:
        raise L.GnnmpError(f"{name}: expected a GPU tensor (the HIP path has no CPU fallback), got {t.device}")
    if t.dtype != dtype:
        raise L.GnnmpError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise L.GnnmpError(f"{name}: must be contiguous")
    if ndim is not None and t.dim() != ndim:
        raise L.GnnmpError(f"{name}: expected {ndim} dims, got shape {tuple(t.shape)}")


def _need_rows(t: Tensor, name: str) -> None:
    """2-D fp32 GPU tensor whose rows are contiguous (stride 1 along a row, leading dimension >= the row length)."""
    if not t.is_cuda:
        raise L.GnnmpError(f"{name}: expected a GPU tensor (the HIP path has no CPU fallback), got {t.device}")
    if t.dtype != torch.float32 or t.dim() != 2:
        raise L.GnnmpError(f"{name}: expected a 2-D {torch.float32} tensor, got {t.dtype} {tuple(t.shape)}")
    if t.numel() and (t.stride(1) != 1 or (t.size(0) > 1 and t.stride(0) < t.size(1))):
        raise L.GnnmpError(f"{name}: rows must be contiguous (strides {t.stride()})")


def _ws(nbytes: int, device) -> Tensor:
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


class CSR(NamedTuple):
    """Both orientations of one edge_index (int32, device)."""
    rowptr: Tensor     # [N+1] grouped by target
    col: Tensor        # [E]   sources
    perm: Tensor       # [E]   COO edge id per slot
    rowptr_t: Tensor   # grouped by source (transposed graph)
    col_t: Tensor
    perm_t: Tensor
    num_nodes: int
    status: Tensor     # int32[1]: endpoints out of range (0 when well formed)


def csr_build(edge_index: Tensor, num_nodes: int) -> CSR:
    _need(edge_index, torch.int64, "edge_index", 2)
    if edge_index.size(0) != 2:
        raise L.GnnmpError(f"edge_index must be [2,E], got {tuple(edge_index.shape)}")
    E, dev = edge_index.size(1), edge_index.device
    mk = lambda n: torch.empty(n, dtype=torch.int32, device=dev)
    out = CSR(mk(num_nodes + 1), mk(E), mk(E), mk(num_nodes + 1), mk(E), mk(E), num_nodes, mk(1))
    l = L.lib()
    ws = _ws(l.gmp_csr_build_workspace_bytes(num_nodes, E), dev)
    L.check(l.gmp_csr_build(_ptr(edge_index), num_nodes, E, _ptr(out.rowptr), _ptr(out.col), _ptr(out.perm),
                            _ptr(out.rowptr_t), _ptr(out.col_t), _ptr(out.perm_t), _ptr(out.status),
                            _ptr(ws), ws.numel(), _stream(edge_index)), "gmp_csr_build")
    return out


def _feat_ok(x: Tensor, name: str) -> int:
    _need(x, torch.float32, name, 2)
    F = x.size(1)
    if F % 4 or F > 1024:
        raise L.GnnmpError(f"{name}: feature width {F} must be a multiple of 4 and <= 1024")
    return F


def gin_aggregate_fwd(x: Tensor, rowptr: Tensor, col: Tensor, eps: Tensor) -> Tensor:
    F = _feat_ok(x, "x")
    _need(rowptr, torch.int32, "rowptr", 1); _need(col, torch.int32, "col", 1); _need(eps, torch.float32, "eps")
    if rowptr.numel() != x.size(0) + 1:
        raise L.GnnmpError(f"rowptr has {rowptr.numel()} entries for {x.size(0)} rows")
    out = torch.empty_like(x)
    L.check(L.lib().gmp_gin_aggregate_fwd(_ptr(x), _ptr(rowptr), _ptr(col), _ptr(eps), _ptr(out), x.size(0), F,
                                          _stream(x)), "gmp_gin_aggregate_fwd")
    return out


def gin_aggregate_bwd(g_out: Tensor, rowptr_t: Tensor, col_t: Tensor, eps: Tensor,
                      x: Optional[Tensor]) -> Tuple[Tensor, Optional[Tensor]]:
    F = _feat_ok(g_out, "g_out")
    _need(rowptr_t, torch.int32, "rowptr_t", 1); _need(col_t, torch.int32, "col_t", 1)
    if rowptr_t.numel() != g_out.size(0) + 1:
        raise L.GnnmpError("rowptr_t / g_out row mismatch")
    g_x = torch.empty_like(g_out)
    g_eps = None
    if x is not None:
        _feat_ok(x, "x")
        if x.shape != g_out.shape:
            raise L.GnnmpError("x / g_out shape mismatch")
        g_eps = torch.empty(1, dtype=torch.float32, device=g_out.device)
    l = L.lib()
    ws = _ws(l.gmp_gin_aggregate_bwd_workspace_bytes(g_out.size(0), F), g_out.device)
    L.check(l.gmp_gin_aggregate_bwd(_ptr(g_out), _ptr(rowptr_t), _ptr(col_t), _ptr(eps), _ptr(x), _ptr(g_x),
                                    _ptr(g_eps), g_out.size(0), F, _ptr(ws), ws.numel(), _stream(g_out)),
            "gmp_gin_aggregate_bwd")
    return g_x, g_eps


def segment_sum(src: Tensor, ptr: Tensor, idx: Optional[Tensor], mean: bool = False,
                out: Optional[Tensor] = None, accumulate: bool = False) -> Tensor:
    _need(src, torch.float32, "src", 2)
    F = src.size(1)
    _need(ptr, torch.int32, "ptr", 1)
    if idx is not None:
        _need(idx, torch.int32, "idx", 1)
    nseg = ptr.numel() - 1
    if out is None:
        out = torch.empty(nseg, F, dtype=torch.float32, device=src.device)
        accumulate = False
    L.check(L.lib().gmp_segment_sum(_ptr(src), _ptr(ptr), _ptr(idx), _ptr(out), nseg, F, int(mean), int(accumulate),
                                    _stream(src)), "gmp_segment_sum")
    return out


def row_gather(src: Tensor, idx: Tensor, seg_ptr: Optional[Tensor] = None) -> Tensor:
    _need(src, torch.float32, "src", 2)
    F = src.size(1)
    _need(idx, torch.int64, "idx", 1)
    if seg_ptr is not None:
        _need(seg_ptr, torch.int32, "seg_ptr", 1)
        if seg_ptr.numel() != src.size(0) + 1:
            raise L.GnnmpError("seg_ptr must have one entry per source row + 1")
    out = torch.empty(idx.numel(), F, dtype=torch.float32, device=src.device)
    L.check(L.lib().gmp_row_gather(_ptr(src), _ptr(idx), _ptr(seg_ptr), _ptr(out), idx.numel(), src.size(0), F,
                                   _stream(src)), "gmp_row_gather")
    return out


def segment_max_fwd(x: Tensor, ptr: Tensor) -> Tensor:
    F = _feat_ok(x, "x")
    _need(ptr, torch.int32, "ptr", 1)
    out = torch.empty(ptr.numel() - 1, F, dtype=torch.float32, device=x.device)
    L.check(L.lib().gmp_segment_max_fwd(_ptr(x), _ptr(ptr), _ptr(out), ptr.numel() - 1, F, _stream(x)),
            "gmp_segment_max_fwd")
    return out


def segment_max_bwd(g_out: Tensor, x: Tensor, out: Tensor, ptr: Tensor) -> Tensor:
    F = _feat_ok(x, "x")
    _need(g_out, torch.float32, "g_out", 2); _need(out, torch.float32, "out", 2); _need(ptr, torch.int32, "ptr", 1)
    if g_out.shape != out.shape or out.size(0) != ptr.numel() - 1 or out.size(1) != F:
        raise L.GnnmpError("segment_max_bwd: shape mismatch")
    g_x = torch.empty_like(x)
    L.check(L.lib().gmp_segment_max_bwd(_ptr(g_out), _ptr(x), _ptr(out), _ptr(ptr), _ptr(g_x), ptr.numel() - 1, F, 0,
                                        _stream(x)), "gmp_segment_max_bwd")
    return g_x


NT, NN, TN = 0, 1, 2


def gemm(mode: int, A: Tensor, B: Tensor, bias: Optional[Tensor] = None, out: Optional[Tensor] = None,
         alpha: float = 1.0, accumulate: bool = False, relu: bool = False) -> Tensor:
    """NT: A[M,K] B[N,K]^T ; NN: A[M,K] B[K,N] ; TN: A[K,M]^T B[K,N].  Operands may be row-strided views (unit stride along a row, any
    leading dimension >= the row length): the fine-tune engine keeps the encoder weight as a [256, 1433] view of its K-padded [256, 1440]
    slot, and evaluation through the module multiplies by that view."""
    _need_rows(A, "A"); _need_rows(B, "B")
    if mode == NT:
        M, K, N = A.size(0), A.size(1), B.size(0); kb = B.size(1)
    elif mode == NN:
        M, K, N = A.size(0), A.size(1), B.size(1); kb = B.size(0)
    elif mode == TN:
        M, K, N = A.size(1), A.size(0), B.size(1); kb = B.size(0)
    else:
        raise L.GnnmpError(f"gemm mode {mode}")
    if kb != K:
        raise L.GnnmpError(f"gemm: inner dims differ ({K} vs {kb}) for mode {mode}")
    if bias is not None:
        _need(bias, torch.float32, "bias", 1)
        if bias.numel() != N:
            raise L.GnnmpError("gemm: bias length")
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=A.device)
        accumulate = False
    else:
        _need(out, torch.float32, "out", 2)
        if tuple(out.shape) != (M, N):
            raise L.GnnmpError("gemm: out shape")
    l = L.lib()
    wsb = l.gmp_gemm_f32_workspace_bytes(mode, M, N, K)
    ws = _ws(wsb, A.device) if wsb else None
    L.check(l.gmp_gemm_f32(mode, _ptr(A), _ptr(B), _ptr(bias), _ptr(out), M, N, K, A.stride(0) if A.size(0) > 1 else A.size(1),
                           B.stride(0) if B.size(0) > 1 else B.size(1), N,
                           float(alpha), int(accumulate), int(relu), _ptr(ws), wsb, _stream(A)), "gmp_gemm_f32")
    return out


def colsum(A: Tensor, out: Optional[Tensor] = None, accumulate: bool = False) -> Tensor:
    _need(A, torch.float32, "A", 2)
    M, N = A.shape
    if out is None:
        out = torch.empty(N, dtype=torch.float32, device=A.device)
        accumulate = False
    l = L.lib()
    wsb = l.gmp_colsum_workspace_bytes(M, N)
    ws = _ws(wsb, A.device)
    L.check(l.gmp_colsum(_ptr(A), _ptr(out), M, N, N, int(accumulate), _ptr(ws), ws.numel(), _stream(A)), "gmp_colsum")
    return out


def make_bn_config(training: bool, relu: bool, dropout_p: float = 0.0, seed: int = 0, stream_id: int = 0,
                   eps: float = 1e-5, momentum: float = 0.1, sync: Optional[Tensor] = None) -> L.BnConfig:
    """sync: zero-filled int32 device tensor of bn_sync_words(channels, segments) words (gmp_bn_config.sync): segments of 1,025-4,096
    rows then run as 128-row slabs over the whole chip.  The caller keeps the tensor alive while launches that got it are in flight."""
    if sync is None:
        return L.BnConfig(int(training), int(relu), eps, momentum, float(dropout_p), seed & (2 ** 64 - 1), stream_id)
    _need(sync, torch.int32, "sync", 1)
    return L.BnConfig(int(training), int(relu), eps, momentum, float(dropout_p), seed & (2 ** 64 - 1), stream_id, None, sync.data_ptr(),
                      sync.numel())


def bn_sync_words(channels: int, num_segments: int) -> int:
    return L.lib().gmp_bn_sync_bytes(channels, num_segments) // 4


def bn_fwd(x: Tensor, residual: Optional[Tensor], seg_ptr: Tensor, max_seg_rows: int, gamma: Tensor, beta: Tensor,
           running_mean: Optional[Tensor], running_var: Optional[Tensor], cfg: L.BnConfig):
    """Returns (y, save_mean, save_rstd); the saved stats are None in eval mode."""
    _need(x, torch.float32, "x", 2)
    rows, Cc = x.shape
    if residual is not None:
        _need(residual, torch.float32, "residual", 2)
        if residual.shape != x.shape:
            raise L.GnnmpError("bn: residual shape")
    _need(seg_ptr, torch.int32, "seg_ptr", 1)
    S = seg_ptr.numel() - 1
    for t, n in ((gamma, "gamma"), (beta, "beta")):
        _need(t, torch.float32, n, 1)
        if t.numel() != Cc:
            raise L.GnnmpError(f"bn: {n} length")
    y = torch.empty_like(x)
    sm = sr = None
    if cfg.training:
        sm = torch.empty(S, Cc, dtype=torch.float32, device=x.device)
        sr = torch.empty(S, Cc, dtype=torch.float32, device=x.device)
    l = L.lib()
    ws = _ws(l.gmp_bn_workspace_bytes(rows, Cc, S, max_seg_rows), x.device)
    L.check(l.gmp_bn_fwd(_ptr(x), _ptr(residual), _ptr(seg_ptr), None, S, max_seg_rows, rows, Cc, _ptr(gamma), _ptr(beta),
                         _ptr(running_mean), _ptr(running_var), _ptr(sm), _ptr(sr), _ptr(y), C.byref(cfg),
                         _ptr(ws), ws.numel(), _stream(x)), "gmp_bn_fwd")
    return y, sm, sr


def bn_bwd(g_y: Tensor, x: Tensor, residual: Optional[Tensor], seg_ptr: Tensor, max_seg_rows: int, gamma: Tensor,
           beta: Tensor, running_mean, running_var, save_mean, save_rstd, cfg: L.BnConfig, group_seg_ptr=None):
    """Returns (g_u, g_gamma [G,C], g_beta [G,C]); group_seg_ptr: python list of segment offsets (default one group)."""
    _need(g_y, torch.float32, "g_y", 2); _need(x, torch.float32, "x", 2)
    rows, Cc = x.shape
    S = seg_ptr.numel() - 1
    grp = [0, S] if group_seg_ptr is None else list(group_seg_ptr)
    G = len(grp) - 1
    arr = (C.c_int32 * (G + 1))(*grp)
    g_u = torch.empty_like(x)
    gg = torch.empty(G, Cc, dtype=torch.float32, device=x.device)
    gb = torch.empty(G, Cc, dtype=torch.float32, device=x.device)
    l = L.lib()
    ws = _ws(l.gmp_bn_workspace_bytes(rows, Cc, S, max_seg_rows), x.device)
    L.check(l.gmp_bn_bwd(_ptr(g_y), _ptr(x), _ptr(residual), _ptr(seg_ptr), None, S, max_seg_rows, rows, Cc, _ptr(gamma),
                         _ptr(beta), _ptr(running_mean), _ptr(running_var), _ptr(save_mean), _ptr(save_rstd),
                         _ptr(g_u), _ptr(gg), _ptr(gb), C.cast(arr, C.c_void_p), None, None, G, C.byref(cfg), _ptr(ws), ws.numel(),
                         _stream(x)), "gmp_bn_bwd")
    return g_u, gg, gb


def lp_edge_features_fwd(h: Tensor, edges: Tensor) -> Tensor:
    F = _feat_ok(h, "h")
    _need(edges, torch.int64, "edges", 2)
    K = edges.size(1)
    feat = torch.empty(K, 3 * F, dtype=torch.float32, device=h.device)
    L.check(L.lib().gmp_lp_edge_features_fwd(_ptr(h), _ptr(edges), _ptr(feat), h.size(0), K, F, _stream(h)),
            "gmp_lp_edge_features_fwd")
    return feat


def lp_edge_features_bwd(g_feat: Tensor, h: Tensor, edges: Tensor) -> Tuple[Tensor, Tensor]:
    F = _feat_ok(h, "h")
    _need(g_feat, torch.float32, "g_feat", 2); _need(edges, torch.int64, "edges", 2)
    K = edges.size(1)
    if tuple(g_feat.shape) != (K, 3 * F):
        raise L.GnnmpError("lp_edge_features_bwd: g_feat shape")
    ghs = torch.empty(K, F, dtype=torch.float32, device=h.device)
    ghd = torch.empty_like(ghs)
    L.check(L.lib().gmp_lp_edge_features_bwd(_ptr(g_feat), _ptr(h), _ptr(edges), _ptr(ghs), _ptr(ghd), h.size(0), K, F,
                                             _stream(h)), "gmp_lp_edge_features_bwd")
    return ghs, ghd


def nt_xent_fwd(z1: Tensor, z2: Tensor, temperature: float) -> Tuple[Tensor, Tensor]:
    """Returns (loss_sum [1], workspace) -- the workspace feeds nt_xent_bwd."""
    _need(z1, torch.float32, "z1", 2); _need(z2, torch.float32, "z2", 2)
    if z1.shape != z2.shape:
        raise L.GnnmpError("nt_xent: z1/z2 shapes differ")
    n, d = z1.shape
    l = L.lib()
    ws = _ws(l.gmp_nt_xent_workspace_bytes(n, d), z1.device)
    loss = torch.empty(1, dtype=torch.float32, device=z1.device)
    L.check(l.gmp_nt_xent_fwd(_ptr(z1), _ptr(z2), n, d, float(temperature), _ptr(loss), _ptr(ws), ws.numel(),
                              _stream(z1)), "gmp_nt_xent_fwd")
    return loss, ws


def nt_xent_bwd(z1: Tensor, z2: Tensor, temperature: float, g_scale: Tensor, ws: Tensor) -> Tuple[Tensor, Tensor]:
    n, d = z1.shape
    _need(g_scale, torch.float32, "g_scale")
    g1, g2 = torch.empty_like(z1), torch.empty_like(z2)
    L.check(L.lib().gmp_nt_xent_bwd(_ptr(z1), _ptr(z2), n, d, float(temperature), _ptr(g_scale), _ptr(g1), _ptr(g2),
                                    _ptr(ws), ws.numel(), _stream(z1)), "gmp_nt_xent_bwd")
    return g1, g2


def hard_negative_topk(emb: Tensor, existing_edges: Tensor, k: int, return_scores: bool = False,
                       return_matrix: bool = False):
    """Top-k cosine-similarity non-edges (finetune.py:45-75).  Returns edges [2,k] int64 (score-descending), and
    optionally the selected scores [k] and the masked n x n similarity matrix."""
    _need(emb, torch.float32, "emb", 2); _need(existing_edges, torch.int64, "existing_edges", 2)
    n, d = emb.shape
    E = existing_edges.size(1)
    l = L.lib()
    ws = _ws(l.gmp_hard_negative_workspace_bytes(n, d), emb.device)
    out = torch.empty(2, k, dtype=torch.int64, device=emb.device)
    sc = torch.empty(k, dtype=torch.float32, device=emb.device) if return_scores else None
    mat = torch.empty(n, n, dtype=torch.float32, device=emb.device) if return_matrix else None
    L.check(l.gmp_hard_negative_topk(_ptr(emb), n, d, _ptr(existing_edges) if E else None, E, k, _ptr(out) if k else None,
                                     _ptr(sc) if sc is not None else None, _ptr(mat) if mat is not None else None,
                                     _ptr(ws), ws.numel(), _stream(emb)), "gmp_hard_negative_topk")
    res = (out,)
    if return_scores:
        res += (sc,)
    if return_matrix:
        res += (mat,)
    return res if len(res) > 1 else out


def nt_xent_grouped(z: Tensor, ns, row_offsets, temperature: float, g_scale: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """Several NT-Xent problems in one call: group g's rows [z1; z2] (2*ns[g] of them) start at row_offsets[g] of z.
    Returns (g_z [same shape as z; rows outside every group untouched = 0], loss_sums [G], loss_total [1])."""
    _need(z, torch.float32, "z", 2); _need(g_scale, torch.float32, "g_scale")
    G, d = len(ns), z.size(1)
    l = L.lib()
    ws = _ws(l.gmp_nt_xent_grouped_workspace_bytes(G, max(max(ns), 1), d), z.device)
    gz = torch.zeros_like(z)
    sums = torch.empty(G, dtype=torch.float32, device=z.device)
    total = torch.empty(1, dtype=torch.float32, device=z.device)
    n_arr = (C.c_int32 * G)(*[int(n) for n in ns])
    o_arr = (C.c_int64 * G)(*[int(o) for o in row_offsets])
    L.check(l.gmp_nt_xent_grouped(_ptr(z), _ptr(gz), G, n_arr, o_arr, d, float(temperature), _ptr(g_scale), _ptr(sums), _ptr(total),
                                  _ptr(ws), ws.numel(), _stream(z)), "gmp_nt_xent_grouped")
    return gz, sums, total


def dropout_fwd(x: Tensor, p: float, seed: int, stream_id: int) -> Tensor:
    _need(x, torch.float32, "x")
    if x.numel() % 4:
        raise L.GnnmpError("dropout: numel must be a multiple of 4")
    y = torch.empty_like(x)
    L.check(L.lib().gmp_dropout_fwd(_ptr(x), _ptr(y), x.numel(), float(p), seed & (2 ** 64 - 1), stream_id, _stream(x)),
            "gmp_dropout_fwd")
    return y


def relu_dropout_bwd(g: Tensor, act: Tensor, p: float, seed: int, stream_id: int) -> Tensor:
    _need(g, torch.float32, "g"); _need(act, torch.float32, "act")
    if g.shape != act.shape or g.numel() % 4:
        raise L.GnnmpError("relu_dropout_bwd: shapes")
    out = torch.empty_like(g)
    L.check(L.lib().gmp_relu_dropout_bwd(_ptr(g), _ptr(act), _ptr(out), g.numel(), float(p), seed & (2 ** 64 - 1),
                                         stream_id, _stream(g)), "gmp_relu_dropout_bwd")
    return out


def _loss_ws(n: int, device) -> Tensor:
    return _ws(L.lib().gmp_loss_workspace_bytes(n), device)


def mse_sum_fwd(a: Tensor, b: Tensor) -> Tensor:
    _need(a, torch.float32, "a"); _need(b, torch.float32, "b")
    if a.shape != b.shape:
        raise L.GnnmpError(f"mse_sum: shapes {tuple(a.shape)} vs {tuple(b.shape)}")
    loss = torch.empty(1, dtype=torch.float32, device=a.device)
    ws = _loss_ws(a.numel(), a.device)
    L.check(L.lib().gmp_mse_sum_fwd(_ptr(a), _ptr(b), a.numel(), _ptr(loss), _ptr(ws), ws.numel(), _stream(a)), "gmp_mse_sum_fwd")
    return loss


def mse_sum_bwd(a: Tensor, b: Tensor, g_scale: Tensor) -> Tensor:
    ga = torch.empty_like(a)
    L.check(L.lib().gmp_mse_sum_bwd(_ptr(a), _ptr(b), _ptr(g_scale), _ptr(ga), a.numel(), _stream(a)), "gmp_mse_sum_bwd")
    return ga


def sigmoid_fwd(x: Tensor) -> Tensor:
    _need(x, torch.float32, "x")
    y = torch.empty_like(x)
    L.check(L.lib().gmp_sigmoid_fwd(_ptr(x), _ptr(y), x.numel(), _stream(x)), "gmp_sigmoid_fwd")
    return y


def sigmoid_bwd(g: Tensor, y: Tensor) -> Tensor:
    _need(g, torch.float32, "g")
    out = torch.empty_like(y)
    L.check(L.lib().gmp_sigmoid_bwd(_ptr(g), _ptr(y), _ptr(out), y.numel(), _stream(y)), "gmp_sigmoid_bwd")
    return out


def bce_sum_fwd(p: Tensor, labels: Tensor) -> Tensor:
    _need(p, torch.float32, "p"); _need(labels, torch.float32, "labels")
    if p.shape != labels.shape:
        raise L.GnnmpError("bce_sum: shapes differ")
    loss = torch.empty(1, dtype=torch.float32, device=p.device)
    ws = _loss_ws(p.numel(), p.device)
    L.check(L.lib().gmp_bce_sum_fwd(_ptr(p), _ptr(labels), p.numel(), _ptr(loss), _ptr(ws), ws.numel(), _stream(p)), "gmp_bce_sum_fwd")
    return loss


def bce_sum_bwd(p: Tensor, labels: Tensor, g_scale: Tensor) -> Tensor:
    gp = torch.empty_like(p)
    L.check(L.lib().gmp_bce_sum_bwd(_ptr(p), _ptr(labels), _ptr(g_scale), _ptr(gp), p.numel(), _stream(p)), "gmp_bce_sum_bwd")
    return gp


def cross_entropy_sum_fwd(logits: Tensor, target: Tensor) -> Tensor:
    _need(logits, torch.float32, "logits", 2); _need(target, torch.int64, "target", 1)
    if target.numel() != logits.size(0):
        raise L.GnnmpError("cross_entropy: target length")
    loss = torch.empty(1, dtype=torch.float32, device=logits.device)
    ws = _loss_ws(logits.size(0), logits.device)
    L.check(L.lib().gmp_cross_entropy_sum_fwd(_ptr(logits), _ptr(target), logits.size(0), logits.size(1), _ptr(loss),
                                              _ptr(ws), ws.numel(), _stream(logits)), "gmp_cross_entropy_sum_fwd")
    return loss


def cross_entropy_sum_bwd(logits: Tensor, target: Tensor, g_scale: Tensor) -> Tensor:
    gl = torch.empty_like(logits)
    L.check(L.lib().gmp_cross_entropy_sum_bwd(_ptr(logits), _ptr(target), logits.size(0), logits.size(1), _ptr(g_scale),
                                              _ptr(gl), _stream(logits)), "gmp_cross_entropy_sum_bwd")
    return gl


def row_fill_(dst: Tensor, idx: Tensor, src: Tensor, broadcast: bool) -> Tensor:
    F = _feat_ok(dst, "dst")
    _need(idx, torch.int64, "idx", 1); _need(src, torch.float32, "src")
    if broadcast:
        if src.numel() != F:
            raise L.GnnmpError("row_fill: broadcast source must have F elements")
    elif tuple(src.shape) != (idx.numel(), F):
        raise L.GnnmpError("row_fill: source shape")
    L.check(L.lib().gmp_row_fill(_ptr(dst), _ptr(idx), _ptr(src), idx.numel(), dst.size(0), F, int(broadcast), _stream(dst)),
            "gmp_row_fill")
    return dst


# ---- device-side augmentation and masking (csrc/augment.hip) ---------------------------------------------
def view_sizes(ptr_host) -> Tuple[list, list]:
    """Per-graph kept-node counts of an augmented view and masked-node counts of node-feature masking: functions of the graph
    sizes alone (augmentations.py:49-50, pretrain_model.py:75-76), so the host lays out offsets without asking the device."""
    n = [int(b - a) for a, b in zip(ptr_host[:-1], ptr_host[1:])]
    kept = [v - max(1, int(v * 0.2)) if v >= 3 else v for v in n]
    masked = [max(1, int(v * 0.15)) if v >= 3 else 0 for v in n]
    return kept, masked


class DeviceViews(NamedTuple):
    """gmp_aug_two_views outputs of one domain batch (all on the device): per view rows / rowmask / edges [2, ecap] / common, plus
    counts [3G] (edges view 1, edges view 2, common, per graph) and totals [5] (E1, E2, common, mask flag 1, mask flag 2)."""
    rows: Tuple[Tensor, Tensor]
    rowmask: Tuple[Tensor, Tensor]
    edges: Tuple[Tensor, Tensor]
    common: Tuple[Tensor, Tensor]
    counts: Tensor
    totals: Tensor
    view_ptr: list


def aug_node_masks(ptr: Tensor, ptr_host, seed: int, stream_id: int) -> Tensor:
    """Node-feature-masking indices of a domain batch, drawn on the device (pretrain_model.py:71-80's per-graph loop)."""
    _need(ptr, torch.int64, "ptr", 1)
    _, masked = view_sizes(ptr_host)
    off = [0]
    for m in masked:
        off.append(off[-1] + m)
    out = torch.empty(off[-1], dtype=torch.int64, device=ptr.device)
    if off[-1] == 0:
        return out
    out_ptr = torch.tensor(off, dtype=torch.int64).to(ptr.device)
    G = len(ptr_host) - 1
    L.check(L.lib().gmp_aug_node_masks(_ptr(ptr), _ptr(out_ptr), G, max(int(b - a) for a, b in zip(ptr_host[:-1], ptr_host[1:])),
                                       seed & (2 ** 64 - 1), stream_id & 0xffffffff, _ptr(out), _stream(ptr)), "gmp_aug_node_masks")
    return out


def aug_two_views(ptr: Tensor, eptr: Tensor, edge_index: Tensor, ptr_host, eptr_host, num_features: int, seed: int,
                  stream_id: int) -> DeviceViews:
    """GraphAugmentor.create_two_views (augmentations.py:88-111) of a domain batch on the device."""
    _need(ptr, torch.int64, "ptr", 1); _need(eptr, torch.int64, "eptr", 1); _need(edge_index, torch.int64, "edge_index", 2)
    dev, G = ptr.device, len(ptr_host) - 1
    N, E = int(ptr_host[-1]), int(edge_index.size(1))
    kept, _ = view_sizes(ptr_host)
    vp = [0]
    for k in kept:
        vp.append(vp[-1] + k)
    vptr = torch.tensor(vp, dtype=torch.int64).to(dev)
    mk = lambda n, dt=torch.int64: torch.empty(n, dtype=dt, device=dev)
    rows, masks = (mk(vp[-1]), mk(vp[-1])), (mk(vp[-1]), mk(vp[-1]))
    edges, common = (mk(2 * max(E, 1)).view(2, -1), mk(2 * max(E, 1)).view(2, -1)), (mk(vp[-1]), mk(vp[-1]))
    counts, totals = mk(5 * max(G, 1), torch.int32), mk(5, torch.int32)
    l = L.lib()
    ws = _ws(l.gmp_aug_workspace_bytes(N, E, G), dev)
    L.check(l.gmp_aug_two_views(_ptr(ptr), _ptr(eptr), _ptr(edge_index.contiguous()), N, E, _ptr(vptr), G,
                                max([int(b - a) for a, b in zip(ptr_host[:-1], ptr_host[1:])] + [0]),
                                max([int(b - a) for a, b in zip(eptr_host[:-1], eptr_host[1:])] + [0]), num_features,
                                seed & (2 ** 64 - 1), stream_id & 0xffffffff, _ptr(rows[0]), _ptr(rows[1]), _ptr(masks[0]), _ptr(masks[1]),
                                _ptr(edges[0]), _ptr(edges[1]), max(E, 1), _ptr(common[0]), _ptr(common[1]), _ptr(counts), _ptr(totals),
                                _ptr(ws), ws.numel(), _stream(ptr)), "gmp_aug_two_views")
    return DeviceViews(rows, masks, edges, common, counts, totals, vp)


def aug_node_masks_batch(jobs, seed: int):
    """gmp_aug_node_masks for several domain batches in ONE launch.  jobs: [(ptr, ptr_host, stream_id)]; returns a list of index tensors
    identical to aug_node_masks(ptr, ptr_host, seed, stream_id) per job."""
    outs, cjobs, keep, nmax = [], [], [], 1
    for ptr, ptr_host, sid in jobs:
        _need(ptr, torch.int64, "ptr", 1)
        _, masked = view_sizes(ptr_host)
        off = [0]
        for m in masked:
            off.append(off[-1] + m)
        out = torch.empty(off[-1], dtype=torch.int64, device=ptr.device)
        outs.append(out)
        if off[-1] == 0:
            continue
        out_ptr = torch.tensor(off, dtype=torch.int64).to(ptr.device)
        keep.append(out_ptr)
        nmax = max(nmax, max(int(b - a) for a, b in zip(ptr_host[:-1], ptr_host[1:])))
        cjobs.append(L.AugMasksJob(_ptr(ptr), _ptr(out_ptr), len(ptr_host) - 1, sid & 0xffffffff, _ptr(out)))
    if cjobs:
        L.check(L.lib().gmp_aug_node_masks_batch((L.AugMasksJob * len(cjobs))(*cjobs), len(cjobs), nmax, seed & (2 ** 64 - 1), _stream(jobs[0][0])),
                "gmp_aug_node_masks_batch")
    return outs


def aug_two_views_batch(jobs, seed: int):
    """gmp_aug_two_views for several domain batches in TWO launches.  jobs: [(ptr, eptr, edge_index, ptr_host, eptr_host, num_features,
    stream_id)]; returns a list of DeviceViews identical to aug_two_views(..., seed, stream_id) per job."""
    res, cjobs, keep, nmax, emax = [], [], [], 1, 0
    l = L.lib()
    for ptr, eptr, edge_index, ptr_host, eptr_host, F, sid in jobs:
        dev, G = ptr.device, len(ptr_host) - 1
        N, E = int(ptr_host[-1]), int(edge_index.size(1))
        kept, _ = view_sizes(ptr_host)
        vp = [0]
        for k in kept:
            vp.append(vp[-1] + k)
        vptr = torch.tensor(vp, dtype=torch.int64).to(dev)
        mk = lambda n, dt=torch.int64: torch.empty(n, dtype=dt, device=dev)
        rows, masks = (mk(vp[-1]), mk(vp[-1])), (mk(vp[-1]), mk(vp[-1]))
        edges, common = (mk(2 * max(E, 1)).view(2, -1), mk(2 * max(E, 1)).view(2, -1)), (mk(vp[-1]), mk(vp[-1]))
        counts, totals = mk(5 * max(G, 1), torch.int32), mk(5, torch.int32)
        ws = _ws(l.gmp_aug_workspace_bytes(N, E, G), dev).clone()          # (one region per job: the jobs of a launch run concurrently)
        ei = edge_index.contiguous()
        keep += [vptr, ws, ei]
        nmax = max([nmax] + [int(b - a) for a, b in zip(ptr_host[:-1], ptr_host[1:])])
        emax = max([emax] + [int(b - a) for a, b in zip(eptr_host[:-1], eptr_host[1:])])
        cjobs.append(L.AugViewsJob(_ptr(ptr), _ptr(eptr), _ptr(ei), N, E, _ptr(vptr), G, F, sid & 0xffffffff, _ptr(rows[0]), _ptr(rows[1]),
                                   _ptr(masks[0]), _ptr(masks[1]), _ptr(edges[0]), _ptr(edges[1]), max(E, 1), _ptr(common[0]), _ptr(common[1]),
                                   _ptr(counts), _ptr(totals), _ptr(ws), ws.numel()))
        res.append(DeviceViews(rows, masks, edges, common, counts, totals, vp))
    if cjobs:
        L.check(l.gmp_aug_two_views_batch((L.AugViewsJob * len(cjobs))(*cjobs), len(cjobs), nmax, emax, seed & (2 ** 64 - 1), _stream(jobs[0][0])),
                "gmp_aug_two_views_batch")
        torch.cuda.current_stream(jobs[0][0].device).synchronize()          # (the per-job scratch above dies with this frame)
    return res


def dropout_rowdot_fwd(x: Tensor, w: Tensor, bias: Optional[Tensor], p: float, seed: int, stream_id: int):
    """(dropout(x), dropout(x) @ w + bias) for a Linear(F, 1) (gmp_dropout_rowdot_fwd); the dropped copy is x itself when p == 0."""
    _need(x, torch.float32, "x", 2)
    rows, F = x.shape
    d = torch.empty_like(x) if p > 0 else x
    y = torch.empty(rows, dtype=torch.float32, device=x.device)
    L.check(L.lib().gmp_dropout_rowdot_fwd(_ptr(x), _ptr(w), _ptr(bias), _ptr(d), _ptr(y), rows, F, float(p), seed & (2 ** 64 - 1), stream_id,
                                           _stream(x)), "gmp_dropout_rowdot_fwd")
    return d, y


def outer_relu_dropout_bwd(g: Tensor, w: Tensor, act: Tensor, p: float, seed: int, stream_id: int) -> Tensor:
    """g[m] * w[c] pushed through the dropout mask of the forward and the ReLU whose output was `act` (gmp_outer_relu_dropout_bwd)."""
    rows, F = act.shape
    out = torch.empty_like(act)
    L.check(L.lib().gmp_outer_relu_dropout_bwd(_ptr(g), _ptr(w), _ptr(act), _ptr(out), rows, F, float(p), seed & (2 ** 64 - 1), stream_id,
                                               _stream(act)), "gmp_outer_relu_dropout_bwd")
    return out


def weighted_colsum(g: Tensor, x: Tensor):
    """(sum_m g[m] x[m, :], sum_m g[m]) -- weight and bias gradient of a Linear(F, 1) (gmp_weighted_colsum)."""
    rows, F = x.shape
    ow = torch.empty(F, dtype=torch.float32, device=x.device)
    ob = torch.empty(1, dtype=torch.float32, device=x.device)
    l = L.lib()
    ws = _ws(l.gmp_weighted_colsum_workspace_bytes(rows, F), x.device)
    L.check(l.gmp_weighted_colsum(_ptr(g), _ptr(x), _ptr(ow), _ptr(ob), rows, F, _ptr(ws), ws.numel(), _stream(x)), "gmp_weighted_colsum")
    return ow, ob


def lp_pair_head(y1: Tensor, w: Tensor, bias: Tensor, pos: Tensor, sign: Tensor, g_scale: Tensor, p: float, seed: int, stream_id: int):
    """The link-prediction scorer's tail over MERGED rows (one per unordered pair), one dropout mask / score / BCE term per ORDERED row of the
    reference's list (heads.py:44-52, tasks.py:111-120; gnnmp.h gmp_lp_pair_*).  y1 [K, F] = ReLU output of the 768 -> 256 layer, pos [2, K] int32
    ordered positions (second -1 = none), sign [K] (> 0: positive pair).  Returns (y2 [2, K], loss_sum [1], g_y2 [2, K], g_y1 [K, F], g_w [F], g_b [1])."""
    _need(y1, torch.float32, "y1", 2)
    _need(pos, torch.int32, "pos", 2)
    K, F = y1.shape
    l, dev, st, sd = L.lib(), y1.device, _stream(y1), seed & (2 ** 64 - 1)
    y2, g_y2, prob = (torch.empty(2, K, dtype=torch.float32, device=dev) for _ in range(3))
    loss, g_w, g_b, g_y1 = torch.zeros(1, device=dev), torch.empty(F, device=dev), torch.empty(1, device=dev), torch.empty_like(y1)
    lws = _ws(l.gmp_loss_workspace_bytes(2 * K), dev)
    L.check(l.gmp_lp_pair_rowdot_fwd(_ptr(y1), _ptr(w), _ptr(bias), _ptr(pos), _ptr(y2), K, F, float(p), sd, stream_id, st), "gmp_lp_pair_rowdot_fwd")
    L.check(l.gmp_lp_pair_sigmoid_bce_fwd_bwd(_ptr(y2), _ptr(sign), _ptr(pos), K, _ptr(g_scale), _ptr(loss), _ptr(prob), _ptr(g_y2), _ptr(lws), lws.numel(), st),
            "gmp_lp_pair_sigmoid_bce_fwd_bwd")
    L.check(l.gmp_lp_pair_outer_bwd(_ptr(g_y2), _ptr(w), _ptr(y1), _ptr(pos), _ptr(g_y1), K, F, float(p), sd, stream_id, st), "gmp_lp_pair_outer_bwd")
    ws = _ws(l.gmp_lp_pair_colsum_workspace_bytes(K, F), dev).clone()
    L.check(l.gmp_lp_pair_weighted_colsum(_ptr(g_y2), _ptr(y1), _ptr(pos), _ptr(g_w), _ptr(g_b), K, F, float(p), sd, stream_id, _ptr(ws), ws.numel(), st),
            "gmp_lp_pair_weighted_colsum")
    return y2, loss, g_y2, g_y1, g_w, g_b
