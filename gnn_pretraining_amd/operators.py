"""Operator layer: the torch-geometric / torch operators the reference's hot path calls,
re-implemented as autograd Functions over libgnnmp (SURVEY.md section 8b "operator layer").

    gin_aggregate(x, edge_index, eps)      GINConv's propagate + (1+eps) x       gnn.py:29-41
    global_mean_pool / global_max_pool     PyG read-out pooling                  tasks.py:241-246,299
    linear(x, W, b)                        nn.Linear on the f32 MFMA             gnn.py:14,31,34 heads.py:42
    batch_norm_act(...)                    BatchNorm1d (+residual,+ReLU,+dropout) gnn.py:19-22,42-43
    take_rows(h, idx)                      h[idx]                                tasks.py:80
    lp_edge_features(h, edges)             [hs+hd | hs*hd | |hs-hd|]             heads.py:58-66
    nt_xent(z1, z2, T)                     SimCLR loss                           tasks.py:192-213
    relu_dropout(x, p, training)           the ReLU->Dropout pair of MLPHead     heads.py:43-45

Everything raises GnnmpError on CPU tensors -- there is no eager fallback.
"""
from __future__ import annotations

import weakref
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor

from . import ops
from ._lib import GnnmpError

# --------------------------------------------------------------------------- #
# CSR cache: the 5 GIN layers of a forward (and every task sharing a batch)
# reuse one edge_index tensor, so the int32 CSR is built once per tensor object.
# --------------------------------------------------------------------------- #
_CSR_CACHE: Dict[int, Tuple[weakref.ref, int, int, ops.CSR]] = {}
_CSR_CACHE_MAX = 256


def csr_of(edge_index: Tensor, num_nodes: int) -> ops.CSR:
    key = id(edge_index)
    hit = _CSR_CACHE.get(key)
    if hit is not None:
        ref, version, n, csr = hit
        if ref() is edge_index and version == edge_index._version and n == num_nodes:
            return csr
    if not edge_index.is_contiguous():
        edge_index = edge_index.contiguous()
    csr = ops.csr_build(edge_index, num_nodes)
    if len(_CSR_CACHE) >= _CSR_CACHE_MAX:
        for k in [k for k, v in _CSR_CACHE.items() if v[0]() is None] or list(_CSR_CACHE)[: _CSR_CACHE_MAX // 2]:
            _CSR_CACHE.pop(k, None)
    try:
        _CSR_CACHE[key] = (weakref.ref(edge_index), edge_index._version, num_nodes, csr)
    except TypeError:
        pass
    return csr


# --------------------------------------------------------------------------- #
# dropout seeds: (seed, stream_id) per dropout site; the mask is regenerated in
# the backward from the same pair, never stored.
# --------------------------------------------------------------------------- #
class _DropoutCounter:
    def __init__(self) -> None:
        self.count = 0

    def next(self) -> Tuple[int, int]:
        self.count += 1
        return (torch.initial_seed() + 0x9E3779B97F4A7C15 * (self.count >> 20)) & (2 ** 64 - 1), self.count & 0xFFFFF


DROPOUT = _DropoutCounter()


# --------------------------------------------------------------------------- #
class _GinAggregate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, eps, csr: ops.CSR):
        x = x.contiguous()
        ctx.csr = csr
        ctx.save_for_backward(x, eps)
        return ops.gin_aggregate_fwd(x, csr.rowptr, csr.col, eps)

    @staticmethod
    def backward(ctx, g):
        x, eps = ctx.saved_tensors
        csr = ctx.csr
        need_eps = ctx.needs_input_grad[1]
        gx, ge = ops.gin_aggregate_bwd(g.contiguous(), csr.rowptr_t, csr.col_t, eps, x if need_eps else None)
        return gx, ge, None


def gin_aggregate(x: Tensor, edge_index: Tensor, eps: Tensor) -> Tensor:
    """sum_{j->i} x_j + (1+eps) x_i ; edge_index[0]=source, edge_index[1]=target (int64 [2,E])."""
    return _GinAggregate.apply(x, eps, csr_of(edge_index, x.size(0)))


# --------------------------------------------------------------------------- #
def _ptr_from_batch(batch: Tensor, size: Optional[int]) -> Tensor:
    """Segment offsets from a sorted graph-id vector (PyG passes only `batch`).  Costs one host sync for
    B = batch.max()+1 exactly like PyG's scatter; callers holding a Batch pass ptr32 instead."""
    B = int(batch.max().item()) + 1 if size is None else size
    counts = torch.bincount(batch, minlength=B)
    ptr = torch.zeros(B + 1, dtype=torch.int32, device=batch.device)
    ptr[1:] = counts.cumsum(0)
    return ptr


class _MeanPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, batch, ptr):
        ctx.save_for_backward(batch, ptr)
        return ops.segment_sum(x.contiguous(), ptr, None, mean=True)

    @staticmethod
    def backward(ctx, g):
        batch, ptr = ctx.saved_tensors
        return ops.row_gather(g.contiguous(), batch, ptr), None, None


class _MaxPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ptr):
        x = x.contiguous()
        out = ops.segment_max_fwd(x, ptr)
        ctx.save_for_backward(x, out, ptr)
        return out

    @staticmethod
    def backward(ctx, g):
        x, out, ptr = ctx.saved_tensors
        return ops.segment_max_bwd(g.contiguous(), x, out, ptr), None


def global_mean_pool(x: Tensor, batch: Tensor, size: Optional[int] = None, ptr32: Optional[Tensor] = None) -> Tensor:
    ptr = ptr32 if ptr32 is not None else _ptr_from_batch(batch, size)
    return _MeanPool.apply(x, batch, ptr)


def global_max_pool(x: Tensor, batch: Tensor, size: Optional[int] = None, ptr32: Optional[Tensor] = None) -> Tensor:
    ptr = ptr32 if ptr32 is not None else _ptr_from_batch(batch, size)
    return _MaxPool.apply(x, ptr)


# --------------------------------------------------------------------------- #
class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, relu: bool):
        x = x.contiguous()
        y = ops.gemm(ops.NT, x, weight, bias, relu=relu)
        ctx.relu = relu
        ctx.save_for_backward(x, weight, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, g):
        x, weight, y = ctx.saved_tensors
        g = g.contiguous()
        if ctx.relu:
            g = ops.relu_dropout_bwd(g, y, 0.0, 0, 0) if g.numel() % 4 == 0 else g * (y > 0)
        gx = ops.gemm(ops.NN, g, weight) if ctx.needs_input_grad[0] else None
        gw = ops.gemm(ops.TN, g, x) if ctx.needs_input_grad[1] else None
        gb = ops.colsum(g) if (ctx.needs_input_grad[2]) else None
        return gx, gw, gb, None


def linear(x: Tensor, weight: Tensor, bias: Optional[Tensor], relu: bool = False) -> Tensor:
    """x W^T + b (torch.nn.Linear layout: weight [out,in]); optional fused ReLU."""
    return _Linear.apply(x, weight, bias, relu)


# --------------------------------------------------------------------------- #
class _BatchNormAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, residual, gamma, beta, running_mean, running_var, seg_ptr, max_seg_rows, training, relu,
                dropout_p, seed, stream_id):
        x = x.contiguous()
        residual = None if residual is None else residual.contiguous()
        cfg = ops.make_bn_config(training, relu, dropout_p if training else 0.0, seed, stream_id)
        y, sm, sr = ops.bn_fwd(x, residual, seg_ptr, max_seg_rows, gamma, beta, running_mean, running_var, cfg)
        ctx.cfg, ctx.max_seg_rows, ctx.has_res = cfg, max_seg_rows, residual is not None
        ctx.save_for_backward(x, residual, gamma, beta, running_mean, running_var, sm, sr, seg_ptr)
        ctx.mark_non_differentiable()
        return y

    @staticmethod
    def backward(ctx, g):
        x, residual, gamma, beta, rm, rv, sm, sr, seg_ptr = ctx.saved_tensors
        gu, gg, gb = ops.bn_bwd(g.contiguous(), x, residual, seg_ptr, ctx.max_seg_rows, gamma, beta, rm, rv, sm, sr,
                                ctx.cfg)
        return (gu, gu if ctx.has_res else None, gg[0], gb[0]) + (None,) * 9


_one_segment: dict = {}


def _whole_input_segment(rows: int, device) -> Tensor:
    """[0, rows] as a device int32 tensor, cached: built per call it was a blocking host-to-device copy in front of every
    BatchNorm of the module path (11 per fine-tune step)."""
    key = (rows, device)
    t = _one_segment.get(key)
    if t is None:
        if len(_one_segment) > 256:
            _one_segment.clear()
        t = _one_segment[key] = torch.tensor([0, rows], dtype=torch.int32, device=device)
    return t


def batch_norm_act(x: Tensor, bn: torch.nn.BatchNorm1d, *, residual: Optional[Tensor] = None, relu: bool = True,
                   dropout_p: float = 0.0, training: Optional[bool] = None, seg_ptr: Optional[Tensor] = None,
                   max_seg_rows: Optional[int] = None) -> Tensor:
    """dropout(relu(BatchNorm1d(x + residual))) in one launch, statistics per segment
    (default: one segment = the whole input, i.e. exactly nn.BatchNorm1d)."""
    training = bn.training if training is None else training
    if seg_ptr is None:
        seg_ptr = _whole_input_segment(x.size(0), x.device)
        max_seg_rows = x.size(0)
    if training and x.size(0) <= 1:
        raise ValueError("Expected more than 1 value per channel when training")   # torch's own check
    seed, sid = DROPOUT.next() if (training and dropout_p > 0) else (0, 0)
    y = _BatchNormAct.apply(x, residual, bn.weight, bn.bias, bn.running_mean, bn.running_var, seg_ptr, max_seg_rows,
                            training, relu, dropout_p, seed, sid)
    if training and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(seg_ptr.numel() - 1)
    return y


# --------------------------------------------------------------------------- #
class _TakeRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, idx):
        h = h.contiguous()
        ctx.n = h.size(0)
        ctx.save_for_backward(idx)
        return ops.row_gather(h, idx)

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        m, n = idx.numel(), ctx.n
        if m == 0:
            return torch.zeros(n, g.size(1), dtype=g.dtype, device=g.device), None
        pairs = torch.stack([torch.arange(m, device=idx.device), idx])
        csr = ops.csr_build(pairs, max(m, n))
        return ops.segment_sum(g.contiguous(), csr.rowptr[: n + 1].contiguous(), csr.col), None


def take_rows(h: Tensor, idx: Tensor) -> Tensor:
    """h[idx] for an int64 index vector (repeats allowed; the backward accumulates deterministically)."""
    return _TakeRows.apply(h, idx.contiguous())


def take_rows_mask(h: Tensor, mask: Tensor) -> Tensor:
    """h[mask] for a boolean row mask (the common-node selection of tasks.py:153-164)."""
    return take_rows(h, mask.nonzero().squeeze(1))


# --------------------------------------------------------------------------- #
class _LpEdgeFeatures(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, edges):
        h = h.contiguous()
        ctx.save_for_backward(h, edges)
        return ops.lp_edge_features_fwd(h, edges)

    @staticmethod
    def backward(ctx, g):
        h, edges = ctx.saved_tensors
        ghs, ghd = ops.lp_edge_features_bwd(g.contiguous(), h, edges)
        csr = ops.csr_build(edges, h.size(0))
        gh = ops.segment_sum(ghs, csr.rowptr_t, csr.perm_t)
        gh = ops.segment_sum(ghd, csr.rowptr, csr.perm, out=gh, accumulate=True)
        return gh, None


def lp_edge_features(h: Tensor, edges: Tensor) -> Tensor:
    return _LpEdgeFeatures.apply(h, edges.contiguous())


# --------------------------------------------------------------------------- #
class _NtXent(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z1, z2, temperature: float):
        z1, z2 = z1.contiguous(), z2.contiguous()
        loss, ws = ops.nt_xent_fwd(z1, z2, temperature)
        ctx.t = temperature
        ctx.save_for_backward(z1, z2, ws)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        z1, z2, ws = ctx.saved_tensors
        g1, g2 = ops.nt_xent_bwd(z1, z2, ctx.t, g.reshape(1).contiguous().float(), ws)
        return g1, g2, None


def nt_xent(z1: Tensor, z2: Tensor, temperature: float) -> Tuple[Tensor, int]:
    """(sum of the 2n cross-entropies, 2n)."""
    return _NtXent.apply(z1, z2, float(temperature)), 2 * z1.size(0)


# --------------------------------------------------------------------------- #
class _Dropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed, sid):
        ctx.p, ctx.seed, ctx.sid = p, seed, sid
        return ops.dropout_fwd(x.contiguous(), p, seed, sid)

    @staticmethod
    def backward(ctx, g):
        return ops.dropout_fwd(g.contiguous(), ctx.p, ctx.seed, ctx.sid), None, None, None


def dropout(x: Tensor, p: float, training: bool) -> Tensor:
    if not training or p <= 0.0:
        return x
    if x.numel() % 4:
        raise GnnmpError("dropout: numel must be a multiple of 4")
    seed, sid = DROPOUT.next()
    return _Dropout.apply(x, p, seed, sid)


# --------------------------------------------------------------------------- #
# losses (sum reductions, scalar outputs)
# --------------------------------------------------------------------------- #
class _MseSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        ctx.save_for_backward(a, b)
        return ops.mse_sum_fwd(a, b).reshape(())

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        ga = ops.mse_sum_bwd(a, b, g.reshape(1).contiguous())
        return ga, (-ga if ctx.needs_input_grad[1] else None)


def mse_loss_sum(pred: Tensor, target: Tensor) -> Tensor:
    return _MseSum.apply(pred, target)


class _Sigmoid(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y = ops.sigmoid_fwd(x.contiguous())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        return ops.sigmoid_bwd(g.contiguous(), y)


def sigmoid(x: Tensor) -> Tensor:
    return _Sigmoid.apply(x)


class _BceSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p, labels):
        p, labels = p.contiguous(), labels.contiguous()
        ctx.save_for_backward(p, labels)
        return ops.bce_sum_fwd(p, labels).reshape(())

    @staticmethod
    def backward(ctx, g):
        p, labels = ctx.saved_tensors
        return ops.bce_sum_bwd(p, labels, g.reshape(1).contiguous()), None


def binary_cross_entropy_sum(probs: Tensor, labels: Tensor) -> Tensor:
    return _BceSum.apply(probs, labels)


class _CeSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target):
        logits = logits.contiguous()
        ctx.save_for_backward(logits, target)
        return ops.cross_entropy_sum_fwd(logits, target).reshape(())

    @staticmethod
    def backward(ctx, g):
        logits, target = ctx.saved_tensors
        return ops.cross_entropy_sum_bwd(logits, target, g.reshape(1).contiguous()), None


def cross_entropy_sum(logits: Tensor, target: Tensor) -> Tensor:
    return _CeSum.apply(logits, target.contiguous())


class _MaskRows(torch.autograd.Function):
    """out = h0 with rows idx replaced by `token` (pretrain_model.py:82-85)."""

    @staticmethod
    def forward(ctx, h0, idx, token):
        out = h0.detach().clone()
        ops.row_fill_(out, idx, token.contiguous(), broadcast=True)
        ctx.save_for_backward(idx)
        ctx.h0_grad = h0.requires_grad
        return out

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        g = g.contiguous()
        g_token = ops.colsum(ops.row_gather(g, idx))
        g_h0 = None
        if ctx.h0_grad:
            g_h0 = g.clone()
            ops.row_fill_(g_h0, idx, torch.zeros(g.size(1), device=g.device), broadcast=True)
        return g_h0, None, g_token


def mask_rows(h0: Tensor, idx: Tensor, token: Tensor) -> Tensor:
    return _MaskRows.apply(h0, idx.contiguous(), token)


class _GradReverse(torch.autograd.Function):
    """heads.py:16-24 -- identity forward, -lambda * g backward (no kernel: a view and a scale)."""

    @staticmethod
    def forward(ctx, x, lam):
        ctx.lam = lam
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.neg() * ctx.lam, None


def grad_reverse(x: Tensor, lam: float) -> Tensor:
    return _GradReverse.apply(x, lam)
