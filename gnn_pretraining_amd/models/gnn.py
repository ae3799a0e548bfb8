"""GIN backbone on libgnnmp.  Mirrors src/models/gnn.py (InputEncoder, GINLayer, GINBackbone,
forward signatures, state_dict keys gin_conv.nn.{0,1,3}.*, gin_conv.eps [1], batch_norm.*) but
each layer is five fused launches instead of ~11 eager ops:

    aggregate (CSR gather+sum+(1+eps)x) -> Linear 256->512 (f32 MFMA) -> BN+ReLU ->
    Linear 512->256 (f32 MFMA) -> residual+BN+ReLU+dropout

``seg_ptr`` (optional everywhere) marks independent forward() calls stacked into one launch:
BatchNorm statistics are taken per segment, so a stacked call equals the separate calls.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn
from torch import Tensor

from .. import operators as O

DROPOUT_RATE = 0.2
GNN_HIDDEN_DIM = 256
GNN_NUM_LAYERS = 5


class Linear(nn.Linear):
    """nn.Linear whose forward runs the f32-MFMA GEMM (parameters/keys unchanged)."""

    def forward(self, x: Tensor, relu: bool = False) -> Tensor:
        return O.linear(x, self.weight, self.bias, relu)


class BatchNorm1d(nn.BatchNorm1d):
    def forward(self, x: Tensor) -> Tensor:
        return O.batch_norm_act(x, self, relu=False)


class InputEncoder(nn.Module):
    def __init__(self, dim_in: int) -> None:
        super().__init__()
        self.linear = Linear(dim_in, GNN_HIDDEN_DIM)
        self.batch_norm = BatchNorm1d(GNN_HIDDEN_DIM)
        self.dropout = nn.Dropout(DROPOUT_RATE)      # parameter-free; kept for module-tree parity

    def forward(self, x: Tensor, seg_ptr: Optional[Tensor] = None, max_seg_rows: Optional[int] = None) -> Tensor:
        z = self.linear(x)
        return O.batch_norm_act(z, self.batch_norm, relu=True, dropout_p=self.dropout.p, training=self.training,
                                seg_ptr=seg_ptr, max_seg_rows=max_seg_rows)


class GINConv(nn.Module):
    """PyG GINConv(nn, train_eps=True): nn((1 + eps) x_i + sum_{j->i} x_j)."""

    def __init__(self, mlp: nn.Module, eps: float = 0.0, train_eps: bool = False) -> None:
        super().__init__()
        self.nn = mlp
        if train_eps:
            self.eps = nn.Parameter(torch.full((1,), float(eps)))
        else:
            self.register_buffer("eps", torch.full((1,), float(eps)))

    def _is_gin_mlp(self) -> bool:
        m = self.nn
        return (isinstance(m, nn.Sequential) and len(m) == 4 and isinstance(m[0], Linear)
                and isinstance(m[1], nn.BatchNorm1d) and isinstance(m[2], nn.ReLU) and isinstance(m[3], Linear))

    def forward(self, x: Tensor, edge_index: Tensor, seg_ptr: Optional[Tensor] = None,
                max_seg_rows: Optional[int] = None) -> Tensor:
        a = O.gin_aggregate(x, edge_index, self.eps)
        if not self._is_gin_mlp():
            return self.nn(a)
        m = self.nn
        r1 = O.batch_norm_act(m[0](a), m[1], relu=True, seg_ptr=seg_ptr, max_seg_rows=max_seg_rows)
        return m[3](r1)


class GINLayer(nn.Module):
    def __init__(self) -> None:
        super().__init__()
        self.gin_conv = GINConv(nn.Sequential(Linear(GNN_HIDDEN_DIM, 2 * GNN_HIDDEN_DIM),
                                              BatchNorm1d(2 * GNN_HIDDEN_DIM), nn.ReLU(),
                                              Linear(2 * GNN_HIDDEN_DIM, GNN_HIDDEN_DIM)), train_eps=True)
        self.batch_norm = BatchNorm1d(GNN_HIDDEN_DIM)
        self.dropout_p = DROPOUT_RATE

    def forward(self, h: Tensor, edge_index: Tensor, seg_ptr: Optional[Tensor] = None,
                max_seg_rows: Optional[int] = None) -> Tensor:
        z = self.gin_conv(h, edge_index, seg_ptr, max_seg_rows)
        # (z + h) -> BatchNorm -> ReLU -> dropout, one kernel
        return O.batch_norm_act(z, self.batch_norm, residual=h, relu=True, dropout_p=self.dropout_p,
                                training=self.training, seg_ptr=seg_ptr, max_seg_rows=max_seg_rows)


class GINBackbone(nn.Module):
    def __init__(self) -> None:
        super().__init__()
        self.layers = nn.ModuleList([GINLayer() for _ in range(GNN_NUM_LAYERS)])

    def forward(self, h: Tensor, edge_index: Tensor, seg_ptr: Optional[Tensor] = None,
                max_seg_rows: Optional[int] = None) -> Tensor:
        for layer in self.layers:
            h = layer(h, edge_index, seg_ptr, max_seg_rows)
        return h
