"""Task heads on libgnnmp; mirrors src/models/heads.py (MLPHead keys mlp.{0,3,...},
MLPLinkPredictor.predictor, DomainClassifierHead.classifier, gradient reversal)."""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
import torch.nn as nn
from torch import Tensor

from .. import operators as O
from ..constants import PRETRAIN_TUDATASETS
from .gnn import DROPOUT_RATE, GNN_HIDDEN_DIM, Linear

CONTRASTIVE_PROJ_DIM = 128
DOMAIN_CLASSIFIER_DROPOUT_RATE = 0.5
DOMAIN_CLASSIFIER_HIDDEN_DIM = 128
GRAPH_PROP_HIDDEN_DIM = 512


class GradientReversalLayer(nn.Module):
    def forward(self, x: Tensor, lambda_val: float) -> Tensor:
        return O.grad_reverse(x, lambda_val)


class MLPHead(nn.Module):
    """Linear (ReLU Dropout Linear)* -- the ReLU is fused into the GEMM epilogue, the dropout
    mask is counter-based and regenerated in the backward."""

    def __init__(self, dims: Sequence[int], dropout_rates: Optional[Sequence[float]] = None) -> None:
        super().__init__()
        mods: List[nn.Module] = []
        n = len(dims) - 1
        for i in range(n):
            mods.append(Linear(dims[i], dims[i + 1]))
            if i < n - 1:
                mods.append(nn.ReLU())
                mods.append(nn.Dropout(DROPOUT_RATE if dropout_rates is None else dropout_rates[i]))
        self.mlp = nn.Sequential(*mods)

    def forward(self, x: Tensor) -> Tensor:
        mods = list(self.mlp)
        i = 0
        while i < len(mods):
            lin = mods[i]
            fuse_relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
            x = lin(x, relu=fuse_relu)
            i += 2 if fuse_relu else 1
            if i < len(mods) and isinstance(mods[i], nn.Dropout):
                x = O.dropout(x, mods[i].p, self.training)
                i += 1
        return x


class MLPLinkPredictor(nn.Module):
    def __init__(self) -> None:
        super().__init__()
        self.predictor = MLPHead([3 * GNN_HIDDEN_DIM, GNN_HIDDEN_DIM, 1])

    def forward(self, h: Tensor, edge_index: Tensor) -> Tensor:
        feats = O.lp_edge_features(h, edge_index)          # [K, 768] = [hs+hd | hs*hd | |hs-hd|]
        return O.sigmoid(self.predictor(feats).squeeze(-1))


class DomainClassifierHead(nn.Module):
    def __init__(self) -> None:
        super().__init__()
        self.grl = GradientReversalLayer()
        self.classifier = MLPHead([GNN_HIDDEN_DIM, DOMAIN_CLASSIFIER_HIDDEN_DIM, len(PRETRAIN_TUDATASETS)],
                                  dropout_rates=[DOMAIN_CLASSIFIER_DROPOUT_RATE])

    def forward(self, x: Tensor, lambda_val: float) -> Tensor:
        return self.classifier(self.grl(x, lambda_val))
