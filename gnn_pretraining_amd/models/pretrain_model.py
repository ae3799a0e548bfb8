"""PretrainableGNN on libgnnmp; mirrors src/models/pretrain_model.py:23-99."""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn
from torch import Tensor

from .. import operators as O
from ..constants import DOMAIN_DIMENSIONS, GRAPH_PROPERTY_DIM
from .gnn import GINBackbone, GNN_HIDDEN_DIM, InputEncoder
from .heads import CONTRASTIVE_PROJ_DIM, GRAPH_PROP_HIDDEN_DIM, DomainClassifierHead, MLPHead, MLPLinkPredictor

MASK_TOKEN_INIT_STD = 0.1
NODE_FEATURE_MASKING_MASK_RATE = 0.15
NODE_FEATURE_MASKING_MIN_NUM_NODES = 3

_HEAD_DIMS = {
    "node_feat_mask": [GNN_HIDDEN_DIM, GNN_HIDDEN_DIM, GNN_HIDDEN_DIM],
    "node_contrast": [GNN_HIDDEN_DIM, GNN_HIDDEN_DIM, CONTRASTIVE_PROJ_DIM],
    "graph_contrast": [2 * GNN_HIDDEN_DIM, GNN_HIDDEN_DIM, CONTRASTIVE_PROJ_DIM],
    "graph_prop": [GNN_HIDDEN_DIM, GRAPH_PROP_HIDDEN_DIM, GRAPH_PROPERTY_DIM],
}


def draw_mask_indices(ptr_host: List[int], generator: torch.Generator) -> Tensor:
    """RNG half of apply_node_masking (pretrain_model.py:71-80), on the HOST offsets: per graph with
    n >= 3 nodes, randperm(n, generator)[:max(1, int(.15 n))] + offset.  Same draws, same order and
    therefore the same indices as the reference for an equal generator state -- without the two
    ``.item()`` device syncs per graph."""
    parts = []
    for s, e in zip(ptr_host[:-1], ptr_host[1:]):
        n = e - s
        if n >= NODE_FEATURE_MASKING_MIN_NUM_NODES:
            k = max(1, int(n * NODE_FEATURE_MASKING_MASK_RATE))
            parts.append(torch.randperm(n, generator=generator)[:k] + s)
    return torch.cat(parts) if parts else torch.empty(0, dtype=torch.long)


class PretrainableGNN(nn.Module):
    def __init__(self, device: torch.device, domain_names: List[str], task_names: List[str]) -> None:
        super().__init__()
        self.device = device
        self.input_encoders = nn.ModuleDict({d: InputEncoder(DOMAIN_DIMENSIONS[d]) for d in domain_names})
        self.mask_token = nn.Parameter(torch.zeros(GNN_HIDDEN_DIM))
        nn.init.normal_(self.mask_token, std=MASK_TOKEN_INIT_STD)
        self.gnn_backbone = GINBackbone()
        self.heads = nn.ModuleDict()
        for t in task_names:
            if t in _HEAD_DIMS:
                self.heads[t] = nn.ModuleDict({d: MLPHead(_HEAD_DIMS[t]) for d in domain_names})
            elif t == "link_pred":
                self.heads[t] = MLPLinkPredictor()
            elif t == "domain_adv":
                self.heads[t] = DomainClassifierHead()
        self.to(self.device)

    def mask_with_indices(self, batch, domain_name: str, mask_indices: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
        """Deterministic half of apply_node_masking: encoder under no_grad in the current mode (dropout
        live, BN running statistics updated), masked rows replaced by the mask token."""
        with torch.no_grad():
            original_h0 = self.input_encoders[domain_name](batch.x)
        idx = mask_indices.to(self.device)
        if idx.numel() == 0:
            return original_h0, idx, torch.empty(0, original_h0.size(1), device=self.device)
        target = O.take_rows(original_h0, idx)
        return O.mask_rows(original_h0, idx, self.mask_token), idx, target

    def apply_node_masking(self, batch, domain_name: str, generator: torch.Generator):
        idx = draw_mask_indices(batch.ptr_host, generator)
        return self.mask_with_indices(batch, domain_name, idx)

    def forward(self, batch, domain_name: str) -> Tensor:
        return self.gnn_backbone(self.input_encoders[domain_name](batch.x), batch.edge_index)

    def forward_with_h0(self, h_0: Tensor, edge_index: Tensor) -> Tensor:
        return self.gnn_backbone(h_0, edge_index)

    def get_head(self, task_name: str, domain_name: Optional[str] = None) -> nn.Module:
        head = self.heads[task_name]
        return head if domain_name is None else head[domain_name]
