"""Oracle restatement of src/models/{gnn,heads,pretrain_model,finetune_model}.py.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Module trees and attribute names
are kept identical to the reference so ``state_dict()`` keys line up with the
product modules and with reference checkpoints.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor

from . import gates, graph_ops as G

# constants: src/models/gnn.py:6-8, heads.py:10-13, pretrain_model.py:18-20,
# finetune_model.py:14-17, src/data/data_setup.py:26,31-59, graph_properties dim 12
HIDDEN = 256
NUM_LAYERS = 5
DROPOUT = 0.2
PROJ_DIM = 128
DOMAIN_CLS_DROPOUT = 0.5
DOMAIN_CLS_HIDDEN = 128
GRAPH_PROP_HIDDEN = 512
GRAPH_PROP_DIM = 12
FINETUNE_HIDDEN = 128
MASK_TOKEN_STD = 0.1
NFM_RATE = 0.15
NFM_MIN_NODES = 3
PRETRAIN_DOMAINS = ["MUTAG", "PROTEINS", "NCI1", "ENZYMES"]
DOMAIN_DIMS = {"MUTAG": 7, "PROTEINS": 4, "NCI1": 37, "ENZYMES": 21, "PTC_MR": 18,
               "Cora_NC": 1433, "CiteSeer_NC": 3703, "Cora_LP": 1433, "CiteSeer_LP": 3703}
NUM_CLASSES = {"ENZYMES": 6, "PTC_MR": 2, "Cora_NC": 7, "CiteSeer_NC": 6, "Cora_LP": 2, "CiteSeer_LP": 2}
TASK_TYPES = {"ENZYMES": "graph_classification", "PTC_MR": "graph_classification",
              "Cora_NC": "node_classification", "CiteSeer_NC": "node_classification",
              "Cora_LP": "link_prediction", "CiteSeer_LP": "link_prediction"}


class InputEncoder(nn.Module):
    """gnn.py:11-23 -- dropout(relu(BN(Linear(x))))."""

    def __init__(self, dim_in: int) -> None:
        super().__init__()
        self.linear = nn.Linear(dim_in, HIDDEN)
        self.batch_norm = nn.BatchNorm1d(HIDDEN)
        self.dropout = nn.Dropout(DROPOUT)

    def forward(self, x: Tensor) -> Tensor:
        return self.dropout(gates.relu(self.batch_norm(self.linear(x))))


class GINConv(nn.Module):
    """PyG GINConv(nn, train_eps=True): nn((1+eps) x_i + sum_j x_j); eps is a
    Parameter of shape [1] initialised to 0 (gnn.py:29-37)."""

    def __init__(self, mlp: nn.Module) -> None:
        super().__init__()
        self.nn = mlp
        self.eps = nn.Parameter(torch.zeros(1))

    def forward(self, x: Tensor, edge_index: Tensor) -> Tensor:
        return self.nn(G.gin_aggregate(x, edge_index, self.eps))


class GINLayer(nn.Module):
    """gnn.py:26-43."""

    def __init__(self) -> None:
        super().__init__()
        self.gin_conv = GINConv(nn.Sequential(
            nn.Linear(HIDDEN, 2 * HIDDEN), nn.BatchNorm1d(2 * HIDDEN), gates.ReLU(),
            nn.Linear(2 * HIDDEN, HIDDEN)))
        self.batch_norm = nn.BatchNorm1d(HIDDEN)
        self.dropout_p = DROPOUT        # plain attribute so parity tests can switch dropout off under train-mode BN

    def forward(self, h: Tensor, edge_index: Tensor) -> Tensor:
        u = self.gin_conv(h, edge_index) + h
        return F.dropout(gates.relu(self.batch_norm(u)), p=self.dropout_p, training=self.training)


class GINBackbone(nn.Module):
    """gnn.py:46-54."""

    def __init__(self) -> None:
        super().__init__()
        self.layers = nn.ModuleList([GINLayer() for _ in range(NUM_LAYERS)])

    def forward(self, h: Tensor, edge_index: Tensor) -> Tensor:
        for layer in self.layers:
            h = layer(h, edge_index)
        return h


class MLPHead(nn.Module):
    """heads.py:35-50 -- Linear (ReLU Dropout Linear)*; Sequential indices 0,3,6.."""

    def __init__(self, dims: Sequence[int], dropout_rates: Optional[Sequence[float]] = None) -> None:
        super().__init__()
        mods: List[nn.Module] = []
        last = len(dims) - 2
        for i in range(len(dims) - 1):
            mods.append(nn.Linear(dims[i], dims[i + 1]))
            if i < last:
                mods += [gates.ReLU(), nn.Dropout(DROPOUT if dropout_rates is None else dropout_rates[i])]
        self.mlp = nn.Sequential(*mods)

    def forward(self, x: Tensor) -> Tensor:
        return self.mlp(x)


class MLPLinkPredictor(nn.Module):
    """heads.py:53-67 -- sigmoid(MLP([hs+hd | hs*hd | |hs-hd|]))."""

    def __init__(self) -> None:
        super().__init__()
        self.predictor = MLPHead([3 * HIDDEN, HIDDEN, 1])

    def forward(self, h: Tensor, edge_index: Tensor) -> Tensor:
        hs, hd = h[edge_index[0]], h[edge_index[1]]
        feats = torch.cat([hs + hd, hs * hd, gates.abs(hs - hd)], dim=1)
        return torch.sigmoid(self.predictor(feats).squeeze(-1))


class _GradReverse(torch.autograd.Function):
    """heads.py:16-24."""

    @staticmethod
    def forward(ctx, x, lam):
        ctx.lam = lam
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return -g * ctx.lam, None


class GradientReversalLayer(nn.Module):
    def forward(self, x: Tensor, lam: float) -> Tensor:
        return _GradReverse.apply(x, lam)


class DomainClassifierHead(nn.Module):
    """heads.py:70-82."""

    def __init__(self) -> None:
        super().__init__()
        self.grl = GradientReversalLayer()
        self.classifier = MLPHead([HIDDEN, DOMAIN_CLS_HIDDEN, len(PRETRAIN_DOMAINS)],
                                  dropout_rates=[DOMAIN_CLS_DROPOUT])

    def forward(self, x: Tensor, lam: float) -> Tensor:
        return self.classifier(self.grl(x, lam))


def _per_domain(domains, dims):
    return nn.ModuleDict({d: MLPHead(dims) for d in domains})


class PretrainableGNN(nn.Module):
    """pretrain_model.py:23-99."""

    def __init__(self, device, domain_names: List[str], task_names: List[str]) -> None:
        super().__init__()
        self.device = device
        self.input_encoders = nn.ModuleDict({d: InputEncoder(DOMAIN_DIMS[d]) for d in domain_names})
        self.mask_token = nn.Parameter(torch.zeros(HIDDEN))
        nn.init.normal_(self.mask_token, std=MASK_TOKEN_STD)
        self.gnn_backbone = GINBackbone()
        self.heads = nn.ModuleDict()
        for t in task_names:
            if t == "node_feat_mask":
                self.heads[t] = _per_domain(domain_names, [HIDDEN, HIDDEN, HIDDEN])
            elif t == "link_pred":
                self.heads[t] = MLPLinkPredictor()
            elif t == "node_contrast":
                self.heads[t] = _per_domain(domain_names, [HIDDEN, HIDDEN, PROJ_DIM])
            elif t == "graph_contrast":
                self.heads[t] = _per_domain(domain_names, [2 * HIDDEN, HIDDEN, PROJ_DIM])
            elif t == "graph_prop":
                self.heads[t] = _per_domain(domain_names, [HIDDEN, GRAPH_PROP_HIDDEN, GRAPH_PROP_DIM])
            elif t == "domain_adv":
                self.heads[t] = DomainClassifierHead()
        self.to(device)

    @staticmethod
    def draw_mask_indices(ptr: Tensor, generator: torch.Generator) -> Tensor:
        """RNG half of apply_node_masking (pretrain_model.py:71-80): per graph with
        n >= 3, randperm(n)[:max(1,int(.15 n))] + ptr[g], concatenated in graph order."""
        out = []
        for g in range(ptr.numel() - 1):
            s, e = int(ptr[g]), int(ptr[g + 1])
            n = e - s
            if n >= NFM_MIN_NODES:
                k = max(1, int(n * NFM_RATE))
                out.append(torch.randperm(n, generator=generator)[:k] + s)
        return torch.cat(out) if out else torch.empty(0, dtype=torch.long)

    def mask_with_indices(self, batch, domain: str, idx: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
        """Deterministic half of apply_node_masking (pretrain_model.py:68-69,82-88).
        The encoder runs under no_grad but in the module's current mode (dropout
        active and BN running stats updated when training)."""
        with torch.no_grad():
            h0 = self.input_encoders[domain](batch.x)
        if idx.numel() == 0:
            return h0, idx, torch.empty(0, h0.size(1))
        masked = h0.clone()
        masked[idx] = self.mask_token.expand(idx.numel(), -1)
        return masked, idx, h0[idx].detach()

    def apply_node_masking(self, batch, domain: str, generator: torch.Generator):
        # NB the reference runs the encoder before the draws; the draws do not
        # depend on it, so the order is immaterial for the generator stream --
        # except for dropout, which uses the *global* torch RNG, not `generator`.
        with torch.no_grad():
            h0 = self.input_encoders[domain](batch.x)
        idx = self.draw_mask_indices(batch.ptr, generator)
        if idx.numel() == 0:
            return h0, idx, torch.empty(0, h0.size(1))
        masked = h0.clone()
        masked[idx] = self.mask_token.expand(idx.numel(), -1)
        return masked, idx, h0[idx].detach()

    def forward(self, batch, domain: str) -> Tensor:
        return self.gnn_backbone(self.input_encoders[domain](batch.x), batch.edge_index)

    def forward_with_h0(self, h0: Tensor, edge_index: Tensor) -> Tensor:
        return self.gnn_backbone(h0, edge_index)

    def get_head(self, task: str, domain: Optional[str] = None) -> nn.Module:
        head = self.heads[task]
        return head if domain is None else head[domain]


class FinetuneGNN(nn.Module):
    """finetune_model.py:20-80 (freezing rules and param groups included)."""

    LR_BACKBONE = 1e-4
    LR_FINETUNE = 1e-3

    def __init__(self, device, domain_name: str, finetune_strategy: str) -> None:
        super().__init__()
        self.device = device
        self.domain_name = domain_name
        self.input_encoder = InputEncoder(DOMAIN_DIMS[domain_name])
        self.gnn_backbone = GINBackbone()
        kind = TASK_TYPES[domain_name]
        if kind == "graph_classification":
            self.classification_head = MLPHead([HIDDEN, FINETUNE_HIDDEN, NUM_CLASSES[domain_name]])
        elif kind == "node_classification":
            self.classification_head = MLPHead([HIDDEN, NUM_CLASSES[domain_name]])
        else:
            self.classification_head = MLPLinkPredictor()
        self.param_groups: List[Dict] = []
        if domain_name == "ENZYMES":
            for p in self.input_encoder.parameters():
                p.requires_grad = False
        else:
            self.param_groups.append({"params": self.input_encoder.parameters(),
                                      "lr": self.LR_FINETUNE, "name": "encoder"})
        if finetune_strategy == "linear_probe":
            for p in self.gnn_backbone.parameters():
                p.requires_grad = False
        else:
            self.param_groups.append({"params": self.gnn_backbone.parameters(),
                                      "lr": self.LR_BACKBONE, "name": "backbone"})
        self.param_groups.append({"params": self.classification_head.parameters(),
                                  "lr": self.LR_FINETUNE, "name": "head"})
        self.to(device)

    def forward(self, batch, edge_index: Optional[Tensor] = None,
                message_passing_edges: Optional[Tensor] = None) -> Tensor:
        h0 = self.input_encoder(batch.x)
        mp = batch.edge_index if message_passing_edges is None else message_passing_edges
        h = self.gnn_backbone(h0, mp)
        kind = TASK_TYPES[self.domain_name]
        if kind == "graph_classification":
            return self.classification_head(G.global_mean_pool(h, batch.batch))
        if kind == "node_classification":
            return self.classification_head(h)
        return self.classification_head(h, edge_index)


def load_pretrained_state(model: FinetuneGNN, pretrained_state: Dict[str, Tensor]) -> None:
    """finetune_model.py:128-146 without the wandb download: copy gnn_backbone.*,
    and input_encoders.ENZYMES.* -> input_encoder.* when the domain is ENZYMES."""
    sd = model.state_dict()
    for k, v in pretrained_state.items():
        if k.startswith("gnn_backbone.") and k in sd:
            sd[k] = v
    if model.domain_name == "ENZYMES":
        pre = "input_encoders.ENZYMES."
        for k, v in pretrained_state.items():
            if k.startswith(pre):
                k2 = "input_encoder." + k[len(pre):]
                if k2 in sd:
                    sd[k2] = v
    model.load_state_dict(sd, strict=False)
