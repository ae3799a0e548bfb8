"""FinetuneGNN on libgnnmp; mirrors src/models/finetune_model.py:20-80,128-152.
The wandb artifact download of the reference (finetune_model.py:88-118) is a network fetch and is
not reproduced: a missing checkpoint raises FileNotFoundError."""
from __future__ import annotations

from pathlib import Path
from typing import Dict, Optional

import torch
import torch.nn as nn
from torch import Tensor

from .. import operators as O
from ..constants import DOMAIN_DIMENSIONS, NUM_CLASSES, TASK_TYPES
from .gnn import GINBackbone, GNN_HIDDEN_DIM, InputEncoder
from .heads import MLPHead, MLPLinkPredictor

FINETUNE_HIDDEN_DIM = 128
LR_BACKBONE = 1e-4
LR_FINETUNE = 1e-3
PRETRAIN_OUTPUT_DIR = Path(__file__).resolve().parents[2] / "outputs" / "pretrain"


class FinetuneGNN(nn.Module):
    def __init__(self, device: torch.device, domain_name: str, finetune_strategy: str) -> None:
        super().__init__()
        self.device, self.domain_name = device, domain_name
        self.input_encoder = InputEncoder(DOMAIN_DIMENSIONS[domain_name])
        self.gnn_backbone = GINBackbone()
        kind = TASK_TYPES[domain_name]
        if kind == "graph_classification":
            self.classification_head = MLPHead([GNN_HIDDEN_DIM, FINETUNE_HIDDEN_DIM, NUM_CLASSES[domain_name]])
        elif kind == "node_classification":
            self.classification_head = MLPHead([GNN_HIDDEN_DIM, NUM_CLASSES[domain_name]])
        else:
            self.classification_head = MLPLinkPredictor()

        self.param_groups = []
        if domain_name == "ENZYMES":                       # encoder comes frozen from pre-training
            for p in self.input_encoder.parameters():
                p.requires_grad = False
        else:
            self.param_groups.append({"params": self.input_encoder.parameters(), "lr": LR_FINETUNE, "name": "encoder"})
        if finetune_strategy == "linear_probe":
            for p in self.gnn_backbone.parameters():
                p.requires_grad = False
        else:
            self.param_groups.append({"params": self.gnn_backbone.parameters(), "lr": LR_BACKBONE, "name": "backbone"})
        self.param_groups.append({"params": self.classification_head.parameters(), "lr": LR_FINETUNE, "name": "head"})
        self.to(self.device)

    def forward(self, batch, edge_index: Optional[Tensor] = None,
                message_passing_edges: Optional[Tensor] = None) -> Tensor:
        mp = batch.edge_index if message_passing_edges is None else message_passing_edges
        h = self.gnn_backbone(self.input_encoder(batch.x), mp)
        kind = TASK_TYPES[self.domain_name]
        if kind == "graph_classification":
            ptr32 = getattr(batch, "ptr32", None)
            return self.classification_head(O.global_mean_pool(h, batch.batch, ptr32=ptr32))
        if kind == "node_classification":
            return self.classification_head(h)
        return self.classification_head(h, edge_index)


def load_pretrained_state(model: FinetuneGNN, pretrained_state: Dict[str, Tensor]) -> None:
    """Key-prefix transfer of finetune_model.py:134-146."""
    sd = model.state_dict()
    for k, v in pretrained_state.items():
        if k.startswith("gnn_backbone.") and k in sd:
            sd[k] = v
    if model.domain_name == "ENZYMES":
        pre = "input_encoders.ENZYMES."
        for k, v in pretrained_state.items():
            if k.startswith(pre) and "input_encoder." + k[len(pre):] in sd:
                sd["input_encoder." + k[len(pre):]] = v
    model.load_state_dict(sd, strict=False)


def load_pretrained_weights(model: FinetuneGNN, pretrained_scheme: str, seed: int,
                            directory: Path = PRETRAIN_OUTPUT_DIR) -> None:
    path = Path(directory) / f"model_{pretrained_scheme}_{seed}.pt"
    if not path.exists():
        raise FileNotFoundError(f"{path}: run run_pretrain.py --exp_name {pretrained_scheme} --seed {seed} first "
                                "(the reference would fetch it from wandb; there is no network here)")
    ckpt = torch.load(path, map_location=model.device, weights_only=True)
    load_pretrained_state(model, ckpt["model_state_dict"])


def create_finetune_model(device: torch.device, cfg) -> FinetuneGNN:
    model = FinetuneGNN(device, cfg.domain_name, cfg.finetune_strategy)
    if cfg.pretrained_scheme != "b1":
        load_pretrained_weights(model, cfg.pretrained_scheme, cfg.seed)
    return model
