"""Fine-tuning loaders with the behaviour of src/data/finetune_data_loaders.py: graph batches for the TUDatasets,
(graph, node ids, labels) for Planetoid node classification, (graph, edge columns, labels) for link prediction.  None
of the reference's loaders shuffles (no `shuffle=True` is passed at :72,89,100), so batches are consecutive slices."""
from __future__ import annotations

from pathlib import Path
from typing import Dict, Iterator, Optional, Tuple

import torch
from torch import Tensor

from ..constants import TASK_TYPES
from ..graph import Batch
from .data_setup import processed_dir
from .pretrain_data_loaders import GraphDataset, SequentialGraphLoader
from .store import GraphStore, load_splits


class NodeDataset:
    """finetune_data_loaders.py:14-25"""

    def __init__(self, data: Batch, indices) -> None:
        self.data = data
        self.indices = torch.as_tensor(indices, dtype=torch.long).reshape(-1)

    def __len__(self) -> int:
        return int(self.indices.numel())

    def __getitem__(self, idx: int) -> tuple:
        node = int(self.indices[idx])
        return self.data, node, self.data.y[node]


class LinkPredictionDataset:
    """finetune_data_loaders.py:28-52: train = positive edges only (negatives are mined per batch); val / test =
    positives followed by the stored negatives."""

    def __init__(self, data: Batch, split_edges: Dict[str, Tensor], split: str) -> None:
        self.data, self.split = data, split
        self.train_edges = split_edges["train_pos"]
        if split == "train":
            self.edges = split_edges["train_pos"]
            self.labels = torch.ones(self.edges.size(1))
        else:
            self.pos_edges, self.neg_edges = split_edges[f"{split}_pos"], split_edges[f"{split}_neg"]
            self.edges = torch.cat([self.pos_edges, self.neg_edges], dim=1)
            self.labels = torch.cat([torch.ones(self.pos_edges.size(1)), torch.zeros(self.neg_edges.size(1))])

    def __len__(self) -> int:
        return int(self.edges.size(1))

    def __getitem__(self, idx: int) -> tuple:
        return self.data, self.edges[:, idx], self.labels[idx]


class NodeLoader:
    def __init__(self, dataset: NodeDataset, batch_size: int) -> None:
        self.dataset, self.batch_size = dataset, batch_size

    def __iter__(self) -> Iterator[Tuple[Batch, Tensor, Tensor]]:
        ds = self.dataset
        for s in range(0, len(ds), self.batch_size):
            idx = ds.indices[s:s + self.batch_size]
            yield ds.data, idx, ds.data.y[idx].to(torch.long)

    def __len__(self) -> int:
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size


class LinkLoader:
    def __init__(self, dataset: LinkPredictionDataset, batch_size: int) -> None:
        self.dataset, self.batch_size = dataset, batch_size

    def __iter__(self) -> Iterator[Tuple[Batch, Tensor, Tensor]]:
        ds = self.dataset
        for s in range(0, len(ds), self.batch_size):
            yield ds.data, ds.edges[:, s:s + self.batch_size], ds.labels[s:s + self.batch_size].to(torch.float)

    def __len__(self) -> int:
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size


def create_graph_classification_loader(domain_name: str, split: str, batch_size: int, generator: torch.Generator,
                                       root: Optional[Path] = None) -> SequentialGraphLoader:
    d = processed_dir(domain_name, root)
    return SequentialGraphLoader(GraphDataset(GraphStore.load(d, with_properties=False), load_splits(d)[split]), batch_size)


def _single_graph(domain_name: str, root: Optional[Path]):
    d = processed_dir(domain_name, root)
    return GraphStore.load(d, with_properties=False).collate([0]), load_splits(d)


def create_node_classification_loader(domain_name: str, split: str, batch_size: int, generator: torch.Generator,
                                      root: Optional[Path] = None) -> NodeLoader:
    data, splits = _single_graph(domain_name, root)
    ds = NodeDataset(data, splits[split])
    return NodeLoader(ds, len(ds) if batch_size == -1 else batch_size)


def create_link_prediction_loader(domain_name: str, split: str, batch_size: int, generator: torch.Generator,
                                  root: Optional[Path] = None) -> LinkLoader:
    data, splits = _single_graph(domain_name, root)
    return LinkLoader(LinkPredictionDataset(data, splits, split), batch_size)


def create_finetune_data_loader(domain_name: str, split: str, batch_size: int, generator: torch.Generator,
                                root: Optional[Path] = None):
    kind = TASK_TYPES[domain_name]
    if kind == "graph_classification":
        return create_graph_classification_loader(domain_name, split, batch_size, generator, root)
    if kind == "node_classification":
        return create_node_classification_loader(domain_name, split, batch_size, generator, root)
    return create_link_prediction_loader(domain_name, split, batch_size, generator, root)
