"""Pre-training loaders with the behaviour of src/data/pretrain_data_loaders.py over a GraphStore.

`BalancedMultiDomainSampler` draws, per step and per domain in dict order, `torch.randint(0, len, (32 // D,))` from
the shared CPU generator -- with replacement, exactly the reference's draw sequence (:35-43) -- so a run over the same
processed data picks the same graphs.  A batch is then one vectorised `GraphStore.collate` instead of
`Batch.from_data_list` over Python objects."""
from __future__ import annotations

from pathlib import Path
from typing import Dict, Iterator, List, Optional

import torch

from ..graph import Batch, Data
from .data_setup import processed_dir
from .store import GraphStore, load_splits

BATCH_SIZE = 32


class GraphDataset:
    """pretrain_data_loaders.py:16-25: a split of a processed dataset."""

    def __init__(self, graphs: GraphStore, indices) -> None:
        self.graphs = graphs
        self.indices = torch.as_tensor(indices, dtype=torch.long).reshape(-1)

    def __len__(self) -> int:
        return int(self.indices.numel())

    def __getitem__(self, idx: int) -> Data:
        return self.graphs.graph(int(self.indices[idx]))

    def collate(self, positions) -> Batch:
        return self.graphs.collate(self.indices[torch.as_tensor(positions, dtype=torch.long)])


class BalancedMultiDomainSampler:
    def __init__(self, domain_datasets: Dict[str, GraphDataset], generator: torch.Generator) -> None:
        self.domain_datasets = domain_datasets
        self.generator = generator
        self.samples_per_domain = BATCH_SIZE // len(domain_datasets)
        self.num_steps = max(len(d) for d in domain_datasets.values()) // self.samples_per_domain

    def __iter__(self) -> Iterator[Dict[str, Batch]]:
        for _ in range(self.num_steps):
            out = {}
            for domain, dataset in self.domain_datasets.items():
                pick = torch.randint(0, len(dataset), (self.samples_per_domain,), generator=self.generator)
                out[domain] = dataset.collate(pick)
            yield out

    def __len__(self) -> int:
        return self.num_steps


class SequentialGraphLoader:
    """torch_geometric DataLoader(dataset, batch_size, generator=generator) as the reference builds it (:65): no shuffle, last
    batch kept.  One detail of torch's DataLoader is part of the contract because the reference hands its SHARED generator to the
    validation loaders: every `iter(loader)` draws a 64-bit base seed from `loader.generator`
    (torch/utils/data/dataloader.py, _BaseDataLoaderIter.__init__: `torch.empty((), dtype=torch.int64).random_(generator=...)`),
    i.e. once per (task, domain) pass of run_evaluation (pretrain.py:214), interleaved with that pass's mask / view / negative
    draws.  `draw_base_seed` makes that draw; `batches()` lists the batches without it (for callers that order the draws
    themselves, run_evaluation_engine)."""

    def __init__(self, dataset: GraphDataset, batch_size: int, generator: Optional[torch.Generator] = None) -> None:
        self.dataset, self.batch_size, self.generator = dataset, batch_size, generator

    def draw_base_seed(self) -> None:
        if self.generator is not None:
            torch.empty((), dtype=torch.int64).random_(generator=self.generator)

    def batches(self) -> List[Batch]:
        return [self.dataset.collate(torch.arange(s, min(s + self.batch_size, len(self.dataset))))
                for s in range(0, len(self.dataset), self.batch_size)]

    def __iter__(self) -> Iterator[Batch]:
        self.draw_base_seed()
        for s in range(0, len(self.dataset), self.batch_size):
            yield self.dataset.collate(torch.arange(s, min(s + self.batch_size, len(self.dataset))))

    def __len__(self) -> int:
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size


def _split_dataset(domain_name: str, split: str, root: Optional[Path]) -> GraphDataset:
    d = processed_dir(domain_name, root)
    return GraphDataset(GraphStore.load(d), load_splits(d)[split])


def create_val_data_loader(domain_name: str, generator: torch.Generator, root: Optional[Path] = None) -> SequentialGraphLoader:
    return SequentialGraphLoader(_split_dataset(domain_name, "val", root), BATCH_SIZE, generator)


def create_train_data_loader(domains: List[str], generator: torch.Generator, root: Optional[Path] = None) -> BalancedMultiDomainSampler:
    return BalancedMultiDomainSampler({d: _split_dataset(d, "train", root) for d in domains}, generator)
