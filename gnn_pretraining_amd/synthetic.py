"""Synthetic inputs of the shapes SURVEY.md section 8d names (there is no network for
the TUDataset / Planetoid downloads of src/data/data_setup.py).  Everything is
drawn from a CPU torch.Generator so CPU-oracle and GPU runs see identical tensors.
"""
from __future__ import annotations

from typing import Dict, List

import torch

from .graph import Batch, Data

GRAPH_PROPERTY_DIM = 12
# name -> (input dim [data_setup.py:31-41], mean nodes, mean undirected edges) -- TUDataset public statistics
DOMAIN_SHAPES = {"MUTAG": (7, 18.0, 20.0), "PROTEINS": (4, 39.0, 73.0), "NCI1": (37, 30.0, 32.0),
                 "ENZYMES": (21, 33.0, 62.0)}
ENZYMES_SHAPE = (33.0, 62.0)


def random_graph(gen: torch.Generator, dim: int, mean_nodes: float = 33.0, mean_edges: float = 62.0,
                 num_classes: int = 6) -> Data:
    """One 'G-ENZ'-style graph: n ~ clip(round(N(mean, (8*mean/33)^2)), 3, 126); round(mean_edges*n/mean_nodes) >= 2
    undirected pairs sampled without replacement, stored in both directions sorted by (src, dst);
    x ~ clip(N(0,1), -3, 3) (StandardScaler + clip, data_setup.py:17-18,93-100)."""
    std = 8.0 * mean_nodes / 33.0
    n = int(torch.clamp(torch.round(torch.randn(1, generator=gen) * std + mean_nodes), 3, 126).item())
    pairs = n * (n - 1) // 2
    m = min(max(2, int(round(mean_edges * n / mean_nodes))), pairs)
    pick = torch.randperm(pairs, generator=gen)[:m]
    iu = torch.triu_indices(n, n, offset=1)
    a, b = iu[0][pick], iu[1][pick]
    src, dst = torch.cat([a, b]), torch.cat([b, a])
    order = torch.argsort(src * n + dst)
    ei = torch.stack([src[order], dst[order]])
    x = torch.randn(n, dim, generator=gen).clamp_(-3.0, 3.0)
    gp = torch.randn(GRAPH_PROPERTY_DIM, generator=gen)
    y = torch.randint(0, num_classes, (1,), generator=gen)
    return Data(x, ei, y, gp)


def domain_batch(gen: torch.Generator, dim: int, graphs: int = 8, mean_nodes: float = 33.0,
                 mean_edges: float = 62.0) -> Batch:
    return Batch.from_data_list([random_graph(gen, dim, mean_nodes, mean_edges) for _ in range(graphs)])


def pretrain_step_batches(gen: torch.Generator, domains: List[str], graphs_per_domain: int = 8,
                          enzymes_shaped: bool = True) -> Dict[str, Batch]:
    """One training step's input: {domain: Batch of 8 graphs} (pretrain_data_loaders.py:35-43).
    enzymes_shaped=True is the headline 'synthetic ENZYMES-shaped' workload (every domain keeps its
    own input width but uses the ENZYMES size distribution); False uses each dataset's own sizes."""
    out = {}
    for d in domains:
        dim, mn, me = DOMAIN_SHAPES[d]
        if enzymes_shaped:
            mn, me = ENZYMES_SHAPE
        out[d] = domain_batch(gen, dim, graphs_per_domain, mn, me)
    return out


def cora_like(gen: torch.Generator, num_nodes: int = 2708, undirected_edges: int = 5429, dim: int = 1433,
              density: float = 0.0127, num_classes: int = 7) -> Data:
    """Cora-shaped single graph (README.md:128): 5,429 undirected pairs stored both ways, sorted;
    x non-negative, ~1.27 % dense, rows normalised to sum 1 (NormalizeFeatures, data_setup.py:154)."""
    n = num_nodes
    seen = torch.randint(0, n * n, (undirected_edges * 3,), generator=gen)
    a, b = seen // n, seen % n
    keep = a < b
    key = torch.unique(a[keep] * n + b[keep])
    key = key[torch.randperm(key.numel(), generator=gen)[:undirected_edges]]
    a, b = key // n, key % n
    src, dst = torch.cat([a, b]), torch.cat([b, a])
    order = torch.argsort(src * n + dst)
    ei = torch.stack([src[order], dst[order]])
    x = (torch.rand(n, dim, generator=gen) < density).float()
    x[torch.arange(n), torch.randint(0, dim, (n,), generator=gen)] = 1.0     # no empty rows
    x = x / x.sum(dim=1, keepdim=True)
    y = torch.randint(0, num_classes, (n,), generator=gen)
    return Data(x, ei, y, None)
