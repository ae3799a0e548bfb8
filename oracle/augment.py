"""Oracle restatement of src/pretrain/augmentations.py plus the negative-edge
sampler used where the reference calls PyG's batched_negative_sampling.

TEST INFRASTRUCTURE (see oracle/__init__.py).

RNG contract (augmentations.py:17-74): all draws come from the caller's CPU
``torch.Generator`` in this order, per graph, view 1 then view 2:
    randperm(n)                     if n >= 3            (node drop, always applied)
    rand(1)                         always               (edge-drop coin, p = .2)
    randperm(E')                    if coin and E' >= 3  (E' = edges left after node drop)
    rand(1)                         always               (attr-mask coin, p = .2)
    randperm(F)                     if coin and F >= 3
Because the draws are CPU draws from a seeded generator, a GPU run of the
reference produces the same indices as a CPU run -- index parity is bit-exact.
"""
from __future__ import annotations

from typing import List, Tuple

import torch
from torch import Tensor

from .graph_ops import Batch, Data, subgraph

NODE_DROP_RATE = 0.2
NODE_DROP_MIN = 3
EDGE_DROP_P = 0.2
EDGE_DROP_RATE = 0.2
EDGE_DROP_MIN = 3
ATTR_MASK_P = 0.2
ATTR_MASK_RATE = 0.2
ATTR_MASK_MIN = 3


def make_view(graph: Data, gen: torch.Generator) -> Tuple[Data, Tensor]:
    """augmentations.py:44-74 (_node_drop, _edge_drop, _attribute_mask, _create_augmented_view)."""
    x, ei = graph.x.clone(), graph.edge_index.clone()
    n = x.size(0)
    if n >= NODE_DROP_MIN:
        keep_n = n - max(1, int(n * NODE_DROP_RATE))
        kept = torch.randperm(n, generator=gen)[:keep_n].sort()[0]
        ei = subgraph(kept, ei, n)
        x = x[kept]
    else:
        kept = torch.arange(n)
    if torch.rand(1, generator=gen).item() < EDGE_DROP_P:
        e = ei.size(1)
        if e >= EDGE_DROP_MIN:
            keep_e = e - max(1, int(e * EDGE_DROP_RATE))
            ei = ei[:, torch.randperm(e, generator=gen)[:keep_e]]
    if torch.rand(1, generator=gen).item() < ATTR_MASK_P:
        f = x.size(1)
        if f >= ATTR_MASK_MIN:
            cols = torch.randperm(f, generator=gen)[:max(1, int(f * ATTR_MASK_RATE))]
            x[:, cols] = 0.0
    return Data(x, ei, graph.y, graph.graph_properties), kept


def common_masks(kept1: Tensor, kept2: Tensor) -> Tuple[Tensor, Tensor]:
    """augmentations.py:77-85: boolean masks over each view's nodes marking the
    original nodes that survived in both views."""
    both = torch.cat([kept1, kept2])
    uniq, cnt = both.unique(return_counts=True)
    common = uniq[cnt == 2]
    return torch.isin(kept1, common), torch.isin(kept2, common)


def create_two_views(batch: Batch, gen: torch.Generator) -> Tuple[Batch, Batch, List[Tensor], List[Tensor]]:
    """augmentations.py:88-111."""
    v1, v2, m1, m2 = [], [], [], []
    for g in batch.to_data_list():
        a, ka = make_view(g, gen)
        b, kb = make_view(g, gen)
        ma, mb = common_masks(ka, kb)
        v1.append(a); v2.append(b); m1.append(ma); m2.append(mb)
    return Batch.from_data_list(v1), Batch.from_data_list(v2), m1, m2


def sample_negative_edges(batch: Batch, gen: torch.Generator) -> Tensor:
    """Stand-in for ``batched_negative_sampling(to_undirected(pos), batch,
    num_neg_samples=E)`` (tasks.py:107-111).  PyG's sampler draws from Python's
    ``random``/NumPy, not from the seeded generator, so its output cannot be
    reproduced; parity tests therefore take negatives as an input.  This sampler
    implements the benchmark rule of SURVEY.md section 8d: per graph, as many
    negatives as the graph has directed COO entries, drawn uniformly without
    replacement from ordered pairs (i, j), i != j, that are not adjacent in
    either direction; graph order, then draw order."""
    outs = []
    for g in range(batch.num_graphs):
        s, e = int(batch.ptr[g]), int(batch.ptr[g + 1])
        n = e - s
        es, ee = int(batch.edge_ptr[g]), int(batch.edge_ptr[g + 1])
        ei = batch.edge_index[:, es:ee] - s
        adj = torch.zeros(n, n, dtype=torch.bool)
        adj[ei[0], ei[1]] = True
        adj[ei[1], ei[0]] = True
        adj.fill_diagonal_(True)
        cand = (~adj).flatten().nonzero().squeeze(1)
        k = min(ee - es, cand.numel())
        if k == 0:
            continue
        pick = cand[torch.randperm(cand.numel(), generator=gen)[:k]]
        outs.append(torch.stack([pick // n, pick % n]) + s)
    return torch.cat(outs, dim=1) if outs else torch.empty(2, 0, dtype=torch.long)
