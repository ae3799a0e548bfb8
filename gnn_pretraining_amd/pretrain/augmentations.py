"""Two-view graph augmentation (node drop, edge drop, attribute mask) with the reference's exact
RNG contract, done on HOST index arrays instead of per-graph device tensors.

Reference: src/pretrain/augmentations.py:17-111.  Draw order per graph, view 1 then view 2:
randperm(n) [n>=3] ; rand(1) ; randperm(E') [coin<.2 and E'>=3] ; rand(1) ; randperm(F) [coin<.2 and F>=3].
All draws come from the caller's CPU generator, so for an equal generator state the produced
views (node sets, edge lists, masked columns) are identical to the reference's, bit for bit.
The reference clones every graph onto the device and runs `subgraph` there (dozens of tiny
launches and syncs per graph); here the whole batch is cut on the host with numpy and each view is
uploaded once.
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np
import torch
from torch import Tensor

from ..graph import Batch

ATTR_MASK_MIN_NUM_FEATURES = 3
ATTR_MASK_PROB = 0.2
ATTR_MASK_RATE = 0.2
EDGE_DROP_MIN_NUM_EDGES = 3
EDGE_DROP_PROB = 0.2
EDGE_DROP_RATE = 0.2
NODE_DROP_MIN_NUM_NODES = 3
NODE_DROP_RATE = 0.2


class _View:
    """One augmented graph, as host arrays in LOCAL node numbering of the view."""
    __slots__ = ("kept", "edges", "masked_cols")

    def __init__(self, kept: np.ndarray, edges: np.ndarray, masked_cols) -> None:
        self.kept, self.edges, self.masked_cols = kept, edges, masked_cols


def _augment_one(n: int, edges: np.ndarray, num_features: int, gen: torch.Generator) -> _View:
    if n >= NODE_DROP_MIN_NUM_NODES:
        keep_n = n - max(1, int(n * NODE_DROP_RATE))
        kept = np.sort(torch.randperm(n, generator=gen)[:keep_n].numpy())
        relabel = np.full(n, -1, dtype=np.int64)
        relabel[kept] = np.arange(keep_n)
        both = (relabel[edges[0]] >= 0) & (relabel[edges[1]] >= 0)          # subgraph(): edge order preserved
        edges = relabel[edges[:, both]]
    else:
        kept = np.arange(n)
    if torch.rand(1, generator=gen).item() < EDGE_DROP_PROB:
        e = edges.shape[1]
        if e >= EDGE_DROP_MIN_NUM_EDGES:
            keep_e = e - max(1, int(e * EDGE_DROP_RATE))
            edges = edges[:, torch.randperm(e, generator=gen)[:keep_e].numpy()]
    cols = None
    if torch.rand(1, generator=gen).item() < ATTR_MASK_PROB:
        if num_features >= ATTR_MASK_MIN_NUM_FEATURES:
            cols = torch.randperm(num_features, generator=gen)[:max(1, int(num_features * ATTR_MASK_RATE))].numpy()
    return _View(kept, edges, cols)


def _assemble(batch_host: Batch, views: List[_View], device) -> Batch:
    """Batch.from_data_list of the augmented graphs, built in one go."""
    x_host = batch_host.x
    ptr, sizes, rows, eds, eptr = [0], [], [], [], [0]
    for g, v in enumerate(views):
        rows.append(v.kept + batch_host.ptr_host[g])
        eds.append(v.edges + ptr[-1])
        sizes.append(len(v.kept))
        ptr.append(ptr[-1] + len(v.kept))
        eptr.append(eptr[-1] + v.edges.shape[1])
    rows_t = torch.from_numpy(np.concatenate(rows))
    x = x_host[rows_t]                                                     # fresh host copy
    for g, v in enumerate(views):
        if v.masked_cols is not None:
            x[ptr[g]:ptr[g + 1], torch.from_numpy(v.masked_cols)] = 0.0
    ei = torch.from_numpy(np.concatenate(eds, axis=1)) if eds else torch.empty(2, 0, dtype=torch.long)
    bvec = torch.repeat_interleave(torch.arange(len(views)), torch.tensor(sizes))
    out = Batch(x, ei, bvec, torch.tensor(ptr, dtype=torch.long), ptr, eptr,
                batch_host.y, batch_host.graph_properties)
    return out.to(device) if device is not None and torch.device(device).type != "cpu" else out


def common_masks(kept1: np.ndarray, kept2: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """augmentations.py:77-85: per view, which of its nodes also survive in the other view."""
    return np.isin(kept1, kept2), np.isin(kept2, kept1)


class GraphAugmentor:
    @staticmethod
    def create_two_views(batch: Batch, generator: torch.Generator) -> Tuple[Batch, Batch, List[Tensor], List[Tensor]]:
        host = batch.host()
        ei = host.edge_index.numpy()
        F = host.x.size(1)
        v1, v2, m1, m2 = [], [], [], []
        for g in range(host.num_graphs):
            s, e = host.ptr_host[g], host.ptr_host[g + 1]
            es, ee = host.edge_ptr_host[g], host.edge_ptr_host[g + 1]
            local = ei[:, es:ee] - s
            a = _augment_one(e - s, local, F, generator)
            b = _augment_one(e - s, local, F, generator)
            ma, mb = common_masks(a.kept, b.kept)
            v1.append(a); v2.append(b)
            m1.append(torch.from_numpy(ma)); m2.append(torch.from_numpy(mb))
        dev = batch.x.device
        return _assemble(host, v1, dev), _assemble(host, v2, dev), m1, m2
