"""Oracle restatement of the torch-geometric operators the hot path calls.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Third-party dependency restated:
torch-geometric >= 2.3.0 (requirements.txt:3, unpinned, not vendored) -> parity
unpinned at this boundary; semantics follow PyG's published behaviour as listed
in SURVEY.md section 8c.

Call sites in the reference:
  GINConv                       src/models/gnn.py:29-41
  global_mean_pool/max_pool     src/pretrain/tasks.py:241-246,299,331 ; src/models/finetune_model.py:75
  to_undirected                 src/pretrain/tasks.py:107-111
  subgraph                      src/pretrain/augmentations.py:56
  Batch / Data                  src/pretrain/augmentations.py:91,108-109 ; src/data/pretrain_data_loaders.py:41
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np
import torch
from torch import Tensor


# --------------------------------------------------------------------------- #
# containers
# --------------------------------------------------------------------------- #
class Data:
    """One graph: x [n,d] f32, edge_index [2,e] i64 (+ optional y, graph_properties)."""

    def __init__(self, x: Tensor, edge_index: Tensor, y: Optional[Tensor] = None,
                 graph_properties: Optional[Tensor] = None) -> None:
        self.x = x
        self.edge_index = edge_index
        self.y = y
        self.graph_properties = graph_properties

    @property
    def num_nodes(self) -> int:
        return int(self.x.size(0))

    @property
    def num_edges(self) -> int:
        return int(self.edge_index.size(1))

    @property
    def num_node_features(self) -> int:
        return int(self.x.size(1))

    def clone(self) -> "Data":
        return Data(self.x.clone(), self.edge_index.clone(),
                    None if self.y is None else self.y.clone(),
                    None if self.graph_properties is None else self.graph_properties.clone())


class Batch:
    """Block-diagonal collation of graphs (PyG ``Batch.from_data_list`` semantics):
    x concatenated, edge_index offset by cumulative node counts, ``batch`` = graph
    id per node, ``ptr`` = node offsets, per-graph attributes concatenated along
    dim 0 (so graph_properties becomes [B*12], cf. tasks.py:303 ``.view(B,12)``)."""

    def __init__(self, x, edge_index, batch, ptr, edge_ptr, y=None, graph_properties=None):
        self.x = x
        self.edge_index = edge_index
        self.batch = batch
        self.ptr = ptr
        self.edge_ptr = edge_ptr          # edges of graph g are edge_index[:, edge_ptr[g]:edge_ptr[g+1]]
        self.y = y
        self.graph_properties = graph_properties

    @property
    def num_graphs(self) -> int:
        return int(self.ptr.numel() - 1)

    @property
    def num_nodes(self) -> int:
        return int(self.x.size(0))

    @staticmethod
    def from_data_list(graphs: Sequence[Data]) -> "Batch":
        sizes = [g.num_nodes for g in graphs]
        ptr = torch.zeros(len(graphs) + 1, dtype=torch.long)
        ptr[1:] = torch.tensor(sizes, dtype=torch.long).cumsum(0)
        eptr = torch.zeros(len(graphs) + 1, dtype=torch.long)
        eptr[1:] = torch.tensor([g.num_edges for g in graphs], dtype=torch.long).cumsum(0)
        x = torch.cat([g.x for g in graphs], dim=0)
        ei = torch.cat([g.edge_index + int(ptr[i]) for i, g in enumerate(graphs)], dim=1)
        batch = torch.repeat_interleave(torch.arange(len(graphs)), torch.tensor(sizes))
        y = None
        if graphs[0].y is not None:
            y = torch.cat([g.y.reshape(-1) for g in graphs])
        gp = None
        if graphs[0].graph_properties is not None:
            gp = torch.cat([g.graph_properties.reshape(-1) for g in graphs])
        return Batch(x, ei, batch, ptr, eptr, y, gp)

    def to_data_list(self) -> List[Data]:
        out = []
        B = self.num_graphs
        gp = None if self.graph_properties is None else self.graph_properties.view(B, -1)
        for g in range(B):
            s, e = int(self.ptr[g]), int(self.ptr[g + 1])
            es, ee = int(self.edge_ptr[g]), int(self.edge_ptr[g + 1])
            out.append(Data(self.x[s:e], self.edge_index[:, es:ee] - s,
                            None if self.y is None else self.y[g:g + 1],
                            None if gp is None else gp[g]))
        return out


# --------------------------------------------------------------------------- #
# edge utilities
# --------------------------------------------------------------------------- #
def to_undirected(edge_index: Tensor, num_nodes: Optional[int] = None) -> Tensor:
    """PyG ``to_undirected``: concatenate both directions, then coalesce
    (sort by (row, col), drop duplicates)."""
    row, col = edge_index[0], edge_index[1]
    row2 = torch.cat([row, col])
    col2 = torch.cat([col, row])
    n = int(max(int(row2.max()) + 1 if row2.numel() else 0, num_nodes or 0))
    key = torch.unique(row2 * n + col2)            # sorted ascending == (row, col) order
    return torch.stack([key // n, key % n])


def subgraph(kept: Tensor, edge_index: Tensor, num_nodes: int) -> Tensor:
    """PyG ``subgraph(subset, edge_index, relabel_nodes=True, num_nodes=n)``:
    keep edges whose endpoints are both kept (edge order preserved) and relabel
    endpoints to their rank in ``kept`` (kept is sorted: augmentations.py:54)."""
    keep = torch.zeros(num_nodes, dtype=torch.bool)
    keep[kept] = True
    emask = keep[edge_index[0]] & keep[edge_index[1]]
    relabel = torch.full((num_nodes,), -1, dtype=torch.long)
    relabel[kept] = torch.arange(kept.numel())
    return relabel[edge_index[:, emask]]


def coo_to_csr(edge_index: Tensor, num_nodes: int, by: str = "dst"):
    """Integer oracle for the device CSR builder (bit-exact contract).

    Groups edges by destination (by='dst': the forward aggregation reads, for row
    i, the sources j of edges j->i) or by source (by='src': the transposed graph
    used by the backward pass).  Within a row, entries keep ascending COO edge id
    (stable counting sort).  Returns (rowptr i32 [N+1], col i32 [E], perm i32 [E])
    where perm[k] is the COO edge id stored at CSR slot k."""
    src = edge_index[0].numpy()
    dst = edge_index[1].numpy()
    key, other = (dst, src) if by == "dst" else (src, dst)
    perm = np.argsort(key, kind="stable").astype(np.int32)
    counts = np.bincount(key, minlength=num_nodes)
    rowptr = np.zeros(num_nodes + 1, dtype=np.int32)
    rowptr[1:] = np.cumsum(counts)
    col = other[perm].astype(np.int32)
    return torch.from_numpy(rowptr), torch.from_numpy(col), torch.from_numpy(perm)


# --------------------------------------------------------------------------- #
# message passing + read-out
# --------------------------------------------------------------------------- #
def gin_aggregate(x: Tensor, edge_index: Tensor, eps: Tensor) -> Tensor:
    """GINConv before its nn: sum_{(j->i) in E} x_j + (1+eps) * x_i, with
    edge_index[0] = source j and edge_index[1] = target i (PyG flow
    source_to_target; gnn.py:29-41)."""
    agg = torch.zeros_like(x).index_add_(0, edge_index[1], x[edge_index[0]])
    return agg + (1.0 + eps) * x


def global_mean_pool(x: Tensor, batch: Tensor, size: Optional[int] = None) -> Tensor:
    """PyG scatter(reduce='mean'): segment sum / clamp(count, 1); B = batch.max()+1."""
    B = int(batch.max()) + 1 if size is None else size
    s = torch.zeros(B, x.size(1), dtype=x.dtype).index_add_(0, batch, x)
    c = torch.zeros(B, dtype=x.dtype).index_add_(0, batch, torch.ones_like(batch, dtype=x.dtype))
    return s / c.clamp(min=1).unsqueeze(1)


def global_max_pool(x: Tensor, batch: Tensor, size: Optional[int] = None) -> Tensor:
    """PyG scatter(reduce='max') on the CPU path: ``new_zeros(size).scatter_reduce_
    (0, index, src, 'amax', include_self=False)``.  Its autograd splits the
    incoming gradient EVENLY between tied maxima (torch derivative of
    scatter_reduce amax) -- that is the tie rule the HIP kernel reproduces.  Measured
    here (torch 2.10): the zero-initialised output counts as one more tie whenever
    the segment maximum is exactly 0, even with include_self=False."""
    B = int(batch.max()) + 1 if size is None else size
    idx = batch.view(-1, 1).expand_as(x)
    return x.new_zeros(B, x.size(1)).scatter_reduce(0, idx, x, reduce="amax", include_self=False)
