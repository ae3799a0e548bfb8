"""Graph containers with the attribute surface the reference uses from
torch_geometric.data (Data / Batch): x, edge_index, batch, ptr, num_graphs,
graph_properties, y, .to(device), to_data_list(), Batch.from_data_list()
(src/pretrain/augmentations.py:91,108-109; src/data/pretrain_data_loaders.py:41;
src/pretrain/pretrain.py:116-117).

Additions for the HIP path: host copies of ``ptr`` / ``edge_ptr`` (so no
``.item()`` sync is ever needed to walk the graphs of a batch) and a per-object
cache of the int32 CSR built by libgnnmp.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
from torch import Tensor


class Data:
    def __init__(self, x: Tensor, edge_index: Tensor, y: Optional[Tensor] = None,
                 graph_properties: Optional[Tensor] = None) -> None:
        self.x, self.edge_index, self.y, self.graph_properties = x, edge_index, y, graph_properties

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
        c = lambda t: None if t is None else t.clone()
        return Data(self.x.clone(), self.edge_index.clone(), c(self.y), c(self.graph_properties))

    def to(self, device) -> "Data":
        m = lambda t: None if t is None else t.to(device)
        return Data(self.x.to(device), self.edge_index.to(device), m(self.y), m(self.graph_properties))


class Batch:
    def __init__(self, x, edge_index, batch, ptr, ptr_host: List[int], edge_ptr_host: List[int],
                 y=None, graph_properties=None) -> None:
        self.x, self.edge_index, self.batch, self.ptr = x, edge_index, batch, ptr
        self.ptr_host, self.edge_ptr_host = ptr_host, edge_ptr_host
        self.y, self.graph_properties = y, graph_properties
        self._cache = {}

    # ---- reference-visible surface -------------------------------------------------
    @property
    def num_graphs(self) -> int:
        return len(self.ptr_host) - 1

    @property
    def num_nodes(self) -> int:
        return int(self.x.size(0))

    @property
    def num_edges(self) -> int:
        return int(self.edge_index.size(1))

    @property
    def device(self):
        return self.x.device

    @staticmethod
    def from_data_list(graphs: Sequence[Data]) -> "Batch":
        sizes = [g.num_nodes for g in graphs]
        ptr_h = [0]
        for s in sizes:
            ptr_h.append(ptr_h[-1] + s)
        eptr_h = [0]
        for g in graphs:
            eptr_h.append(eptr_h[-1] + g.num_edges)
        dev = graphs[0].x.device
        x = torch.cat([g.x for g in graphs], dim=0)
        ei = torch.cat([g.edge_index + off for g, off in zip(graphs, ptr_h)], dim=1)
        batch = torch.repeat_interleave(torch.arange(len(graphs), device=dev), torch.tensor(sizes, device=dev))
        ptr = torch.tensor(ptr_h, dtype=torch.long, device=dev)
        y = torch.cat([g.y.reshape(-1) for g in graphs]) if graphs[0].y is not None else None
        gp = (torch.cat([g.graph_properties.reshape(-1) for g in graphs])
              if graphs[0].graph_properties is not None else None)
        return Batch(x, ei, batch, ptr, ptr_h, eptr_h, y, gp)

    @staticmethod
    def empty(num_node_features: int, with_properties: bool = True) -> "Batch":
        """A batch of zero graphs (a domain absent from a step, e.g. single-domain validation passes on the stacked engine)."""
        z = torch.zeros(0, dtype=torch.long)
        return Batch(torch.zeros(0, num_node_features), torch.zeros(2, 0, dtype=torch.long), z, torch.zeros(1, dtype=torch.long), [0], [0],
                     z.clone(), torch.zeros(0) if with_properties else None)

    def to_data_list(self) -> List[Data]:
        B = self.num_graphs
        gp = None if self.graph_properties is None else self.graph_properties.view(B, -1)
        out = []
        for g in range(B):
            s, e = self.ptr_host[g], self.ptr_host[g + 1]
            es, ee = self.edge_ptr_host[g], self.edge_ptr_host[g + 1]
            out.append(Data(self.x[s:e], self.edge_index[:, es:ee] - s,
                            None if self.y is None else self.y[g:g + 1], None if gp is None else gp[g]))
        return out

    def to(self, device) -> "Batch":
        m = lambda t: None if t is None else t.to(device, non_blocking=True)
        out = Batch(m(self.x), m(self.edge_index), m(self.batch), m(self.ptr), self.ptr_host, self.edge_ptr_host,
                    m(self.y), m(self.graph_properties))
        if self.x.device.type == "cpu":
            out._cache["host"] = self        # index work (augmentation, sampling) stays on the host copy
        return out

    def host(self) -> "Batch":
        """The CPU twin of this batch (kept when the batch was moved with .to(); otherwise one D2H copy)."""
        if self.x.device.type == "cpu":
            return self
        if "host" not in self._cache:
            m = lambda t: None if t is None else t.cpu()
            self._cache["host"] = Batch(m(self.x), m(self.edge_index), m(self.batch), m(self.ptr), self.ptr_host,
                                        self.edge_ptr_host, m(self.y), m(self.graph_properties))
        return self._cache["host"]

    # ---- HIP-path extras -----------------------------------------------------------
    @property
    def ptr32(self) -> Tensor:
        """Graph row offsets as device int32 (segment pointer for the pooling kernels)."""
        if "ptr32" not in self._cache:
            self._cache["ptr32"] = torch.tensor(self.ptr_host, dtype=torch.int32).to(self.x.device)
        return self._cache["ptr32"]

    @property
    def max_graph_nodes(self) -> int:
        return max((b - a for a, b in zip(self.ptr_host[:-1], self.ptr_host[1:])), default=0)
