"""Flat on-disk / in-memory form of a processed dataset.

The reference keeps a processed dataset as a pickled Python list of torch_geometric ``Data`` objects
(``data/processed/{D}/data.pt``, src/data/data_setup.py:66-72) and rebuilds a ``Batch`` from 8-32 of them per
step with ``Batch.from_data_list`` (src/data/pretrain_data_loaders.py:41).  Here a dataset is five flat arrays,

    x          [sum_n, d]  float32   node features of all graphs, graph after graph
    edge_index [2, sum_e]  int64     COO pairs, node ids LOCAL to their graph
    node_ptr   [G + 1]     int64     first row of graph g in x
    edge_ptr   [G + 1]     int64     first column of graph g in edge_index
    y          [G] or [sum_n] int64  graph labels, or node labels for the single-graph Planetoid sets

stored in one ``data.safetensors`` (no pickle: a loader that executes nothing from the file), so collating a
batch is four vectorised gathers instead of a Python loop over objects, and the same arrays can be kept resident
on the GPU.  ``collate`` returns exactly what ``Batch.from_data_list([store.graph(i) for i in indices])``
returns (tests/test_data.py).
"""
from __future__ import annotations

from pathlib import Path
from typing import Dict, List, Optional, Sequence, Union

import numpy as np
import torch
from torch import Tensor

from ..graph import Batch, Data

DATA_FILE, SPLITS_FILE, PROPS_FILE = "data.safetensors", "splits.safetensors", "graph_properties.safetensors"


class GraphStore:
    def __init__(self, x: Tensor, edge_index: Tensor, node_ptr: Tensor, edge_ptr: Tensor, y: Optional[Tensor] = None,
                 graph_properties: Optional[Tensor] = None) -> None:
        if node_ptr.numel() != edge_ptr.numel() or int(node_ptr[-1]) != x.size(0) or int(edge_ptr[-1]) != edge_index.size(1):
            raise ValueError("GraphStore: pointer arrays do not match x / edge_index")
        self.x, self.edge_index = x.contiguous(), edge_index.contiguous()
        self.node_ptr, self.edge_ptr = node_ptr.to(torch.long), edge_ptr.to(torch.long)
        self.y, self.graph_properties = y, graph_properties
        self._node_ptr_h: List[int] = self.node_ptr.tolist()
        self._edge_ptr_h: List[int] = self.edge_ptr.tolist()

    # ---- construction --------------------------------------------------------------------------------------------
    @staticmethod
    def from_data_list(graphs: Sequence[Data]) -> "GraphStore":
        nptr, eptr = [0], [0]
        for g in graphs:
            nptr.append(nptr[-1] + g.num_nodes)
            eptr.append(eptr[-1] + g.num_edges)
        y = None
        if graphs[0].y is not None:
            y = torch.cat([g.y.reshape(-1) for g in graphs])
        gp = None
        if graphs[0].graph_properties is not None:
            gp = torch.stack([g.graph_properties.reshape(-1) for g in graphs])
        return GraphStore(torch.cat([g.x for g in graphs]), torch.cat([g.edge_index for g in graphs], dim=1),
                          torch.tensor(nptr), torch.tensor(eptr), y, gp)

    # ---- access ----------------------------------------------------------------------------------------------------
    def __len__(self) -> int:
        return len(self._node_ptr_h) - 1

    @property
    def num_node_features(self) -> int:
        return int(self.x.size(1))

    @property
    def node_level_labels(self) -> bool:
        return self.y is not None and self.y.numel() == self.x.size(0) and len(self) != self.x.size(0)

    def graph(self, i: int) -> Data:
        """One graph as a Data object (views into the flat arrays; edge ids are graph-local, as in the reference)."""
        s, e = self._node_ptr_h[i], self._node_ptr_h[i + 1]
        es, ee = self._edge_ptr_h[i], self._edge_ptr_h[i + 1]
        y = None
        if self.y is not None:
            y = self.y[s:e] if self.node_level_labels else self.y[i:i + 1]
        gp = None if self.graph_properties is None else self.graph_properties[i]
        return Data(self.x[s:e], self.edge_index[:, es:ee], y, gp)

    def collate(self, indices: Union[Sequence[int], Tensor]) -> Batch:
        """Batch of the graphs `indices` (repeats allowed: the balanced sampler draws with replacement)."""
        idx = np.asarray(indices, dtype=np.int64).reshape(-1)       # numpy throughout: these are tiny index ops, and
        B = idx.size                                                # torch's OpenMP fan-out costs more than they do
        nptr, eptr = self.node_ptr.numpy(), self.edge_ptr.numpy()
        n0, e0 = nptr[idx], eptr[idx]
        nsz, esz = nptr[idx + 1] - n0, eptr[idx + 1] - e0
        out_ptr = np.zeros(B + 1, dtype=np.int64)
        np.cumsum(nsz, out=out_ptr[1:])
        out_eptr = np.zeros(B + 1, dtype=np.int64)
        np.cumsum(esz, out=out_eptr[1:])
        N, E = int(out_ptr[-1]), int(out_eptr[-1])
        batch = np.repeat(np.arange(B, dtype=np.int64), nsz)
        rows = np.arange(N, dtype=np.int64) + np.repeat(n0 - out_ptr[:-1], nsz)
        cols = np.arange(E, dtype=np.int64) + np.repeat(e0 - out_eptr[:-1], esz)
        ei = self.edge_index.numpy()[:, cols] + np.repeat(out_ptr[:-1], esz)
        y = None
        if self.y is not None:
            y = torch.from_numpy(self.y.numpy()[rows if self.node_level_labels else idx])
        gp = None
        if self.graph_properties is not None:
            gp = torch.from_numpy(self.graph_properties.numpy()[idx].reshape(-1))
        return Batch(torch.from_numpy(self.x.numpy()[rows]), torch.from_numpy(ei), torch.from_numpy(batch),
                     torch.from_numpy(out_ptr), out_ptr.tolist(), out_eptr.tolist(), y, gp)

    # ---- disk ------------------------------------------------------------------------------------------------------
    def save(self, directory: Union[str, Path]) -> None:
        from safetensors.torch import save_file
        d = Path(directory)
        d.mkdir(parents=True, exist_ok=True)
        t = {"x": self.x, "edge_index": self.edge_index, "node_ptr": self.node_ptr, "edge_ptr": self.edge_ptr}
        if self.y is not None:
            t["y"] = self.y.contiguous()
        save_file(t, str(d / DATA_FILE))
        if self.graph_properties is not None:
            save_file({"graph_properties": self.graph_properties.contiguous()}, str(d / PROPS_FILE))

    @staticmethod
    def load(directory: Union[str, Path], with_properties: bool = True) -> "GraphStore":
        from safetensors.torch import load_file
        d = Path(directory)
        if not (d / DATA_FILE).exists():
            raise FileNotFoundError(f"{d / DATA_FILE}: run `python -m gnn_pretraining_amd.data.data_setup` (synthetic "
                                    "stand-ins) or export the reference's processed data (INTEGRATION.md section 5)")
        t = load_file(str(d / DATA_FILE))
        gp = None
        if with_properties and (d / PROPS_FILE).exists():
            gp = load_file(str(d / PROPS_FILE))["graph_properties"]
        return GraphStore(t["x"], t["edge_index"], t["node_ptr"], t["edge_ptr"], t.get("y"), gp)


def save_splits(directory: Union[str, Path], splits: Dict[str, Tensor]) -> None:
    from safetensors.torch import save_file
    Path(directory).mkdir(parents=True, exist_ok=True)
    save_file({k: v.contiguous() for k, v in splits.items()}, str(Path(directory) / SPLITS_FILE))


def load_splits(directory: Union[str, Path]) -> Dict[str, Tensor]:
    from safetensors.torch import load_file
    return load_file(str(Path(directory) / SPLITS_FILE))
