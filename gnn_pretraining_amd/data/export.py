"""Bridge from the reference's on-disk data contract to this library's.

The reference writes, per dataset D (src/data/data_setup.py:66-72, read back by src/data/pretrain_data_loaders.py:28-46 and
src/data/finetune_data_loaders.py):

    data/processed/D/data.pt              torch.save(list of torch_geometric.data.Data)      -- pickled objects
    data/processed/D/splits.pt            torch.save({'train': idx, 'val': idx[, 'test': idx]} or the link-prediction edge splits)
    data/processed/D/graph_properties.pt  torch.save(float tensor [G, 12])                   (pre-training datasets only)

Here a dataset is a `GraphStore` (data/store.py: five flat arrays in one data.safetensors -- nothing executes on load).
`export_dataset` turns the reference's objects into that form.  It is duck-typed on the attributes the reference pickles
(`x`, `edge_index` with graph-LOCAL ids, `y`, `num_nodes`), so it runs on real `torch_geometric.data.Data` lists in the
reference's environment -- where this module needs nothing but torch and safetensors -- and on this repository's own `Data`
objects (tests/test_data.py round-trips it without PyG).  `export_processed_tree` walks a whole data/processed directory;
splits.pt / graph_properties.pt are read with the weights-only loader; data.pt (a pickle of objects) only behind an explicit
`allow_pickle=True` -- for files the caller generated himself (`python -m src.data.data_setup`), never for downloaded ones.
INTEGRATION.md section 5 shows the call."""
from __future__ import annotations

from pathlib import Path
from typing import Callable, Dict, Optional, Sequence, Union

import torch
from torch import Tensor

from .store import GraphStore, save_splits


def _num_nodes(g) -> int:
    n = getattr(g, "num_nodes", None)
    return int(n) if n is not None else int(g.x.size(0))


def export_dataset(graphs: Sequence, splits: Dict[str, Tensor], graph_properties: Optional[Tensor], out_dir: Union[str, Path]) -> GraphStore:
    """graphs: the list the reference pickles into data.pt (TUDatasets: one Data per graph, y = graph label [1]; Planetoid: ONE
    Data with node labels y [N]); splits: the dict of splits.pt (index vectors, or [2, k] edge lists for the *_LP datasets);
    graph_properties: the tensor of graph_properties.pt or None.  Writes out_dir/{data,splits,graph_properties}.safetensors."""
    if len(graphs) == 0:
        raise ValueError("export_dataset: empty dataset")
    nptr, eptr = [0], [0]
    for g in graphs:
        if g.edge_index.numel() and int(g.edge_index.max()) >= _num_nodes(g):
            raise ValueError("export_dataset: edge ids must be local to their graph (as the reference stores them)")
        nptr.append(nptr[-1] + _num_nodes(g))
        eptr.append(eptr[-1] + int(g.edge_index.size(1)))
    y = None
    if getattr(graphs[0], "y", None) is not None:
        y = torch.cat([g.y.reshape(-1) for g in graphs]).to(torch.long).contiguous()
    gp = None
    if graph_properties is not None:
        gp = torch.as_tensor(graph_properties).to(torch.float32).reshape(len(graphs), -1).contiguous()
    store = GraphStore(torch.cat([g.x for g in graphs]).to(torch.float32).contiguous(),
                       torch.cat([g.edge_index for g in graphs], dim=1).to(torch.long).contiguous(),
                       torch.tensor(nptr, dtype=torch.long), torch.tensor(eptr, dtype=torch.long), y, gp)
    store.save(out_dir)
    save_splits(out_dir, {k: torch.as_tensor(v).to(torch.long) for k, v in splits.items()})
    return store


def _load_plain(path: Path):
    """A .pt file that holds only tensors / dicts / lists of tensors (splits.pt, graph_properties.pt): nothing in it executes."""
    return torch.load(path, weights_only=True)


def export_processed_tree(processed_dir: Union[str, Path], out_root: Union[str, Path],
                          load: Optional[Callable[[Path], object]] = None, allow_pickle: bool = False) -> Dict[str, GraphStore]:
    """Every dataset directory under the reference's data/processed -> out_root/D.

    splits.pt and graph_properties.pt are plain tensors / dicts and are read with torch.load(weights_only=True).  data.pt is a pickled list
    of Data objects: reading it EXECUTES whatever the file contains, so it is never done by default -- pass `allow_pickle=True` (files you
    produced yourself with `python -m src.data.data_setup`, never downloaded ones), or `load=` with a reader of your own for data.pt."""
    if load is None:
        if not allow_pickle:
            raise ValueError("export_processed_tree: data.pt is a pickle of Data objects and unpickling executes code from the file. "
                             "Pass allow_pickle=True only for files you generated yourself, or load=<your reader of data.pt>.")
        load = lambda p: torch.load(p, weights_only=False)          # noqa: E731  (explicit opt-in above)
    out = {}
    for d in sorted(p for p in Path(processed_dir).iterdir() if p.is_dir() and (p / "data.pt").exists()):
        gp = _load_plain(d / "graph_properties.pt") if (d / "graph_properties.pt").exists() else None
        out[d.name] = export_dataset(load(d / "data.pt"), _load_plain(d / "splits.pt"), gp, Path(out_root) / d.name)
    return out
