"""Dataset processing with the logic of src/data/data_setup.py, minus its downloads.

What the reference does per dataset (data_setup.py:75-176) is kept: the train/val(/test) splits with sklearn's
ShuffleSplit / StratifiedShuffleSplit at random_state 42, StandardScaler + clip[-3, 3] of the continuous TUDatasets
fitted on the train graphs, the 12 standardised graph properties, the Planetoid node-classification splits and the
link-prediction edge splits with sampled negatives.  What cannot be kept is the source of the raw graphs
(`TUDataset(root=...)` / `Planetoid(root=...)` fetch from the network): `process_*` take a raw GraphStore, which is
either an export of the real data or a synthetic stand-in of the public size statistics
(`synthetic_tu_store`, `synthetic_planetoid`).  Output: data/processed/{D}/{data,splits,graph_properties}.safetensors.
"""
from __future__ import annotations

import argparse
import os
from pathlib import Path
from typing import Dict, Optional

import numpy as np
import torch
from torch import Tensor

from .. import synthetic
from ..constants import DOMAIN_DIMENSIONS
from ..graph import Data
from .graph_properties import GraphPropertyCalculator, standardize
from .store import GraphStore, save_splits

MIN_SCALE, MAX_SCALE = -3.0, 3.0
RANDOM_SEED = 42
VAL_FRACTION = 0.1
VAL_TEST_FRACTION = 0.2
VAL_TEST_SPLIT_RATIO = 0.5

CONTINUOUS_TUDATASETS = ["PROTEINS", "ENZYMES"]
DOWNSTREAM_TUDATASETS = ["ENZYMES", "PTC_MR"]
PRETRAIN_TUDATASETS = ["MUTAG", "PROTEINS", "NCI1", "ENZYMES"]
TUDATASETS = ["MUTAG", "PROTEINS", "NCI1", "ENZYMES", "PTC_MR"]
PLANETOID_DATASETS = ["Cora", "CiteSeer"]

DATA_ROOT_DIR = Path(os.environ.get("GNNMP_DATA_ROOT", Path(__file__).resolve().parents[2] / "data"))
RAW_DIR = DATA_ROOT_DIR / "raw"
PROCESSED_DIR = DATA_ROOT_DIR / "processed"

# public TUDataset / Planetoid statistics (not in the reference tree):
# graphs, mean nodes, mean undirected edges, classes
TU_STATS = {"MUTAG": (188, 17.9, 19.8, 2), "PROTEINS": (1113, 39.1, 72.8, 2), "NCI1": (4110, 29.9, 32.3, 2),
            "ENZYMES": (600, 32.6, 62.1, 6), "PTC_MR": (344, 14.3, 14.7, 2)}
PLANETOID_STATS = {"Cora": (2708, 5278, 7), "CiteSeer": (3327, 4552, 6)}      # nodes, undirected edges, classes


def processed_dir(name: str, root: Optional[Path] = None) -> Path:
    return (Path(root) if root is not None else PROCESSED_DIR) / name


# ---------------------------------------------------------------------------------------------------------------------
# splits and scaling (data_setup.py:75-127)
# ---------------------------------------------------------------------------------------------------------------------
def pretrain_only_splits(num_graphs: int) -> Dict[str, Tensor]:
    from sklearn.model_selection import ShuffleSplit
    ss = ShuffleSplit(n_splits=1, test_size=VAL_FRACTION, random_state=RANDOM_SEED)
    train_idx, val_idx = next(ss.split(np.arange(num_graphs)))
    return {"train": torch.tensor(train_idx, dtype=torch.long), "val": torch.tensor(val_idx, dtype=torch.long)}


def downstream_splits(labels: np.ndarray) -> Dict[str, Tensor]:
    from sklearn.model_selection import StratifiedShuffleSplit
    n = len(labels)
    s1 = StratifiedShuffleSplit(n_splits=1, test_size=VAL_TEST_FRACTION, random_state=RANDOM_SEED)
    train_idx, val_test_idx = next(s1.split(np.arange(n), labels))
    s2 = StratifiedShuffleSplit(n_splits=1, test_size=VAL_TEST_SPLIT_RATIO, random_state=RANDOM_SEED)
    v, t = next(s2.split(np.arange(len(val_test_idx)), labels[val_test_idx]))
    return {"train": torch.tensor(train_idx, dtype=torch.long), "val": torch.tensor(val_test_idx[v], dtype=torch.long),
            "test": torch.tensor(val_test_idx[t], dtype=torch.long)}


def standardize_clip_features(store: GraphStore, train_idx: np.ndarray) -> None:
    """In place: scaler fitted on the node rows of the train graphs, applied to every graph, clipped (:93-100)."""
    nptr = store.node_ptr.numpy()
    rows = np.concatenate([np.arange(nptr[i], nptr[i + 1]) for i in train_idx])
    x = store.x.numpy().astype(np.float64)
    store.x = torch.from_numpy(np.clip(standardize(x, rows), MIN_SCALE, MAX_SCALE)).to(store.x.dtype)


def process_tu_store(name: str, store: GraphStore, root: Optional[Path] = None) -> Dict[str, Tensor]:
    """process_tudatasets (:75-127) for one dataset whose raw graphs are already in `store`."""
    needs_pretrain, needs_downstream = name in PRETRAIN_TUDATASETS, name in DOWNSTREAM_TUDATASETS
    if needs_downstream:
        splits = downstream_splits(store.y.numpy())
        if name in CONTINUOUS_TUDATASETS:
            standardize_clip_features(store, splits["train"].numpy())
    else:
        splits = pretrain_only_splits(len(store))
    if needs_pretrain:
        graphs = [store.graph(i) for i in range(len(store))]
        store.graph_properties = GraphPropertyCalculator().compute_and_standardize_for_dataset(graphs, splits["train"].numpy())
    store.save(processed_dir(name, root))
    save_splits(processed_dir(name, root), splits)
    return splits


# ---------------------------------------------------------------------------------------------------------------------
# Planetoid (data_setup.py:130-176)
# ---------------------------------------------------------------------------------------------------------------------
def sample_non_edges(edge_index: Tensor, num_nodes: int, count: int, generator: torch.Generator) -> Tensor:
    """`count` distinct ordered pairs (s != d) that are not columns of `edge_index` -- the role of
    torch_geometric.utils.negative_sampling at data_setup.py:137-141 (which draws from the global RNG, so its exact
    picks are not reproducible; the distribution -- uniform over non-edges -- is)."""
    n = num_nodes
    key = torch.unique(edge_index[0] * n + edge_index[1])
    picked = torch.empty(0, dtype=torch.long)
    while picked.numel() < count:
        cand = torch.randint(0, n * n, (2 * (count - picked.numel()) + 16,), generator=generator)
        cand = cand[(cand // n) != (cand % n)]
        pos = torch.searchsorted(key, cand).clamp_(max=max(key.numel() - 1, 0))
        if key.numel():
            cand = cand[key[pos] != cand]
        merged = torch.cat([picked, cand])
        uniq, first = np.unique(merged.numpy(), return_index=True)            # keep draw order
        picked = merged[torch.from_numpy(np.sort(first))]
    picked = picked[:count]
    return torch.stack([picked // n, picked % n])


def to_undirected_host(edge_index: Tensor, num_nodes: int) -> Tensor:
    """torch_geometric.utils.to_undirected on the host: both directions, duplicates merged, sorted by (src, dst)."""
    s, d = torch.cat([edge_index[0], edge_index[1]]), torch.cat([edge_index[1], edge_index[0]])
    key = torch.unique(s * num_nodes + d)
    return torch.stack([key // num_nodes, key % num_nodes])


def create_link_prediction_splits(data: Data) -> Dict[str, Tensor]:
    g = torch.Generator()
    g.manual_seed(RANDOM_SEED)
    num_edges = data.num_edges
    num_val_test = int(num_edges * VAL_TEST_FRACTION)
    num_val = int(num_val_test * VAL_TEST_SPLIT_RATIO)
    perm = torch.randperm(num_edges, generator=g)
    train_edges = data.edge_index[:, perm[num_val_test:]]
    val_test_edges = data.edge_index[:, perm[:num_val_test]]
    neg = sample_non_edges(to_undirected_host(train_edges, data.num_nodes), data.num_nodes, num_val_test, g)
    return {"train_pos": train_edges, "val_pos": val_test_edges[:, :num_val], "val_neg": neg[:, :num_val],
            "test_pos": val_test_edges[:, num_val:], "test_neg": neg[:, num_val:]}


def process_planetoid_store(name: str, store: GraphStore, nc_splits: Dict[str, Tensor], root: Optional[Path] = None) -> None:
    """process_planetoid_datasets (:152-167): the same graph saved twice, with node and with edge splits."""
    store.save(processed_dir(f"{name}_NC", root))
    save_splits(processed_dir(f"{name}_NC", root), nc_splits)
    store.save(processed_dir(f"{name}_LP", root))
    save_splits(processed_dir(f"{name}_LP", root), create_link_prediction_splits(store.graph(0)))


# ---------------------------------------------------------------------------------------------------------------------
# synthetic raw data (stand-ins for the downloads)
# ---------------------------------------------------------------------------------------------------------------------
def synthetic_tu_store(name: str, gen: torch.Generator, num_graphs: Optional[int] = None) -> GraphStore:
    """Raw (unscaled) TUDataset stand-in: public graph count / sizes, one-hot node labels for the discrete sets,
    per-column N(mu_j, sigma_j) attributes for the continuous ones (so the scaler has something to do)."""
    count, mean_nodes, mean_edges, classes = TU_STATS[name]
    dim = DOMAIN_DIMENSIONS[name]
    graphs = [synthetic.random_graph(gen, dim, mean_nodes, mean_edges, classes) for _ in range(num_graphs or count)]
    if name in CONTINUOUS_TUDATASETS:
        mu, sigma = torch.randn(dim, generator=gen) * 5.0, torch.rand(dim, generator=gen) * 4.0 + 0.5
        for g in graphs:
            g.x = torch.randn(g.num_nodes, dim, generator=gen) * sigma + mu
    else:
        for g in graphs:
            hot = torch.randint(0, dim, (g.num_nodes,), generator=gen)
            g.x = torch.nn.functional.one_hot(hot, dim).float()
    labels = (torch.arange(len(graphs)) % classes)[torch.randperm(len(graphs), generator=gen)]     # balanced classes
    for g, y in zip(graphs, labels):
        g.graph_properties, g.y = None, y.reshape(1)
    return GraphStore.from_data_list(graphs)


def synthetic_planetoid(name: str, gen: torch.Generator, num_nodes: Optional[int] = None):
    nodes, edges, classes = PLANETOID_STATS[name]
    if num_nodes is not None:
        edges = max(8, edges * num_nodes // nodes)
        nodes = num_nodes
    g = synthetic.cora_like(gen, nodes, edges, DOMAIN_DIMENSIONS[f"{name}_NC"], num_classes=classes)
    perm = torch.randperm(nodes, generator=gen)
    k = min(20 * classes, nodes // 4)                      # Planetoid public split: 20 per class / 500 / 1000
    v = min(500, nodes // 4)
    t = min(1000, nodes - k - v)
    splits = {"train": perm[:k].sort().values, "val": perm[k:k + v].sort().values, "test": perm[k + v:k + v + t].sort().values}
    return GraphStore.from_data_list([g]), splits


def process_synthetic(root: Optional[Path] = None, seed: int = RANDOM_SEED, scale: float = 1.0) -> None:
    gen = torch.Generator()
    gen.manual_seed(seed)
    for name in TUDATASETS:
        process_tu_store(name, synthetic_tu_store(name, gen, max(120, int(TU_STATS[name][0] * scale))), root)
    for name in PLANETOID_DATASETS:
        store, splits = synthetic_planetoid(name, gen, None if scale >= 1.0 else max(200, int(PLANETOID_STATS[name][0] * scale)))
        process_planetoid_store(name, store, splits, root)


def ensure_processed(names, root: Optional[Path] = None, scale: float = 1.0) -> None:
    """Create the synthetic stand-ins if data/processed lacks any of `names` (the CLIs call this; real data, once
    exported, is left alone)."""
    from .store import DATA_FILE
    if all((processed_dir(n, root) / DATA_FILE).exists() for n in names):
        return
    process_synthetic(root, scale=scale)


def main() -> None:
    p = argparse.ArgumentParser(description="write data/processed/* (synthetic stand-ins; there is no network)")
    p.add_argument("--root", type=str, default=None)
    p.add_argument("--scale", type=float, default=1.0, help="fraction of the public dataset sizes to generate")
    a = p.parse_args()
    process_synthetic(Path(a.root) if a.root else None, scale=a.scale)
    print(f"wrote {Path(a.root) if a.root else PROCESSED_DIR}")


if __name__ == "__main__":
    main()
