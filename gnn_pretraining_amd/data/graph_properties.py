"""The 12 structural targets of the graph-property task (src/data/graph_properties.py:17-96), computed from the
sparse adjacency with numpy / scipy instead of a networkx object per graph.

Order (graph_properties.py:62-75): N, E, density, mean degree, degree variance, max degree, average clustering,
transitivity, #components, diameter of the largest component, degree assortativity, degree centralisation -- on the
simple undirected graph (self loops removed, both directions merged, graph_properties.py:21-24).
tests/test_data.py holds this against oracle/graph_properties.py (the networkx restatement of the reference)."""
from __future__ import annotations

import math
from typing import Sequence

import numpy as np
import scipy.sparse as sp
import torch
from scipy.sparse import csgraph
from torch import Tensor

from ..graph import Data

GRAPH_PROPERTY_DIM = 12


def _simple_adjacency(edge_index: np.ndarray, n: int) -> sp.csr_matrix:
    s, d = edge_index[0], edge_index[1]
    keep = s != d
    s, d = s[keep], d[keep]
    a = sp.coo_matrix((np.ones(2 * s.size, dtype=np.float64), (np.concatenate([s, d]), np.concatenate([d, s]))),
                      shape=(n, n)).tocsr()
    a.data[:] = 1.0              # duplicates were summed by tocsr(); the graph is simple
    return a


class GraphPropertyCalculator:
    def __call__(self, graph: Data) -> Tensor:
        n = int(graph.num_nodes)
        a = _simple_adjacency(graph.edge_index.cpu().numpy().astype(np.int64), n)
        deg = np.asarray(a.sum(axis=1)).reshape(-1)
        e = float(deg.sum() / 2.0)
        deg_mean, deg_var, deg_max = float(deg.mean()), float(deg.var()), float(deg.max())
        density = 0.0 if (n <= 1 or e == 0) else 2.0 * e / (n * (n - 1))

        # t2[v] = 2 * (#triangles through v) = sum over neighbours of the common-neighbour count
        t2 = np.asarray((a @ a).multiply(a).sum(axis=1)).reshape(-1)
        pairs = deg * (deg - 1.0)
        local = np.divide(t2, pairs, out=np.zeros_like(t2), where=pairs > 0)
        clustering = float(local.mean())
        transitivity = 0.0
        if n > 2 and t2.sum() > 0:
            transitivity = float(t2.sum() / pairs.sum())

        ncomp, label = csgraph.connected_components(a, directed=False)
        sizes = np.bincount(label, minlength=ncomp)
        # largest component; ties go to the one holding the smallest node id (networkx yields components in that order)
        first = np.full(ncomp, n, dtype=np.int64)
        np.minimum.at(first, label, np.arange(n))
        big = min(range(ncomp), key=lambda c: (-sizes[c], first[c]))
        members = np.nonzero(label == big)[0]
        if members.size <= 1:
            diameter = 0.0
        else:
            dist = csgraph.shortest_path(a[members][:, members], method="D", unweighted=True)
            diameter = float(dist.max())

        assort = 0.0
        if deg_var != 0.0 and e > 0:
            coo = a.tocoo()
            x, y = deg[coo.row], deg[coo.col]          # every undirected edge appears in both directions
            vx = float((x * x).mean() - x.mean() ** 2)
            if vx > 0:
                assort = float(((x * y).mean() - x.mean() * y.mean()) / vx)
            if math.isnan(assort) or math.isinf(assort):
                assort = 0.0

        central = float((deg_max - deg).sum() / ((n - 1) * (n - 2))) if n > 2 else 0.0
        return torch.tensor([float(n), e, density, deg_mean, deg_var, deg_max, clustering, transitivity, float(ncomp),
                             diameter, assort, central], dtype=torch.float32)

    def compute_for_dataset(self, dataset_list: Sequence[Data]) -> Tensor:
        out = torch.zeros((len(dataset_list), GRAPH_PROPERTY_DIM), dtype=torch.float32)
        for i, g in enumerate(dataset_list):
            out[i] = self(g)
        return out

    def compute_and_standardize_for_dataset(self, dataset_list: Sequence[Data], train_idx: np.ndarray) -> Tensor:
        """graph_properties.py:88-96: StandardScaler fitted on the train split (population std; zero scale -> 1)."""
        props = self.compute_for_dataset(dataset_list).numpy().astype(np.float64)
        return torch.from_numpy(standardize(props, np.asarray(train_idx))).float()


def standardize(values: np.ndarray, fit_rows: np.ndarray) -> np.ndarray:
    """sklearn StandardScaler().fit(values[fit_rows]).transform(values) with scale_==0 replaced by 1."""
    ref = values[fit_rows]
    mean, scale = ref.mean(axis=0), ref.std(axis=0)
    scale[scale == 0] = 1.0
    return (values - mean) / scale
