"""TEST INFRASTRUCTURE (see oracle/__init__.py) -- CPU restatement of src/data/graph_properties.py:17-96 on networkx,
the library the reference itself calls (networkx is importable in this image; torch_geometric's remove_self_loops /
to_undirected / to_networkx are restated by building the simple undirected nx.Graph directly).

Pinning: the reference holds no fixture for these values ("parity unpinned" by the reference); the restatement calls
the same networkx functions in the same order, and tests/test_data.py additionally pins closed-form cases (path, cycle,
star, complete graph, two triangles)."""
from __future__ import annotations

import math

import networkx as nx
import numpy as np

GRAPH_PROPERTY_DIM = 12


def simple_graph(edge_index: np.ndarray, num_nodes: int) -> nx.Graph:
    g = nx.Graph()
    g.add_nodes_from(range(num_nodes))                                   # to_networkx keeps isolated nodes
    g.add_edges_from((int(s), int(d)) for s, d in zip(edge_index[0], edge_index[1]) if s != d)   # :21-22
    return g


def graph_properties(edge_index: np.ndarray, num_nodes: int) -> np.ndarray:
    G = simple_graph(edge_index, num_nodes)                              # :21-24
    N, E = G.number_of_nodes(), G.number_of_edges()                      # :26-27
    degrees = np.array(list(dict(G.degree()).values()), dtype=float)     # :29-30
    deg_mean, deg_var, deg_max = float(degrees.mean()), float(degrees.var()), float(degrees.max())   # :31-33
    density = float(nx.density(G))                                       # :35
    clustering_global = float(nx.average_clustering(G))                  # :37
    transitivity = float(nx.transitivity(G)) if N > 2 else 0.0           # :38
    num_components = float(nx.number_connected_components(G))            # :40
    try:                                                                 # :42-47
        components = [G.subgraph(c).copy() for c in nx.connected_components(G)]
        H = max(components, key=lambda g: g.number_of_nodes())
        diameter = float(nx.diameter(H))
    except (nx.NetworkXError, ValueError):
        diameter = 0.0
    if deg_var == 0.0:                                                   # :49-54
        assortativity = 0.0
    else:
        with np.errstate(all="ignore"):
            assortativity = float(nx.degree_assortativity_coefficient(G))
        if math.isnan(assortativity) or math.isinf(assortativity):
            assortativity = 0.0
    if N > 2:                                                            # :56-61
        degree_centralization = float((degrees.max() - degrees).sum()) / float((N - 1) * (N - 2))
    else:
        degree_centralization = 0.0
    return np.array([float(N), float(E), density, deg_mean, deg_var, deg_max, clustering_global, transitivity,
                     num_components, diameter, assortativity, degree_centralization], dtype=np.float32)   # :62-77


def standardize(all_props: np.ndarray, train_idx: np.ndarray) -> np.ndarray:
    """:88-96 with sklearn's StandardScaler, as the reference does."""
    from sklearn.preprocessing import StandardScaler
    scaler = StandardScaler()
    scaler.fit(all_props[train_idx])
    scaler.scale_[scaler.scale_ == 0] = 1.0
    return scaler.transform(all_props).astype(np.float32)
