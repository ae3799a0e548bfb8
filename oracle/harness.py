"""Bridges between the product's containers and the oracle's, for the checkers that compare the two.  TEST INFRASTRUCTURE
(see oracle/__init__.py): imported by tests/, by __graft_entry__.smoke() and by bench.py's cpu_baseline leg only -- it lives
here rather than under tests/ so that the driver's entry points do not depend on test files."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from . import graph_ops as OG, tasks as OTk


def set_dropout(model: nn.Module, p: float) -> None:
    """Parity runs keep train-mode BatchNorm but switch dropout off: GPU and CPU
    dropout masks cannot coincide."""
    for m in model.modules():
        if isinstance(m, nn.Dropout):
            m.p = p
        if hasattr(m, "dropout_p"):
            m.dropout_p = p


def to_oracle(b) -> OG.Batch:
    """gnn_pretraining_amd.graph.Batch -> oracle.graph_ops.Batch (host tensors)."""
    h = b.host()
    return OG.Batch(h.x, h.edge_index, h.batch, h.ptr, torch.tensor(h.edge_ptr_host), h.y, h.graph_properties)


def copy_state(dst: nn.Module, src: nn.Module) -> None:
    dst.load_state_dict({k: v.detach().cpu().clone() for k, v in src.state_dict().items()})


def view_to_oracle(host_batch, v):
    """ViewArrays -> the oracle's Batch of augmented graphs (features gathered and attribute-masked on the host)."""
    rows = torch.from_numpy(np.asarray(v.rows, dtype=np.int64))
    x = host_batch.x[rows].clone()
    if v.rowmask is not None:
        F = x.size(1)
        bits = ((v.rowmask[:, None] >> np.arange(F, dtype=np.uint64)[None, :]) & np.uint64(1)).astype(bool)
        x[torch.from_numpy(bits)] = 0.0
    ptr = torch.from_numpy(np.asarray(v.ptr, dtype=np.int64))
    batch = torch.repeat_interleave(torch.arange(len(ptr) - 1), ptr[1:] - ptr[:-1])
    ei = torch.from_numpy(np.asarray(v.edges, dtype=np.int64))
    eptr = torch.zeros(len(ptr), dtype=torch.long)      # not used by the oracle losses
    return OG.Batch(x, ei, batch, ptr, eptr, host_batch.y, host_batch.graph_properties)


def oracle_artefacts(art, host):
    out = {}
    for t, a in art.items():
        if t in ("node_contrast", "graph_contrast"):
            conv = {}
            for d, views in a.items():
                if views is None:
                    conv[d] = None
                    continue
                b1, b2 = view_to_oracle(host[d], views[0]), view_to_oracle(host[d], views[1])
                c1 = torch.zeros(b1.num_nodes, dtype=torch.bool); c1[torch.from_numpy(views[0].common)] = True
                c2 = torch.zeros(b2.num_nodes, dtype=torch.bool); c2[torch.from_numpy(views[1].common)] = True
                conv[d] = OTk.TwoViews(b1, b2, c1, c2)
            out[t] = conv
        else:
            out[t] = {d: torch.from_numpy(np.asarray(v, dtype=np.int64)) for d, v in a.items()}
    return out


