"""Pre-training tasks on libgnnmp.  Same classes and ``compute_loss(domain_batches, generator)``
contract as src/pretrain/tasks.py:61-343, restructured so that every task is

    draw(domain_batches, generator) -> artefacts      (all RNG, on the host, reference draw order)
    loss(domain_batches, artefacts) -> (total, per_domain)

which lets parity tests feed this path and the CPU oracle identical artefacts, and lets a training
loop prepare step t+1's artefacts while the GPU still runs step t.  No ``.item()`` / host sync occurs
inside ``loss``: sizes are known on the host from the artefacts.
"""
from __future__ import annotations

from abc import ABC
from typing import Dict, List, NamedTuple, Tuple

import numpy as np
import torch
from torch import Tensor

from .. import operators as O
from ..constants import GRAPH_PROPERTY_DIM
from ..graph import Batch
from ..models.gnn import GNN_HIDDEN_DIM
from ..models.pretrain_model import PretrainableGNN, draw_mask_indices
from .augmentations import GraphAugmentor


class TwoViews(NamedTuple):
    v1: Batch
    v2: Batch
    common1: Tensor     # int64 row ids of v1 that survive in both views (host)
    common2: Tensor


def negative_sampling_local(ei: np.ndarray, n: int, num_neg: int, rng) -> np.ndarray:
    """torch_geometric.utils.negative_sampling(edge_index, n, num_neg) for ONE graph in local numbering ("sparse" method,
    not bipartite, force_undirected=False; PyG >= 2.3, requirements.txt:3).  `ei` = to_undirected edges [2, E] without
    duplicates.  Index space: the n (n - 1) ordered pairs (i, j), i != j (self loops are never sampled); when the over-sample
    size int(1.1 * num_neg / prob) reaches that population every non-edge is returned, in index order, with no random draw;
    otherwise `rng.sample(range(population), size)` -- Python's random, as PyG -- up to three times.  -> [2, k] int64."""
    pop = n * n - n
    row, col = ei[0], ei[1]
    keep = row != col
    row, col = row[keep], col[keep]
    idx = row * (n - 1) + (col - (row < col))
    if idx.size >= pop:
        return np.zeros((2, 0), dtype=np.int64)
    prob = 1.0 - idx.size / pop
    size = int(1.1 * num_neg / prob)
    neg = None
    for _ in range(3):
        rnd = np.arange(pop, dtype=np.int64) if pop <= size else np.asarray(rng.sample(range(pop), size), dtype=np.int64)
        m = np.isin(rnd, idx)
        if neg is not None:
            m |= np.isin(rnd, neg)
        rnd = rnd[~m]
        neg = rnd if neg is None else np.concatenate([neg, rnd])
        if neg.size >= num_neg:
            neg = neg[:num_neg]
            break
    r = neg // (n - 1)
    c = neg % (n - 1)
    c = c + (r <= c)
    return np.stack([r, c])


def sample_negative_edges(batch: Batch, rng=None) -> Tensor:
    """The reference's negatives (tasks.py:105-110): batched_negative_sampling(to_undirected(pos_edges), batch.batch,
    num_neg_samples=pos_edges.size(1)) -- PyG applies num_neg_samples to every graph of the batch on its own, so graph i gives
    min(E_batch, n_i (n_i - 1) - E_i) negatives (for TUDataset-sized graphs: ALL its non-edges, ~8x its positives at 8 graphs per
    batch).  Drawn from Python's `random` (`rng`: anything with .sample; default the global module, as PyG does), NOT from the
    shared torch generator.  Restated from PyG's published source; parity unpinned (no fixture in the reference)."""
    import random as _random
    rng = rng or _random
    host = batch.host()
    ei = host.edge_index.numpy()
    num_neg = int(ei.shape[1])
    outs = []
    for g in range(host.num_graphs):
        s, e = host.ptr_host[g], host.ptr_host[g + 1]
        es, ee = host.edge_ptr_host[g], host.edge_ptr_host[g + 1]
        n = e - s
        if n < 2:
            continue
        loc = ei[:, es:ee] - s
        und = np.unique(np.concatenate([loc[0] * n + loc[1], loc[1] * n + loc[0]]))      # to_undirected: both directions, coalesced
        neg = negative_sampling_local(np.stack([und // n, und % n]), n, num_neg, rng)
        if neg.shape[1]:
            outs.append(neg + s)
    if not outs:
        return torch.empty(2, 0, dtype=torch.long)
    return torch.from_numpy(np.concatenate(outs, axis=1))


def _scalar0(device) -> Tensor:
    return torch.zeros((), device=device)


class BasePretrainTask(ABC):
    name = ""

    def __init__(self, model: PretrainableGNN) -> None:
        self.model = model

    def draw(self, domain_batches: Dict[str, Batch], generator: torch.Generator):
        return None

    def loss(self, domain_batches: Dict[str, Batch], artefacts) -> Tuple[Tensor, Dict[str, Tensor]]:
        raise NotImplementedError

    def compute_loss(self, domain_batches: Dict[str, Batch], generator: torch.Generator) -> Tuple[Tensor, Dict[str, Tensor]]:
        return self.loss(domain_batches, self.draw(domain_batches, generator))


def _pool(total: Tensor, size: int) -> Tensor:
    return total / size if size > 0 else total


class NodeFeatureMaskingTask(BasePretrainTask):
    """tasks.py:69-94.  Encoder under no_grad -> mask rows with the token -> backbone -> MLP on the
    masked rows -> MSE(sum) against the pre-mask encoder output; total = sum / sum(M*256)."""
    name = "node_feat_mask"

    def draw(self, domain_batches, generator):
        return {d: draw_mask_indices(b.ptr_host, generator) for d, b in domain_batches.items()}

    def loss(self, domain_batches, mask_idx):
        dev = self.model.device
        total, size, per = _scalar0(dev), 0, {}
        for d, b in domain_batches.items():
            masked, idx, target = self.model.mask_with_indices(b, d, mask_idx[d])
            m = idx.numel()
            if m == 0:
                per[d] = _scalar0(dev)
                continue
            h = self.model.forward_with_h0(masked, b.edge_index)
            rec = self.model.get_head(self.name, d)(O.take_rows(h, idx))
            l = O.mse_loss_sum(rec, target)
            total = total + l
            size += m * GNN_HIDDEN_DIM
            per[d] = l / (m * GNN_HIDDEN_DIM)
        return _pool(total, size), per


class LinkPredictionTask(BasePretrainTask):
    """tasks.py:96-127.  Scores positives + sampled negatives with the edge MLP; BCE(sum) on the
    sigmoid outputs (torch's -100 log clamp); total = sum / sum(K)."""
    name = "link_pred"

    py_rng = None        # the Python-random stream PyG's sampler draws from; None = the global `random` module, as in the reference

    def draw(self, domain_batches, generator, rng=None):
        """(the shared torch generator is NOT advanced: PyG's sampler draws from Python's `random`)"""
        return {d: sample_negative_edges(b, rng or self.py_rng) for d, b in domain_batches.items()}

    def loss(self, domain_batches, neg_edges):
        dev = self.model.device
        total, size, per = _scalar0(dev), 0, {}
        decoder = self.model.get_head(self.name)
        for d, b in domain_batches.items():
            neg = neg_edges[d].to(dev)
            npos, nneg = b.edge_index.size(1), neg.size(1)
            edges = torch.cat([b.edge_index, neg], dim=1)
            labels = torch.zeros(npos + nneg, device=dev)
            labels[:npos] = 1.0
            probs = decoder(self.model(b, d), edges)
            l = O.binary_cross_entropy_sum(probs, labels)
            total = total + l
            size += npos + nneg
            per[d] = l / (npos + nneg)
        return total / size, per


def _views(batch: Batch, generator: torch.Generator) -> TwoViews:
    v1, v2, m1, m2 = GraphAugmentor.create_two_views(batch, generator)
    # per-graph h[batch==g][mask_g] concatenated over g == one row selection (batch is sorted)
    return TwoViews(v1, v2, torch.cat(m1).nonzero().squeeze(1), torch.cat(m2).nonzero().squeeze(1))


class NodeContrastiveTask(BasePretrainTask):
    """tasks.py:130-213."""
    name = "node_contrast"

    def __init__(self, model, temperature_scheduler) -> None:
        super().__init__(model)
        self.temperature_scheduler = temperature_scheduler

    def draw(self, domain_batches, generator):
        return {d: _views(b, generator) for d, b in domain_batches.items()}

    def loss(self, domain_batches, views):
        dev = self.model.device
        total, size, per = _scalar0(dev), 0, {}
        t = self.temperature_scheduler()
        for d, v in views.items():
            h1 = self.model(v.v1, d)
            h2 = self.model(v.v2, d)
            if v.common1.numel() < 2 or v.common2.numel() < 2:
                per[d] = _scalar0(dev)
                continue
            proj = self.model.get_head(self.name, d)
            z1 = proj(O.take_rows(h1, v.common1.to(dev)))
            z2 = proj(O.take_rows(h2, v.common2.to(dev)))
            l, n = O.nt_xent(z1, z2, t)
            total = total + l
            size += n
            per[d] = l / n
        return _pool(total, size), per


class GraphContrastiveTask(BasePretrainTask):
    """tasks.py:216-287: mean || max read-out of each view -> projection -> NT-Xent."""
    name = "graph_contrast"

    def __init__(self, model, temperature_scheduler=None) -> None:
        super().__init__(model)
        self.temperature_scheduler = temperature_scheduler

    def draw(self, domain_batches, generator):
        # a domain with < 2 graphs is skipped before create_two_views and draws nothing (tasks.py:233-238)
        return {d: (_views(b, generator) if b.num_graphs >= 2 else None) for d, b in domain_batches.items()}

    def loss(self, domain_batches, views):
        dev = self.model.device
        total, size, per = _scalar0(dev), 0, {}
        t = self.temperature_scheduler()
        for d, v in views.items():
            if v is None:
                per[d] = _scalar0(dev)
                continue
            s = []
            for vb in (v.v1, v.v2):
                h = self.model(vb, d)
                s.append(torch.cat([O.global_mean_pool(h, vb.batch, ptr32=vb.ptr32),
                                    O.global_max_pool(h, vb.batch, ptr32=vb.ptr32)], dim=1))
            proj = self.model.get_head(self.name, d)
            l, n = O.nt_xent(proj(s[0]), proj(s[1]), t)
            total = total + l
            size += n
            per[d] = l / n
        return _pool(total, size), per


class GraphPropertyPredictionTask(BasePretrainTask):
    """tasks.py:290-312."""
    name = "graph_prop"

    def loss(self, domain_batches, _=None):
        dev = self.model.device
        total, size, per = _scalar0(dev), 0, {}
        for d, b in domain_batches.items():
            emb = O.global_mean_pool(self.model(b, d), b.batch, ptr32=b.ptr32)
            pred = self.model.get_head(self.name, d)(emb)
            labels = b.graph_properties.to(torch.float32).to(dev).view(b.num_graphs, GRAPH_PROPERTY_DIM)
            l = O.mse_loss_sum(pred, labels)
            n = b.num_graphs * GRAPH_PROPERTY_DIM
            total = total + l
            size += n
            per[d] = l / n
        return total / size, per


class DomainAdversarialTask(BasePretrainTask):
    """tasks.py:315-343 (scheme s5 only)."""
    name = "domain_adv"

    def __init__(self, model, grl_scheduler=None) -> None:
        super().__init__(model)
        self.domain_to_idx = {n: i for i, n in enumerate(self.model.input_encoders.keys())}
        self.grl_scheduler = grl_scheduler

    def loss(self, domain_batches, _=None):
        dev = self.model.device
        total, size, per = _scalar0(dev), 0, {}
        lam = self.grl_scheduler() if self.grl_scheduler is not None else 0.0
        for d, b in domain_batches.items():
            emb = O.global_mean_pool(self.model(b, d), b.batch, ptr32=b.ptr32)
            logits = self.model.get_head(self.name)(emb, lam)
            labels = torch.full((b.num_graphs,), self.domain_to_idx[d], device=dev, dtype=torch.long)
            l = O.cross_entropy_sum(logits, labels)
            total = total + l
            size += b.num_graphs
            per[d] = l / b.num_graphs
        return total / size, per


def instantiate_tasks(model, active_tasks: List[str], grl_scheduler, temperature_scheduler) -> Dict[str, BasePretrainTask]:
    """src/pretrain/pretrain.py:77-93."""
    out: Dict[str, BasePretrainTask] = {}
    for n in active_tasks:
        if n == "node_feat_mask":
            out[n] = NodeFeatureMaskingTask(model)
        elif n == "link_pred":
            out[n] = LinkPredictionTask(model)
        elif n == "node_contrast":
            out[n] = NodeContrastiveTask(model, temperature_scheduler)
        elif n == "graph_contrast":
            out[n] = GraphContrastiveTask(model, temperature_scheduler)
        elif n == "graph_prop":
            out[n] = GraphPropertyPredictionTask(model)
        elif n == "domain_adv":
            out[n] = DomainAdversarialTask(model, grl_scheduler)
    return out
