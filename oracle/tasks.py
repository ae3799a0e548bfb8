"""Oracle restatement of src/pretrain/tasks.py.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Every task is split into
``draw`` (consume the RNG, produce index artefacts) and ``loss`` (pure function
of model, batches and artefacts) so that parity tests can feed the HIP path
and this oracle the *same* artefacts (SURVEY.md section 7 "hard parts").
``compute_loss(domain_batches, generator)`` composes the two in the reference's
order and has the reference signature (tasks.py:66).
"""
from __future__ import annotations

from typing import Dict, List, NamedTuple, Optional, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor

from . import graph_ops as G
from .augment import create_two_views, sample_negative_edges
from .models import GRAPH_PROP_DIM, HIDDEN, PretrainableGNN


class TwoViews(NamedTuple):
    v1: G.Batch
    v2: G.Batch
    common1: Tensor      # bool over v1 nodes (concatenated per-graph masks)
    common2: Tensor


def _views(batch, gen) -> TwoViews:
    v1, v2, m1, m2 = create_two_views(batch, gen)
    return TwoViews(v1, v2, torch.cat(m1), torch.cat(m2))


def nt_xent(z1: Tensor, z2: Tensor, temperature: float) -> Tuple[Tensor, int]:
    """tasks.py:192-213 == 265-287.  normalize(eps 1e-12) -> z z^T / T -> diagonal
    -inf -> cross_entropy(sum) against the other view's row."""
    n = z1.size(0)
    z = torch.cat([F.normalize(z1, dim=1), F.normalize(z2, dim=1)], dim=0)
    sim = (z @ z.t()) / temperature
    sim = sim.masked_fill(torch.eye(2 * n, dtype=torch.bool), float("-inf"))
    target = torch.cat([torch.arange(n, 2 * n), torch.arange(0, n)])
    return F.cross_entropy(sim, target, reduction="sum"), 2 * n


def _finish(total: Tensor, size: int) -> Tensor:
    return total / size if size > 0 else total


# ----------------------------------------------------------------- NFM (a7) --
def nfm_draw(model: PretrainableGNN, batches, gen) -> Dict[str, Tensor]:
    return {d: model.draw_mask_indices(b.ptr, gen) for d, b in batches.items()}


def nfm_loss(model: PretrainableGNN, batches, mask_idx: Dict[str, Tensor]):
    """tasks.py:70-94."""
    total, size, per = torch.tensor(0.0), 0, {}
    for d, b in batches.items():
        masked, idx, target = model.mask_with_indices(b, d, mask_idx[d])
        if idx.numel() == 0:
            per[d] = torch.tensor(0.0)
            continue
        h = model.forward_with_h0(masked, b.edge_index)
        rec = model.get_head("node_feat_mask", d)(h[idx])
        l = F.mse_loss(rec, target, reduction="sum")
        total = total + l
        size += idx.numel() * HIDDEN
        per[d] = l / (idx.numel() * HIDDEN)
    return _finish(total, size), per


# ------------------------------------------------------------------ LP (a9) --
def lp_draw(batches, gen=None, rng=None) -> Dict[str, Tensor]:
    """PyG's negative sampler draws from Python's `random` (rng; default the global module), never from the torch generator."""
    return {d: sample_negative_edges(b, rng) for d, b in batches.items()}


def lp_loss(model: PretrainableGNN, batches, neg_edges: Dict[str, Tensor]):
    """tasks.py:97-127.  F.binary_cross_entropy on probabilities (log clamped at -100)."""
    total, size, per = torch.tensor(0.0), 0, {}
    dec = model.get_head("link_pred")
    for d, b in batches.items():
        pos, neg = b.edge_index, neg_edges[d]
        edges = torch.cat([pos, neg], dim=1)
        labels = torch.cat([torch.ones(pos.size(1)), torch.zeros(neg.size(1))])
        probs = dec(model(b, d), edges)
        l = F.binary_cross_entropy(probs, labels.to(probs.dtype), reduction="sum")      # (fp64 runs of the oracle: tests' arbiter)
        total = total + l
        size += labels.numel()
        per[d] = l / labels.numel()
    return total / size, per


# ----------------------------------------------------------------- NC (a11) --
def nc_draw(batches, gen) -> Dict[str, TwoViews]:
    return {d: _views(b, gen) for d, b in batches.items()}


def nc_loss(model: PretrainableGNN, views: Dict[str, TwoViews], temperature: float):
    """tasks.py:136-190.  Per-graph ``h[batch==g][mask_g]`` concatenated over g is
    one boolean-mask row selection because ``batch`` is sorted."""
    total, size, per = torch.tensor(0.0), 0, {}
    for d, v in views.items():
        h1 = model(v.v1, d)
        h2 = model(v.v2, d)
        c1, c2 = h1[v.common1], h2[v.common2]
        if c1.size(0) < 2 or c2.size(0) < 2:
            per[d] = torch.tensor(0.0)
            continue
        proj = model.get_head("node_contrast", d)
        l, s = nt_xent(proj(c1), proj(c2), temperature)
        total = total + l
        size += s
        per[d] = l / s
    return _finish(total, size), per


# ----------------------------------------------------------------- GC (a12) --
def gc_draw(batches, gen) -> Dict[str, Optional[TwoViews]]:
    """tasks.py:233-238: a domain with fewer than 2 graphs is skipped *before*
    create_two_views, i.e. it draws nothing."""
    return {d: (_views(b, gen) if b.num_graphs >= 2 else None) for d, b in batches.items()}


def gc_loss(model: PretrainableGNN, views: Dict[str, Optional[TwoViews]], temperature: float):
    """tasks.py:222-262."""
    total, size, per = torch.tensor(0.0), 0, {}
    for d, v in views.items():
        if v is None:
            per[d] = torch.tensor(0.0)
            continue
        s = []
        for vb in (v.v1, v.v2):
            h = model(vb, d)
            s.append(torch.cat([G.global_mean_pool(h, vb.batch), G.global_max_pool(h, vb.batch)], dim=1))
        proj = model.get_head("graph_contrast", d)
        l, n = nt_xent(proj(s[0]), proj(s[1]), temperature)
        total = total + l
        size += n
        per[d] = l / n
    return _finish(total, size), per


# ----------------------------------------------------------------- GP (a13) --
def gp_loss(model: PretrainableGNN, batches):
    """tasks.py:291-312."""
    total, size, per = torch.tensor(0.0), 0, {}
    for d, b in batches.items():
        emb = G.global_mean_pool(model(b, d), b.batch)
        pred = model.get_head("graph_prop", d)(emb)
        labels = b.graph_properties.to(torch.float32).view(emb.size(0), GRAPH_PROP_DIM)
        l = F.mse_loss(pred, labels.to(pred.dtype), reduction="sum")
        n = emb.size(0) * GRAPH_PROP_DIM
        total = total + l
        size += n
        per[d] = l / n
    return total / size, per


# ----------------------------------------------------------------- DA (a15) --
def da_loss(model: PretrainableGNN, batches, lam: float):
    """tasks.py:315-343 (s5 only)."""
    total, size, per = torch.tensor(0.0), 0, {}
    index = {name: i for i, name in enumerate(model.input_encoders.keys())}
    for d, b in batches.items():
        emb = G.global_mean_pool(model(b, d), b.batch)
        logits = model.get_head("domain_adv")(emb, lam)
        labels = torch.full((emb.size(0),), index[d], dtype=torch.long)
        l = F.cross_entropy(logits, labels, reduction="sum")
        total = total + l
        size += labels.numel()
        per[d] = l / labels.numel()
    return total / size, per


# ------------------------------------------------------ reference-shaped API --
class Task:
    """Mirror of BasePretrainTask (tasks.py:61-67): compute_loss(domain_batches, generator)."""

    def __init__(self, name: str, model: PretrainableGNN, temperature=None, grl=None) -> None:
        self.name, self.model, self.temperature, self.grl = name, model, temperature, grl
        self.last_artifacts = None
        self.py_rng = None          # link_pred: the Python-random stream PyG's sampler draws from (None: the global `random` module)

    def draw(self, batches, gen):
        n = self.name
        if n == "node_feat_mask":
            return nfm_draw(self.model, batches, gen)
        if n == "link_pred":
            return lp_draw(batches, gen, self.py_rng)
        if n == "node_contrast":
            return nc_draw(batches, gen)
        if n == "graph_contrast":
            return gc_draw(batches, gen)
        return None

    def loss(self, batches, art):
        n, m = self.name, self.model
        if n == "node_feat_mask":
            return nfm_loss(m, batches, art)
        if n == "link_pred":
            return lp_loss(m, batches, art)
        if n == "node_contrast":
            return nc_loss(m, art, self.temperature())
        if n == "graph_contrast":
            return gc_loss(m, art, self.temperature())
        if n == "graph_prop":
            return gp_loss(m, batches)
        if n == "domain_adv":
            return da_loss(m, batches, self.grl() if self.grl is not None else 0.0)
        raise KeyError(n)

    def compute_loss(self, batches, gen):
        self.last_artifacts = self.draw(batches, gen)
        return self.loss(batches, self.last_artifacts)


def instantiate_tasks(model, active: List[str], grl, temperature) -> Dict[str, Task]:
    """src/pretrain/pretrain.py:77-93."""
    return {n: Task(n, model, temperature, grl) for n in active}
