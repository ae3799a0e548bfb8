"""TEST INFRASTRUCTURE (see oracle/__init__.py) -- CPU restatement of LinkPredictionHardNegativeMiner
(src/finetune/finetune.py:45-106) in torch-CPU / numpy, line by line.

Two places where the reference leaves the result open are fixed here and documented in include/gnnmp.h:
  * torch.topk does not define the order of equal scores (and the similarity matrix is symmetric, so every score
    appears at least twice): ties go to the lower flat index i*n+j (a stable sort of the row-major score list);
  * the random remainder uses torch.randperm on the model's device from the global RNG: here the permutation is a
    parameter.
Pinning: the reference has no fixture for the miner ("parity unpinned" by the reference); tests/test_gpu_miner.py pins
the restatement on hand-checkable cases (exact arithmetic, known top pairs)."""
from __future__ import annotations

from typing import Callable, Optional

import numpy as np
import torch
import torch.nn.functional as F

HARD_NEGATIVE_RATIO = 0.3      # finetune.py:39
MIN_HARD_NEGATIVES = 8         # finetune.py:40


def similarity_and_mask(node_embeddings: torch.Tensor, existing_edges: torch.Tensor):
    n = node_embeddings.size(0)
    zn = F.normalize(node_embeddings, dim=1)                                  # :50
    sim = torch.mm(zn, zn.t())                                                # :51
    edge_mask = torch.zeros(n, n, dtype=torch.bool)                           # :53
    if existing_edges.size(1) > 0:                                            # :55-57
        edge_mask[existing_edges[0], existing_edges[1]] = True
        edge_mask[existing_edges[1], existing_edges[0]] = True
    edge_mask.fill_diagonal_(True)                                            # :59
    return sim, ~edge_mask                                                    # :61


def num_hard(num_potential: int, num_negatives: int) -> int:
    k = max(MIN_HARD_NEGATIVES, int(num_potential * HARD_NEGATIVE_RATIO))     # :69
    return min(k, num_potential, num_negatives)                               # :70


def mine_hard_negatives_for_edges(node_embeddings: torch.Tensor, positive_edges: torch.Tensor, num_negatives: int,
                                  existing_edges: torch.Tensor,
                                  randperm: Optional[Callable[[int], torch.Tensor]] = None,
                                  similarity: Optional[torch.Tensor] = None) -> torch.Tensor:
    """`similarity`: use this score matrix instead of computing it (to check the index work bit-exactly on scores
    produced elsewhere)."""
    sim, potential = similarity_and_mask(node_embeddings, existing_edges)
    if similarity is not None:
        sim = similarity
    scores = sim[potential]                                                   # :63 (row-major order)
    idx = torch.where(potential)                                              # :64
    if scores.numel() == 0:                                                   # :66-67
        return torch.empty(2, 0, dtype=torch.long)
    k = num_hard(scores.numel(), num_negatives)
    if k > 0:                                                                 # :72-76
        order = torch.from_numpy(np.argsort(-scores.numpy(), kind="stable")[:k].copy())
        hard_src, hard_dst = idx[0][order], idx[1][order]
        hard = torch.stack([hard_src, hard_dst], dim=0)
    else:
        hard = torch.empty(2, 0, dtype=torch.long)
    remaining = num_negatives - k                                             # :80
    if remaining <= 0:
        return hard
    rem_mask = potential.clone()                                              # :82-85
    if k > 0:
        rem_mask[hard_src, hard_dst] = False
        rem_mask[hard_dst, hard_src] = False
    rem_idx = torch.where(rem_mask)                                           # :87
    avail = rem_idx[0].numel()
    if avail == 0:                                                            # :103
        return hard
    take = min(remaining, avail)                                              # :91-92
    perm = (randperm or torch.randperm)(avail)[:take]
    rand = torch.stack([rem_idx[0][perm], rem_idx[1][perm]], dim=0)           # :93-95
    return torch.cat([hard, rand], dim=1) if k > 0 else rand                  # :97-100
