"""Forced ReLU gates for flip-free gradient comparisons.  TEST INFRASTRUCTURE (see oracle/__init__.py).

Two correct fp32 implementations of a ReLU network disagree on the gate of a pre-activation that lies within rounding of
zero (about one per backbone forward at the step's sizes); the gradient of the graph containing it then moves by ~1e-3
(scripts/diag_precision.py, diag_tasks_fp64.py) -- which is why the step-level gradient tests had to accept 1e-2.  With
the gates of one implementation imposed on the other the comparison has no such discontinuity left and every gradient can
be held to a few 1e-4 (tests/test_gpu_engine.py::test_engine_gradients_with_shared_relu_gates).

Every ReLU of the oracle (InputEncoder, both ReLUs of a GINLayer, MLPHead) goes through `relu()` here, and the one other
kink on the path -- |hs - hd| of the link-prediction edge features (heads.py:62), whose subgradient sign(hs - hd) flips when two
embeddings agree to rounding: 7.7 M such elements per step -- through `abs()`.  Outside a `use_tape` block they are F.relu /
torch.abs.  Inside, the k-th call takes the k-th mask of the tape (shape-checked, so a wrong call order cannot pass silently):
y = x * mask, dy/dx = mask (mask: 0/1 gates for relu, -1/0/+1 signs for abs)."""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor

_TAPE: Optional["GateTape"] = None


class GateTape:
    def __init__(self, masks: List[Tensor]) -> None:
        self.masks, self.pos, self.flips = list(masks), 0, 0

    def take(self, x: Tensor) -> Tensor:
        if self.pos >= len(self.masks):
            raise AssertionError(f"gate tape exhausted at call {self.pos} (shape {tuple(x.shape)})")
        m = self.masks[self.pos]
        if tuple(m.shape) != tuple(x.shape):
            raise AssertionError(f"gate tape: call {self.pos} has shape {tuple(x.shape)}, mask {tuple(m.shape)}")
        self.pos += 1
        own = (x.detach() > 0) if m.dtype == torch.bool else torch.sign(x.detach()).to(m.dtype)
        self.flips += int((own != m).sum())                      # gates this implementation would have chosen differently
        return m

    def done(self) -> bool:
        return self.pos == len(self.masks)


class use_tape:
    def __init__(self, tape: GateTape) -> None:
        self.tape = tape

    def __enter__(self) -> GateTape:
        global _TAPE
        self.prev, _TAPE = _TAPE, self.tape
        return self.tape

    def __exit__(self, *exc) -> None:
        global _TAPE
        _TAPE = self.prev


def relu(x: Tensor) -> Tensor:
    if _TAPE is None:
        return F.relu(x)
    return x * _TAPE.take(x).to(x.dtype)


def abs(x: Tensor) -> Tensor:          # noqa: A001 (mirrors torch.abs)
    if _TAPE is None:
        return torch.abs(x)
    return x * _TAPE.take(x).to(x.dtype)


class ReLU(nn.Module):
    """nn.ReLU stand-in (no parameters: state_dict keys of the enclosing Sequential are unchanged)."""

    def forward(self, x: Tensor) -> Tensor:
        return relu(x)
