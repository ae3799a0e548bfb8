"""Shared helpers for the module/step parity tests."""
import torch
import torch.nn as nn

from gnn_pretraining_amd.graph import Batch
from oracle import graph_ops as OG


def set_dropout(model: nn.Module, p: float) -> None:
    """Parity runs keep train-mode BatchNorm but switch dropout off: GPU and CPU
    dropout masks cannot coincide."""
    for m in model.modules():
        if isinstance(m, nn.Dropout):
            m.p = p
        if hasattr(m, "dropout_p"):
            m.dropout_p = p


def to_oracle(b: Batch) -> OG.Batch:
    h = b.host()
    return OG.Batch(h.x, h.edge_index, h.batch, h.ptr, torch.tensor(h.edge_ptr_host), h.y, h.graph_properties)


def rel_err(got: torch.Tensor, want: torch.Tensor) -> float:
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    assert got.shape == want.shape, (got.shape, want.shape)
    return (got - want).abs().max().item() / max(want.abs().max().item(), 1e-30)


def assert_close(got, want, rtol, what=""):
    e = rel_err(got, want)
    assert e <= rtol, f"{what}: max rel err {e:.3e} > {rtol}"


def copy_state(dst: nn.Module, src: nn.Module) -> None:
    dst.load_state_dict({k: v.detach().cpu().clone() for k, v in src.state_dict().items()})
