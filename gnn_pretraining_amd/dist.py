"""Data-parallel glue: one process per GPU, gradients averaged with ONE flat all-reduce per step
(backend "nccl" == RCCL over xGMI on ROCm; "gloo" for the CPU tests).

The reference has no distributed training at all; this is the new piece
BASELINE.json asks for.  Semantics: every rank draws its own 4x8-graph step,
computes per-task gradients, the ranks average them task by task, and every rank then applies the
identical PCGrad projection / clip / AdamW, so replicas stay bit-identical without a broadcast.
All per-task gradients of a step travel in one buffer (36 MB for s4): at this size the collective is
latency- and link-bound, so one launch beats 300 small ones.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import torch
import torch.distributed as dist
from torch import Tensor


def init_from_env(backend: Optional[str] = None) -> int:
    """Join the job torchrun started (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*); returns world size."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # the host driver on this pool only supports dmabuf IPC: without this RCCL's peer mappings fail with
    # "hipIpcGetMemHandle: invalid argument" (it is already exported on the GPU boxes; harmless to repeat)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
            torch.cuda.set_device(local)
            dist.init_process_group(backend=backend, device_id=torch.device("cuda", local))   # binds the communicator to this rank's GPU
        else:
            dist.init_process_group(backend=backend)
    return world


def world_size() -> int:
    return dist.get_world_size() if dist.is_initialized() else 1


def _exchange_needed() -> bool:
    """False on a single rank -- unless GMP_DP_FORCE=1 asks for the pack / all-reduce / unpack sequence anyway
    (scripts/diag_dp_overlap.py: what the exchange machinery costs a step on one GPU)."""
    return world_size() > 1 or (dist.is_initialized() and os.environ.get("GMP_DP_FORCE") == "1")


def rank() -> int:
    return dist.get_rank() if dist.is_initialized() else 0


class FlatGradSync:
    """Average a set of gradient tensors across ranks through one flat buffer."""

    def __init__(self) -> None:
        self._buf: Optional[Tensor] = None

    def _buffer(self, numel: int, like: Tensor) -> Tensor:
        if self._buf is None or self._buf.numel() < numel or self._buf.device != like.device:
            self._buf = torch.empty(numel, dtype=torch.float32, device=like.device)
        return self._buf[:numel]

    def average_(self, tensors: List[Tensor]) -> None:
        """In place: every tensor becomes the mean over ranks.  Tensor order/shapes must agree on all ranks."""
        w = world_size()
        if w == 1 or not tensors:
            return
        total = sum(t.numel() for t in tensors)
        flat = self._buffer(total, tensors[0])
        torch.cat([t.reshape(-1) for t in tensors], out=flat)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.mul_(1.0 / w)
        off = 0
        for t in tensors:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t))
            off += n

    def average_task_grads_(self, grads: Dict[str, Dict[str, Tensor]]) -> None:
        """grads[task][param_name] -> tensor; deterministic (task, name) order."""
        self.average_([grads[t][n] for t in grads for n in sorted(grads[t])])

    def average_model_grads_(self, model: torch.nn.Module) -> None:
        self.average_([p.grad for _, p in model.named_parameters() if p.grad is not None])


class PackedGradSync:
    """The same exchange for the stacked engine, where every slice lives in ONE device buffer (the [tasks, params] per-task
    gradient matrix): a table of (offset, length) slices is fixed at construction, one kernel packs them into the message,
    one all-reduce, one kernel writes the mean back (libgnnmp gmp_segments_pack / _unpack) -- instead of torch.cat plus a
    copy_ per slice.  Offsets / lengths in floats, multiples of 4 (the engine aligns every tensor to 4 floats)."""

    def __init__(self, base: Tensor, slices) -> None:
        from . import _lib as L
        self._L = L
        self.base = base
        slices = list(slices)
        offs = [int(o) for o, _ in slices]
        lens = [int(n) for _, n in slices]
        if any(o % 4 or n % 4 for o, n in zip(offs, lens)):
            raise ValueError("PackedGradSync: slice offsets and lengths must be multiples of 4 floats")
        pre = [0]
        for n in lens:
            pre.append(pre[-1] + n)
        self.n, self.total = len(offs), pre[-1]
        self.table = torch.tensor(offs + pre, dtype=torch.int64, device=base.device)
        self.packed = torch.empty(self.total, dtype=torch.float32, device=base.device)
        self._ptrs = (self.base.data_ptr(), self.packed.data_ptr(), self.table.data_ptr())
        self.share = 1.0           # this message's share of the step's whole exchange (OverlappedGradSync sets it per part)
        # diagnostic (scripts/diag_dp_overlap.py): stand in for the collective's run time on a box with one GPU by a spin kernel
        # of GMP_DP_FAKE_US microseconds for the whole exchange, split over the parts by size
        self._fake_us = float(os.environ.get("GMP_DP_FAKE_US", "0"))

    def average_(self, stream_handle: Optional[int] = None) -> None:
        """stream_handle: the hipStream_t of torch's CURRENT stream when the caller already has it (saves the lookup)."""
        w = world_size()
        if not _exchange_needed() or self.total == 0:
            return
        L = self._L
        st = stream_handle if stream_handle is not None else torch.cuda.current_stream(self.base.device).cuda_stream
        lib, (pb, pp, pt) = L.lib(), self._ptrs            # (pointers cached: the launcher thread has little host time to spare)
        rc = lib.gmp_segments_pack(pb, pp, pt, self.n, self.total, st)
        if rc:
            L.check(rc, "gmp_segments_pack")
        dist.all_reduce(self.packed, op=dist.ReduceOp.SUM)
        if self._fake_us > 0:
            L.check(lib.gmp_spin_us(int(round(self._fake_us * self.share)), st), "gmp_spin_us")
        rc = lib.gmp_segments_unpack(pb, pp, pt, self.n, self.total, 1.0 / w, st)
        if rc:
            L.check(rc, "gmp_segments_unpack")


class OverlappedGradSync:
    """The exchange of the stacked engine run BESIDE the stacked backward (libgnnmp
    gmp_step_wait_grads).  The per-task gradient matrix is cut into parts in the order the step finishes them -- the task
    heads (final before the backward starts), then backbone layers 4 ... 0, then mask token + encoders -- and
    every part is packed, all-reduced and written back on `comm` (one of the engine's two head streams, idle during the
    backward: the runtime has four hardware queues and the step already uses four streams, so no new stream is created)
    as soon as the events of that part have fired.  Only the last part (0.4 of 36 MB for s4) and the tail of layer 0's are exposed after
    the backward; RCCL runs a synchronous collective on the caller's current stream, so every collective is ordered behind
    its own pack kernel and before its unpack kernel without host involvement.  The result is bitwise the one-message
    exchange: the same elementwise sums, only sent in pieces."""

    def __init__(self, base: Tensor, parts, comm: "torch.cuda.Stream") -> None:
        # Consecutive parts may travel together (one pack / collective / unpack, sent when the LAST of them is final): every
        # collective costs the launcher thread ~40 us, and at 1.6 ms per step the host has little to spare.  Default: heads |
        # layers 4+3 | layers 2+1 | layer 0 | mask token + encoders -- the two parts whose collectives would otherwise be exposed
        # after the backward stay on their own.  GMP_DP_GROUPS="0|1|2|3|4|5|6" sends every part by itself.
        spec = os.environ.get("GMP_DP_GROUPS", "0|1,2|3,4|5|6")
        groups = [[int(x) for x in g.split(",")] for g in spec.split("|")]
        if sorted(i for g in groups for i in g) != list(range(len(parts))) or any(g != sorted(g) for g in groups):
            raise ValueError(f"GMP_DP_GROUPS={spec!r} must list the parts 0..{len(parts) - 1} once each, in order")
        self.groups = groups
        self.wait_part = [g[-1] for g in groups]                  # the last part of a group is the one to wait for
        self.parts = [PackedGradSync(base, [sl for i in g for sl in parts[i]]) for g in groups]
        self.comm = comm
        self.total = sum(p.total for p in self.parts)
        self.host_s, self.calls = 0.0, 0              # host time spent enqueueing the exchange (scripts/diag_dp_overlap.py)
        for p in self.parts:
            p.share = p.total / max(self.total, 1)

    def average_(self, lib, main: "torch.cuda.Stream", gate=None, exchange: bool = True, after_message=None) -> None:
        """gate: (flags data_ptr, epoch) of the engine when its streams sit on hardware queues of their own -- the main stream
        then waits for the exchange through a gate (flag 38) instead of an event (record + wait: ~6 us on this hardware).
        after_message(parts, stream_handle): called after every message's unpack with the parts it carried, to enqueue what
        follows them on the exchange stream (the engine's per-part PCGrad).  exchange=False: no collectives (one rank), only the
        waits and the callbacks."""
        exchange = exchange and _exchange_needed()
        if not exchange and after_message is None:
            return
        from . import _lib as L
        import time as _t
        t0 = _t.perf_counter()
        with torch.cuda.stream(self.comm):
            handle = self.comm.cuda_stream
            for group, part in zip(self.groups, self.parts):
                L.check(lib.gmp_step_wait_grads(group[-1], handle), "gmp_step_wait_grads")
                if exchange:
                    part.average_(handle)
                if after_message is not None:
                    after_message(group, handle)
        if gate is not None:
            flags, epoch = gate
            L.check(lib.gmp_gate_open(flags + 4 * 38, epoch, self.comm.cuda_stream), "gmp_gate_open")
            L.check(lib.gmp_gate_wait(flags, 1 << 38, epoch, flags + 4 * 63, main.cuda_stream), "gmp_gate_wait")
        else:
            main.wait_stream(self.comm)
        self.host_s += _t.perf_counter() - t0
        self.calls += 1
