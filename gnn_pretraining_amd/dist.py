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
            # RCCL needs a GPU per rank ("Duplicate GPU detected" otherwise): more local ranks than devices = a functional rehearsal over gloo
            local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
            backend = "nccl" if torch.cuda.is_available() and torch.cuda.device_count() >= local_world else "gloo"
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


def split_by_weight(weights: List[int], world: int) -> List[int]:
    """Cut points c[0] = 0 <= c[1] <= ... <= c[world] = len(weights) of a run of items into `world` CONTIGUOUS shares of about equal
    weight (share j = items [c[j], c[j + 1])): item i goes to the share its midpoint falls into.  Deterministic, every rank computes the same."""
    total = float(sum(weights))
    cuts, acc, j = [0], 0.0, 0
    for i, w in enumerate(weights):
        mid = acc + w / 2.0
        want = min(world - 1, int(mid * world / total)) if total > 0 else 0
        while j < want:
            cuts.append(i)
            j += 1
        acc += w
    while len(cuts) < world + 1:
        cuts.append(len(weights))
    return cuts


class ShardedGradSync:
    """The data-parallel exchange with PCGrad SHARDED over the ranks.

    `OverlappedGradSync` all-reduces every per-task gradient (shared tensors once per task: 36.2 MB for s4) and then every rank runs the
    identical PCGrad over all tensors.  PCGrad works tensor by tensor (gradient_surgery.py:70-103 projects per parameter tensor), so it shards:
    every tensor has ONE owner rank.  Per message (the same parts, in the order the step finishes them, on the same idle head stream):

        pack (rank-major)  ->  reduce-scatter: the owner receives the SUM of its tensors' per-task gradients  ->  unpack x 1/W
        ->  Gram / solve / combine for the owned tensors only (gmp_mt_pcgrad_clip_adamw_ex, phase bit 0)
        ->  pack the owned combined gradients  ->  all-gather  ->  unpack into final_grad on every rank
        ->  the foreign pass for the tensors of the other ranks (phase bit 2: flags, step counts, norm partials from the gradient in place)

    and, after the last message, total norm + clip + AdamW on every rank (phase bit 1) as before: replicas stay bit-identical, and the
    result equals the all-reduce path's whenever the sum of W addends does (W = 2: always; tests/test_gpu_dist.py).  Ring traffic per rank
    and step: (W-1)/W x (36.2 + 14.7) MB instead of 2 (W-1)/W x 36.2 MB (-30 %), PCGrad's Gram / combine sweeps 1/W per rank.  What it adds:
    the last message's all-gather is exposed after the backward (the all-reduce path exposes its last all-reduce instead).
    No multi-GPU box in rounds 1-3: exercised with two ranks over gloo on one GPU only -- opt-in (StepEngine(dp_mode="sharded"), GMP_DP_MODE=sharded).

    The owner's rule for "which tensors get a gradient" uses its own availability table; the foreign pass uses the local one.  They agree
    unless a (task, domain) pair dropped out on one rank only (fewer than two common nodes in a whole domain batch) -- as on the all-reduce path.

    parts[q] = (k0, k1): the run of tensor indices of part q; msg(k) -> [(offset, length)] slices of tensor k in the per-task gradient
    matrix (floats, multiples of 4); fin(k) -> (offset, length) of tensor k in final_grad."""

    def __init__(self, task_grads: Tensor, final_grad: Tensor, parts, msg, fin, comm: "torch.cuda.Stream", groups_spec: Optional[str] = None) -> None:
        from . import _lib as L
        self._L = L
        self.world, self.rank = world_size(), rank()
        W, r, dev = self.world, self.rank, task_grads.device
        spec = groups_spec or os.environ.get("GMP_DP_GROUPS", "0|1,2|3,4|5|6")
        groups = [[int(x) for x in g.split(",")] for g in spec.split("|")]
        if sorted(i for g in groups for i in g) != list(range(len(parts))) or any(g != sorted(g) for g in groups):
            raise ValueError(f"GMP_DP_GROUPS={spec!r} must list the parts 0..{len(parts) - 1} once each, in order")
        self.groups, self.comm, self.base, self.final = groups, comm, task_grads, final_grad
        self.host_s, self.calls = 0.0, 0

        def merged(sl):
            out = []
            for o, n in sl:
                if n <= 0:
                    continue
                if o % 4 or n % 4:
                    raise ValueError("ShardedGradSync: slice offsets and lengths must be multiples of 4 floats")
                if out and out[-1][0] + out[-1][1] == o:
                    out[-1][1] += n
                else:
                    out.append([o, n])
            return out

        def table(slices, pad_to, base_numel):
            """`slices` laid end to end, padded to pad_to floats by re-reading valid floats of the base (the receiver never looks at them)"""
            sl = [list(x) for x in slices]
            have = sum(n for _, n in sl)
            while have < pad_to:
                n = min(pad_to - have, base_numel // 4 * 4)
                sl.append([0, n])
                have += n
            return sl

        self.msgs = []
        self.total_rs = self.total_ag = 0
        for g in groups:
            own_k, foreign_k, rs_shards, ag_shards = [], [], [[] for _ in range(W)], [[] for _ in range(W)]
            for q in g:
                k0, k1 = parts[q]
                ks = list(range(k0, k1))
                cuts = split_by_weight([sum(n for _, n in msg(k)) for k in ks], W)
                for j in range(W):
                    a, b = k0 + cuts[j], k0 + cuts[j + 1]
                    for k in range(a, b):
                        rs_shards[j] += msg(k)
                        ag_shards[j].append(fin(k))
                    if j == r and b > a:
                        own_k.append((a, b))
                a, b = k0 + cuts[r], k0 + cuts[r + 1]
                if a > k0:
                    foreign_k.append((k0, a))
                if k1 > b:
                    foreign_k.append((b, k1))
            # shards as sorted, merged slice lists (a shared tensor's T copies lie P floats apart: sorting by offset groups them per task row)
            rs_shards = [merged(sorted(sl)) for sl in rs_shards]
            ag_shards = [merged(sorted(sl)) for sl in ag_shards]
            Lr = max(sum(n for _, n in sl) for sl in rs_shards)
            La = max(sum(n for _, n in sl) for sl in ag_shards)
            m = {"own_k": own_k, "foreign_k": foreign_k, "Lr": Lr, "La": La, "wait_part": g[-1]}
            if Lr:
                send = [x for j in range(W) for x in table(rs_shards[j], Lr, task_grads.numel())]
                m["rs_send"] = self._tab(send, dev)
                m["rs_recv"] = self._tab(rs_shards[r], dev) if rs_shards[r] else None
                m["rs_buf"] = torch.empty(W * Lr, dtype=torch.float32, device=dev)
                m["rs_out"] = torch.empty(Lr, dtype=torch.float32, device=dev)
            if La:
                m["ag_send"] = self._tab(table(ag_shards[r], La, final_grad.numel()), dev)
                # receiving side: shard j's pieces land at j * La; its padding must not be written anywhere real -> unpack shard by shard, pieces only
                m["ag_recv"] = [(self._tab(ag_shards[j], dev), j * La) for j in range(W) if ag_shards[j] and j != r]
                m["ag_buf"] = torch.empty(W * La, dtype=torch.float32, device=dev)
                m["ag_out"] = torch.empty(La, dtype=torch.float32, device=dev)
            self.msgs.append(m)
            self.total_rs += W * Lr
            self.total_ag += W * La

    def _tab(self, slices, dev):
        offs = [int(o) for o, _ in slices]
        pre = [0]
        for _, n in slices:
            pre.append(pre[-1] + int(n))
        if len(offs) > 256:
            raise ValueError("ShardedGradSync: more than 256 slices in one message")
        return (torch.tensor(offs + pre, dtype=torch.int64, device=dev), len(offs), pre[-1])

    def average_(self, lib, main: "torch.cuda.Stream", gate, own_pass, foreign_pass) -> None:
        """own_pass(k0, k1, stream_handle) / foreign_pass(k0, k1, stream_handle): enqueue the optimizer's phase bit 0 / bit 2 for a run of
        tensors on the exchange stream.  gate: (flags data_ptr, epoch) or None, as OverlappedGradSync.average_."""
        L = self._L
        import time as _t
        t0 = _t.perf_counter()
        W = self.world
        bp, fp = self.base.data_ptr(), self.final.data_ptr()
        with torch.cuda.stream(self.comm):
            h = self.comm.cuda_stream
            for m in self.msgs:
                L.check(lib.gmp_step_wait_grads(m["wait_part"], h), "gmp_step_wait_grads")
                if m["Lr"]:
                    tab, n, tot = m["rs_send"]
                    L.check(lib.gmp_segments_pack(bp, m["rs_buf"].data_ptr(), tab.data_ptr(), n, tot, h), "pack (reduce-scatter)")
                    dist.reduce_scatter_tensor(m["rs_out"], m["rs_buf"], op=dist.ReduceOp.SUM)
                    if m["rs_recv"] is not None:
                        tab, n, tot = m["rs_recv"]
                        L.check(lib.gmp_segments_unpack(bp, m["rs_out"].data_ptr(), tab.data_ptr(), n, tot, 1.0 / W, h), "unpack (own shard)")
                for k0, k1 in m["own_k"]:
                    own_pass(k0, k1, h)
                if m["La"]:
                    tab, n, tot = m["ag_send"]
                    L.check(lib.gmp_segments_pack(fp, m["ag_out"].data_ptr(), tab.data_ptr(), n, tot, h), "pack (all-gather)")
                    dist.all_gather_into_tensor(m["ag_buf"], m["ag_out"])
                    for (tab, n, tot), at in m["ag_recv"]:
                        L.check(lib.gmp_segments_unpack(fp, m["ag_buf"].data_ptr() + 4 * at, tab.data_ptr(), n, tot, 1.0, h), "unpack (foreign shard)")
                for k0, k1 in m["foreign_k"]:
                    foreign_pass(k0, k1, h)
        if gate is not None:
            flags, epoch = gate
            L.check(lib.gmp_gate_open(flags + 4 * 38, epoch, self.comm.cuda_stream), "gmp_gate_open")
            L.check(lib.gmp_gate_wait(flags, 1 << 38, epoch, flags + 4 * 63, main.cuda_stream), "gmp_gate_wait")
        else:
            main.wait_stream(self.comm)
        self.host_s += _t.perf_counter() - t0
        self.calls += 1
