"""Stacked-step engine: one whole pre-training step (all tasks x all domains) as ~250 launches.

The reference runs 28 separate encoder+backbone forwards per s4 step (7 per domain: NFM 1, LP 1, NC 2,
GC 2, GP 1 -- src/pretrain/pretrain.py:124-129 over tasks.py), then 5 full backwards for PCGrad.
All 28 share the backbone weights, so here they are STACKED into one block-diagonal pass:

  * every reference forward() call becomes a *segment* of rows; BatchNorm statistics are taken per
    segment (gmp_bn_fwd seg_ptr), so the stacked pass computes exactly what the 28 calls compute,
    running statistics included (segments are applied in the reference's call order);
  * segments are laid out task-major, so each task owns one contiguous row range: ONE backward pass
    produces all per-task weight gradients PCGrad needs (grouped TN GEMMs / grouped reductions that
    write straight into a [tasks, params] gradient buffer) instead of five backward passes;
  * per-domain heads run as grouped GEMMs (one group per domain);
  * PCGrad + clip + AdamW are five multi-tensor launches over flat buffers, with the reference's
    "which tensors receive a gradient" rule (gradient_surgery.py:61) reproduced exactly.

Host work per step is index bookkeeping only (augmentation / masks / negatives with the reference's CPU
RNG contract) and ships to the device in two packed copies; nothing in the step reads back from the GPU.
No autograd: forward and backward are explicit kernel sequences over a preallocated arena.
"""
from __future__ import annotations

import ctypes as C
import os
import random
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch import Tensor

from . import _lib as L
from .constants import DOMAIN_DIMENSIONS, GRAPH_PROPERTY_DIM
from .graph import Batch
from .models.gnn import DROPOUT_RATE, GNN_HIDDEN_DIM, GNN_NUM_LAYERS
from .models.pretrain_model import PretrainableGNN, draw_mask_indices
from .pretrain.augmentations import _augment_one
from .pretrain.control import DEFAULT_LR, DEFAULT_WEIGHT_DECAY, TASK_SPECIFIC_LR
from .pretrain.tasks import sample_negative_edges

H = GNN_HIDDEN_DIM
NT, NN, TN = 0, 1, 2
MAXT = 8
SEG_CSR_MAX_ROWS, SEG_CSR_MAX_EDGES = 8192, 24576        # gmp_csr_build_segmented: what one workgroup's LDS holds

# rough length (us) of each head's kernel chain at the s4 workload, used only to balance heads over the four streams
HEAD_CHAIN_US = {"node_contrast": 350.0, "link_pred": 330.0, "graph_contrast": 250.0, "node_feat_mask": 130.0, "graph_prop": 100.0,
                 "domain_adv": 100.0}
SUPPORTED_TASKS = ("node_feat_mask", "link_pred", "node_contrast", "graph_contrast", "graph_prop", "domain_adv")
DA_HIDDEN, DA_DROPOUT = 128, 0.5          # heads.py:11-12


_ARR_CACHE: Dict[Tuple, "C.Array"] = {}


def _i32(xs) -> "C.Array":
    """ctypes int32 array; memoised by content (offset/row tables repeat every step)."""
    key = (32,) + tuple(xs)
    a = _ARR_CACHE.get(key)
    if a is None:
        if len(_ARR_CACHE) > 20000:
            _ARR_CACHE.clear()
        a = _ARR_CACHE[key] = (C.c_int32 * len(xs))(*[int(v) for v in xs])
    return a


def _i64(xs) -> "C.Array":
    key = (64,) + tuple(xs)
    a = _ARR_CACHE.get(key)
    if a is None:
        if len(_ARR_CACHE) > 20000:
            _ARR_CACHE.clear()
        a = _ARR_CACHE[key] = (C.c_int64 * len(xs))(*[int(v) for v in xs])
    return a


class StepInputs:
    """Device-resident input of one step (the benchmark keeps a pool of these in HBM):
    all domains' features padded into one matrix + per-domain offsets, plus host twins for index work."""

    def __init__(self, batches: Dict[str, Batch], device, dpad: int) -> None:
        self.domains = list(batches)
        self.host = {d: b.host() for d, b in batches.items()}
        rows, self.row_off = 0, {}
        for d in self.domains:
            self.row_off[d] = rows
            rows += self.host[d].num_nodes
        pin = torch.device(device).type == "cuda"
        x = torch.zeros(rows, dpad, pin_memory=pin)           # pinned (cached by torch's host allocator): the copy below is asynchronous
        gp = []
        for d in self.domains:
            hb = self.host[d]
            x[self.row_off[d]:self.row_off[d] + hb.num_nodes, :hb.x.size(1)] = hb.x
            gp.append(hb.graph_properties.to(torch.float32).view(hb.num_graphs, GRAPH_PROPERTY_DIM))
        self.x_all = x.to(device, non_blocking=pin)
        g = torch.cat(gp)
        self.graph_props = (g.pin_memory() if pin else g).to(device, non_blocking=pin)           # [sum B, 12] in domain order
        self.device = torch.device(device)
        self._dev_graph: Dict[str, tuple] = {}

    def dev_graph(self, d: str):
        """(ptr, eptr, edge_index, view_ptr, mask_ptr) of domain d's batch as int64 device tensors (batch-local numbering; view_ptr /
        mask_ptr = exclusive scans of the per-graph kept-node / masked-node counts, functions of the graph sizes alone) + the two scans
        as numpy arrays -- what the device-side augmentation kernels read; uploaded once per input, on first use (a pageable-memory
        upload synchronises with the stream it is made on: never do it per step)."""
        t = self._dev_graph.get(d)
        if t is None:
            hb = self.host[d]
            dev = self.device
            kept, masked = L_view_sizes(hb.ptr_host)
            vptr = np.concatenate([[0], np.cumsum(kept)]).astype(np.int64)
            moff = np.concatenate([[0], np.cumsum(masked)]).astype(np.int64)
            t = self._dev_graph[d] = (torch.tensor(hb.ptr_host, dtype=torch.long).to(dev), torch.tensor(hb.edge_ptr_host, dtype=torch.long).to(dev),
                                      hb.edge_index.contiguous().to(dev), torch.from_numpy(vptr).to(dev), torch.from_numpy(moff).to(dev), vptr, moff)
        return t


def _merge_runs(ranges):
    """[k0, k1) tensor-index ranges -> the fewest contiguous runs covering them (one PCGrad launch set per run)."""
    out = []
    for a, b in sorted(r for r in ranges if r[1] > r[0]):
        if out and out[-1][1] == a:
            out[-1][1] = b
        else:
            out.append([a, b])
    return out


class StepPlan:
    """Host-side description of one stacked step (pure index data)."""
    pass


class Artefacts(dict):
    """draw()'s result {task: {domain: arrays}}; `raw` keeps the native module's own tuples (hostdraw.draw_step) for the native layout
    step (hostdraw.plan_step), which never touches the numpy copies -- so those are only built (`fill`) when somebody reads the dict
    (tests, the oracle harness, the Python layout)."""
    raw = None
    fill = None

    def _need(self) -> None:
        f, self.fill = self.fill, None
        if f is not None:
            f(self)

    def __getitem__(self, k):
        self._need()
        return dict.__getitem__(self, k)

    def __iter__(self):
        self._need()
        return dict.__iter__(self)

    def __len__(self):
        self._need()
        return dict.__len__(self)

    def __contains__(self, k):
        self._need()
        return dict.__contains__(self, k)

    def get(self, k, default=None):
        self._need()
        return dict.get(self, k, default)

    def keys(self):
        self._need()
        return dict.keys(self)

    def items(self):
        self._need()
        return dict.items(self)

    def values(self):
        self._need()
        return dict.values(self)


class _Views(dict):
    """Named views into a packed upload image (StepPlan.a32 / a64 of a natively planned step), built on first use."""

    def __init__(self, cat: np.ndarray, lay) -> None:
        super().__init__()
        self._cat, self._lay = cat, {n: (o, k) for n, o, k in lay}

    _SHAPES = {"lp_edges": (2, -1), "edge_index": (2, -1), "tiles": (-1, 2), "lp_pos": (2, -1)}

    def __missing__(self, name):
        o, k = self._lay[name]
        v = self._cat[o:o + k]
        v = v.reshape(self._SHAPES[name]) if name in self._SHAPES else v
        self[name] = v
        return v

    def __contains__(self, name) -> bool:
        return name in self._lay


class ViewArrays:
    """One augmented view of a whole domain batch as flat index arrays (what Batch.from_data_list of the
    augmented graphs would hold, minus the features): rows = kept nodes (domain-local ids of the base batch),
    edges = [2, e'] in view-local numbering, ptr = per-graph node offsets, rowmask = per-row bitmask of zeroed
    feature columns (None if no graph drew an attribute mask), common = view-local ids of nodes kept in BOTH views."""
    __slots__ = ("rows", "edges", "ptr", "rowmask", "common")

    def __init__(self, rows, edges, ptr, rowmask, common) -> None:
        self.rows, self.edges, self.ptr, self.rowmask, self.common = rows, edges, ptr, rowmask, common


def device_views_to_host(dv) -> Tuple["ViewArrays", "ViewArrays"]:
    """ops.DeviceViews (gmp_aug_two_views outputs, on the device) -> the two ViewArrays the planner takes.  One small read-back:
    the counts say how much of the edge / common arrays is valid."""
    tot = dv.totals.cpu().tolist()
    out = []
    for v in range(2):
        rm = dv.rowmask[v].cpu().numpy().view(np.uint64) if tot[3 + v] else None
        out.append(ViewArrays(dv.rows[v].cpu().numpy(), dv.edges[v][:, :tot[v]].cpu().numpy(), np.asarray(dv.view_ptr, dtype=np.int64),
                              rm, dv.common[v][:tot[2]].cpu().numpy()))
    return out[0], out[1]


class DrawTicket:
    """One step's device-side draws in flight: the pinned slot they land in, where each (task, domain) piece sits, the flag value
    that says they have landed (StepEngine.enqueue_draws / collect_draws)."""
    __slots__ = ("slot", "layout", "epoch")

    def __init__(self, slot, layout, epoch) -> None:
        self.slot, self.layout, self.epoch = slot, layout, epoch


def L_view_sizes(ptr_host):
    from .ops import view_sizes
    return view_sizes(ptr_host)


_HOSTDRAW, _HOSTDRAW_TRIED = None, False


def merge_mirrored_pairs(b: Batch, neg: np.ndarray, offset: int, ord_base: int = 0) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Unordered pairs [2, K'] (+offset), signed multiplicities (+w positive, -w negative; w = 1 or 2) and ordered positions [2, K'] int32 of
    one domain batch's scored pairs (its edges, then the drawn negatives): the LP scorer's features (heads.py:57-61) are symmetric in
    (src, dst), so (i, j) and (j, i) need one row through the 768 -> 256 layer, not two.  The reference drops every ORDERED row with a mask of
    its own (heads.py:44-52), so each merged row remembers the one or two ordered rows it stands for: ord[0] = position of the first
    occurrence in the reference's list (counted from ord_base: this domain's positives, then its negatives), ord[1] = of the second (-1: none).
    A third occurrence of a pair (a duplicated edge) starts a row of its own.  Native (csrc_host/hostdraw.cpp) when built, numpy otherwise;
    both keep first-occurrence order, positives first."""
    H = hostdraw()
    if H is not None and hasattr(H, "merge_mirrored_pairs"):
        ptr, _, ei = _host_tensors(b)
        pairs, w, ord_ = H.merge_mirrored_pairs(ei, torch.from_numpy(np.ascontiguousarray(neg)), ptr, int(offset), int(ord_base))
        return pairs.numpy(), w.numpy(), ord_.numpy()
    out_p, out_w, out_o, n = [], [], [], max(b.num_nodes, 1)
    base = int(ord_base)
    for sign, e in ((1.0, b.edge_index.numpy()), (-1.0, neg)):
        lo, hi = np.minimum(e[0], e[1]), np.maximum(e[0], e[1])
        key = lo * n + hi
        order = np.argsort(key, kind="stable")
        sk = key[order]
        ar = np.arange(len(sk))
        start = np.ones(len(sk), dtype=bool)
        start[1:] = sk[1:] != sk[:-1]
        rank = ar - np.maximum.accumulate(np.where(start, ar, 0))           # occurrence number of every entry among its equals
        lead = np.flatnonzero(rank % 2 == 0)                                    # sorted positions that open a row
        nxt = np.minimum(lead + 1, len(sk) - 1)
        has2 = (lead + 1 < len(sk)) & (sk[nxt] == sk[lead]) if len(sk) else np.zeros(0, dtype=bool)
        first = order[lead]
        second = np.where(has2, order[nxt], -1)
        emit = np.argsort(first, kind="stable")                                 # rows in first-occurrence order
        first, second, has2 = first[emit], second[emit], has2[emit]
        out_p.append(np.stack([lo[first], hi[first]]) + offset)
        out_w.append((sign * (1 + has2)).astype(np.float32))
        out_o.append(np.stack([first + base, np.where(has2, second + base, -1)]).astype(np.int32))
        base += e.shape[1]
    return np.concatenate(out_p, axis=1), np.concatenate(out_w), np.concatenate(out_o, axis=1)


def hostdraw():
    """The native module for the reference-order draws (csrc_host/hostdraw.cpp), or None when it has not been built: the
    Python implementations it mirrors then run instead (bit-identical, ~10x slower, and they hold the GIL)."""
    global _HOSTDRAW, _HOSTDRAW_TRIED
    if not _HOSTDRAW_TRIED:
        _HOSTDRAW_TRIED = True
        if os.environ.get("GMP_NO_HOSTDRAW") is None:
            try:
                from . import _hostdraw
                _HOSTDRAW = _hostdraw
            except ImportError:
                _HOSTDRAW = None
    return _HOSTDRAW


def _host_tensors(b: Batch):
    """(node ptr, edge ptr, contiguous edge_index) of a host batch as the int64 tensors the native draw module takes, built once
    per batch object (they were rebuilt for every task of every step: 24 torch.tensor calls per step)."""
    t = b.__dict__.get("_host_tensors")
    if t is None:
        t = b.__dict__["_host_tensors"] = (torch.tensor(b.ptr_host, dtype=torch.long), torch.tensor(b.edge_ptr_host, dtype=torch.long),
                                           b.edge_index.contiguous())
    return t


def _empty_views():
    z, e = np.zeros(0, dtype=np.int64), np.zeros((2, 0), dtype=np.int64)
    return (ViewArrays(z, e, np.zeros(1, dtype=np.int64), None, z), ViewArrays(z.copy(), e.copy(), np.zeros(1, dtype=np.int64), None, z.copy()))


# what draw() records for a domain that has no graph in this step
_EMPTY_ART = {"node_feat_mask": lambda: np.zeros(0, dtype=np.int64), "link_pred": lambda: np.zeros((2, 0), dtype=np.int64),
              "node_contrast": _empty_views, "graph_contrast": lambda: None, "domain_adv": lambda: None}


class StepEngine:
    def __init__(self, model: PretrainableGNN, tasks: Sequence[str], domains: Sequence[str], device,
                 max_rows: int = 16384, max_edges: int = 131072, seed: int = 0, shuffle_rng: Optional[random.Random] = None,
                 grad_sync=None, rng_mode: str = "reference", native: bool = True, neg_rng: Optional[random.Random] = None,
                 dp_mode: Optional[str] = None) -> None:
        self.native = native       # True: csrc/step.hip enqueues the step; False: the same launches one by one from Python
        # data-parallel exchange: "allreduce" (every rank all-reduces all per-task gradients and runs the whole PCGrad: dist.OverlappedGradSync)
        # or "sharded" (reduce-scatter to the owner of each tensor, PCGrad on the owned tensors, all-gather of the combined gradient:
        # dist.ShardedGradSync -- 30 % fewer bytes, PCGrad 1/W per rank; needs the native executor)
        self.dp_mode = dp_mode or os.environ.get("GMP_DP_MODE", "allreduce")
        if self.dp_mode not in ("allreduce", "sharded"):
            raise ValueError("dp_mode must be 'allreduce' or 'sharded'")
        self._shard_sync_obj = None
        if rng_mode not in ("reference", "vectorized", "device"):
            raise ValueError("rng_mode must be 'reference', 'vectorized' or 'device'")
        self.rng_mode, self._nprng = rng_mode, None
        # Link-prediction negatives: PyG's sampler draws from Python's `random` (the global, unseeded module in the reference), never
        # from the shared torch generator (pretrain/tasks.py sample_negative_edges).  The engine keeps a stream of its own.
        self.neg_rng = neg_rng if neg_rng is not None else random.Random(0x9E3779B1 * (seed + 1))
        self._neg_native = None
        # score each unordered pair once (the scorer is symmetric in (src, dst)); GMP_LP_MERGE=0 keeps the reference's ordered list
        self.lp_merge = os.environ.get("GMP_LP_MERGE", "1") != "0"
        # row ranges of the stacked forward (gnnmp_step.h fwd_cut_*): GMP_FWD_RANGES = 1 (one pass on main), 2 (default), 3 (measured equal or
        # slightly worse: 1.474-1.484 against 1.455-1.485 ms per step)
        self.fwd_ranges = max(1, min(3, int(os.environ.get("GMP_FWD_RANGES", "2"))))
        self.native_plan = os.environ.get("GMP_NATIVE_PLAN", "1") != "0"
        for t in tasks:
            if t not in SUPPORTED_TASKS:
                raise NotImplementedError(f"StepEngine covers {SUPPORTED_TASKS}; '{t}' runs on the module path")
        self.model, self.tasks, self.domains, self.device = model, list(tasks), list(domains), torch.device(device)
        self.T, self.D = len(self.tasks), len(self.domains)
        self._bn_calls_dom = [0] * self.D
        self.lib = L.lib()
        self.dpad = 40 if max(DOMAIN_DIMENSIONS[d] for d in domains) <= 40 else 64
        self.max_rows, self.max_edges = max_rows, max_edges
        self.seed, self.step_count = seed, 0
        self.shuffle_rng = shuffle_rng
        self.grad_sync = grad_sync
        self._packed_sync = None
        self._bn_calls = 0
        self.temperature = 0.5
        self.da_dropout = DA_DROPOUT
        self.grl_lambda = 0.0              # gradient-reversal strength of the domain-adversarial task (GRLScheduler)
        self._p_cache, self._tg_cache = {}, {}
        self._stream_handle = torch.cuda.current_stream(self.device).cuda_stream if torch.cuda.is_available() else 0
        self.host_ms = {"draw": 0.0, "plan": 0.0, "upload": 0.0, "launch": 0.0, "steps": 0}   # host time per phase (upload includes ring waits)
        self.dropout_p = DROPOUT_RATE
        self.max_grad_norm = 0.5
        self._flatten_parameters()
        self._build_tables()
        self._alloc()

    @property
    def upload_on_aux(self) -> bool:
        """Native executor with gates: the step's index arrays go up on the aux stream (16 us of PCIe reads off the main stream)."""
        return bool(self.use_gates and self.native and os.environ.get("GMP_UPLOAD_ON_AUX", "1") != "0")

    @property
    def parts_beside_backward(self) -> bool:
        """(self.native may be flipped after construction -- the tests do: only the native executor publishes the part flags)"""
        return bool(self._parts_ok and self.native)

    # ------------------------------------------------------------------ parameters
    def _flatten_parameters(self) -> None:
        m, dev = self.model, self.device
        named = dict(m.named_parameters())
        enc = [m.input_encoders[d] for d in self.domains]
        order: List[Tuple[str, Tensor]] = []
        for d, e in zip(self.domains, enc):
            order.append((f"input_encoders.{d}.linear.weight", e.linear.weight))
        for key in ("linear.bias", "batch_norm.weight", "batch_norm.bias"):      # [D][256] blocks (grouped BN layout)
            for d in self.domains:
                order.append((f"input_encoders.{d}.{key}", named[f"input_encoders.{d}.{key}"]))
        seen = {n for n, _ in order}
        for n, p in named.items():
            if n not in seen:
                order.append((n, p))
        al4 = lambda v: (v + 3) // 4 * 4           # every tensor starts 16-byte aligned (float4 / MFMA tile loads)
        self.off: Dict[str, int] = {}
        o = 0
        for n, p in order:
            self.off[n] = o
            o += al4(p.numel())
        self.P = o
        self.P_shared = min(self.off[n] for n, _ in order if n.startswith("heads.")) if any(n.startswith("heads.") for n, _ in order) else o
        self.flat = torch.zeros(self.P, device=dev)
        for n, p in order:
            o = self.off[n]
            self.flat[o:o + p.numel()].copy_(p.data.reshape(-1))
            p.data = self.flat[o:o + p.numel()].view_as(p)
        self.names = [n for n, _ in order]
        self.numel = {n: p.numel() for n, p in order}
        # encoder running statistics as [D][256] blocks
        self.enc_rm = torch.zeros(self.D, H, device=dev)
        self.enc_rv = torch.ones(self.D, H, device=dev)
        for i, e in enumerate(enc):
            self.enc_rm[i].copy_(e.batch_norm.running_mean)
            self.enc_rv[i].copy_(e.batch_norm.running_var)
            e.batch_norm.running_mean.data = self.enc_rm[i]
            e.batch_norm.running_var.data = self.enc_rv[i]
        self.exp_avg = torch.zeros(self.P, device=dev)
        self.exp_avg_sq = torch.zeros(self.P, device=dev)

    def _build_tables(self) -> None:
        dev, T = self.device, self.T
        K = len(self.names)
        self.K = K
        self.t_off = torch.tensor([self.off[n] for n in self.names], dtype=torch.int64, device=dev)
        self.t_len = torch.tensor([self.numel[n] for n in self.names], dtype=torch.int32, device=dev)
        has = np.zeros((K, MAXT), dtype=np.uint8)
        lr = np.zeros(K, dtype=np.float32)
        for k, n in enumerate(self.names):
            lr[k] = DEFAULT_LR
            if n.startswith("input_encoders."):
                for t, task in enumerate(self.tasks):
                    has[k, t] = task != "node_feat_mask"        # NFM runs the encoder under no_grad
            elif n == "mask_token":
                for t, task in enumerate(self.tasks):
                    has[k, t] = task == "node_feat_mask"
            elif n.startswith("gnn_backbone."):
                has[k, :T] = 1
            elif n.startswith("heads."):
                for t, task in enumerate(self.tasks):
                    if f"heads.{task}" in n:
                        has[k, t] = 1
                        lr[k] = TASK_SPECIFIC_LR[task]
        self.has_static = has
        self.has = torch.from_numpy(has.copy()).to(dev)
        self.lr = torch.from_numpy(lr).to(dev)
        self.wd = torch.full((K,), DEFAULT_WEIGHT_DECAY, dtype=torch.float32, device=dev)
        self.steps = torch.zeros(K, dtype=torch.float32, device=dev)
        self.name_index = {n: k for k, n in enumerate(self.names)}

    def _alloc(self) -> None:
        dev, R = self.device, self.max_rows
        f = lambda *shape: torch.empty(*shape, device=dev)
        self.task_grads = torch.zeros(self.T, self.P, device=dev)
        self.final_grad = torch.zeros(self.P, device=dev)
        self.normsq = torch.zeros(1, device=dev)
        self.metrics = torch.zeros(2, dtype=torch.int32, device=dev)
        self.flags = torch.zeros(self.K, dtype=torch.int32, device=dev)
        self.mt_ws = torch.empty(self.lib.gmp_mt_workspace_bytes(self.K), dtype=torch.uint8, device=dev)
        Lr = GNN_NUM_LAYERS
        self.z0 = f(R, H)
        self.h = [f(R, H) for _ in range(Lr + 1)]
        self.a = [f(R, H) for _ in range(Lr)]
        self.z1 = [f(R, 2 * H) for _ in range(Lr)]
        self.r1 = [f(R, 2 * H) for _ in range(Lr)]
        self.z2 = [f(R, H) for _ in range(Lr)]
        self.S_MAX = 64
        self.stat = {k: f(Lr, self.S_MAX, c) for k, c in (("m1", 2 * H), ("s1", 2 * H), ("m2", H), ("s2", H))}
        self.enc_mean, self.enc_rstd = f(self.S_MAX, H), f(self.S_MAX, H)
        self.gA, self.gB = f(R, H), f(R, H)                  # ping-pong [R,256] gradients
        self.ga = f(R, H)                                    # gradient w.r.t. a layer's aggregated input (own buffer: activations stay intact)
        self.gW = f(R, 2 * H)                                # [R,512] gradients
        self.gW2 = f(R, 2 * H)
        self.gB2, self.gW3 = f(R, H), f(R, 2 * H)
        # per-layer g_u / g_z1 of the native executor's backward: aux never holds main back (gnnmp_step.h)
        self.gu_l, self.gz1_l = [f(R, H) for _ in range(Lr)], [f(R, 2 * H) for _ in range(Lr)]
        self.rowdot = f(Lr * R)                              # one slice per backward layer for the native executor (gnnmp_step.h)
        self.gemm_ws = torch.empty(48 << 20, dtype=torch.uint8, device=dev)     # slice partials of the grouped weight-gradient GEMMs (aux | second weight-gradient stream) | encoder backward (main)
        # the task heads are independent of each other: each has its own scratch and they share four streams
        # The runtime multiplexes HIP streams onto 4 hardware queues (GPU_MAX_HW_QUEUES; raising it made the step 2.4x
        # slower), and a per-task stream layout left three task heads serialised on one queue (profiles/README.md).  So:
        # exactly four streams -- main, aux and two more -- and the heads are packed onto them by estimated chain length
        # (longest first onto the least-loaded stream; main and aux are idle while the heads run).
        # ... and which hardware queue a stream lands on is measured, not assumed (streams.py): a kernel trace showed two of the
        # four streams sharing the main stream's queue, their heads running behind main's instead of beside them.
        from . import streams as ST
        self.aux_stream, *extra = ST.concurrent_streams(dev, 3)
        # Cross-stream dependencies inside the native step are carried by GATES (a sleeping wave on a flag word, csrc/streams.hip)
        # instead of events when all four streams were measured on hardware queues of their own: a queue parked on an event wait
        # costs every running queue ~2 us per kernel boundary, and with the host several steps ahead two or three queues were
        # parked most of the time (scripts/diag_blocked_queues.py).  In one in-order queue a gate ahead of its opener would never
        # open, hence the condition.  GMP_STEP_GATES=0 keeps the events (needed under tools that serialise kernels, e.g. --pmc).
        self.sync_flags = torch.zeros(64, dtype=torch.int32, device=dev)
        self._epoch = 0
        self.use_gates = bool(ST.last_report.get("calibrated") and ST.last_report.get("own_queue") == 3
                              and os.environ.get("GMP_STEP_GATES", "1") != "0")
        # PCGrad (and the data-parallel exchange) part by part beside the backward: needs the gates and the native executor
        # (opt-in, GMP_OPT_OVERLAP=1: measured 1.58 against 1.54 ms/step on one GPU -- 28 small launches on the exchange stream beside
        # the backward and a longer hand-over chain at the end cost more than the 58 us of Gram / solve / combine they take off it)
        self._parts_ok = (self.use_gates and os.environ.get("GMP_DP_OVERLAP", "1") != "0" and os.environ.get("GMP_OPT_OVERLAP", "0") == "1")
        self.comm_stream = extra[1]        # data-parallel exchange beside the backward (the head streams are idle by then)
        bins = [[0.0, extra[0]], [0.0, extra[1]], [0.0, self.aux_stream], [0.0, None]]      # None = the main stream
        self.task_streams = [None] * self.T
        for ti in sorted(range(self.T), key=lambda i: -HEAD_CHAIN_US.get(self.tasks[i], 100.0)):
            b = min(bins, key=lambda x: x[0])
            b[0] += HEAD_CHAIN_US.get(self.tasks[ti], 100.0)
            self.task_streams[ti] = b[1]
        self.task_gemm_ws = [torch.empty(24 << 20, dtype=torch.uint8, device=dev) for _ in range(self.T)]
        self.task_loss_ws = [torch.empty(self.lib.gmp_loss_workspace_bytes(R * H), dtype=torch.uint8, device=dev) for _ in range(self.T)]
        self._cur_gemm_ws = self.gemm_ws
        # one slice per BatchNorm (11) + spare, each large enough for the long-segment regime (segments up to R rows): the
        # native executor keeps every BN backward's per-segment sums until the aux stream has reduced them per task
        self.bn_ws = torch.empty(12 * self.lib.gmp_bn_workspace_bytes(R, 2 * H, self.S_MAX, R), dtype=torch.uint8, device=dev)
        self.csr_ws = torch.empty(self.lib.gmp_csr_build_workspace_bytes(R, self.max_edges), dtype=torch.uint8, device=dev)
        i32 = lambda n: torch.empty(n, dtype=torch.int32, device=dev)
        self.csr = [i32(R + 1), i32(self.max_edges), i32(self.max_edges), i32(R + 1), i32(self.max_edges), i32(self.max_edges)]
        self.csr_status = i32(1)
        self.lp_csr_status = i32(1)
        self.lp_csr_ws = torch.empty(self.lib.gmp_csr_build_workspace_bytes(R, self.max_edges), dtype=torch.uint8, device=dev)
        self.lp_csr = [i32(R + 1), i32(self.max_edges), i32(self.max_edges), i32(R + 1), i32(self.max_edges), i32(self.max_edges)]
        self.loss_sums = torch.zeros(MAXT, device=dev)       # per-task loss SUMS of the last step
        self.loss_ws = torch.empty(self.lib.gmp_loss_workspace_bytes(R * H), dtype=torch.uint8, device=dev)
        # head workspaces (rows bounded by max_rows / edges)
        self.KMAX = self.max_edges
        self.hd = {k: f(n, c) for k, (n, c) in {
            "nfm_in": (R, H), "nfm_y1": (R, H), "nfm_d1": (R, H), "nfm_y2": (R, H), "nfm_tgt": (R, H), "nfm_g": (R, H), "nfm_g1": (R, H), "nfm_gin": (R, H),
            "lp_feat": (self.KMAX, 3 * H), "lp_y1": (self.KMAX, H), "lp_d1": (self.KMAX, H), "lp_gy1": (self.KMAX, H),
            "lp_gfeat": (self.KMAX, 3 * H), "lp_ghs": (self.KMAX, H), "lp_ghd": (self.KMAX, H),
            "nc_in": (2 * R, H), "nc_y1": (2 * R, H), "nc_d1": (2 * R, H), "nc_z": (2 * R, 128), "nc_gz": (2 * R, 128), "nc_g1": (2 * R, H), "nc_gin": (2 * R, H),
            "gc_mean": (1024, H), "gc_max": (1024, H), "gc_in": (1024, 2 * H), "gc_y1": (1024, H), "gc_d1": (1024, H), "gc_z": (1024, 128),
            "gc_gz": (1024, 128), "gc_g1": (1024, H), "gc_gin": (1024, 2 * H), "gc_gmean": (1024, H), "gc_gmax": (1024, H),
            "da_in": (1024, H), "da_y1": (1024, DA_HIDDEN), "da_d1": (1024, DA_HIDDEN), "da_logits": (1024, 8), "da_glogits": (1024, 8),
            "da_g1": (1024, DA_HIDDEN), "da_gin": (1024, H),
            "gp_in": (1024, H), "gp_y1": (1024, 2 * H), "gp_d1": (1024, 2 * H), "gp_y2": (1024, 16), "gp_g2": (1024, 16), "gp_g1": (1024, 2 * H), "gp_gin": (1024, H),
        }.items()}
        # (lp_lab: upload set below; merged rows keep two scores each -- one per ordered row of the reference's list)
        self.lp_y2, self.lp_p, self.lp_gp, self.lp_gy2 = f(2 * self.KMAX), f(2 * self.KMAX), f(self.KMAX), f(2 * self.KMAX)
        self.gp_y2 = f(1024, GRAPH_PROPERTY_DIM)
        self.gp_g2 = f(1024, GRAPH_PROPERTY_DIM)
        self.ntx_ws = [torch.empty(self.lib.gmp_nt_xent_grouped_workspace_bytes(self.D, 512, 128), dtype=torch.uint8, device=dev) for _ in range(2 * self.D)]
        # packed per-step index uploads (pinned staging)
        self.i32_cap, self.i64_cap = 4 * R + 8 * self.S_MAX + 65536 + 2 * self.max_edges, 4 * self.max_edges + 8 * R
        # The host runs several steps ahead of the GPU (nothing in a step syncs), so the pinned staging buffers
        # form a ring: a slot is refilled only after the copy that last read it has completed (event per slot).
        self.STAGES = 4
        self.stage = [{"pin32": torch.empty(self.i32_cap, dtype=torch.int32).pin_memory(),
                       "pin64": torch.empty(self.i64_cap, dtype=torch.int64).pin_memory(),
                       "pinf": torch.empty(64 + self.KMAX, dtype=torch.float32).pin_memory(),
                       "event": None} for _ in range(self.STAGES)]
        # Two sets of upload destinations, alternating by step: the native executor uploads step t+1's arrays on the AUX stream,
        # behind aux's last work of step t and beside main's tail / optimizer, into the set step t is not using.
        self._up_sets = [(torch.empty(self.i32_cap, dtype=torch.int32, device=dev), torch.empty(self.i64_cap, dtype=torch.int64, device=dev),
                          torch.zeros(64, device=dev), f(self.KMAX)) for _ in range(2)]
        self.dev32, self.dev64, self.scal, self.lp_lab = self._up_sets[0]

    # ------------------------------------------------------------------ host: draw + plan one step
    def draw(self, inp: StepInputs, gen: torch.Generator) -> Dict[str, object]:
        """All RNG of one step as index arrays, tasks in ACTIVE_TASKS order, domains in dict order.
        rng_mode 'reference': the reference's exact draw sequence from the caller's CPU torch.Generator (bit-identical
        indices for an equal generator state; per-graph Python loop).  rng_mode 'vectorized': the same distributions
        drawn for all graphs of a domain at once with numpy (different stream, ~10x less host time)."""
        if self.rng_mode == "vectorized":
            return self._draw_vectorized(inp, gen)
        if self.rng_mode == "device":
            return self.collect_draws(inp, self.enqueue_draws(inp))
        art: Dict[str, object] = {}
        host = {d: inp.host[d] for d in self.domains}
        H = hostdraw()
        if H is not None and hasattr(H, "draw_step"):
            # one native call for the whole step (GIL released once, generator locked once): the launcher thread's Python is not held up
            # by sixteen hand-overs per step
            if self._neg_native is None:
                self._neg_native = H.PyRandom()
                self._neg_native.setstate(torch.tensor(self.neg_rng.getstate()[1], dtype=torch.long))
            kinds = [self.DRAWN_TASKS.index(t) for t in self.tasks if t in self.DRAWN_TASKS]
            doms = inp.__dict__.get("_draw_args")
            if doms is None:
                doms = inp.__dict__["_draw_args"] = [_host_tensors(b) + (int(b.x.size(1)),) for b in host.values()]
            art = Artefacts()
            art.raw = H.draw_step(kinds, doms, gen, self._neg_native)
            tasks, names = [t for t in self.tasks if t in self.DRAWN_TASKS], list(host)

            def fill(a: Artefacts) -> None:
                for t, row in zip(tasks, a.raw):
                    out = {}
                    for d, r in zip(names, row):
                        if r is None:
                            out[d] = None
                        elif len(r) == 0:
                            out[d] = _EMPTY_ART[t]()
                        elif len(r) == 1:
                            out[d] = r[0].numpy()
                        else:
                            out[d] = tuple(ViewArrays(*(x.numpy() for x in r[5 * v:5 * v + 3]),
                                                      r[5 * v + 3].numpy().view(np.uint64) if r[5 * v + 3].numel() else None, r[5 * v + 4].numpy())
                                           for v in range(2))
                    dict.__setitem__(a, t, out)

            art.fill = fill
            return art
        if H is not None:
            args = {d: _host_tensors(b) for d, b in host.items() if b.num_graphs}
        for t in self.tasks:
            if t == "node_feat_mask":
                art[t] = {d: (_EMPTY_ART[t]() if not b.num_graphs else
                              H.mask_indices(args[d][0], gen).numpy() if H is not None else draw_mask_indices(b.ptr_host, gen).numpy())
                          for d, b in host.items()}
            elif t == "link_pred":
                art[t] = {d: (_EMPTY_ART[t]() if not b.num_graphs else self._negatives(b)) for d, b in host.items()}
            elif t in ("node_contrast", "graph_contrast"):
                art[t] = {d: (_EMPTY_ART[t]() if b.num_graphs == 0 else
                              self._draw_views(b, gen) if (t == "node_contrast" or b.num_graphs >= 2) else None)
                          for d, b in host.items()}
        return art

    DRAWN_TASKS = ("node_feat_mask", "link_pred", "node_contrast", "graph_contrast")

    def draw_task(self, t: str, b: Batch, gen: torch.Generator):
        """The reference-order draws of ONE task for ONE domain batch -- what task.compute_loss({domain: batch}, generator)
        consumes from the generator (validation walks tasks x domains x batches in that order, pretrain.py:211-221)."""
        if b.num_graphs == 0:
            return _EMPTY_ART[t]()
        H = hostdraw()
        if t == "node_feat_mask":
            return (H.mask_indices(_host_tensors(b)[0], gen) if H is not None else draw_mask_indices(b.ptr_host, gen)).numpy()
        if t == "link_pred":
            return self._negatives(b)
        if t in ("node_contrast", "graph_contrast"):
            return self._draw_views(b, gen) if (t == "node_contrast" or b.num_graphs >= 2) else None
        raise KeyError(t)

    @staticmethod
    def empty_art(t: str):
        return _EMPTY_ART[t]()

    def _negatives(self, b: Batch) -> np.ndarray:
        """batched_negative_sampling(to_undirected(pos), batch, num_neg_samples=E) of one domain batch from the engine's Python-random
        stream: the native CPython-compatible MT19937 (csrc_host/hostdraw.cpp PyRandom, seeded from self.neg_rng's state at first
        use and the owner of the stream from then on) when the module is built, pretrain/tasks.py otherwise -- same negatives."""
        H = hostdraw()
        if H is None or not hasattr(H, "PyRandom"):
            return sample_negative_edges(b, self.neg_rng).numpy()
        if self._neg_native is None:
            self._neg_native = H.PyRandom()
            self._neg_native.setstate(torch.tensor(self.neg_rng.getstate()[1], dtype=torch.long))
        return self._neg_native.negative_edges(*_host_tensors(b)).numpy()

    def rng_state(self) -> Dict[str, object]:
        """Everything random the engine owns besides the caller's torch.Generator, as plain Python data (pretrain() stores it in the checkpoint
        next to the generator state): the link-prediction negatives' Python-random stream and the device-draw sequence numbers."""
        return {"neg_rng": self.sync_neg_rng().getstate(), "draw_seq": dict(getattr(self, "_draw_seq", {True: 0, False: 0})),
                "step_count": int(self.step_count)}

    def set_rng_state(self, st: Dict[str, object]) -> None:
        v, key, g = st["neg_rng"]
        self.neg_rng.setstate((int(v), tuple(int(x) for x in key), g))
        if self._neg_native is not None:
            self._neg_native.setstate(torch.tensor(self.neg_rng.getstate()[1], dtype=torch.long))
        self._draw_seq = {bool(k): int(n) for k, n in st.get("draw_seq", {}).items()} or {True: 0, False: 0}
        self.step_count = int(st.get("step_count", self.step_count))

    def sync_neg_rng(self) -> random.Random:
        """Write the native stream's state back into self.neg_rng (checkpointing / tests) and return it."""
        if self._neg_native is not None:
            st = self.neg_rng.getstate()
            self.neg_rng.setstate((st[0], tuple(int(v) for v in self._neg_native.getstate().tolist()), st[2]))
        return self.neg_rng

    @staticmethod
    def _draw_views(b: Batch, gen: torch.Generator) -> Tuple[ViewArrays, ViewArrays]:
        """Same draws as GraphAugmentor.create_two_views (augmentations.py:88-111), kept as index arrays: the native module
        when it is built, the Python loop below otherwise (bit-identical, tests/test_hostdraw.py)."""
        H = hostdraw()
        if H is None:
            return StepEngine._draw_views_python(b, gen)
        r = H.draw_views(*_host_tensors(b), int(b.x.size(1)), gen)
        out = []
        for vi in range(2):
            rows, edges, vptr, rowmask, common = (t.numpy() for t in r[5 * vi:5 * vi + 5])
            out.append(ViewArrays(rows, edges, vptr, rowmask.view(np.uint64) if rowmask.size else None, common))
        return out[0], out[1]

    @staticmethod
    def _draw_views_python(b: Batch, gen: torch.Generator) -> Tuple[ViewArrays, ViewArrays]:
        ei = b.edge_index.numpy()
        F = b.x.size(1)
        acc = [dict(rows=[], edges=[], ptr=[0], masks=[], common=[]) for _ in range(2)]
        for g in range(b.num_graphs):
            s, e = b.ptr_host[g], b.ptr_host[g + 1]
            es, ee = b.edge_ptr_host[g], b.edge_ptr_host[g + 1]
            loc = ei[:, es:ee] - s
            pair = (_augment_one(e - s, loc, F, gen), _augment_one(e - s, loc, F, gen))
            flags = np.zeros((2, e - s), dtype=bool)
            flags[0, pair[0].kept] = True
            flags[1, pair[1].kept] = True
            for vi, v in enumerate(pair):
                a = acc[vi]
                base = a["ptr"][-1]
                a["rows"].append(v.kept + s)
                a["edges"].append(v.edges + base)
                a["common"].append(np.flatnonzero(flags[1 - vi, v.kept]) + base)
                m = 0
                if v.masked_cols is not None:
                    for c in v.masked_cols:
                        m |= 1 << int(c)
                a["masks"].append(np.full(len(v.kept), m, dtype=np.uint64))
                a["ptr"].append(base + len(v.kept))
        out = []
        for a in acc:
            rm = np.concatenate(a["masks"])
            out.append(ViewArrays(np.concatenate(a["rows"]), np.concatenate(a["edges"], axis=1), np.asarray(a["ptr"], dtype=np.int64),
                                  rm if rm.any() else None, np.concatenate(a["common"])))
        return out[0], out[1]

    # ---- device draws (csrc/augment.hip): masks and views built on the GPU, one ticket per step ----------------------------
    DRAW_SLOTS = 8

    def enqueue_draws(self, inp: StepInputs) -> "DrawTicket":
        """Enqueue the device-side draws of ONE step (rng_mode 'device'): node-feature-masking indices and the two augmented views
        of every contrastive (task, domain) pair, by gmp_aug_node_masks / gmp_aug_two_views on the aux stream, followed by one copy
        kernel that writes the results into a pinned host slot and a one-thread kernel that raises the slot's flag.  Safe to call
        from the prefetch thread while the launcher thread enqueues steps: the kernels depend on nothing a step computes, and
        wherever they land between the aux stream's packets they run at most one step later.  Returns the ticket
        collect_draws() waits on."""
        if not hasattr(self, "_draw_slots"):
            self._draw_slots, self._draw_count, self._draw_ws, self._draw_graveyard = [], 0, None, []
            self._draw_seq = {True: 0, False: 0}
        k = self._draw_count                       # ticket number: ring slot and flag value (every enqueue, probes and evaluation included)
        self._draw_count += 1
        # The random stream is a function of (engine seed, mode, how many inputs of that mode were drawn), not of the ticket number:
        # evaluation passes and verify_gates' probe steps (which rewinds its own) leave the training sequence where it was
        mode = bool(self.model.training)
        seq = self._draw_seq[mode]
        self._draw_seq[mode] = seq + 1
        self._bury()                               # free outgrown buffers the aux stream has passed
        lay, off = [], 0

        def take(nbytes: int) -> int:
            nonlocal off
            o = off
            off += (nbytes + 15) // 16 * 16
            return o

        for t in self.tasks:
            if t not in ("node_feat_mask", "node_contrast", "graph_contrast"):
                continue
            for d in self.domains:
                hb = inp.host[d]
                G, N, E = hb.num_graphs, hb.num_nodes, hb.num_edges
                if G == 0 or (t == "graph_contrast" and G < 2):
                    continue
                _, _, _, _, _, vptr_h, moff_h = inp.dev_graph(d)
                if t == "node_feat_mask":
                    lay.append((t, d, {"idx": take(8 * int(moff_h[-1])), "m": int(moff_h[-1])}))
                else:
                    V = int(vptr_h[-1])
                    lay.append((t, d, {"rows": (take(8 * V), take(8 * V)), "mask": (take(8 * V), take(8 * V)),
                                       "edges": (take(16 * max(E, 1)), take(16 * max(E, 1))), "common": (take(8 * V), take(8 * V)),
                                       "counts": take(20 * G), "totals": take(32), "V": V, "E": E, "vptr": vptr_h}))
        total = max(off, 16)
        slot_id = k % self.DRAW_SLOTS
        while len(self._draw_slots) <= slot_id:
            self._draw_slots.append(None)
        slot = self._draw_slots[slot_id]
        if slot is None or slot["dev"].numel() < total:
            cap = max(total * 2, 1 << 20)
            self._bury(slot)
            slot = self._draw_slots[slot_id] = {"dev": torch.empty(cap, dtype=torch.uint8, device=self.device),
                                                "pin": torch.empty(cap, dtype=torch.uint8).pin_memory(),
                                                "flag": torch.zeros(4, dtype=torch.int32).pin_memory()}
        lib, aux = self.lib, self.aux_stream.cuda_stream
        base = slot["dev"].data_ptr()
        seed = (self.seed * 1000003 + 0x5bd1e995 * (seq + 1) + (0 if mode else 0x9E3779B97F4A7C15)) & (2 ** 64 - 1)
        if True:
            # every (task, domain) job of the step in three launches (masks | views | emit): on the aux stream beside a running step the
            # twenty per-job launches cost that step 0.3 ms
            mjobs, vjobs, nmax_all, emax_all, ws_need = [], [], 1, 0, 0
            for (t, d, o) in lay:
                hb = inp.host[d]
                nmax_all = max(nmax_all, max(int(b - a) for a, b in zip(hb.ptr_host[:-1], hb.ptr_host[1:])))
                if t != "node_feat_mask":
                    emax_all = max([emax_all] + [int(b - a) for a, b in zip(hb.edge_ptr_host[:-1], hb.edge_ptr_host[1:])])
                    o["ws_off"], o["ws_bytes"] = ws_need, (lib.gmp_aug_workspace_bytes(hb.num_nodes, o["E"], hb.num_graphs) + 255) // 256 * 256
                    ws_need += o["ws_bytes"]
            if ws_need and (self._draw_ws is None or self._draw_ws.numel() < ws_need):
                # (kernels of earlier tickets may still be using the old one on the aux stream: keep it alive.  One region per JOB -- the
                # jobs of a ticket run concurrently; tickets follow each other on the aux stream and share the regions)
                self._bury(self._draw_ws)
                self._draw_ws = torch.empty(max(2 * ws_need, 4 * lib.gmp_aug_workspace_bytes(self.max_rows, self.max_edges, 1024)),
                                            dtype=torch.uint8, device=self.device)
            ws_base = self._draw_ws.data_ptr() if ws_need else 0
            for (t, d, o) in lay:
                ptr, eptr, ei, vptr, optr, _, _ = inp.dev_graph(d)
                hb = inp.host[d]
                sid = 16 * self.tasks.index(t) + 2 * self.domains.index(d) * len(self.tasks) * 16
                if t == "node_feat_mask":
                    if o["m"]:
                        mjobs.append(L.AugMasksJob(ptr.data_ptr(), optr.data_ptr(), hb.num_graphs, sid, base + o["idx"]))
                else:
                    vjobs.append(L.AugViewsJob(ptr.data_ptr(), eptr.data_ptr(), ei.data_ptr(), hb.num_nodes, o["E"], vptr.data_ptr(), hb.num_graphs,
                                               int(hb.x.size(1)), sid, base + o["rows"][0], base + o["rows"][1], base + o["mask"][0], base + o["mask"][1],
                                               base + o["edges"][0], base + o["edges"][1], max(o["E"], 1), base + o["common"][0], base + o["common"][1],
                                               base + o["counts"], base + o["totals"], ws_base + o["ws_off"], o["ws_bytes"]))
            if mjobs:
                self._chk(lib.gmp_aug_node_masks_batch((L.AugMasksJob * len(mjobs))(*mjobs), len(mjobs), nmax_all, seed, aux), "aug_node_masks_batch")
            if vjobs:
                self._chk(lib.gmp_aug_two_views_batch((L.AugViewsJob * len(vjobs))(*vjobs), len(vjobs), nmax_all, emax_all, seed, aux), "aug_two_views_batch")
            src = (C.c_void_p * 1)(base)
            dst = (C.c_void_p * 1)(slot["pin"].data_ptr())
            self._chk(lib.gmp_upload(1, src, dst, (C.c_int64 * 1)((total + 15) // 16 * 16), aux), "draw results -> pinned host")
            self._chk(lib.gmp_gate_open(slot["flag"].data_ptr(), k + 1, aux), "draw flag")
        return DrawTicket(slot, lay, k + 1)

    def _bury(self, obj=None) -> None:
        """Outgrown draw slots / workspaces may still be in use by tickets already on the aux stream: each is kept with an event recorded
        there and dropped once the stream has passed it (they used to live as long as the engine)."""
        if obj is not None:
            ev = torch.cuda.Event()
            ev.record(self.aux_stream)
            self._draw_graveyard.append((obj, ev))
        self._draw_graveyard = [(o, e) for o, e in self._draw_graveyard if not e.query()]

    def collect_draws(self, inp: StepInputs, ticket: "DrawTicket") -> Dict[str, object]:
        """Wait for a ticket's flag (a word in pinned host memory the GPU sets behind its copy: no HIP call, no stream sync) and wrap
        the slot's arrays as the step's artefacts.  Link-prediction negatives stay on the host (mostly "every non-edge", no draw)."""
        import time as _t
        flag = ticket.slot["flag"].numpy()
        t_end = _t.time() + 120.0
        while int(flag[0]) < ticket.epoch:
            if _t.time() > t_end:
                raise L.GnnmpError("engine: device draws did not arrive within two minutes")
            _t.sleep(2e-5)
        buf = ticket.slot["pin"].numpy()
        art: Dict[str, object] = {}
        for t in self.tasks:
            if t == "link_pred":
                art[t] = {d: (_EMPTY_ART[t]() if not inp.host[d].num_graphs else self._negatives(inp.host[d])) for d in self.domains}
            elif t in ("node_feat_mask", "node_contrast", "graph_contrast"):
                art[t] = {d: _EMPTY_ART[t]() for d in self.domains}        # domains without a drawn piece (no graphs / too few)
        for (t, d, o) in ticket.layout:
            if t == "node_feat_mask":
                art[t][d] = buf[o["idx"]:o["idx"] + 8 * o["m"]].view(np.int64).copy()
            else:
                tot = buf[o["totals"]:o["totals"] + 20].view(np.int32)
                views = []
                for v in range(2):
                    V, e = o["V"], int(tot[v])
                    rows = buf[o["rows"][v]:o["rows"][v] + 8 * V].view(np.int64).copy()
                    ed = buf[o["edges"][v]:o["edges"][v] + 16 * max(o["E"], 1)].view(np.int64).reshape(2, -1)[:, :e].copy()
                    rm = buf[o["mask"][v]:o["mask"][v] + 8 * V].view(np.uint64).copy() if tot[3 + v] else None
                    cm = buf[o["common"][v]:o["common"][v] + 8 * int(tot[2])].view(np.int64).copy()
                    views.append(ViewArrays(rows, ed, o["vptr"], rm, cm))
                art[t][d] = (views[0], views[1])
        return art

    # ---- vectorized draws (same distributions, numpy stream) --------------------------------------------
    def _np_rng(self, gen: torch.Generator) -> np.random.Generator:
        if self._nprng is None:
            self._nprng = np.random.default_rng(int(torch.randint(0, 2 ** 62, (1,), generator=gen).item()))
        return self._nprng

    @staticmethod
    def _rank_in_group(keys: np.ndarray, group: np.ndarray, gptr: np.ndarray) -> np.ndarray:
        """rank of every element among the elements of its group when ordered by key (groups are contiguous)."""
        order = np.argsort(group + keys)          # keys in [0,1): one float sort orders by (group, key)
        rank = np.empty(len(keys), dtype=np.int64)
        rank[order] = np.arange(len(keys)) - gptr[group[order]]
        return rank

    def _static(self, inp: StepInputs, d: str) -> Dict[str, np.ndarray]:
        """Per-input structures that do not depend on the step's RNG (cached on the StepInputs)."""
        cache = inp.__dict__.setdefault("_static", {})
        if d not in cache:
            hb = inp.host[d]
            ptr = np.asarray(hb.ptr_host, dtype=np.int64)
            n = np.diff(ptr)
            ei = hb.edge_index.numpy()
            eptr = np.asarray(hb.edge_ptr_host, dtype=np.int64)
            st = {"ptr": ptr, "n": n, "node_graph": np.repeat(np.arange(len(n)), n), "ei": ei, "eptr": eptr,
                  "edge_graph": np.repeat(np.arange(len(n)), np.diff(eptr)), "F": hb.x.size(1)}
            # candidate negative edges: ordered pairs (i, j), i != j, not adjacent in either direction
            cs, cd, cg = [], [], []
            for g in range(len(n)):
                adj = np.zeros((n[g], n[g]), dtype=bool)
                loc = ei[:, eptr[g]:eptr[g + 1]] - ptr[g]
                adj[loc[0], loc[1]] = True
                adj[loc[1], loc[0]] = True
                np.fill_diagonal(adj, True)
                i, j = np.nonzero(~adj)
                cs.append(i + ptr[g]); cd.append(j + ptr[g]); cg.append(np.full(len(i), g))
            st["cand"] = np.stack([np.concatenate(cs), np.concatenate(cd)])
            st["cand_graph"] = np.concatenate(cg)
            st["cand_ptr"] = np.concatenate([[0], np.cumsum(np.bincount(st["cand_graph"], minlength=len(n)))])
            cache[d] = st
        return cache[d]

    def _view_vectorized(self, st, rng: np.random.Generator, keep: np.ndarray) -> Tuple[np.ndarray, ...]:
        """Edge drop + attribute mask for one view of a whole domain batch, given its node keep mask."""
        new_id = np.cumsum(keep) - 1
        rows = np.flatnonzero(keep)
        ei, eg = st["ei"], st["edge_graph"]
        em = keep[ei[0]] & keep[ei[1]]
        edges = new_id[ei[:, em]]
        eg = eg[em]
        G = len(st["n"])
        ecount = np.bincount(eg, minlength=G)
        coin = (rng.random(G) < 0.2) & (ecount >= 3)
        if coin.any():
            eptr = np.concatenate([[0], np.cumsum(ecount)])
            rank = self._rank_in_group(rng.random(len(eg)), eg, eptr)
            keep_e = ecount - np.maximum(1, (ecount * 0.2).astype(np.int64))
            ekeep = ~coin[eg] | (rank < keep_e[eg])
            edges = edges[:, ekeep]
        kept_n = np.bincount(st["node_graph"][keep], minlength=G)
        ptr = np.concatenate([[0], np.cumsum(kept_n)])
        rowmask = None
        F = st["F"]
        coin2 = rng.random(G) < 0.2
        if F >= 3 and coin2.any():
            m = max(1, int(F * 0.2))
            cols = np.argsort(rng.random((G, F)), axis=1)[:, :m]
            bits = np.bitwise_or.reduce(np.uint64(1) << cols.astype(np.uint64), axis=1)
            bits[~coin2] = 0
            rowmask = np.repeat(bits, kept_n)
        return rows, edges, ptr, rowmask, new_id

    def _draw_vectorized(self, inp: StepInputs, gen: torch.Generator) -> Dict[str, object]:
        rng = self._np_rng(gen)
        art: Dict[str, object] = {}
        for t in self.tasks:
            if t == "graph_prop":
                continue
            out = art[t] = {}
            for d in self.domains:
                if inp.host[d].num_graphs == 0:          # a domain absent from this step (single-domain validation passes)
                    out[d] = _EMPTY_ART[t]()
                    continue
                st = self._static(inp, d)
                n, ng, ptr = st["n"], st["node_graph"], st["ptr"]
                if t == "node_feat_mask":
                    k = np.where(n >= 3, np.maximum(1, (n * 0.15).astype(np.int64)), 0)
                    rank = self._rank_in_group(rng.random(len(ng)), ng, ptr)
                    out[d] = np.flatnonzero(rank < k[ng])
                elif t == "link_pred":
                    cptr = st["cand_ptr"]
                    # PyG applies num_neg_samples = E of the WHOLE batch to every graph: min(E, its non-edges) each
                    want = np.minimum(int(st["eptr"][-1]), np.diff(cptr))
                    pick = [cptr[g] + (np.arange(want[g]) if want[g] == cptr[g + 1] - cptr[g] else
                                       rng.choice(cptr[g + 1] - cptr[g], size=want[g], replace=False))
                            for g in range(len(n)) if want[g] > 0]
                    out[d] = st["cand"][:, np.concatenate(pick)] if pick else np.zeros((2, 0), dtype=np.int64)
                else:
                    if t == "graph_contrast" and len(n) < 2:
                        out[d] = None
                        continue
                    keep_n = np.where(n >= 3, n - np.maximum(1, (n * 0.2).astype(np.int64)), n)
                    keeps = [self._rank_in_group(rng.random(len(ng)), ng, ptr) < keep_n[ng] for _ in range(2)]
                    parts = [self._view_vectorized(st, rng, k) for k in keeps]
                    both = keeps[0] & keeps[1]
                    out[d] = tuple(ViewArrays(rows, edges, vptr, rowmask, new_id[both]) for (rows, edges, vptr, rowmask, new_id) in parts)
        return art

    # ---- plan: lay the step out as segments, everything as flat arrays ----------------------------------
    def _plan_native(self, inp: StepInputs, raw) -> StepPlan:
        """hostdraw.plan_step: the layout below in one native call with the GIL released (array for array the same: tests/test_hostdraw.py)."""
        from ._step_desc import TASK_KIND
        D = self.domains
        doms = inp.__dict__.get("_draw_args")
        if doms is None:
            doms = inp.__dict__["_draw_args"] = [_host_tensors(inp.host[d]) + (int(inp.host[d].x.size(1)),) for d in D]
        r = hostdraw().plan_step([TASK_KIND[t] for t in self.tasks], doms, [int(inp.row_off[d]) for d in D], raw, self.lp_merge, self.fwd_ranges,
                                 H, GRAPH_PROPERTY_DIM)
        p = StepPlan()
        p.cat32, p.cat64 = r["cat32"].numpy(), r["cat64"].numpy()
        p.lay32, p.lay64 = {n: o for n, o, _ in r["lay32"]}, {n: o for n, o, _ in r["lay64"]}
        p.a32, p.a64 = _Views(p.cat32, r["lay32"]), _Views(p.cat64, r["lay64"])
        p.seg_ptr, p.seg_dom, p.seg_task, p.task_row = r["seg_ptr"], r["seg_dom"], r["seg_task"], r["task_row"]
        p.sizes = {t: int(v) for t, v in zip(self.tasks, r["sizes"])}
        p.skipped = [(int(ti), D[int(di)]) for ti, di in r["skipped"]]
        p.N, p.S, p.E, p.max_seg, p.max_seg_edges, p.num_tiles = r["N"], r["S"], r["E"], r["max_seg"], r["max_seg_edges"], r["num_tiles"]
        p.fwd_cuts = list(zip(r["fwd_cut_seg"], r["fwd_cut_row"]))
        for k in ("nfm_rows", "nc_rows", "nc_n", "gc_rows", "gc_n", "gc_B", "gc_r0", "gc_M", "gp_rows", "gp_B", "gp_r0", "gp_M", "da_B", "da_r0", "da_M",
                  "lp_K", "lp_S", "lp_rows_end", "lp_max_rows", "lp_max_edges"):
            if k in r:
                setattr(p, k, r[k])
        if "lp_K" in r:
            p.lp_labels = r["lp_labels"].numpy()
            if p.lp_K > self.KMAX:
                raise L.GnnmpError("engine: too many link-prediction edges")
        if p.N > self.max_rows or p.E > self.max_edges or p.S > self.S_MAX:
            raise L.GnnmpError(f"step of {p.N} rows / {p.E} edges / {p.S} segments exceeds the engine capacity "
                               f"({self.max_rows}/{self.max_edges}/{self.S_MAX})")
        if p.cat32.size > getattr(self, "i32_cap", 1 << 62) or p.cat64.size > getattr(self, "i64_cap", 1 << 62):
            raise L.GnnmpError("engine: staging buffer too small")
        return p

    def plan(self, inp: StepInputs, art: Dict[str, object]) -> StepPlan:
        raw = getattr(art, "raw", None)
        if raw is not None and self.native_plan and hasattr(hostdraw(), "plan_step"):
            return self._plan_native(inp, raw)
        p, D = StepPlan(), self.domains
        seg_ptr, seg_dom, seg_task = [0], [], []
        src_rows, edges, rowmasks = [], [], []
        task_row = [0]
        a32: Dict[str, np.ndarray] = {}
        a64: Dict[str, np.ndarray] = {}
        sizes: Dict[str, int] = {}
        p.skipped = []

        def add_segment(ti: int, di: int, rows: np.ndarray, e_local: np.ndarray, rowmask: Optional[np.ndarray]) -> int:
            r0 = seg_ptr[-1]
            src_rows.append(rows)
            edges.append(e_local + r0)
            rowmasks.append((r0, rowmask))
            seg_ptr.append(r0 + len(rows)); seg_dom.append(di); seg_task.append(ti)
            return r0

        for ti, t in enumerate(self.tasks):
            if t in ("node_feat_mask", "link_pred", "graph_prop", "domain_adv"):
                r0s = []
                for di, d in enumerate(D):
                    hb, roff = inp.host[d], inp.row_off[d]
                    r0s.append(add_segment(ti, di, np.arange(roff, roff + hb.num_nodes), hb.edge_index.numpy(), None))
                if t == "node_feat_mask":
                    idx = [np.asarray(art[t][d], dtype=np.int64) + r0 for d, r0 in zip(D, r0s)]
                    rows = np.concatenate([[0], np.cumsum([len(i) for i in idx])]).tolist()
                    p.skipped += [(ti, d) for d, i in zip(D, idx) if len(i) == 0]
                    a64["nfm_idx"] = np.concatenate(idx)
                    p.nfm_rows = rows
                    sizes[t] = rows[-1] * H
                elif t == "link_pred":
                    eds, labs, npos, ordered, ords = [], [], [], 0, []
                    for d, r0 in zip(D, r0s):
                        hb = inp.host[d]
                        neg = np.asarray(art[t][d], dtype=np.int64)
                        ord_base = ordered
                        ordered += hb.edge_index.size(1) + neg.shape[1]
                        if self.lp_merge:
                            pairs, w, od = merge_mirrored_pairs(hb, neg, r0, ord_base)
                            eds.append(pairs); labs.append(w); ords.append(od)
                            npos.append((pairs.shape[1], 0))
                        else:
                            eds += [hb.edge_index.numpy() + r0, neg + r0]
                            npos.append((eds[-2].shape[1], eds[-1].shape[1]))
                            labs += [np.ones(npos[-1][0], dtype=np.float32), -np.ones(npos[-1][1], dtype=np.float32)]
                    e = np.concatenate(eds, axis=1)
                    lab = np.concatenate(labs)
                    if e.shape[1] > self.KMAX:
                        raise L.GnnmpError("engine: too many link-prediction edges")
                    a64["lp_edges"] = e
                    p.lp_labels, p.lp_K = lab, e.shape[1]
                    sizes[t] = ordered                    # the reference's count: BCE is a mean over its ordered list (tasks.py:120)
                    # block diagonal by domain: the decoder CSR is built one workgroup per (domain, orientation)
                    a32["lp_seg_ptr"] = np.asarray(r0s + [seg_ptr[-1]])
                    a32["lp_seg_eptr"] = np.concatenate([[0], np.cumsum([a + b for a, b in npos])])
                    if self.lp_merge:           # [2, K']: the ordered row(s) of the reference's list every merged row stands for
                        a32["lp_pos"] = np.concatenate(ords, axis=1) if ords else np.zeros((2, 0), dtype=np.int32)
                    p.lp_S, p.lp_rows_end = len(r0s), seg_ptr[-1]
                    p.lp_max_rows = int(np.diff(a32["lp_seg_ptr"]).max())
                    p.lp_max_edges = int(np.diff(a32["lp_seg_eptr"]).max())
                else:
                    starts, rows, labels = [], [0], []
                    for di, (d, r0) in enumerate(zip(D, r0s)):
                        ph = inp.host[d].ptr_host
                        starts += [r0 + v for v in ph[:-1]]
                        rows.append(rows[-1] + len(ph) - 1)
                        labels += [di] * (len(ph) - 1)
                    end = seg_ptr[-1]
                    ptr = np.asarray(starts + [end], dtype=np.int64)
                    if t == "graph_prop":
                        a32["gp_ptr"] = ptr
                        a64["gp_gid"] = np.repeat(np.arange(len(starts)), np.diff(ptr))
                        p.gp_rows, p.gp_B, p.gp_r0, p.gp_M = rows, len(starts), task_row[-1], end - task_row[-1]
                        sizes[t] = rows[-1] * GRAPH_PROPERTY_DIM
                    else:                                   # domain_adv: label = index of the graph's domain (tasks.py:333)
                        a32["da_ptr"] = ptr
                        a64["da_gid"] = np.repeat(np.arange(len(starts)), np.diff(ptr))
                        a64["da_labels"] = np.asarray(labels, dtype=np.int64)
                        p.da_B, p.da_r0, p.da_M = len(starts), task_row[-1], end - task_row[-1]
                        sizes[t] = len(starts)
            else:
                idx, rows, ns, starts = [], [0], [], []
                for di, d in enumerate(D):
                    views, roff = art[t][d], inp.row_off[d]
                    if views is None:
                        ns.append(0); rows.append(rows[-1]); p.skipped.append((ti, d))
                        continue
                    r0s = [add_segment(ti, di, v.rows + roff, v.edges, v.rowmask) for v in views]
                    if t == "node_contrast":
                        c1, c2 = views[0].common, views[1].common
                        n = len(c1) if (len(c1) >= 2 and len(c2) >= 2) else 0        # tasks.py:171-173
                        ns.append(n)
                        if n:
                            idx += [c1 + r0s[0], c2 + r0s[1]]
                        else:
                            p.skipped.append((ti, d))
                        rows.append(rows[-1] + 2 * n)
                    else:
                        for v, r0 in zip(views, r0s):
                            starts.append(v.ptr[:-1] + r0)
                        B = len(views[0].ptr) - 1
                        ns.append(B); rows.append(rows[-1] + 2 * B)
                if t == "node_contrast":
                    a64["nc_idx"] = np.concatenate(idx) if idx else np.zeros(0, dtype=np.int64)
                    p.nc_rows, p.nc_n = rows, ns
                else:
                    st = np.concatenate(starts) if starts else np.zeros(0, dtype=np.int64)
                    ptr = np.concatenate([st, [seg_ptr[-1]]])
                    a32["gc_ptr"] = ptr
                    a64["gc_gid"] = np.repeat(np.arange(len(st)), np.diff(ptr))
                    p.gc_rows, p.gc_n, p.gc_B, p.gc_r0, p.gc_M = rows, ns, len(st), task_row[-1], seg_ptr[-1] - task_row[-1]
                sizes[t] = rows[-1]
            task_row.append(seg_ptr[-1])
        p.seg_ptr, p.seg_dom, p.seg_task, p.task_row, p.sizes = seg_ptr, seg_dom, seg_task, task_row, sizes
        p.N, p.S = seg_ptr[-1], len(seg_dom)
        e_all = np.concatenate(edges, axis=1)
        p.E = e_all.shape[1]
        p.max_seg = max(b - a for a, b in zip(seg_ptr[:-1], seg_ptr[1:]))
        # the stacked forward runs as fwd_ranges row ranges on as many streams (gnnmp_step.h fwd_cut_*): cut k at the segment boundary nearest
        # k N / R, kept only while the cuts ascend strictly inside (0, N)
        p.fwd_cuts, R = [], self.fwd_ranges
        for k in range(1, R):
            if p.S <= 1:
                break
            cut = min(range(1, p.S), key=lambda i: abs(R * seg_ptr[i] - k * p.N))
            if (p.fwd_cuts[-1][1] if p.fwd_cuts else 0) < seg_ptr[cut] < p.N:
                p.fwd_cuts.append((cut, seg_ptr[cut]))
        if p.N > self.max_rows or p.E > self.max_edges or p.S > self.S_MAX:
            raise L.GnnmpError(f"step of {p.N} rows / {p.E} edges / {p.S} segments exceeds the engine capacity "
                               f"({self.max_rows}/{self.max_edges}/{self.S_MAX})")
        a32["seg_ptr"], a32["seg_dom"], a32["src_row"] = np.asarray(seg_ptr), np.asarray(seg_dom), np.concatenate(src_rows)
        seg_eptr = np.concatenate([[0], np.cumsum([e.shape[1] for e in edges])])       # the batch is block diagonal: one CSR build per segment
        a32["seg_eptr"] = seg_eptr
        p.max_seg_edges = int(np.diff(seg_eptr).max()) if len(edges) else 0
        nt = [(b - a + 31) // 32 for a, b in zip(seg_ptr[:-1], seg_ptr[1:])]
        tile_seg = np.repeat(np.arange(p.S), nt)
        tile_first = np.concatenate([[0], np.cumsum(nt)])[:-1]
        tile_row = np.asarray(seg_ptr[:-1])[tile_seg] + 32 * (np.arange(len(tile_seg)) - tile_first[tile_seg])
        a32["tiles"] = np.stack([tile_seg, tile_row], axis=1)
        p.num_tiles = len(tile_seg)
        a64["edge_index"] = e_all
        if any(m is not None for _, m in rowmasks):
            rm = np.zeros(p.N, dtype=np.uint64)
            for r0, m in rowmasks:
                if m is not None:
                    rm[r0:r0 + len(m)] = m
            a64["rowmask"] = rm.view(np.int64)
        p.a32, p.a64 = a32, a64
        # the upload image of the step, packed here (prefetch thread) so the launcher thread only makes two block copies into
        # its pinned slot: every array starts 16-byte aligned
        def pack(arrs, dtype, gran, cap, what):
            lay, parts, o = {}, [], 0
            for name, arr in arrs.items():
                a = np.ascontiguousarray(np.asarray(arr).reshape(-1), dtype=dtype)
                lay[name] = o
                pad = (-a.size) % gran
                parts.append(a)
                if pad:
                    parts.append(np.zeros(pad, dtype=dtype))
                o += a.size + pad
            if o > cap:
                raise L.GnnmpError(f"engine: {what} staging buffer too small")
            return (np.concatenate(parts) if parts else np.zeros(0, dtype=dtype)), lay
        p.cat32, p.lay32 = pack(a32, np.int32, 4, getattr(self, "i32_cap", 1 << 62), "int32")     # (no capacities on a host-only engine)
        p.cat64, p.lay64 = pack(a64, np.int64, 2, getattr(self, "i64_cap", 1 << 62), "int64")
        return p

    # ------------------------------------------------------------------ device helpers
    def _st(self):
        """hipStream_t the launches go to.  Cached: torch.cuda.current_stream() costs ~5 us and a step makes ~300 launches."""
        return self._stream_handle

    def _use_stream(self, stream=None) -> None:
        self._stream_handle = (stream or torch.cuda.current_stream(self.device)).cuda_stream

    def _chk(self, rc: int, what: str) -> None:
        if rc:
            L.check(rc, what)

    def _gemm(self, mode, A, B, bias, Cc, M, N, K, lda, ldb, ldc, relu=False, accumulate=False):
        self._chk(self.lib.gmp_gemm_f32(mode, A, B, bias, Cc, M, N, K, lda, ldb, ldc, 1.0, int(accumulate), int(relu), None, 0, self._st()), "gemm")

    def _gemm_g(self, mode, A, B, bias, Cc, rows, boff, biasoff, coff, asum, asumoff, M_tn, N, K, lda, ldb, ldc, relu=False):
        G = len(rows) - 1
        self._chk(self.lib.gmp_gemm_f32_grouped(mode, A, B, bias, Cc, G, _i32(rows), None if boff is None else _i64(boff),
                                                None if biasoff is None else _i64(biasoff), None if coff is None else _i64(coff),
                                                asum, None if asumoff is None else _i64(asumoff), M_tn, N, K, lda, ldb, ldc, 1.0, 0,
                                                int(relu), self._cur_gemm_ws.data_ptr() if mode == TN else None, self._cur_gemm_ws.numel(),
                                                self._st()), "gemm_grouped")

    def _bn_cfg(self, relu: bool, dropout: bool, site: int) -> L.BnConfig:
        p = self.dropout_p if (dropout and self.model.training) else 0.0
        return L.BnConfig(int(self.model.training), int(relu), 1e-5, 0.1, p, (self.seed * 1000003 + self.step_count) & (2 ** 64 - 1), site)

    # ------------------------------------------------------------------ one step
    def prepare(self, inp: StepInputs, gen: torch.Generator, ticket: Optional[DrawTicket] = None):
        """Host half of a step (all RNG draws + the segment layout); may run on another thread ahead of time.  `ticket`: the
        device-side draws of this input, enqueued earlier (rng_mode 'device'; StepPrefetcher keeps a few inputs' tickets in flight)."""
        art = self.collect_draws(inp, ticket) if ticket is not None else self.draw(inp, gen)
        return art, self.plan(inp, art)

    def step(self, inp: StepInputs, gen: torch.Generator, art: Optional[Dict[str, object]] = None,
             order: Optional[List[str]] = None, apply_update: bool = True, prepared=None) -> None:
        """Forward, backward, PCGrad, clip, AdamW for one step.  Nothing is read back: losses stay in
        self.loss_sums / self.plan_sizes until someone asks (losses())."""
        import time as _t
        self._use_stream()
        if self.rng_mode == "device":
            # Device draws ride the aux stream, i.e. they run behind whatever the launcher has already enqueued there: the launcher
            # therefore keeps at most two steps in front of the GPU (enough to keep it fed: enqueueing a step takes half a step), so a
            # ticket is served within two steps and the prefetcher's three tickets in flight cover it
            lead = getattr(self, "_lead", None)
            if lead is None:
                lead = self._lead = {}
            ev = lead.pop(self.step_count - int(os.environ.get("GMP_DEVICE_LEAD", "2")), None)
            if ev is not None:
                ev.synchronize()
        t0 = _t.perf_counter()
        if prepared is not None:
            art, p = prepared
            t1 = t2 = _t.perf_counter()
        else:
            if art is None:
                art = self.draw(inp, gen)
            t1 = _t.perf_counter()
            p = self.plan(inp, art)
            t2 = _t.perf_counter()
        self._upload(p, inp, art)
        t3 = _t.perf_counter()
        if self.native:
            self._forward_backward_native(p, inp)        # one C call enqueues the whole forward/heads/backward
        else:
            self._forward(p, inp)
            self._heads_and_backward(p, inp)
        self._optimizer(p, order, apply_update)
        t4 = _t.perf_counter()
        h = self.host_ms
        h["draw"] += (t1 - t0) * 1e3; h["plan"] += (t2 - t1) * 1e3; h["upload"] += (t3 - t2) * 1e3; h["launch"] += (t4 - t3) * 1e3
        h["steps"] += 1
        if self.rng_mode == "device":
            e = torch.cuda.Event()
            e.record(torch.cuda.current_stream(self.device))
            self._lead[self.step_count] = e
        self.step_count += 1
        self.last_plan, self.last_inputs = p, inp
        if self.model.training:                      # BatchNorm call counters (one per forward() the reference would have made)
            lens = np.diff(p.a32["seg_ptr"])
            dom = p.a32["seg_dom"]
            self._bn_calls += int((lens > 0).sum())
            for di in range(self.D):
                self._bn_calls_dom[di] += int(((lens > 0) & (dom == di)).sum())

    def flush_counters(self) -> None:
        """Bring every BatchNorm's num_batches_tracked up to date (it only matters for the saved state_dict -- momentum is
        fixed -- so the engine counts calls on the host and writes them when asked: pretrain() does before a checkpoint)."""
        if self._bn_calls:
            for l in self.model.gnn_backbone.layers:
                l.batch_norm.num_batches_tracked += self._bn_calls
                l.gin_conv.nn[1].num_batches_tracked += self._bn_calls
            self._bn_calls = 0
        for di, d in enumerate(self.domains):
            if self._bn_calls_dom[di]:
                self.model.input_encoders[d].batch_norm.num_batches_tracked += self._bn_calls_dom[di]
                self._bn_calls_dom[di] = 0

    # ---- upload ------------------------------------------------------------------------------------
    def _upload(self, p: StepPlan, inp: StepInputs, art) -> None:
        slot = self.stage[self.step_count % self.STAGES]
        if slot["event"] is not None:
            slot["event"].synchronize()                  # the copy that read this slot STAGES steps ago is done
        # device scalars (scal: [0:8) 1/size per task, [16:48) per-domain NT-Xent sums), LP labels and the index arrays of THIS step
        self.dev32, self.dev64, self.scal, self.lp_lab = self._up_sets[self.step_count & 1]
        on_aux = self.upload_on_aux
        up_stream = self.aux_stream if on_aux else torch.cuda.current_stream(self.device)
        pin32, pin64, pinf = slot["pin32"], slot["pin64"], slot["pinf"]
        o32, o64, lay32, lay64 = p.cat32.size, p.cat64.size, p.lay32, p.lay64
        pin32.numpy()[:o32] = p.cat32
        pin64.numpy()[:o64] = p.cat64
        nf = pinf.numpy()
        nf[:64] = 0.0
        for ti, t in enumerate(self.tasks):
            nf[ti] = 1.0 / max(p.sizes[t], 1)            # [0:8) = 1/size per task, [16:48) = per-domain NT-Xent sums (zeroed)
        nlab = 0
        if "link_pred" in self.tasks:
            nf[64:64 + p.lp_K] = p.lp_labels
            nlab = (p.lp_K + 3) // 4 * 4
        # one kernel reads the four pinned pieces over PCIe (gmp_upload) instead of four hand-overs to the copy engine
        src = (C.c_void_p * 4)(pin32.data_ptr(), pin64.data_ptr(), pinf.data_ptr(), pinf.data_ptr() + 256)
        dst = (C.c_void_p * 4)(self.dev32.data_ptr(), self.dev64.data_ptr(), self.scal.data_ptr(), self.lp_lab.data_ptr())
        nbytes = (C.c_int64 * 4)(4 * o32, 8 * o64, 256, 4 * nlab)
        self._chk(self.lib.gmp_upload(4, src, dst, nbytes, up_stream.cuda_stream), "gmp_upload")
        ev = torch.cuda.Event()
        ev.record(up_stream)
        slot["event"] = ev
        b32, b64 = self.dev32.data_ptr(), self.dev64.data_ptr()
        p.d32 = {k: b32 + 4 * o for k, o in lay32.items()}
        p.d64 = {k: b64 + 8 * o for k, o in lay64.items()}
        # per-step gradient-availability table (static unless a (task, domain) pair dropped out)
        if p.skipped:
            has = self.has_static.copy()
            for ti, d in p.skipped:
                t = self.tasks[ti]
                k = [self.name_index[n] for n in self.names if n.startswith(f"heads.{t}.{d}.")]
                has[k, ti] = 0
            self.has.copy_(torch.from_numpy(has).to(self.device))
            self._has_dirty = True
        elif getattr(self, "_has_dirty", False):
            self.has.copy_(torch.from_numpy(self.has_static).to(self.device))
            self._has_dirty = False

    # ---- forward -----------------------------------------------------------------------------------
    def _P(self, name: str) -> int:
        v = self._p_cache.get(name)
        if v is None:
            v = self._p_cache[name] = self.flat.data_ptr() + 4 * self.off[name]
        return v

    def _TG(self, t: int, name: str) -> int:
        """Float offset of (task t, tensor) inside the [T, P] per-task gradient buffer."""
        return t * self.P + self.off[name]

    def _TGs(self, name: str) -> "C.Array":
        """Per-task offsets of one tensor (static: cached as a ready ctypes array)."""
        a = self._tg_cache.get(name)
        if a is None:
            a = self._tg_cache[name] = _i64([t * self.P + self.off[name] for t in range(self.T)])
        return a

    @staticmethod
    def _lp_segmented(p: StepPlan) -> bool:
        return p.lp_S > 0 and p.lp_max_rows <= SEG_CSR_MAX_ROWS and p.lp_max_edges <= SEG_CSR_MAX_EDGES

    def _forward(self, p: StepPlan, inp: StepInputs) -> None:
        lib, st, N, D, P = self.lib, self._st(), p.N, self.domains, self._P
        c = self.csr
        main = torch.cuda.current_stream(self.device)
        # both CSR builds depend only on the uploaded indices: they run beside the encoders on the aux stream
        ev_up = torch.cuda.Event(); ev_up.record(main)
        self.aux_stream.wait_event(ev_up)
        with torch.cuda.stream(self.aux_stream):
            ast = self.aux_stream.cuda_stream
            if p.max_seg <= SEG_CSR_MAX_ROWS and p.max_seg_edges <= SEG_CSR_MAX_EDGES:      # block diagonal: one workgroup per segment
                self._chk(lib.gmp_csr_build_segmented(p.d64["edge_index"], N, p.E, p.d32["seg_ptr"], p.d32["seg_eptr"], p.S, p.max_seg,
                                                      p.max_seg_edges, c[0].data_ptr(), c[1].data_ptr(), c[2].data_ptr(), c[3].data_ptr(),
                                                      c[4].data_ptr(), c[5].data_ptr(), self.csr_status.data_ptr(), ast), "csr_build_segmented")
            else:
                self._chk(lib.gmp_csr_build(p.d64["edge_index"], N, p.E, c[0].data_ptr(), c[1].data_ptr(), c[2].data_ptr(), c[3].data_ptr(),
                                            c[4].data_ptr(), c[5].data_ptr(), self.csr_status.data_ptr(), self.csr_ws.data_ptr(),
                                            self.csr_ws.numel(), ast), "csr_build")
            ev_csr = torch.cuda.Event(); ev_csr.record(self.aux_stream)
            if "link_pred" in self.tasks:
                lc = self.lp_csr
                if self._lp_segmented(p):
                    self._chk(lib.gmp_csr_build_segmented(p.d64["lp_edges"], p.lp_rows_end, p.lp_K, p.d32["lp_seg_ptr"], p.d32["lp_seg_eptr"], p.lp_S,
                                                          p.lp_max_rows, p.lp_max_edges, lc[0].data_ptr(), lc[1].data_ptr(), lc[2].data_ptr(),
                                                          lc[3].data_ptr(), lc[4].data_ptr(), lc[5].data_ptr(), self.lp_csr_status.data_ptr(), ast), "lp csr (segmented)")
                else:
                    self._chk(lib.gmp_csr_build(p.d64["lp_edges"], N, p.lp_K, lc[0].data_ptr(), lc[1].data_ptr(), lc[2].data_ptr(), lc[3].data_ptr(),
                                                lc[4].data_ptr(), lc[5].data_ptr(), self.lp_csr_status.data_ptr(), self.lp_csr_ws.data_ptr(),
                                                self.lp_csr_ws.numel(), ast), "lp csr")
            p.ev_lpcsr = torch.cuda.Event(); p.ev_lpcsr.record(self.aux_stream)
        w_off = [self.off[f"input_encoders.{d}.linear.weight"] for d in D]
        b_off = [self.off[f"input_encoders.{d}.linear.bias"] for d in D]
        d_in = [DOMAIN_DIMENSIONS[d] for d in D]
        self._chk(lib.gmp_encoder_fwd(inp.x_all.data_ptr(), inp.x_all.size(0), N, p.S, p.d32["src_row"], p.d32["seg_ptr"], p.d32["seg_dom"], p.d64.get("rowmask"),
                                      p.d32["tiles"], p.num_tiles, self.flat.data_ptr(), len(D), _i64(w_off), _i64(b_off), _i32(d_in),
                                      self.dpad, self.z0.data_ptr(), st), "encoder_fwd")
        cfg = self._bn_cfg(True, True, 1)
        e0 = f"input_encoders.{D[0]}."
        self._chk(lib.gmp_bn_fwd(self.z0.data_ptr(), None, p.d32["seg_ptr"], p.d32["seg_dom"], p.S, p.max_seg, N, H,
                                 P(e0 + "batch_norm.weight"), P(e0 + "batch_norm.bias"), self.enc_rm.data_ptr(), self.enc_rv.data_ptr(),
                                 self.enc_mean.data_ptr(), self.enc_rstd.data_ptr(), self.h[0].data_ptr(), C.byref(cfg),
                                 self.bn_ws.data_ptr(), self.bn_ws.numel(), st), "bn_fwd encoders")
        if "node_feat_mask" in self.tasks and p.nfm_rows[-1]:
            M = p.nfm_rows[-1]
            self._chk(lib.gmp_row_gather(self.h[0].data_ptr(), p.d64["nfm_idx"], None, self.hd["nfm_tgt"].data_ptr(), M, N, H, st), "nfm target")
            self._chk(lib.gmp_row_fill(self.h[0].data_ptr(), p.d64["nfm_idx"], P("mask_token"), M, N, H, 1, st), "nfm mask")
        main.wait_event(ev_csr)
        for l in range(GNN_NUM_LAYERS):
            pre = f"gnn_backbone.layers.{l}."
            layer = self.model.gnn_backbone.layers[l]
            self._chk(lib.gmp_gin_aggregate_fwd(self.h[l].data_ptr(), c[0].data_ptr(), c[1].data_ptr(), P(pre + "gin_conv.eps"),
                                                self.a[l].data_ptr(), N, H, st), "aggregate")
            self._gemm(NT, self.a[l].data_ptr(), P(pre + "gin_conv.nn.0.weight"), P(pre + "gin_conv.nn.0.bias"), self.z1[l].data_ptr(),
                       N, 2 * H, H, H, H, 2 * H)
            bn1 = layer.gin_conv.nn[1]
            cfg = self._bn_cfg(True, False, 0)
            self._chk(lib.gmp_bn_fwd(self.z1[l].data_ptr(), None, p.d32["seg_ptr"], None, p.S, p.max_seg, N, 2 * H,
                                     P(pre + "gin_conv.nn.1.weight"), P(pre + "gin_conv.nn.1.bias"), bn1.running_mean.data_ptr(),
                                     bn1.running_var.data_ptr(), self.stat["m1"][l].data_ptr(), self.stat["s1"][l].data_ptr(),
                                     self.r1[l].data_ptr(), C.byref(cfg), self.bn_ws.data_ptr(), self.bn_ws.numel(), st), "bn1")
            self._gemm(NT, self.r1[l].data_ptr(), P(pre + "gin_conv.nn.3.weight"), P(pre + "gin_conv.nn.3.bias"), self.z2[l].data_ptr(),
                       N, H, 2 * H, 2 * H, 2 * H, H)
            bn2 = layer.batch_norm
            cfg = self._bn_cfg(True, True, 10 + l)
            self._chk(lib.gmp_bn_fwd(self.z2[l].data_ptr(), self.h[l].data_ptr(), p.d32["seg_ptr"], None, p.S, p.max_seg, N, H,
                                     P(pre + "batch_norm.weight"), P(pre + "batch_norm.bias"), bn2.running_mean.data_ptr(),
                                     bn2.running_var.data_ptr(), self.stat["m2"][l].data_ptr(), self.stat["s2"][l].data_ptr(),
                                     self.h[l + 1].data_ptr(), C.byref(cfg), self.bn_ws.data_ptr(), self.bn_ws.numel(), st), "bn2")

    # ---- head helpers ------------------------------------------------------------------------------
    def _drop(self, src: Tensor, dst: Tensor, numel: int, site: int, p: Optional[float] = None) -> Tensor:
        """dropout(src) -> dst (returns the tensor holding the result; p == 0 aliases src)."""
        p = self.dropout_p if p is None else p
        if not self.model.training or p <= 0:
            return src
        self._chk(self.lib.gmp_dropout_fwd(src.data_ptr(), dst.data_ptr(), numel, p,
                                           (self.seed * 1000003 + self.step_count) & (2 ** 64 - 1), site, self._st()), "dropout")
        return dst

    def _relu_drop_bwd(self, g: Tensor, act: Tensor, out: Tensor, numel: int, site: int, p: Optional[float] = None) -> None:
        p = (self.dropout_p if p is None else p) if self.model.training else 0.0
        self._chk(self.lib.gmp_relu_dropout_bwd(g.data_ptr(), act.data_ptr(), out.data_ptr(), numel, p,
                                                (self.seed * 1000003 + self.step_count) & (2 ** 64 - 1), site, self._st()), "relu_dropout_bwd")

    def _mlp2_grouped(self, ti: int, task: str, x: Tensor, rows: List[int], k_in: int, k_hid: int, k_out: int, bufs, site: int):
        """Per-domain two-layer MLPHead (Linear-ReLU-Dropout-Linear) over row groups; returns output tensor."""
        y1, d1, y2 = bufs
        D = self.domains
        w0 = [self.off[f"heads.{task}.{d}.mlp.0.weight"] for d in D]
        b0 = [self.off[f"heads.{task}.{d}.mlp.0.bias"] for d in D]
        w3 = [self.off[f"heads.{task}.{d}.mlp.3.weight"] for d in D]
        b3 = [self.off[f"heads.{task}.{d}.mlp.3.bias"] for d in D]
        fp = self.flat.data_ptr()
        self._gemm_g(NT, x.data_ptr(), fp, fp, y1.data_ptr(), rows, w0, b0, None, None, None, 0, k_hid, k_in, k_in, k_in, k_hid, relu=True)
        d1 = self._drop(y1, d1, rows[-1] * k_hid, site)
        self._gemm_g(NT, d1.data_ptr(), fp, fp, y2.data_ptr(), rows, w3, b3, None, None, None, 0, k_out, k_hid, k_hid, k_hid, k_out)
        return d1

    def _mlp2_grouped_bwd(self, ti: int, task: str, x: Tensor, rows: List[int], k_in: int, k_hid: int, k_out: int, y1: Tensor, d1: Tensor,
                          g_out: Tensor, g_hid: Tensor, g_in: Tensor, site: int) -> None:
        """Backward of _mlp2_grouped: per-domain weight/bias gradients go straight into task_grads[ti]."""
        D, tg, fp = self.domains, self.task_grads.data_ptr(), self.flat.data_ptr()
        TG = self._TG
        w0 = [self.off[f"heads.{task}.{d}.mlp.0.weight"] for d in D]
        w3 = [self.off[f"heads.{task}.{d}.mlp.3.weight"] for d in D]
        # dW3 = g_out^T d1, db3 = colsum(g_out)
        self._gemm_g(TN, g_out.data_ptr(), d1.data_ptr(), None, tg, rows, None, None, [TG(ti, f"heads.{task}.{d}.mlp.3.weight") for d in D],
                     tg, [TG(ti, f"heads.{task}.{d}.mlp.3.bias") for d in D], k_out, k_hid, 0, k_out, k_hid, k_hid)
        # g_d1 = g_out W3
        self._gemm_g(NN, g_out.data_ptr(), fp, None, g_hid.data_ptr(), rows, w3, None, None, None, None, 0, k_hid, k_out, k_out, k_hid, k_hid)
        self._relu_drop_bwd(g_hid, y1, g_hid, rows[-1] * k_hid, site)
        self._gemm_g(TN, g_hid.data_ptr(), x.data_ptr(), None, tg, rows, None, None, [TG(ti, f"heads.{task}.{d}.mlp.0.weight") for d in D],
                     tg, [TG(ti, f"heads.{task}.{d}.mlp.0.bias") for d in D], k_hid, k_in, 0, k_hid, k_in, k_in)
        self._gemm_g(NN, g_hid.data_ptr(), fp, None, g_in.data_ptr(), rows, w0, None, None, None, None, 0, k_in, k_hid, k_hid, k_in, k_in)

    # ---- heads + backward --------------------------------------------------------------------------
    def _heads_and_backward(self, p: StepPlan, inp: StepInputs) -> None:
        lib, st, N, D, P, TG = self.lib, self._st(), p.N, self.domains, self._P, self._TG
        hd, tg = self.hd, self.task_grads.data_ptr()
        hL = self.h[GNN_NUM_LAYERS]
        gH = self.gA
        gH[:N].zero_()
        sc = self.scal.data_ptr()
        T_ = float(self.temperature)
        main = torch.cuda.current_stream(self.device)
        ev_fwd = torch.cuda.Event(); ev_fwd.record(main)
        done = []
        order = [ti for ti in range(self.T) if self.task_streams[ti] is not None] + [ti for ti in range(self.T) if self.task_streams[ti] is None]
        for ti in order:                                     # heads on other streams first, the ones packed onto main last
            t = self.tasks[ti]
            ts = main if self.task_streams[ti] is None else self.task_streams[ti]
            if ts is not main:
                ts.wait_event(ev_fwd)
            if t == "link_pred":
                ts.wait_event(p.ev_lpcsr)
            with torch.cuda.stream(ts):
                self._use_stream(ts)
                self._cur_gemm_ws, self.loss_ws = self.task_gemm_ws[ti], self.task_loss_ws[ti]
                self._task_head(p, inp, ti, t, hL, gH, sc, T_)
                if ts is not main:
                    ev = torch.cuda.Event(); ev.record(ts)
                    done.append(ev)
        self._use_stream(main)
        self._cur_gemm_ws = self.gemm_ws
        self.loss_ws = self.task_loss_ws[0]
        for ev in done:
            main.wait_event(ev)
        main.wait_event(p.ev_lpcsr)
        self._backbone_backward(p, inp)

    def _task_head(self, p: StepPlan, inp: StepInputs, ti: int, t: str, hL: Tensor, gH: Tensor, sc: int, T_: float) -> None:
        """Head forward, loss, and head backward of ONE task (writes its rows of gH and its slots of task_grads)."""
        lib, st, N, D, P, TG = self.lib, self._st(), p.N, self.domains, self._P, self._TG
        hd, tg = self.hd, self.task_grads.data_ptr()
        gs = sc + 4 * ti                                        # device scalar 1/size_t: d total_t / d loss_sum
        ls = self.loss_sums.data_ptr() + 4 * ti
        if t == "node_feat_mask":
            rows, M = p.nfm_rows, p.nfm_rows[-1]
            if M == 0:
                return
            self._chk(lib.gmp_row_gather(hL.data_ptr(), p.d64["nfm_idx"], None, hd["nfm_in"].data_ptr(), M, N, H, st), "nfm gather")
            d1 = self._mlp2_grouped(ti, t, hd["nfm_in"], rows, H, H, H, (hd["nfm_y1"], hd["nfm_d1"], hd["nfm_y2"]), 100 + ti)
            self._chk(lib.gmp_mse_sum_fwd(hd["nfm_y2"].data_ptr(), hd["nfm_tgt"].data_ptr(), M * H, ls, self.loss_ws.data_ptr(), self.loss_ws.numel(), st), "mse")
            self._chk(lib.gmp_mse_sum_bwd(hd["nfm_y2"].data_ptr(), hd["nfm_tgt"].data_ptr(), gs, hd["nfm_g"].data_ptr(), M * H, st), "mse bwd")
            self._mlp2_grouped_bwd(ti, t, hd["nfm_in"], rows, H, H, H, hd["nfm_y1"], d1, hd["nfm_g"], hd["nfm_g1"], hd["nfm_y2"], 100 + ti)
            self._chk(lib.gmp_row_fill(gH.data_ptr(), p.d64["nfm_idx"], hd["nfm_y2"].data_ptr(), M, N, H, 0, st), "nfm scatter")
        elif t == "link_pred":
            K = p.lp_K
            w0, b0 = P("heads.link_pred.predictor.mlp.0.weight"), P("heads.link_pred.predictor.mlp.0.bias")
            w3, b3 = P("heads.link_pred.predictor.mlp.3.weight"), P("heads.link_pred.predictor.mlp.3.bias")
            self._chk(lib.gmp_lp_edge_features_fwd(hL.data_ptr(), p.d64["lp_edges"], hd["lp_feat"].data_ptr(), N, K, H, st), "lp feat")
            self._gemm(NT, hd["lp_feat"].data_ptr(), w0, b0, hd["lp_y1"].data_ptr(), K, H, 3 * H, 3 * H, 3 * H, H, relu=True)
            # the 256 -> 1 layer as a row dot product / outer product / weighted column sum (csrc/elementwise.hip), as in csrc/step.hip
            pdrop = self.dropout_p if (self.model.training and self.dropout_p > 0) else 0.0
            dseed = (self.seed * 1000003 + self.step_count) & (2 ** 64 - 1)
            d1 = hd["lp_d1"] if pdrop > 0 else hd["lp_y1"]
            pos = p.d32.get("lp_pos")            # merged rows: the ordered row(s) each stands for (a dropout mask per ordered row)
            if pos is not None:
                self._chk(lib.gmp_lp_pair_rowdot_fwd(hd["lp_y1"].data_ptr(), w3, b3, pos, self.lp_y2.data_ptr(), K, H, pdrop, dseed, 100 + ti, st), "lp pair rowdot")
                self._chk(lib.gmp_lp_pair_sigmoid_bce_fwd_bwd(self.lp_y2.data_ptr(), self.lp_lab.data_ptr(), pos, K, gs, ls, self.lp_p.data_ptr(),
                                                              self.lp_gy2.data_ptr(), self.loss_ws.data_ptr(), self.loss_ws.numel(), st), "pair sigmoid+bce")
                self._chk(lib.gmp_lp_pair_outer_bwd(self.lp_gy2.data_ptr(), w3, hd["lp_y1"].data_ptr(), pos, hd["lp_gy1"].data_ptr(), K, H, pdrop, dseed,
                                                    100 + ti, st), "lp pair outer")
            else:
                self._chk(lib.gmp_dropout_rowdot_fwd(hd["lp_y1"].data_ptr(), w3, b3, hd["lp_d1"].data_ptr(), self.lp_y2.data_ptr(), K, H, pdrop, dseed,
                                                     100 + ti, st), "lp rowdot")
                self._chk(lib.gmp_sigmoid_bce_signed_sum_fwd_bwd(self.lp_y2.data_ptr(), self.lp_lab.data_ptr(), K, gs, ls, self.lp_p.data_ptr(),
                                                          self.lp_gy2.data_ptr(), self.loss_ws.data_ptr(), self.loss_ws.numel(), st), "sigmoid+bce")
                self._chk(lib.gmp_outer_relu_dropout_bwd(self.lp_gy2.data_ptr(), w3, hd["lp_y1"].data_ptr(), hd["lp_gy1"].data_ptr(), K, H, pdrop, dseed,
                                                         100 + ti, st), "lp outer")
            one = [0, K]
            # dW0 with db0 riding along (column sums of the A tile already in LDS)
            self._gemm_g(TN, hd["lp_gy1"].data_ptr(), hd["lp_feat"].data_ptr(), None, tg, one, None, None,
                         [TG(ti, "heads.link_pred.predictor.mlp.0.weight")], tg, [TG(ti, "heads.link_pred.predictor.mlp.0.bias")], H, 3 * H, 0, H, 3 * H, 3 * H)
            if pos is not None:
                self._chk(lib.gmp_lp_pair_weighted_colsum(self.lp_gy2.data_ptr(), hd["lp_y1"].data_ptr(), pos, tg + 4 * TG(ti, "heads.link_pred.predictor.mlp.3.weight"),
                                                          tg + 4 * TG(ti, "heads.link_pred.predictor.mlp.3.bias"), K, H, pdrop, dseed, 100 + ti,
                                                          self._cur_gemm_ws.data_ptr(), self._cur_gemm_ws.numel(), st), "lp pair dW3")
            else:
                self._chk(lib.gmp_weighted_colsum(self.lp_gy2.data_ptr(), d1.data_ptr(), tg + 4 * TG(ti, "heads.link_pred.predictor.mlp.3.weight"),
                                                  tg + 4 * TG(ti, "heads.link_pred.predictor.mlp.3.bias"), K, H, self._cur_gemm_ws.data_ptr(),
                                                  self._cur_gemm_ws.numel(), st), "lp dW3")
            self._gemm(NN, hd["lp_gy1"].data_ptr(), w0, None, hd["lp_gfeat"].data_ptr(), K, 3 * H, H, H, 3 * H, 3 * H)
            self._chk(lib.gmp_lp_edge_features_bwd(hd["lp_gfeat"].data_ptr(), hL.data_ptr(), p.d64["lp_edges"], hd["lp_ghs"].data_ptr(),
                                                   hd["lp_ghd"].data_ptr(), N, K, H, st), "lp feat bwd")
            # reduce the per-edge gradients onto nodes -- only over this task's own rows (other tasks' heads are
            # writing their rows of gH concurrently on their own streams)
            c = self.lp_csr
            r0, r1 = p.task_row[ti], p.task_row[ti + 1]
            g_rows = gH.data_ptr() + 4 * H * r0
            self._chk(lib.gmp_segment_sum(hd["lp_ghs"].data_ptr(), c[3].data_ptr() + 4 * r0, c[5].data_ptr(), g_rows, r1 - r0, H, 0, 1, st), "lp g by src")
            self._chk(lib.gmp_segment_sum(hd["lp_ghd"].data_ptr(), c[0].data_ptr() + 4 * r0, c[2].data_ptr(), g_rows, r1 - r0, H, 0, 1, st), "lp g by dst")
        elif t == "node_contrast":
            rows, M = p.nc_rows, p.nc_rows[-1]
            if M == 0:
                return
            self._chk(lib.gmp_row_gather(hL.data_ptr(), p.d64["nc_idx"], None, hd["nc_in"].data_ptr(), M, N, H, st), "nc gather")
            d1 = self._mlp2_grouped(ti, t, hd["nc_in"], rows, H, H, 128, (hd["nc_y1"], hd["nc_d1"], hd["nc_z"]), 100 + ti)
            self._nt_xent_domains(p.nc_n, rows, hd["nc_z"], hd["nc_gz"], gs, ls, T_, 0)
            self._mlp2_grouped_bwd(ti, t, hd["nc_in"], rows, H, H, 128, hd["nc_y1"], d1, hd["nc_gz"], hd["nc_g1"], hd["nc_gin"], 100 + ti)
            self._chk(lib.gmp_row_fill(gH.data_ptr(), p.d64["nc_idx"], hd["nc_gin"].data_ptr(), M, N, H, 0, st), "nc scatter")
        elif t == "graph_contrast":
            rows, B = p.gc_rows, p.gc_B
            if B == 0:
                return
            self._chk(lib.gmp_segment_sum(hL.data_ptr(), p.d32["gc_ptr"], None, hd["gc_mean"].data_ptr(), B, H, 1, 0, st), "gc mean")
            self._chk(lib.gmp_segment_max_fwd(hL.data_ptr(), p.d32["gc_ptr"], hd["gc_max"].data_ptr(), B, H, st), "gc max")
            torch.cat([hd["gc_mean"][:B], hd["gc_max"][:B]], dim=1, out=hd["gc_in"][:B])
            d1 = self._mlp2_grouped(ti, t, hd["gc_in"], rows, 2 * H, H, 128, (hd["gc_y1"], hd["gc_d1"], hd["gc_z"]), 100 + ti)
            self._nt_xent_domains(p.gc_n, rows, hd["gc_z"], hd["gc_gz"], gs, ls, T_, self.D)
            self._mlp2_grouped_bwd(ti, t, hd["gc_in"], rows, 2 * H, H, 128, hd["gc_y1"], d1, hd["gc_gz"], hd["gc_g1"], hd["gc_gin"], 100 + ti)
            hd["gc_gmean"][:B].copy_(hd["gc_gin"][:B, :H])
            hd["gc_gmax"][:B].copy_(hd["gc_gin"][:B, H:])
            g_rows = gH.data_ptr() + 4 * H * p.gc_r0
            self._chk(lib.gmp_row_gather(hd["gc_gmean"].data_ptr(), p.d64["gc_gid"], p.d32["gc_ptr"], g_rows, p.gc_M, B, H, st), "gc mean bwd")
            self._chk(lib.gmp_segment_max_bwd(hd["gc_gmax"].data_ptr(), hL.data_ptr(), hd["gc_max"].data_ptr(), p.d32["gc_ptr"], gH.data_ptr(),
                                              B, H, 1, st), "gc max bwd")
        elif t == "graph_prop":
            rows, B = p.gp_rows, p.gp_B
            G = GRAPH_PROPERTY_DIM
            self._chk(lib.gmp_segment_sum(hL.data_ptr(), p.d32["gp_ptr"], None, hd["gp_in"].data_ptr(), B, H, 1, 0, st), "gp mean")
            d1 = self._mlp2_grouped(ti, t, hd["gp_in"], rows, H, 2 * H, G, (hd["gp_y1"], hd["gp_d1"], self.gp_y2), 100 + ti)
            self._chk(lib.gmp_mse_sum_fwd(self.gp_y2.data_ptr(), inp.graph_props.data_ptr(), B * G, ls, self.loss_ws.data_ptr(), self.loss_ws.numel(), st), "gp mse")
            self._chk(lib.gmp_mse_sum_bwd(self.gp_y2.data_ptr(), inp.graph_props.data_ptr(), gs, self.gp_g2.data_ptr(), B * G, st), "gp mse bwd")
            self._mlp2_grouped_bwd(ti, t, hd["gp_in"], rows, H, 2 * H, G, hd["gp_y1"], d1, self.gp_g2, hd["gp_g1"], hd["gp_gin"], 100 + ti)
            g_rows = gH.data_ptr() + 4 * H * p.gp_r0
            self._chk(lib.gmp_row_gather(hd["gp_gin"].data_ptr(), p.d64["gp_gid"], p.d32["gp_ptr"], g_rows, p.gp_M, B, H, st), "gp mean bwd")
        elif t == "domain_adv":
            # mean read-out -> gradient reversal -> Linear 256->128, ReLU, Dropout(.5), Linear 128->D -> CE(sum)  (heads.py:70-82)
            B, Cc, lam = p.da_B, len(D), float(self.grl_lambda)
            pre = "heads.domain_adv.classifier.mlp."
            w0, b0, w3, b3 = P(pre + "0.weight"), P(pre + "0.bias"), P(pre + "3.weight"), P(pre + "3.bias")
            self._chk(lib.gmp_segment_sum(hL.data_ptr(), p.d32["da_ptr"], None, hd["da_in"].data_ptr(), B, H, 1, 0, st), "da mean")
            self._gemm(NT, hd["da_in"].data_ptr(), w0, b0, hd["da_y1"].data_ptr(), B, DA_HIDDEN, H, H, H, DA_HIDDEN, relu=True)
            d1 = self._drop(hd["da_y1"], hd["da_d1"], B * DA_HIDDEN, 100 + ti, p=self.da_dropout)
            self._gemm(NT, d1.data_ptr(), w3, b3, hd["da_logits"].data_ptr(), B, Cc, DA_HIDDEN, DA_HIDDEN, DA_HIDDEN, Cc)
            self._chk(lib.gmp_cross_entropy_sum_fwd(hd["da_logits"].data_ptr(), p.d64["da_labels"], B, Cc, ls, self.loss_ws.data_ptr(), self.loss_ws.numel(), st), "da ce")
            self._chk(lib.gmp_cross_entropy_sum_bwd(hd["da_logits"].data_ptr(), p.d64["da_labels"], B, Cc, gs, hd["da_glogits"].data_ptr(), st), "da ce bwd")
            one = [0, B]
            self._chk(lib.gmp_gemm_f32_grouped(TN, hd["da_glogits"].data_ptr(), d1.data_ptr(), None, tg, 1, _i32(one), None, None, _i64([TG(ti, pre + "3.weight")]),
                                               tg, _i64([TG(ti, pre + "3.bias")]), Cc, DA_HIDDEN, 0, Cc, DA_HIDDEN, DA_HIDDEN, 1.0, 0, 0, None, 0, st), "da dW3")
            self._gemm(NN, hd["da_glogits"].data_ptr(), w3, None, hd["da_g1"].data_ptr(), B, DA_HIDDEN, Cc, Cc, DA_HIDDEN, DA_HIDDEN)
            self._relu_drop_bwd(hd["da_g1"], hd["da_y1"], hd["da_g1"], B * DA_HIDDEN, 100 + ti, p=self.da_dropout)
            self._chk(lib.gmp_gemm_f32_grouped(TN, hd["da_g1"].data_ptr(), hd["da_in"].data_ptr(), None, tg, 1, _i32(one), None, None, _i64([TG(ti, pre + "0.weight")]),
                                               tg, _i64([TG(ti, pre + "0.bias")]), DA_HIDDEN, H, 0, DA_HIDDEN, H, H, 1.0, 0, 0, None, 0, st), "da dW0")
            self._chk(lib.gmp_gemm_f32(NN, hd["da_g1"].data_ptr(), w0, None, hd["da_gin"].data_ptr(), B, H, DA_HIDDEN, DA_HIDDEN, H, H, -lam, 0, 0, None, 0, st), "da grl")
            g_rows = gH.data_ptr() + 4 * H * p.da_r0
            self._chk(lib.gmp_row_gather(hd["da_gin"].data_ptr(), p.d64["da_gid"], p.d32["da_ptr"], g_rows, p.da_M, B, H, st), "da mean bwd")

    def _nt_xent_domains(self, ns: List[int], rows: List[int], z: Tensor, gz: Tensor, gs: int, ls: int, temperature: float, slot0: int) -> None:
        """One NT-Xent problem per domain on rows [rows[d], rows[d+1]) = [z1 ; z2]; loss sums land in scal[16+slot],
        their total in the task's loss slot."""
        lib, st = self.lib, self._st()
        sc = self.scal.data_ptr()
        ws = self._ntx_workspace(slot0, ns)
        self._chk(lib.gmp_nt_xent_grouped(z.data_ptr(), gz.data_ptr(), len(ns), _i32(list(ns)), _i64([int(r) for r in rows[:len(ns)]]), 128,
                                          temperature, gs, sc + 4 * (16 + slot0), ls, ws.data_ptr(), ws.numel(), st), "nt_xent grouped")

    def _ntx_workspace(self, slot0: int, ns) -> Tensor:
        """One grouped workspace per contrastive task (slot0 = 0 node level, D graph level), grown on demand."""
        need = self.lib.gmp_nt_xent_grouped_workspace_bytes(len(ns), max(max(ns), 1), 128)
        if need > self.ntx_ws[slot0].numel():
            self.ntx_ws[slot0] = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self.ntx_ws[slot0]

    def _backbone_backward(self, p: StepPlan, inp: StepInputs) -> None:
        lib, st, N, D, P, TG, T = self.lib, self._st(), p.N, self.domains, self._P, self._TG, self.T
        tg = self.task_grads.data_ptr()
        c = self.csr
        gcur, gu, ga = self.gA, self.gB, self.ga
        task_seg = [0]
        for ti in range(T):                                            # segments are task-major
            task_seg.append(sum(1 for s in p.seg_task if s <= ti))
        trow = p.task_row
        for l in reversed(range(GNN_NUM_LAYERS)):
            pre = f"gnn_backbone.layers.{l}."
            layer = self.model.gnn_backbone.layers[l]
            bn2 = layer.batch_norm
            cfg = self._bn_cfg(True, True, 10 + l)
            self._chk(lib.gmp_bn_bwd(gcur.data_ptr(), self.z2[l].data_ptr(), self.h[l].data_ptr(), p.d32["seg_ptr"], None, p.S, p.max_seg, N, H,
                                     P(pre + "batch_norm.weight"), P(pre + "batch_norm.bias"), bn2.running_mean.data_ptr(), bn2.running_var.data_ptr(),
                                     self.stat["m2"][l].data_ptr(), self.stat["s2"][l].data_ptr(), gu.data_ptr(), tg, tg, _i32(task_seg),
                                     self._TGs(pre + "batch_norm.weight"), self._TGs(pre + "batch_norm.bias"),
                                     T, C.byref(cfg), self.bn_ws.data_ptr(), self.bn_ws.numel(), st), "bn2 bwd")
            self._gemm_g(TN, gu.data_ptr(), self.r1[l].data_ptr(), None, tg, trow, None, None, [TG(t, pre + "gin_conv.nn.3.weight") for t in range(T)],
                         tg, [TG(t, pre + "gin_conv.nn.3.bias") for t in range(T)], H, 2 * H, 0, H, 2 * H, 2 * H)
            self._gemm(NN, gu.data_ptr(), P(pre + "gin_conv.nn.3.weight"), None, self.gW.data_ptr(), N, 2 * H, H, H, 2 * H, 2 * H)
            bn1 = layer.gin_conv.nn[1]
            cfg = self._bn_cfg(True, False, 0)
            self._chk(lib.gmp_bn_bwd(self.gW.data_ptr(), self.z1[l].data_ptr(), None, p.d32["seg_ptr"], None, p.S, p.max_seg, N, 2 * H,
                                     P(pre + "gin_conv.nn.1.weight"), P(pre + "gin_conv.nn.1.bias"), bn1.running_mean.data_ptr(), bn1.running_var.data_ptr(),
                                     self.stat["m1"][l].data_ptr(), self.stat["s1"][l].data_ptr(), self.gW2.data_ptr(), tg, tg, _i32(task_seg),
                                     self._TGs(pre + "gin_conv.nn.1.weight"), self._TGs(pre + "gin_conv.nn.1.bias"),
                                     T, C.byref(cfg), self.bn_ws.data_ptr(), self.bn_ws.numel(), st), "bn1 bwd")
            self._gemm_g(TN, self.gW2.data_ptr(), self.a[l].data_ptr(), None, tg, trow, None, None, [TG(t, pre + "gin_conv.nn.0.weight") for t in range(T)],
                         tg, [TG(t, pre + "gin_conv.nn.0.bias") for t in range(T)], 2 * H, H, 0, 2 * H, H, H)
            self._gemm(NN, self.gW2.data_ptr(), P(pre + "gin_conv.nn.0.weight"), None, ga.data_ptr(), N, H, 2 * H, 2 * H, H, H)
            self._chk(lib.gmp_gin_aggregate_bwd_ex(ga.data_ptr(), c[3].data_ptr(), c[4].data_ptr(), P(pre + "gin_conv.eps"), self.h[l].data_ptr(),
                                                   gu.data_ptr(), gcur.data_ptr(), self.rowdot.data_ptr(), N, H, st), "aggregate bwd")
            self._chk(lib.gmp_group_sum_1d(self.rowdot.data_ptr(), T, _i32(trow), self._TGs(pre + "gin_conv.eps"), tg, st), "eps grad")
        # ---- below the backbone: mask token (NFM) and the encoders (every task but NFM)
        if "node_feat_mask" in self.tasks and p.nfm_rows[-1]:
            ti, M = self.tasks.index("node_feat_mask"), p.nfm_rows[-1]
            self._chk(lib.gmp_row_gather(gcur.data_ptr(), p.d64["nfm_idx"], None, self.hd["nfm_in"].data_ptr(), M, N, H, st), "token rows")
            self._chk(lib.gmp_colsum(self.hd["nfm_in"].data_ptr(), tg + 4 * TG(ti, "mask_token"), M, H, H, 0, self.loss_ws.data_ptr(),
                                     self.loss_ws.numel(), st), "token grad")
        seg_of: Dict[Tuple[int, int], List[int]] = {}
        for si, (tt, dd) in enumerate(zip(p.seg_task, p.seg_dom)):
            seg_of.setdefault((tt, dd), []).append(si)
        groups = []                                   # (task index, domain, first segment, one-past-last segment)
        for ti, t in enumerate(self.tasks):
            if t == "node_feat_mask":
                continue                              # NFM runs the encoder under no_grad (pretrain_model.py:68-69)
            for di, d in enumerate(D):
                segs = seg_of.get((ti, di), [])
                if not segs:
                    # this (task, domain) pair contributed nothing: its slots must read as zero, not as last step's values
                    for key in ("linear.weight", "linear.bias", "batch_norm.weight", "batch_norm.bias"):
                        n = f"input_encoders.{d}.{key}"
                        self.task_grads[ti, self.off[n]:self.off[n] + self.numel[n]].zero_()
                    continue
                groups.append((ti, d, segs[0], segs[-1] + 1))
        if not groups:
            return
        ptr = [groups[0][2]]
        for (_, _, lo, hi) in groups:
            if lo != ptr[-1]:
                raise L.GnnmpError("engine: encoder gradient groups are not contiguous (segments must be task-major)")
            ptr.append(hi)
        e0 = f"input_encoders.{D[0]}."
        cfg = self._bn_cfg(True, True, 1)
        self._chk(lib.gmp_bn_bwd(gcur.data_ptr(), self.z0.data_ptr(), None, p.d32["seg_ptr"], p.d32["seg_dom"], p.S, p.max_seg, N, H,
                                 P(e0 + "batch_norm.weight"), P(e0 + "batch_norm.bias"), self.enc_rm.data_ptr(), self.enc_rv.data_ptr(),
                                 self.enc_mean.data_ptr(), self.enc_rstd.data_ptr(), gu.data_ptr(), tg, tg, _i32(ptr),
                                 _i64([TG(ti, f"input_encoders.{d}.batch_norm.weight") for (ti, d, _, _) in groups]),
                                 _i64([TG(ti, f"input_encoders.{d}.batch_norm.bias") for (ti, d, _, _) in groups]),
                                 len(groups), C.byref(cfg), self.bn_ws.data_ptr(), self.bn_ws.numel(), st), "bn bwd encoders")
        d_in = [DOMAIN_DIMENSIONS[d] for d in D]
        self._chk(lib.gmp_encoder_bwd(inp.x_all.data_ptr(), inp.x_all.size(0), N, p.S, p.d32["src_row"], p.d32["seg_ptr"], p.d32["seg_dom"], p.d64.get("rowmask"),
                                      gu.data_ptr(), len(D), _i32(d_in), self.dpad, len(groups), _i32(ptr),
                                      _i64([TG(ti, f"input_encoders.{d}.linear.weight") for (ti, d, _, _) in groups]),
                                      _i64([TG(ti, f"input_encoders.{d}.linear.bias") for (ti, d, _, _) in groups]), tg,
                                      self.gemm_ws.data_ptr(), self.gemm_ws.numel(), st), "encoder bwd")

    # ---- optimizer ---------------------------------------------------------------------------------
    def _optimizer(self, p: StepPlan, order: Optional[List[str]], apply_update: bool) -> None:
        names = list(self.tasks)
        main_tasks = [t for t in names if t != "domain_adv"]     # pretrain.py:137-150: PCGrad over the main tasks, then
        extra = names.index("domain_adv") if "domain_adv" in names else -1   # domain_adv_loss.backward() accumulates on top
        if order is None:
            order = list(main_tasks)
            if len(order) > 1:
                (self.shuffle_rng or random).shuffle(order)       # reference: unseeded random.shuffle (gradient_surgery.py:43)
        idx = [names.index(t) for t in order]
        self.last_order = order
        oidx = _i32(idx)

        def pcgrad(k0: int, k1: int, phases: int, stream: int) -> None:
            self._chk(self.lib.gmp_mt_pcgrad_clip_adamw_ex(
                self.task_grads.data_ptr(), self.P, self.T, self.K, self.t_off.data_ptr(), self.t_len.data_ptr(), self.has.data_ptr(),
                oidx, len(idx), names.index(main_tasks[-1]), extra, self.flat.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                self.steps.data_ptr() if apply_update else None, self.lr.data_ptr(), self.wd.data_ptr(), 0.9, 0.999, 1e-8, self.max_grad_norm,
                self.final_grad.data_ptr(), self.normsq.data_ptr(), self.metrics.data_ptr(), self.flags.data_ptr(), self.mt_ws.data_ptr(),
                self.mt_ws.numel(), int(apply_update), k0, k1, phases,
                (self.sync_flags.data_ptr() + 4 * 63) if self.use_gates else None,      # a timed-out gate: no update from this step on
                stream), "mt_pcgrad_clip_adamw")

        if self.grad_sync is not None and self.dp_mode == "sharded" and self.native:
            from . import dist as D
            if D.world_size() > 1:
                sync = self._shard_sync()
                sync.average_(self.lib, torch.cuda.current_stream(self.device),
                              gate=(self.sync_flags.data_ptr(), self._epoch) if self.use_gates else None,
                              own_pass=lambda a, b, st: pcgrad(a, b, 1, st), foreign_pass=lambda a, b, st: pcgrad(a, b, 4, st))
                pcgrad(0, self.K, 2, self._st())
                return
        if self.parts_beside_backward:
            # PCGrad follows the backward part by part on the exchange stream (Gram / solve / combine of a part as soon as its
            # gradients are final -- and, data parallel, averaged); only the total norm, the clip and AdamW wait for the last part
            sync = self._part_sync()
            k_of = self._part_tensors
            sync.average_(self.lib, torch.cuda.current_stream(self.device), gate=(self.sync_flags.data_ptr(), self._epoch),
                          exchange=self.grad_sync is not None,
                          after_message=lambda parts, st: [pcgrad(a, b, 1, st) for a, b in _merge_runs([k_of[q] for q in parts])])
            pcgrad(0, self.K, 2, self._st())
            return
        if self.grad_sync is not None:
            self._sync_task_grads()
        pcgrad(0, self.K, 3, self._st())

    def _head_slices(self):
        out = []
        for k, n in enumerate(self.names):
            if n.startswith("heads."):
                t = int(np.argmax(self.has_static[k]))
                out.append((t * self.P + self.off[n], -(-self.numel[n] // 4) * 4))       # tensors are 4-float aligned
        return out

    def _part_sync(self):
        """The per-task gradient matrix cut into the parts the step finishes one after the other (gmp_step_wait_grads): part 0
        the task heads, part 1 + k backbone layer L-1-k, last part mask token + encoders -- as message slices for the exchange
        (dist.OverlappedGradSync) and as tensor-index ranges for PCGrad (self._part_tensors)."""
        if self._packed_sync is None:
            from .dist import OverlappedGradSync
            ranges: List[List[List[int]]] = [[] for _ in range(GNN_NUM_LAYERS + 2)]
            krange: List[List[int]] = [[self.K, 0] for _ in range(GNN_NUM_LAYERS + 2)]
            shared = [n for n in self.names if self.off[n] < self.P_shared]
            for i, n in enumerate(shared):
                part = GNN_NUM_LAYERS + 1
                if n.startswith("gnn_backbone.layers."):
                    part = GNN_NUM_LAYERS - int(n.split(".")[2])
                end = self.off[shared[i + 1]] if i + 1 < len(shared) else self.P_shared
                r = ranges[part]
                if r and r[-1][1] == self.off[n]:
                    r[-1][1] = end                                   # contiguous with the previous tensor of this part
                else:
                    r.append([self.off[n], end])
                k = self.name_index[n]
                krange[part] = [min(krange[part][0], k), max(krange[part][1], k + 1)]
            hk = [k for k, n in enumerate(self.names) if n.startswith("heads.")]
            krange[0] = [min(hk), max(hk) + 1] if hk else [0, 0]
            for b, (k0, k1) in enumerate(krange):                    # every part must be one run of tensor indices
                if k1 > k0 and sorted(self._part_of(n) for n in self.names[k0:k1]) != [b] * (k1 - k0):
                    raise L.GnnmpError("engine: the tensors of a backward part are not contiguous in the flat layout")
            self._part_tensors = [(k0, k1) if k1 > k0 else (0, 0) for k0, k1 in krange]
            parts = [self._head_slices()] + [[(t * self.P + lo, hi - lo) for t in range(self.T) for lo, hi in ranges[b]]
                                             for b in range(1, GNN_NUM_LAYERS + 2)]
            self._packed_sync = OverlappedGradSync(self.task_grads.view(-1), parts, self.comm_stream)
        return self._packed_sync

    def _shard_sync(self):
        """dist.ShardedGradSync over the same parts as _part_sync: tensor k's message slices = one copy per task that has it (static table),
        its combined gradient = its slot of final_grad."""
        if self._shard_sync_obj is None:
            from .dist import ShardedGradSync
            self._part_sync()                                  # builds self._part_tensors (and checks the parts are runs of tensor indices)
            al4 = lambda v: -(-v // 4) * 4

            def msg(k: int):
                n = self.names[k]
                return [(t * self.P + self.off[n], al4(self.numel[n])) for t in range(self.T) if self.has_static[k][t]]

            fin = lambda k: (self.off[self.names[k]], al4(self.numel[self.names[k]]))
            self._shard_sync_obj = ShardedGradSync(self.task_grads.view(-1), self.final_grad, self._part_tensors, msg, fin, self.comm_stream)
        return self._shard_sync_obj

    @staticmethod
    def _part_of(name: str) -> int:
        if name.startswith("heads."):
            return 0
        if name.startswith("gnn_backbone.layers."):
            return GNN_NUM_LAYERS - int(name.split(".")[2])
        return GNN_NUM_LAYERS + 1

    def _sync_task_grads(self) -> None:
        """Data parallel: average the per-task gradients over ranks BEFORE PCGrad -- shared tensors once per task,
        every head once (only its own task's row is meaningful).  Native executor: in parts beside the backward
        (dist.OverlappedGradSync; GMP_DP_OVERLAP=0 falls back to one flat all-reduce after it)."""
        head_slices = self._head_slices() if self._packed_sync is None else []
        if self.native and os.environ.get("GMP_DP_OVERLAP", "1") != "0":
            self._part_sync().average_(self.lib, torch.cuda.current_stream(self.device),
                                       gate=(self.sync_flags.data_ptr(), self._epoch) if self.use_gates else None)
            return
        if self._packed_sync is None:
            from .dist import PackedGradSync
            slices = [(t * self.P, self.P_shared) for t in range(self.T)] + head_slices
            self._packed_sync = PackedGradSync(self.task_grads.view(-1), slices)
        self._packed_sync.average_()

    # ---- reporting (the only host syncs, and only on request) ---------------------------------------
    def check_gates(self) -> None:
        """A gate that timed out let its stream run ahead of a dependency: results since then are not to be trusted."""
        if self.use_gates and int(self.sync_flags[63].item()) != 0:
            raise L.GnnmpError("engine: a cross-stream gate timed out (streams sharing a hardware queue, or a tool serialising "
                               "kernels?) -- rerun with GMP_STEP_GATES=0")

    def verify_gates(self, inp: "StepInputs", steps: int = 2, timeout_s: float = 5.0) -> bool:
        """Start-up self-check of the gates in the process as it now is (call it AFTER the process group exists: a communicator
        brings streams and queues of its own, and the stream -> hardware-queue calibration this engine was built on predates
        nothing it has not seen).  `steps` probe steps run from the current state with gates (short time-out), then -- from the
        same state, with the same draws and dropout seeds -- with events; gates stay on only when no gate timed out and
        parameters, per-task gradients, losses and running statistics are bitwise equal, on EVERY rank.  The engine's state is
        restored afterwards.  Collective: all ranks of a data-parallel job must call it together."""
        if not self.use_gates:
            return False
        import torch.distributed as tdist
        dev = self.device
        bufs = [self.flat, self.exp_avg, self.exp_avg_sq, self.steps, self.enc_rm, self.enc_rv, self.loss_sums]
        bufs += [b for n, b in self.model.named_buffers() if "running_" in n and not n.startswith("input_encoders.")]
        saved = [b.clone() for b in bufs]
        host_state = (self.step_count, self._bn_calls, list(self._bn_calls_dom), self._nprng, dict(self.host_ms))
        # the link-prediction negatives draw from a stream of their own (neg_rng / its native twin): both passes must see the same one
        neg_state = (self.neg_rng.getstate(), self._neg_native.getstate().clone() if self._neg_native is not None else None)
        seq_state = dict(getattr(self, "_draw_seq", {}))

        def rewind_negatives() -> None:
            if seq_state:
                self._draw_seq = dict(seq_state)           # device draws: both passes (and the run after them) see the same sequence
            self.neg_rng.setstate(neg_state[0])
            if self._neg_native is not None and neg_state[1] is not None:
                self._neg_native.setstate(neg_state[1])
            elif self._neg_native is not None:
                self._neg_native = None                  # created during the first pass: the second re-creates it from neg_rng
        tasks = [t for t in self.tasks if t != "domain_adv"]
        self._chk(self.lib.gmp_gate_set_timeout(float(timeout_s)), "gate_set_timeout")
        outs, timed_out = [], False
        try:
            for gates in (True, False):
                self.use_gates = gates
                for b, s in zip(bufs, saved):
                    b.copy_(s)
                self.step_count = host_state[0]
                self._nprng = None
                rewind_negatives()
                g = torch.Generator().manual_seed(12345)
                for _ in range(steps):
                    self.step(inp, g, order=tasks)
                torch.cuda.synchronize(dev)
                if gates:
                    timed_out = int(self.sync_flags[63].item()) != 0
                    self.sync_flags[63] = 0
                outs.append([self.flat.clone(), self.task_grads.clone(), self.loss_sums.clone()] + [b.clone() for b in bufs[4:]])
        finally:
            for b, s in zip(bufs, saved):
                b.copy_(s)
            self.step_count, self._bn_calls, self._bn_calls_dom, self._nprng = host_state[0], host_state[1], host_state[2], host_state[3]
            rewind_negatives()
            self.host_ms.update(host_state[4])
            self._chk(self.lib.gmp_gate_set_timeout(float(os.environ.get("GMP_GATE_TIMEOUT_S", "120"))), "gate_set_timeout")
        ok = (not timed_out) and all(torch.equal(a, b) for a, b in zip(outs[0], outs[1]))
        if not ok and not timed_out:          # name what differed: a mismatch is a synchronisation bug, not a tuning matter
            labels = ["parameters", "per-task gradients", "loss sums"] + [f"buffer {i}" for i in range(len(outs[0]) - 3)]
            self.gates_mismatch = [(n, float((a.double() - b.double()).abs().max())) for n, a, b in zip(labels, outs[0], outs[1]) if not torch.equal(a, b)]
        if tdist.is_available() and tdist.is_initialized() and tdist.get_world_size() > 1:
            t = torch.tensor([1 if ok else 0], device=dev, dtype=torch.int32)
            tdist.all_reduce(t, op=tdist.ReduceOp.MIN)
            ok = bool(t.item())
        self.use_gates = ok
        self.gates_verified = {"ok": ok, "timed_out": timed_out, "steps": steps}
        if getattr(self, "gates_mismatch", None):
            self.gates_verified["mismatch"] = self.gates_mismatch
        torch.cuda.synchronize(dev)
        return ok

    def losses(self) -> Dict[str, float]:
        self.check_gates()
        sums = self.loss_sums[:self.T].tolist()
        return {t: sums[i] / max(self.last_plan.sizes[t], 1) for i, t in enumerate(self.tasks)}

    def task_gradient(self, task: str, name: str) -> Tensor:
        t = self.tasks.index(task)
        o = self.off[name]
        return self.task_grads[t, o:o + self.numel[name]].view_as(dict(self.model.named_parameters())[name])

    def final_gradient(self, name: str) -> Tensor:
        o = self.off[name]
        return self.final_grad[o:o + self.numel[name]].view_as(dict(self.model.named_parameters())[name])

    # ---- native executor: the same sequence enqueued by one C call (csrc/step.hip) ---------------------
    def _init_desc(self):
        from ._step_desc import StepDesc, TASK_KIND
        d = StepDesc()
        D, T = self.domains, self.T
        ptr = lambda t: t.data_ptr()
        d.num_tasks, d.num_domains, d.dpad, d.hidden = T, len(D), self.dpad, H
        d.flat, d.P, d.task_grads = ptr(self.flat), self.P, ptr(self.task_grads)
        for i in range(6):
            d.csr[i], d.lp_csr[i] = ptr(self.csr[i]), ptr(self.lp_csr[i])
        d.csr_status, d.csr_ws, d.csr_ws_bytes = ptr(self.csr_status), ptr(self.csr_ws), self.csr_ws.numel()
        d.lp_csr_status, d.lp_csr_ws, d.lp_csr_ws_bytes = ptr(self.lp_csr_status), ptr(self.lp_csr_ws), self.lp_csr_ws.numel()
        for i, dom in enumerate(D):
            d.enc_off_w[i] = self.off[f"input_encoders.{dom}.linear.weight"]
            d.enc_off_b[i] = self.off[f"input_encoders.{dom}.linear.bias"]
            d.enc_d_in[i] = DOMAIN_DIMENSIONS[dom]
        d.enc_off_gamma0 = self.off[f"input_encoders.{D[0]}.batch_norm.weight"]
        d.enc_off_beta0 = self.off[f"input_encoders.{D[0]}.batch_norm.bias"]
        d.enc_rm, d.enc_rv, d.enc_mean, d.enc_rstd, d.z0 = ptr(self.enc_rm), ptr(self.enc_rv), ptr(self.enc_mean), ptr(self.enc_rstd), ptr(self.z0)
        d.nfm_task = self.tasks.index("node_feat_mask") if "node_feat_mask" in self.tasks else -1
        if d.nfm_task >= 0:
            d.off_mask_token, d.tg_mask_token = self.off["mask_token"], self._TG(d.nfm_task, "mask_token")
        for l in range(GNN_NUM_LAYERS + 1):
            d.h[l] = ptr(self.h[l])
        names = {"eps": "gin_conv.eps", "w1": "gin_conv.nn.0.weight", "b1": "gin_conv.nn.0.bias", "g1": "gin_conv.nn.1.weight",
                 "be1": "gin_conv.nn.1.bias", "w2": "gin_conv.nn.3.weight", "b2": "gin_conv.nn.3.bias", "g2": "batch_norm.weight", "be2": "batch_norm.bias"}
        for l in range(GNN_NUM_LAYERS):
            Ld, pre, layer = d.layer[l], f"gnn_backbone.layers.{l}.", self.model.gnn_backbone.layers[l]
            for k, n in names.items():
                setattr(Ld, "off_" + k, self.off[pre + n])
                arr = getattr(Ld, "tg_" + k)
                for t in range(T):
                    arr[t] = self._TG(t, pre + n)
            bn1, bn2 = layer.gin_conv.nn[1], layer.batch_norm
            Ld.rm1, Ld.rv1, Ld.rm2, Ld.rv2 = ptr(bn1.running_mean), ptr(bn1.running_var), ptr(bn2.running_mean), ptr(bn2.running_var)
            Ld.a, Ld.z1, Ld.r1, Ld.z2 = ptr(self.a[l]), ptr(self.z1[l]), ptr(self.r1[l]), ptr(self.z2[l])
            Ld.m1, Ld.s1, Ld.m2, Ld.s2 = ptr(self.stat["m1"][l]), ptr(self.stat["s1"][l]), ptr(self.stat["m2"][l]), ptr(self.stat["s2"][l])
        d.gA, d.gB, d.gW, d.gW2, d.rowdot = ptr(self.gA), ptr(self.gB), ptr(self.gW), ptr(self.gW2), ptr(self.rowdot)
        d.ga = ptr(self.ga)
        d.gB2, d.gW3 = ptr(self.gB2), ptr(self.gW3)
        for l in range(GNN_NUM_LAYERS):
            d.gu_l[l], d.gz1_l[l] = ptr(self.gu_l[l]), ptr(self.gz1_l[l])
        d.bn_ws, d.bn_ws_bytes = ptr(self.bn_ws), self.bn_ws.numel()
        d.gemm_ws, d.gemm_ws_bytes = ptr(self.gemm_ws), self.gemm_ws.numel()
        d.loss_ws, d.loss_ws_bytes = ptr(self.task_loss_ws[0]), self.task_loss_ws[0].numel()
        hd = self.hd
        mlp_cfg = {"node_feat_mask": (H, H, H, "nfm_in", "nfm_y1", "nfm_d1", "nfm_y2", "nfm_g", "nfm_g1", "nfm_gin"),   # (y2 is read again by the deferred loss sum)
                   "node_contrast": (H, H, 128, "nc_in", "nc_y1", "nc_d1", "nc_z", "nc_gz", "nc_g1", "nc_gin"),
                   "graph_contrast": (2 * H, H, 128, "gc_in", "gc_y1", "gc_d1", "gc_z", "gc_gz", "gc_g1", "gc_gin"),
                   "graph_prop": (H, 2 * H, GRAPH_PROPERTY_DIM, "gp_in", "gp_y1", "gp_d1", None, None, "gp_g1", "gp_gin")}
        sc = ptr(self.scal)
        for ti, t in enumerate(self.tasks):
            td = d.task[ti]
            td.kind = TASK_KIND[t]
            td.g_scale, td.loss_sum = sc + 4 * ti, ptr(self.loss_sums) + 4 * ti
            td.gemm_ws, td.gemm_ws_bytes = ptr(self.task_gemm_ws[ti]), self.task_gemm_ws[ti].numel()
            td.loss_ws, td.loss_ws_bytes = ptr(self.task_loss_ws[ti]), self.task_loss_ws[ti].numel()
            if t in mlp_cfg:
                k_in, k_hid, k_out, x, y1, d1, y2, g_out, g_hid, g_in = mlp_cfg[t]
                m = td.mlp
                m.k_in, m.k_hid, m.k_out, m.site = k_in, k_hid, k_out, 100 + ti
                for i, dom in enumerate(D):
                    for a, n in (("w0", "mlp.0.weight"), ("b0", "mlp.0.bias"), ("w3", "mlp.3.weight"), ("b3", "mlp.3.bias")):
                        full = f"heads.{t}.{dom}.{n}"
                        getattr(m, "off_" + a)[i] = self.off[full]
                        getattr(m, "tg_" + a)[i] = self._TG(ti, full)
                m.x, m.y1, m.d1, m.g_hid, m.g_in = ptr(hd[x]), ptr(hd[y1]), ptr(hd[d1]), ptr(hd[g_hid]), ptr(hd[g_in])
                m.y2 = ptr(self.gp_y2) if t == "graph_prop" else ptr(hd[y2])
                m.g_out = ptr(self.gp_g2) if t == "graph_prop" else ptr(hd[g_out])
            if t == "node_feat_mask":
                td.nfm_target = ptr(hd["nfm_tgt"])
            if t in ("node_contrast", "graph_contrast"):
                td.ntx_sums = sc + 4 * (16 + (0 if t == "node_contrast" else self.D))
            if t == "graph_contrast":
                td.pool_mean, td.pool_max, td.g_mean, td.g_max = ptr(hd["gc_mean"]), ptr(hd["gc_max"]), ptr(hd["gc_gmean"]), ptr(hd["gc_gmax"])
            if t == "domain_adv":
                m = td.mlp
                m.x, m.y1, m.d1, m.y2, m.g_out, m.g_hid, m.g_in = (ptr(hd[k]) for k in ("da_in", "da_y1", "da_d1", "da_logits", "da_glogits", "da_g1", "da_gin"))
                for a, n in (("w0", "mlp.0.weight"), ("b0", "mlp.0.bias"), ("w3", "mlp.3.weight"), ("b3", "mlp.3.bias")):
                    full = f"heads.domain_adv.classifier.{n}"
                    setattr(td, "lp_off_" + a, self.off[full])
                    setattr(td, "lp_tg_" + a, self._TG(ti, full))
                td.lp_site, td.da_classes, td.da_hidden, td.da_dropout = 100 + ti, len(D), DA_HIDDEN, DA_DROPOUT
            if t == "link_pred":
                td.lp_labels = ptr(self.lp_lab)
                for a in ("feat", "y1", "d1", "gy1", "gfeat", "ghs", "ghd"):
                    setattr(td, "lp_" + a, ptr(hd["lp_" + a]))
                td.lp_y2, td.lp_p, td.lp_gp, td.lp_gy2 = ptr(self.lp_y2), ptr(self.lp_p), ptr(self.lp_gp), ptr(self.lp_gy2)
                for a, n in (("w0", "mlp.0.weight"), ("b0", "mlp.0.bias"), ("w3", "mlp.3.weight"), ("b3", "mlp.3.bias")):
                    full = f"heads.link_pred.predictor.{n}"
                    setattr(td, "lp_off_" + a, self.off[full])
                    setattr(td, "lp_tg_" + a, self._TG(ti, full))
                td.lp_site = 100 + ti
        self._desc = d
        main_h = torch.cuda.current_stream(self.device).cuda_stream
        self._stream_arr = (C.c_void_p * self.T)(*[(s.cuda_stream if s is not None else main_h) for s in self.task_streams])
        return d

    def _fill_desc(self, p: StepPlan, inp: StepInputs):
        d = getattr(self, "_desc", None) or self._init_desc()
        D = self.domains
        d.N, d.E, d.S, d.max_seg, d.num_tiles = p.N, p.E, p.S, p.max_seg, p.num_tiles
        d.max_seg_edges, d.seg_eptr = p.max_seg_edges, p.d32["seg_eptr"]
        for k in range(2):
            d.fwd_cut_seg[k], d.fwd_cut_row[k] = p.fwd_cuts[k] if k < len(p.fwd_cuts) else (0, 0)
        d.training, d.dropout_p = int(self.model.training), float(self.dropout_p)
        d.dp_exchange = int(self.grad_sync is not None or self.parts_beside_backward)     # publish when each part's gradients are final
        d.upload_on_aux = int(self.upload_on_aux)
        sc = self.scal.data_ptr()                    # this step's upload set
        for ti, t in enumerate(self.tasks):
            td = d.task[ti]
            td.g_scale = sc + 4 * ti
            if t in ("node_contrast", "graph_contrast"):
                td.ntx_sums = sc + 4 * (16 + (0 if t == "node_contrast" else self.D))
            if t == "link_pred":
                td.lp_labels = self.lp_lab.data_ptr()
        self._epoch += 1
        d.epoch, d.sync_flags = self._epoch, (self.sync_flags.data_ptr() if self.use_gates else None)
        d.seed = (self.seed * 1000003 + self.step_count) & (2 ** 64 - 1)
        d.seg_ptr, d.seg_dom, d.src_row, d.tiles = p.d32["seg_ptr"], p.d32["seg_dom"], p.d32["src_row"], p.d32["tiles"]
        d.edge_index, d.rowmask = p.d64["edge_index"], p.d64.get("rowmask")
        for i, v in enumerate(p.task_row):
            d.task_row[i] = v
        c = 0
        d.task_seg[0] = 0
        for ti in range(self.T):
            c += sum(1 for s in p.seg_task if s == ti)
            d.task_seg[ti + 1] = c
        d.x_all, d.x_rows = inp.x_all.data_ptr(), inp.x_all.size(0)
        groups = self._encoder_groups(p)
        d.enc_groups = len(groups)
        if groups:
            d.enc_gseg[0] = groups[0][2]
            for g, (ti, dom, lo, hi) in enumerate(groups):
                d.enc_gseg[g + 1] = hi
                d.enc_tg_w[g], d.enc_tg_b[g] = self._TG(ti, f"input_encoders.{dom}.linear.weight"), self._TG(ti, f"input_encoders.{dom}.linear.bias")
                d.enc_tg_gamma[g], d.enc_tg_beta[g] = self._TG(ti, f"input_encoders.{dom}.batch_norm.weight"), self._TG(ti, f"input_encoders.{dom}.batch_norm.bias")
        T_ = float(self.temperature)
        for ti, t in enumerate(self.tasks):
            td = d.task[ti]
            td.row0, td.row1 = p.task_row[ti], p.task_row[ti + 1]
            rows = {"node_feat_mask": getattr(p, "nfm_rows", None), "node_contrast": getattr(p, "nc_rows", None),
                    "graph_contrast": getattr(p, "gc_rows", None), "graph_prop": getattr(p, "gp_rows", None)}.get(t)
            if rows is not None:
                for i, v in enumerate(rows):
                    td.mlp.rows[i] = v
            if t == "node_feat_mask":
                td.idx, td.num_idx = p.d64["nfm_idx"], p.nfm_rows[-1]
            elif t == "node_contrast":
                td.idx, td.num_idx = p.d64["nc_idx"], p.nc_rows[-1]
            if t in ("node_contrast", "graph_contrast"):
                ns, slot0 = (p.nc_n, 0) if t == "node_contrast" else (p.gc_n, self.D)
                td.temperature = T_
                for di, n in enumerate(ns):
                    td.ntx_n[di] = n
                ws = self._ntx_workspace(slot0, ns)
                td.ntx_ws[0], td.ntx_ws_bytes[0] = ws.data_ptr(), ws.numel()
            if t == "graph_contrast":
                td.pool_ptr, td.pool_gid, td.pool_B, td.pool_r0, td.pool_M = p.d32["gc_ptr"], p.d64["gc_gid"], p.gc_B, p.gc_r0, p.gc_M
            if t == "graph_prop":
                td.pool_ptr, td.pool_gid, td.pool_B, td.pool_r0, td.pool_M = p.d32["gp_ptr"], p.d64["gp_gid"], p.gp_B, p.gp_r0, p.gp_M
                td.labels = inp.graph_props.data_ptr()
            if t == "domain_adv":
                td.pool_ptr, td.pool_gid, td.pool_B, td.pool_r0, td.pool_M = p.d32["da_ptr"], p.d64["da_gid"], p.da_B, p.da_r0, p.da_M
                td.da_labels, td.da_lambda, td.da_dropout = p.d64["da_labels"], float(self.grl_lambda), float(self.da_dropout)
            if t == "link_pred":
                td.lp_K, td.lp_edges = p.lp_K, p.d64["lp_edges"]
                td.lp_pos = p.d32.get("lp_pos")
                if self._lp_segmented(p):
                    d.lp_seg_ptr, d.lp_seg_eptr, d.lp_S = p.d32["lp_seg_ptr"], p.d32["lp_seg_eptr"], p.lp_S
                    d.lp_max_seg_rows, d.lp_max_seg_edges, d.lp_rows_end = p.lp_max_rows, p.lp_max_edges, p.lp_rows_end
                else:
                    d.lp_S = 0
        return d

    def _encoder_groups(self, p: StepPlan):
        """(task index, domain, first segment, one-past-last segment) for every (task, domain) pair whose encoder receives a
        gradient this step; pairs that dropped out get their gradient slots zeroed."""
        seg_of: Dict[Tuple[int, int], List[int]] = {}
        for si, (tt, dd) in enumerate(zip(p.seg_task, p.seg_dom)):
            seg_of.setdefault((tt, dd), []).append(si)
        groups = []
        for ti, t in enumerate(self.tasks):
            if t == "node_feat_mask":
                continue
            for di, dom in enumerate(self.domains):
                segs = seg_of.get((ti, di), [])
                if not segs:
                    for key in ("linear.weight", "linear.bias", "batch_norm.weight", "batch_norm.bias"):
                        n = f"input_encoders.{dom}.{key}"
                        self.task_grads[ti, self.off[n]:self.off[n] + self.numel[n]].zero_()
                    continue
                groups.append((ti, dom, segs[0], segs[-1] + 1))
        for a, b in zip(groups[:-1], groups[1:]):
            if a[3] != b[2]:
                raise L.GnnmpError("engine: encoder gradient groups are not contiguous (segments must be task-major)")
        return groups

    def _forward_backward_native(self, p: StepPlan, inp: StepInputs) -> None:
        d = self._fill_desc(p, inp)
        main = torch.cuda.current_stream(self.device)
        self._chk(self.lib.gmp_pretrain_step_fwd_bwd(C.byref(d), main.cuda_stream, self._stream_arr, self.aux_stream.cuda_stream),
                  "gmp_pretrain_step_fwd_bwd")


class StepPrefetcher:
    """Runs engine.prepare() for upcoming steps on a background thread.  The device half of a step is one long C call
    (gmp_pretrain_step_fwd_bwd) that releases the GIL, so index drawing for step t+1 overlaps the launches of step t.
    The RNG stream is unchanged: only this thread draws, in step order."""

    def __init__(self, engine: StepEngine, inputs, gen: torch.Generator, depth: int = 3) -> None:
        import queue
        import threading
        self.q: "queue.Queue" = queue.Queue(maxsize=depth)
        self._err = None

        # producer time in prepare() (host-order draws: in plan(), with draw() on its own thread in draw_s), consumer time blocked in get()
        self.busy_s, self.draw_s, self.wait_s, self.items = 0.0, 0.0, 0.0, 0

        def work() -> None:
            import time as _t
            try:
                if engine.rng_mode == "device":
                    # the draws of the next LOOK inputs are in flight on the GPU while this thread plans the current one: a ticket
                    # enqueued on the aux stream runs within one step, and this thread is `depth` steps ahead of the launcher
                    from collections import deque
                    LOOK, ahead, it = 3, deque(), iter(inputs)
                    done = False
                    while True:
                        while not done and len(ahead) < LOOK:
                            try:
                                nxt = next(it)
                            except StopIteration:
                                done = True
                                break
                            ahead.append((nxt, engine.enqueue_draws(nxt)))
                        if not ahead:
                            break
                        inp, ticket = ahead.popleft()
                        t0 = _t.perf_counter()
                        item = (inp, engine.prepare(inp, gen, ticket))
                        self.busy_s += _t.perf_counter() - t0
                        self.q.put(item)
                    self.q.put(None)
                    return
                for inp in inputs:
                    t0 = _t.perf_counter()
                    art = engine.draw(inp, gen)
                    self.draw_s += _t.perf_counter() - t0
                    t0 = _t.perf_counter()
                    item = (inp, (art, engine.plan(inp, art)))
                    self.busy_s += _t.perf_counter() - t0
                    self.q.put(item)
            except BaseException as e:           # surfaced on the consumer side
                self._err = e
            self.q.put(None)

        # (One thread draws and lays out.  A second, layout-only stage and shorter CPython switch intervals measured equal or worse in round 2
        # -- three Python threads on one GIL -- and were removed in round 3.)
        self.thread = threading.Thread(target=work, daemon=True)
        self.thread.start()

    def __iter__(self):
        import time as _t
        while True:
            t0 = _t.perf_counter()
            item = self.q.get()
            self.wait_s += _t.perf_counter() - t0
            self.items += 1
            if item is None:
                if self._err is not None:
                    raise self._err
                return
            yield item
