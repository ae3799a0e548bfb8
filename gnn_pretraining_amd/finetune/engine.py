"""One full-graph node-classification fine-tune step as an explicit kernel sequence (BASELINE.json configs[4]: Cora_NC).

The reference's epoch on Cora_NC is ONE optimisation step over the whole graph (src/finetune/finetune.py:162-179, batch_size -1):
`model(data)` = InputEncoder (Linear 1433 -> 256, BN, ReLU, dropout) -> 5 GIN layers on all 2,708 nodes -> Linear 256 -> 7,
cross-entropy (mean) on the 140 training nodes, `loss.backward()`, `AdamW(model.param_groups).step()`
(src/models/finetune_model.py:38-64: encoder / head lr 1e-3, backbone 1e-4, AdamW's default weight decay 0.01, no clipping).
The module path (models/finetune_model.py on autograd Functions) runs that in 3.4 ms, nearly all of it host time in ~200 small
autograd nodes.  Here the same step is ~60 launches with no autograd and no host synchronisation: the graph's CSR is built once,
the encoder's K = 1,433 is padded to 1,440 (a multiple of the GEMM's 32-deep K-step; the weight's padded columns stay zero), and
the optimizer is the pre-training engine's multi-tensor AdamW over one flat buffer.

`FinetuneGNN` stays the owner of the parameters (its tensors become views into the flat buffer, `state_dict()` keys unchanged),
so checkpoints, evaluation and the reference-shaped loop around it are untouched.

Round 3: shapes, pointers and the training rows are the same every step, so the whole step is captured ONCE in a hipGraph
(torch.cuda.CUDAGraph over the ctypes launches) and replayed: ~65 ctypes crossings + launches per step (1.35 ms, host-bound) become
one replay.  What changed per step -- the dropout seed -- is read from a device word the graph's last node increments
(gmp_bn_config.seed_dev, gmp_counter_add), so replay k draws exactly the masks the eager step k draws (tests/test_gpu_modules.py).
The encoder GEMM (172 output tiles for 256 CUs) ran as three K-slices (38 us; unsliced since: 33), and the one-segment BatchNorms (2,708 rows) take the medium
regime of csrc/batchnorm.hip (one launch instead of four).  Measured (MI355X, profiles/README.md round 3): the step was NOT host-bound as
round 2 believed -- its kernels add up to 1.43 ms (BatchNorm 0.58, GEMMs 0.56) -- so the replay (host 0.76 ms) runs at the GPU's 1.38 ms;
a parallel graph branch for the weight-gradient GEMMs (GMP_FINETUNE_FORK=1) makes the replay itself cost 1.45 ms of host time.
Second half of round 3: BatchNorm as slabs over the whole chip in one launch (gmp_bn_config.sync: 34 -> 12 us per backward launch) brought the
kernels to 1.0 ms; the eager step with the weight-gradient GEMMs on the side stream then beats the replay (0.88 against 0.95 ms) and is the
default; the capture stays available (GMP_FINETUNE_GRAPH=1) and tested."""
from __future__ import annotations

import ctypes as C
import os as _os_mod
from typing import Dict, List, Optional, Tuple

import torch
from torch import Tensor

from .. import _lib as L, ops
from ..models.finetune_model import LR_BACKBONE, LR_FINETUNE, FinetuneGNN
from ..models.gnn import DROPOUT_RATE, GNN_HIDDEN_DIM, GNN_NUM_LAYERS

H = GNN_HIDDEN_DIM
NT, NN, TN = 0, 1, 2
ADAMW_WEIGHT_DECAY = 0.01           # torch.optim.AdamW default: the reference passes none (finetune.py:363)


def _i32(xs):
    return (C.c_int32 * len(xs))(*[int(v) for v in xs])


def _i64(xs):
    return (C.c_int64 * len(xs))(*[int(v) for v in xs])


class NodeClassificationEngine:
    def __init__(self, model: FinetuneGNN, x: Tensor, edge_index: Tensor, device, seed: int = 0) -> None:
        self.model, self.device, self.lib = model, torch.device(device), L.lib()
        self.seed, self.step_count, self.dropout_p = seed, 0, DROPOUT_RATE
        dev = self.device
        self.N, self.d_in = int(x.size(0)), int(x.size(1))
        self.dpad = (self.d_in + 31) // 32 * 32
        self.x = torch.zeros(self.N, self.dpad, device=dev)
        self.x[:, :self.d_in] = x.to(dev)
        self.csr = ops.csr_build(edge_index.to(dev).contiguous(), self.N)
        self.classes = int(model.classification_head.mlp[0].weight.size(0))
        self._flatten()
        f = lambda *s: torch.empty(*s, device=dev)
        N, Lr = self.N, GNN_NUM_LAYERS
        self.z0, self.h = f(N, H), [f(N, H) for _ in range(Lr + 1)]
        self.a, self.z1, self.r1, self.z2 = [f(N, H) for _ in range(Lr)], [f(N, 2 * H) for _ in range(Lr)], [f(N, 2 * H) for _ in range(Lr)], [f(N, H) for _ in range(Lr)]
        self.stat = {k: f(Lr, c) for k, c in (("m1", 2 * H), ("s1", 2 * H), ("m2", H), ("s2", H))}
        self.enc_mean, self.enc_rstd = f(H), f(H)
        self.logits, self.gA, self.gB, self.ga, self.gW, self.gW2 = f(N, self.classes), f(N, H), f(N, H), f(N, H), f(N, 2 * H), f(N, 2 * H)
        # per-layer g_u / g_z1 (+ one g_u for the encoder): the weight-gradient GEMMs read them on the side stream while the chain moves on
        self.gu_l, self.gz1_l = [f(N, H) for _ in range(Lr + 1)], [f(N, 2 * H) for _ in range(Lr)]
        self.rowdot = f(Lr, N)
        # the side stream must sit on a hardware queue of its own: in a process whose queues are taken (bench.py after the pre-training engine) a fresh
        # pool stream shared the main stream's queue and the forked step took 3.98 ms instead of 0.88 -- measured, not assumed (streams.py)
        if dev.type == "cuda":
            from .. import streams as ST
            self.side = ST.concurrent_streams(dev, 1)[0]
            # main <-> side dependencies by gates (a sleeping wave on a flag word, csrc/streams.hip) instead of events where the two streams
            # were measured on different hardware queues: a queue parked on an event wait costs the running one ~2 us per kernel boundary
            own_queue = not ST.share_queue(torch.cuda.current_stream(dev).cuda_stream, self.side.cuda_stream)
            self._gates_ok = own_queue and _os_mod.environ.get("GMP_FINETUNE_GATES", "1") != "0"
        else:
            self.side, self._gates_ok, own_queue = None, False, False
        self.sync_flags = torch.zeros(64, dtype=torch.int32, device=dev)
        self._epoch = 0
        self.side_ws = torch.empty(32 << 20, dtype=torch.uint8, device=dev)
        self.seed_word = torch.zeros(1, dtype=torch.int64, device=dev)        # device copy of step_count for captured steps (gmp_bn_config.seed_dev)
        import os as _os
        # Default since the slab BatchNorm: launch by launch, the weight-gradient GEMMs on the side stream (0.88 ms per step; the host needs
        # 0.55-0.68 ms for the 120 ctypes launches and keeps up).  GMP_FINETUNE_GRAPH=1: the step captured once and replayed (bitwise the
        # same numbers, host 0.43 ms per replay, 0.95 ms per step on ONE chain: replaying a graph with a second branch costs the host 1.45 ms on
        # this runtime -- ROCm 7.2, profiles/README.md round 3 -- so a captured step keeps its weight gradients in the chain).
        self.use_graph = dev.type == "cuda" and _os.environ.get("GMP_FINETUNE_GRAPH", "0") == "1"
        # (no queue of its own for the side stream -- every hardware queue of the process taken --: the fork would run behind main's kernels in
        # the same in-order queue, slower than the plain chain)
        self.fork_wgrads = _os.environ.get("GMP_FINETUNE_FORK", "0" if (self.use_graph or not own_queue) else "1") == "1"
        self._graph, self._graph_key, self._graph_step, self._graph_seen = None, None, -1, None
        self.seg_ptr = torch.tensor([0, N], dtype=torch.int32, device=dev)
        self.bn_ws = torch.empty(self.lib.gmp_bn_workspace_bytes(N, 2 * H, 1, N), dtype=torch.uint8, device=dev)
        # rendezvous words of the BatchNorm slab form (gmp_bn_config.sync: the graph's 2,708 rows as 128-row slabs over the whole chip, one
        # launch); every BatchNorm of the step runs on the main stream, so one buffer serves them all.  GMP_BN_SLABS=0: one workgroup per strip
        self.bn_sync = (torch.zeros(self.lib.gmp_bn_sync_bytes(2 * H, 1) // 4, dtype=torch.int32, device=dev)
                        if _os.environ.get("GMP_BN_SLABS", "1") != "0" else None)
        self.gemm_ws = torch.empty(32 << 20, dtype=torch.uint8, device=dev)
        self.loss_ws = torch.empty(self.lib.gmp_loss_workspace_bytes(N * H), dtype=torch.uint8, device=dev)
        self.loss_sum, self.g_scale = torch.zeros(1, device=dev), torch.ones(1, device=dev)
        self.mt_ws = torch.empty(self.lib.gmp_mt_workspace_bytes(self.K), dtype=torch.uint8, device=dev)
        self.normsq, self.metrics, self.flags = torch.zeros(1, device=dev), torch.zeros(2, dtype=torch.int32, device=dev), torch.zeros(self.K, dtype=torch.int32, device=dev)
        self._train_idx: Optional[Tensor] = None
        self._bn_calls = 0

    # ------------------------------------------------------------------ parameters: one flat buffer, the module's tensors view into it
    def _flatten(self) -> None:
        m, dev = self.model, self.device
        named = [(n, p) for n, p in m.named_parameters()]
        al4 = lambda v: (v + 3) // 4 * 4
        self.off: Dict[str, int] = {}
        self.numel: Dict[str, int] = {}
        o = 0
        for n, p in named:
            cnt = H * self.dpad if n == "input_encoder.linear.weight" else p.numel()      # encoder weight: rows padded to dpad columns
            self.off[n], self.numel[n] = o, cnt
            o += al4(cnt)
        self.P = o
        self.flat = torch.zeros(self.P, device=dev)
        for n, p in named:
            a = self.off[n]
            if n == "input_encoder.linear.weight":
                view = self.flat[a:a + H * self.dpad].view(H, self.dpad)
                view[:, :self.d_in].copy_(p.data)
                p.data = view[:, :self.d_in]                      # strided view: the padded columns are outside the parameter
            else:
                self.flat[a:a + p.numel()].copy_(p.data.reshape(-1))
                p.data = self.flat[a:a + p.numel()].view_as(p)
        self.names = [n for n, _ in named]
        self.K = len(self.names)
        self.grad = torch.zeros(1, self.P, device=dev)            # [tasks = 1, P]
        self.final_grad = torch.zeros(self.P, device=dev)
        self.exp_avg, self.exp_avg_sq = torch.zeros(self.P, device=dev), torch.zeros(self.P, device=dev)
        self.t_off = torch.tensor([self.off[n] for n in self.names], dtype=torch.int64, device=dev)
        self.t_len = torch.tensor([self.numel[n] for n in self.names], dtype=torch.int32, device=dev)
        trainable = {n: p.requires_grad for n, p in named}
        has = torch.zeros(self.K, 8, dtype=torch.uint8)
        lr = torch.zeros(self.K)
        for k, n in enumerate(self.names):
            has[k, 0] = 1 if trainable[n] else 0
            lr[k] = LR_BACKBONE if n.startswith("gnn_backbone.") else LR_FINETUNE
        self.has, self.lr = has.to(dev), lr.to(dev)
        self.wd = torch.full((self.K,), ADAMW_WEIGHT_DECAY, device=dev)
        self.steps = torch.zeros(self.K, device=dev)

    def _P(self, n: str) -> int:
        return self.flat.data_ptr() + 4 * self.off[n]

    def _G(self, n: str) -> int:
        return self.off[n]

    def _chk(self, rc: int, what: str) -> None:
        if rc:
            L.check(rc, what)

    def _cfg(self, relu: bool, dropout: bool, site: int) -> L.BnConfig:
        """Dropout seed of a launch = seed * 1000003 + step number; in a captured step the step number comes from the device word."""
        p = self.dropout_p if (dropout and self.model.training) else 0.0
        base = self.seed * 1000003
        sync = (self.bn_sync.data_ptr(), self.bn_sync.numel()) if self.bn_sync is not None else (None, 0)
        if self._seed_dev:
            return L.BnConfig(int(self.model.training), int(relu), 1e-5, 0.1, p, base & (2 ** 64 - 1), site, self._seed_dev, *sync)
        return L.BnConfig(int(self.model.training), int(relu), 1e-5, 0.1, p, (base + self.step_count) & (2 ** 64 - 1), site, None, *sync)

    _seed_dev = None

    def _gemm(self, st, mode, A, B, bias, Cc, M, N, K, lda, ldb, ldc, ws: Optional[Tensor] = None):
        self._chk(self.lib.gmp_gemm_f32(mode, A, B, bias, Cc, M, N, K, lda, ldb, ldc, 1.0, 0, 0, None if ws is None else ws.data_ptr(),
                                        0 if ws is None else ws.numel(), st), "gemm")

    def _wgrad(self, st, G, X, w_name: str, b_name: str, M_tn: int, N_out: int, ldx: int, ws: Optional[Tensor] = None):
        """dW = G^T X and db = colsum(G) over all rows (one group), straight into the gradient buffer."""
        g, ws = self.grad.data_ptr(), (self.gemm_ws if ws is None else ws)
        self._chk(self.lib.gmp_gemm_f32_grouped(TN, G, X, None, g, 1, _i32([0, self.N]), None, None, _i64([self._G(w_name)]), g, _i64([self._G(b_name)]),
                                                M_tn, N_out, 0, M_tn, ldx, N_out, 1.0, 0, 0, ws.data_ptr(), ws.numel(), st), "wgrad")

    # ------------------------------------------------------------------ forward (finetune_model.py:68-80, message passing on the full graph)
    def forward(self) -> Tensor:
        lib, N, P, c = self.lib, self.N, self._P, self.csr
        st = torch.cuda.current_stream(self.device).cuda_stream
        enc, sp = self.model.input_encoder, self.seg_ptr.data_ptr()
        # 2,708 x 1,440 -> 256 is 172 output tiles for 256 CUs; unsliced since round 3 (three K-slices + their reduction: 38 us, one launch: 33 --
        # gmp_gemm_f32_workspace_bytes now slices NT / NN only below ~100 tiles)
        self._gemm(st, NT, self.x.data_ptr(), P("input_encoder.linear.weight"), P("input_encoder.linear.bias"), self.z0.data_ptr(), N, H, self.dpad,
                   self.dpad, self.dpad, H, ws=self.gemm_ws)
        cfg = self._cfg(True, True, 1)
        self._chk(lib.gmp_bn_fwd(self.z0.data_ptr(), None, sp, None, 1, N, N, H, P("input_encoder.batch_norm.weight"), P("input_encoder.batch_norm.bias"),
                                 enc.batch_norm.running_mean.data_ptr(), enc.batch_norm.running_var.data_ptr(), self.enc_mean.data_ptr(),
                                 self.enc_rstd.data_ptr(), self.h[0].data_ptr(), C.byref(cfg), self.bn_ws.data_ptr(), self.bn_ws.numel(), st), "bn encoder")
        for l in range(GNN_NUM_LAYERS):
            pre, layer = f"gnn_backbone.layers.{l}.", self.model.gnn_backbone.layers[l]
            self._chk(lib.gmp_gin_aggregate_fwd(self.h[l].data_ptr(), c.rowptr.data_ptr(), c.col.data_ptr(), P(pre + "gin_conv.eps"), self.a[l].data_ptr(), N, H, st), "aggregate")
            self._gemm(st, NT, self.a[l].data_ptr(), P(pre + "gin_conv.nn.0.weight"), P(pre + "gin_conv.nn.0.bias"), self.z1[l].data_ptr(), N, 2 * H, H, H, H, 2 * H)
            bn1, cfg = layer.gin_conv.nn[1], self._cfg(True, False, 0)
            self._chk(lib.gmp_bn_fwd(self.z1[l].data_ptr(), None, sp, None, 1, N, N, 2 * H, P(pre + "gin_conv.nn.1.weight"), P(pre + "gin_conv.nn.1.bias"),
                                     bn1.running_mean.data_ptr(), bn1.running_var.data_ptr(), self.stat["m1"][l].data_ptr(), self.stat["s1"][l].data_ptr(),
                                     self.r1[l].data_ptr(), C.byref(cfg), self.bn_ws.data_ptr(), self.bn_ws.numel(), st), "bn1")
            self._gemm(st, NT, self.r1[l].data_ptr(), P(pre + "gin_conv.nn.3.weight"), P(pre + "gin_conv.nn.3.bias"), self.z2[l].data_ptr(), N, H, 2 * H, 2 * H, 2 * H, H)
            bn2, cfg = layer.batch_norm, self._cfg(True, True, 10 + l)
            self._chk(lib.gmp_bn_fwd(self.z2[l].data_ptr(), self.h[l].data_ptr(), sp, None, 1, N, N, H, P(pre + "batch_norm.weight"), P(pre + "batch_norm.bias"),
                                     bn2.running_mean.data_ptr(), bn2.running_var.data_ptr(), self.stat["m2"][l].data_ptr(), self.stat["s2"][l].data_ptr(),
                                     self.h[l + 1].data_ptr(), C.byref(cfg), self.bn_ws.data_ptr(), self.bn_ws.numel(), st), "bn2")
        self._gemm(st, NT, self.h[GNN_NUM_LAYERS].data_ptr(), P("classification_head.mlp.0.weight"), P("classification_head.mlp.0.bias"), self.logits.data_ptr(),
                   N, self.classes, H, H, H, self.classes)
        if self.model.training:
            self._bn_calls += 1             # num_batches_tracked only matters for a saved state_dict (momentum is fixed): flush_counters()
        return self.logits

    def flush_counters(self) -> None:
        if self._bn_calls:
            enc = self.model.input_encoder
            for bn in [enc.batch_norm] + [b for l in self.model.gnn_backbone.layers for b in (l.gin_conv.nn[1], l.batch_norm)]:
                bn.num_batches_tracked += self._bn_calls
            self._bn_calls = 0

    # ------------------------------------------------------------------ one optimisation step (finetune.py:162-179 + 318-320)
    def step(self, node_indices: Tensor, targets: Tensor, apply_update: bool = True) -> None:
        """loss = cross_entropy(model(data)[node_indices], targets) (mean); backward; AdamW.  Nothing is read back: loss().
        Training steps with an update are replayed from a hipGraph captured at the first such call for these index tensors."""
        if not (self.use_graph and apply_update and self.model.training):
            self._enqueue(node_indices, targets, apply_update, forked=self.fork_wgrads)
            self.step_count += 1
            return
        key = (node_indices.data_ptr(), targets.data_ptr(), int(node_indices.numel()))
        if self._graph_key != key:
            # an eager step first (it IS this call's step): code objects loaded, LDS attributes set, row buffers allocated -- nothing of
            # that may happen inside a capture.  The capture itself (tens of ms) waits for the SECOND call with the same index tensors:
            # a caller that builds fresh tensors per call never pays for it.
            self._enqueue(node_indices, targets, True)
            self.step_count += 1
            if self._graph_seen != key:
                self._graph_seen, self._graph_keep = key, (node_indices, targets)
                return
            main = torch.cuda.current_stream(self.device)
            main.synchronize()
            graph = torch.cuda.CUDAGraph()
            self._seed_dev = self.seed_word.data_ptr()
            try:
                with torch.cuda.graph(graph):
                    self._enqueue(node_indices, targets, True, forked=self.fork_wgrads)
                    self._chk(self.lib.gmp_counter_add(self.seed_word.data_ptr(), 1, torch.cuda.current_stream(self.device).cuda_stream), "seed word")
            finally:
                self._seed_dev = None
            self._bn_calls -= 1                                     # (forward() counted the capture pass, which ran nothing)
            self._graph, self._graph_key, self._graph_step = graph, key, -1
            self._graph_keep = (node_indices, targets)             # the captured launches hold these pointers
            return
        if self._graph_step != self.step_count:                    # eager steps ran in between (evaluation does not count): resynchronise
            self.seed_word.fill_(self.step_count)
        self._graph.replay()
        self.step_count += 1
        self._graph_step = self.step_count
        self._bn_calls += 1

    def _enqueue(self, node_indices: Tensor, targets: Tensor, apply_update: bool, forked: bool = False) -> None:
        """The step's launches on the current stream.  forked: the weight-gradient GEMMs (they only feed the gradient buffer) go to the
        side stream behind an event of the main one -- inside a capture a parallel branch of the graph -- and are joined before AdamW."""
        lib, N, P, c, Cn = self.lib, self.N, self._P, self.csr, self.classes
        main = torch.cuda.current_stream(self.device)
        st = main.cuda_stream
        side = self.side if (forked and self.side is not None) else None
        sst = side.cuda_stream if side is not None else st
        wws = self.side_ws if side is not None else self.gemm_ws

        gates = side is not None and self._gates_ok and not torch.cuda.is_current_stream_capturing()
        if gates:
            self._epoch += 1
        flags, epoch, nfork = self.sync_flags.data_ptr(), self._epoch, [0]

        def fork(by_gemm: bool = False) -> None:
            """The side stream may read what main has produced so far.  by_gemm (gates only): the caller launches a GEMM on MAIN next, which
            carries the signal as it starts (gmp_gate_open_by_next_gemm) -- no one-thread launch on the chain."""
            if side is None:
                return
            if gates:
                k = nfork[0]
                nfork[0] = k + 1
                if by_gemm:
                    self._chk(lib.gmp_gate_open_by_next_gemm(flags + 4 * k, epoch), "gate open by gemm")
                else:
                    self._chk(lib.gmp_gate_open(flags + 4 * k, epoch, st), "gate open")
                self._chk(lib.gmp_gate_wait(flags, 1 << k, epoch, flags + 4 * 63, sst), "gate wait")
            else:
                side.wait_stream(main)

        def flush() -> None:                                        # the GEMM that was to carry the signal launched nothing: open the gate by hand
            if gates and lib.gmp_gate_open_pending():
                self._chk(lib.gmp_gate_open_by_next_gemm(None, 0), "cancel")
                self._chk(lib.gmp_gate_open(flags + 4 * (nfork[0] - 1), epoch, st), "gate open")

        self.forward()
        M = int(node_indices.numel())
        if self._train_idx is None or self._train_idx.numel() != M:
            f = lambda *s: torch.empty(*s, device=self.device)
            self._rows_logits, self._rows_g, self._rows_h = f(M, Cn), f(M, Cn), f(M, H)
            self._rows_gh = f(M, H)
        self._train_idx, self.num_targets = node_indices, M
        self.g_scale.fill_(1.0 / M)
        idx, tgt, g = node_indices.data_ptr(), targets.data_ptr(), self.grad.data_ptr()
        hL = self.h[GNN_NUM_LAYERS]
        self._chk(lib.gmp_row_gather(self.logits.data_ptr(), idx, None, self._rows_logits.data_ptr(), M, N, Cn, st), "logit rows")
        self._chk(lib.gmp_cross_entropy_sum_fwd(self._rows_logits.data_ptr(), tgt, M, Cn, self.loss_sum.data_ptr(), self.loss_ws.data_ptr(), self.loss_ws.numel(), st), "ce")
        self._chk(lib.gmp_cross_entropy_sum_bwd(self._rows_logits.data_ptr(), tgt, M, Cn, self.g_scale.data_ptr(), self._rows_g.data_ptr(), st), "ce bwd")
        # head: dW = g^T h[idx], db = colsum(g), g_h[idx] = g W  (only the M training rows carry a gradient)
        self._chk(lib.gmp_row_gather(hL.data_ptr(), idx, None, self._rows_h.data_ptr(), M, N, H, st), "h rows")
        self._chk(lib.gmp_gemm_f32_grouped(TN, self._rows_g.data_ptr(), self._rows_h.data_ptr(), None, g, 1, _i32([0, M]), None, None,
                                           _i64([self._G("classification_head.mlp.0.weight")]), g, _i64([self._G("classification_head.mlp.0.bias")]),
                                           Cn, H, 0, Cn, H, H, 1.0, 0, 0, None, 0, st), "head wgrad")
        self._gemm(st, NN, self._rows_g.data_ptr(), P("classification_head.mlp.0.weight"), None, self._rows_gh.data_ptr(), M, H, Cn, Cn, H, H)
        gcur, ga = self.gA, self.ga
        gcur.zero_()
        self._chk(lib.gmp_row_fill(gcur.data_ptr(), idx, self._rows_gh.data_ptr(), M, N, H, 0, st), "scatter g_h")
        sp, one = self.seg_ptr.data_ptr(), _i32([0, 1])
        pending_eps = None
        for l in reversed(range(GNN_NUM_LAYERS)):
            pre, layer = f"gnn_backbone.layers.{l}.", self.model.gnn_backbone.layers[l]
            gu, gz1 = self.gu_l[l], self.gz1_l[l]
            bn2, cfg = layer.batch_norm, self._cfg(True, True, 10 + l)
            self._chk(lib.gmp_bn_bwd(gcur.data_ptr(), self.z2[l].data_ptr(), self.h[l].data_ptr(), sp, None, 1, N, N, H, P(pre + "batch_norm.weight"),
                                     P(pre + "batch_norm.bias"), bn2.running_mean.data_ptr(), bn2.running_var.data_ptr(), self.stat["m2"][l].data_ptr(),
                                     self.stat["s2"][l].data_ptr(), gu.data_ptr(), g, g, one, _i64([self._G(pre + "batch_norm.weight")]),
                                     _i64([self._G(pre + "batch_norm.bias")]), 1, C.byref(cfg), self.bn_ws.data_ptr(), self.bn_ws.numel(), st), "bn2 bwd")
            fork(by_gemm=True)
            self._gemm(st, NN, gu.data_ptr(), P(pre + "gin_conv.nn.3.weight"), None, self.gW.data_ptr(), N, 2 * H, H, H, 2 * H, 2 * H)
            flush()
            if pending_eps is not None:                             # (the previous layer's aggregation backward is behind this fork too)
                self._eps_grad(sst, pending_eps)
                pending_eps = None
            self._wgrad(sst, gu.data_ptr(), self.r1[l].data_ptr(), pre + "gin_conv.nn.3.weight", pre + "gin_conv.nn.3.bias", H, 2 * H, 2 * H, wws)
            bn1, cfg = layer.gin_conv.nn[1], self._cfg(True, False, 0)
            self._chk(lib.gmp_bn_bwd(self.gW.data_ptr(), self.z1[l].data_ptr(), None, sp, None, 1, N, N, 2 * H, P(pre + "gin_conv.nn.1.weight"),
                                     P(pre + "gin_conv.nn.1.bias"), bn1.running_mean.data_ptr(), bn1.running_var.data_ptr(), self.stat["m1"][l].data_ptr(),
                                     self.stat["s1"][l].data_ptr(), gz1.data_ptr(), g, g, one, _i64([self._G(pre + "gin_conv.nn.1.weight")]),
                                     _i64([self._G(pre + "gin_conv.nn.1.bias")]), 1, C.byref(cfg), self.bn_ws.data_ptr(), self.bn_ws.numel(), st), "bn1 bwd")
            fork(by_gemm=True)
            self._gemm(st, NN, gz1.data_ptr(), P(pre + "gin_conv.nn.0.weight"), None, ga.data_ptr(), N, H, 2 * H, 2 * H, H, H)
            flush()
            self._wgrad(sst, gz1.data_ptr(), self.a[l].data_ptr(), pre + "gin_conv.nn.0.weight", pre + "gin_conv.nn.0.bias", 2 * H, H, H, wws)
            rowdot = self.rowdot[l]
            self._chk(lib.gmp_gin_aggregate_bwd_ex(ga.data_ptr(), c.rowptr_t.data_ptr(), c.col_t.data_ptr(), P(pre + "gin_conv.eps"), self.h[l].data_ptr(),
                                                   gu.data_ptr(), gcur.data_ptr(), rowdot.data_ptr(), N, H, st), "aggregate bwd")
            pending_eps = l                                          # eps gradient: a 5 us sum that only feeds the gradient buffer -- with the next fork
        self._eps_grad(st, pending_eps)                             # layer 0's: on main, in front of the encoder backward
        enc, cfg = self.model.input_encoder, self._cfg(True, True, 1)
        gu = self.gu_l[GNN_NUM_LAYERS]
        self._chk(lib.gmp_bn_bwd(gcur.data_ptr(), self.z0.data_ptr(), None, sp, None, 1, N, N, H, P("input_encoder.batch_norm.weight"),
                                 P("input_encoder.batch_norm.bias"), enc.batch_norm.running_mean.data_ptr(), enc.batch_norm.running_var.data_ptr(),
                                 self.enc_mean.data_ptr(), self.enc_rstd.data_ptr(), gu.data_ptr(), g, g, one, _i64([self._G("input_encoder.batch_norm.weight")]),
                                 _i64([self._G("input_encoder.batch_norm.bias")]), 1, C.byref(cfg), self.bn_ws.data_ptr(), self.bn_ws.numel(), st), "bn encoder bwd")
        self._wgrad(st, gu.data_ptr(), self.x.data_ptr(), "input_encoder.linear.weight", "input_encoder.linear.bias", H, self.dpad, self.dpad)
        if side is not None:                                        # every weight gradient is in the buffer
            if gates:
                self._chk(lib.gmp_gate_open(flags + 4 * 40, epoch, sst), "gate open")
                self._chk(lib.gmp_gate_wait(flags, 1 << 40, epoch, flags + 4 * 63, st), "gate wait")
            else:
                main.wait_stream(side)
        # AdamW over the flat buffer (the pre-training engine's multi-tensor kernels with one task: no projection, no clipping)
        self._chk(lib.gmp_mt_pcgrad_clip_adamw(g, self.P, 1, self.K, self.t_off.data_ptr(), self.t_len.data_ptr(), self.has.data_ptr(), _i32([0]), 1, 0, -1,
                                               self.flat.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                                               self.steps.data_ptr() if apply_update else None, self.lr.data_ptr(), self.wd.data_ptr(), 0.9, 0.999, 1e-8, 0.0,
                                               self.final_grad.data_ptr(), self.normsq.data_ptr(), self.metrics.data_ptr(), self.flags.data_ptr(),
                                               self.mt_ws.data_ptr(), self.mt_ws.numel(), int(apply_update), st), "adamw")

    def _eps_grad(self, stream: int, l: int) -> None:
        self._chk(self.lib.gmp_group_sum_1d(self.rowdot[l].data_ptr(), 1, _i32([0, self.N]), _i64([self._G(f"gnn_backbone.layers.{l}.gin_conv.eps")]),
                                            self.grad.data_ptr(), stream), "eps grad")

    def loss(self) -> float:
        """Mean cross-entropy of the last step (a read-back: the one place the loop synchronises, so the gates' time-out word and the slab
        BatchNorm's are looked at here too)."""
        v = float(self.loss_sum.item())
        if int(self.sync_flags[63].item()) != 0 or (self.bn_sync is not None and int(self.bn_sync[0].item()) != 0):
            raise L.GnnmpError("fine-tune engine: a cross-stream gate or a BatchNorm slab wait timed out (streams sharing a hardware queue, "
                               "a tool serialising kernels, or a sync buffer shared between streams): results since then are not to be trusted")
        return v / max(self.num_targets, 1)

    def gradient(self, name: str) -> Tensor:
        o = self.off[name]
        if name == "input_encoder.linear.weight":
            return self.final_grad[o:o + H * self.dpad].view(H, self.dpad)[:, :self.d_in]
        p = dict(self.model.named_parameters())[name]
        return self.final_grad[o:o + p.numel()].view_as(p)
