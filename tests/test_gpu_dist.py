"""Data-parallel exchange of the stacked engine on the GPU: two ranks (two processes sharing the one
GPU of the test box, gloo transport) run the same replica on different shards; after the exchange every rank's per-task
gradient matrix equals the arithmetic mean of the two single-rank matrices on every (task, tensor) pair that carries a
gradient, and after the identical PCGrad / clip / AdamW the replicas are bit-identical."""
import os
import random
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _build(seed, sync):
    from gnn_pretraining_amd.engine import StepEngine
    from gnn_pretraining_amd.models.pretrain_model import PretrainableGNN
    from gnn_pretraining_amd.pretrain import pretrain as PT
    dev = torch.device("cuda:0")
    torch.manual_seed(seed)
    model = PretrainableGNN(device=dev, domain_names=PT.PRETRAIN_DOMAINS["s4"], task_names=PT.ACTIVE_TASKS["s4"])
    model.train()
    eng = StepEngine(model, PT.ACTIVE_TASKS["s4"], PT.PRETRAIN_DOMAINS["s4"], dev, seed=seed, shuffle_rng=random.Random(5),
                     grad_sync=sync, rng_mode="vectorized")
    return model, eng


def _inputs(rank, eng):
    from gnn_pretraining_amd import synthetic as S
    from gnn_pretraining_amd.engine import StepInputs
    from gnn_pretraining_amd.pretrain import pretrain as PT
    gen = torch.Generator().manual_seed(1000 + rank)                       # a different shard per rank
    return StepInputs(S.pretrain_step_batches(gen, PT.PRETRAIN_DOMAINS["s4"]), eng.device, eng.dpad), gen


def _worker(rank, world, port, out, overlap, mode="allreduce"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), GMP_DP_OVERLAP=overlap, GMP_DP_MODE=mode)
    import torch.distributed as dist
    from gnn_pretraining_amd import dist as D
    D.init_from_env("gloo")
    model, eng = _build(3, D.FlatGradSync())
    inp, gen = _inputs(rank, eng)
    eng.step(inp, gen)
    torch.cuda.synchronize()
    first = {"tg": eng.task_grads.cpu(), "flat": eng.flat.cpu(), "final": eng.final_grad.cpu(), "normsq": eng.normsq.cpu(), "flags": eng.flags.cpu(),
             "gate_err": int(eng.sync_flags[63].item()) if eng.use_gates else -1}     # 1: a cross-stream gate timed out in the first step (two processes share the GPU here)
    eng.step(inp, gen)                                                      # a second step: the events / message buffers are reused
    torch.cuda.synchronize()
    sync = eng._shard_sync_obj if mode == "sharded" else eng._packed_sync
    parts = [list(zip(p.table[:p.n].tolist(), (p.table[p.n + 1:] - p.table[p.n:-1]).tolist())) for p in getattr(sync, "parts", [sync])] if mode != "sharded" else []
    extra = {"own": [m["own_k"] for m in sync.msgs], "rs_floats": sync.total_rs, "ag_floats": sync.total_ag} if mode == "sharded" else {}
    out[rank] = dict(first, flat2=eng.flat.cpu(), m2=eng.exp_avg.cpu(), v2=eng.exp_avg_sq.cpu(), steps=eng.steps.cpu(), parts=parts, kind=type(sync).__name__, **extra)
    dist.destroy_process_group()


@pytest.fixture(scope="module")
def flat_exchange():
    """The one-message exchange after the backward (GMP_DP_OVERLAP=0): the result the overlapped one must reproduce bitwise."""
    world, port = 2, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_worker, args=(world, port, out, "0"), nprocs=world, join=True)
    return {r: out[r] for r in range(world)}


def test_overlapped_exchange_sends_the_same_slices_and_gives_the_same_bits(flat_exchange):
    world, port = 2, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_worker, args=(world, port, out, "1"), nprocs=world, join=True)
    assert out[0]["kind"] == "OverlappedGradSync" and flat_exchange[0]["kind"] == "PackedGradSync"
    assert len(out[0]["parts"]) == 5                                        # heads | layers 4+3 | layers 2+1 | layer 0 | mask token + encoders
    cover = lambda parts: sorted((o, n) for part in parts for (o, n) in part)
    def merged(sl):                                                         # union of half-open float ranges
        res = []
        for o, n in sl:
            if res and res[-1][1] == o:
                res[-1][1] = o + n
            else:
                assert not res or res[-1][1] < o, "slices overlap"
                res.append([o, o + n])
        return res
    assert merged(cover(out[0]["parts"])) == merged(cover(flat_exchange[0]["parts"]))
    for r in range(world):
        for key in ("tg", "flat", "flat2"):
            assert torch.equal(out[r][key], flat_exchange[r][key]), key
    assert torch.equal(out[0]["flat2"], out[1]["flat2"])


@pytest.mark.parametrize("overlap", ["1"])
def test_engine_dp_exchange_is_the_mean_and_replicas_stay_identical(overlap):
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out, overlap), nprocs=world, join=True)
    # single-rank per-task gradients of the two shards, same replica, no exchange, no update
    def single_rank():
        res = []
        for r in range(world):
            model, eng = _build(3, None)
            inp, gen = _inputs(r, eng)
            eng.step(inp, gen, apply_update=False)
            torch.cuda.synchronize()
            res.append(eng.task_grads.cpu())
        return res, eng
    singles, eng = single_rank()
    has, off, numel, names = eng.has_static, eng.off, eng.numel, eng.names
    mean = (singles[0] + singles[1]) / 2
    got = out[0]["tg"]
    checked, wrong = 0, []
    for k, n in enumerate(names):
        for t in range(got.size(0)):
            if has[k][t]:
                a, b = got[t, off[n]:off[n] + numel[n]], mean[t, off[n]:off[n] + numel[n]]
                scale = max(float(b.abs().max()), 1e-6)
                if float((a - b).abs().max()) > 1e-5 * scale + 1e-9:
                    wrong.append(f"{n} task {t}: {float((a - b).abs().max()):.2e} of {scale:.2e}")
                checked += 1
    if wrong:         # say which side moved: the single-rank reference must at least repeat itself
        again, _ = single_rank()
        rep = [torch.equal(a, b) for a, b in zip(singles, again)]
        raise AssertionError(f"{len(wrong)} of {checked} (task, tensor) pairs off the mean of the single-rank gradients; single-rank runs repeat bitwise: {rep}; "
                             f"gates {eng.use_gates}; gate time-out word of the ranks' first step: {[out[r]['gate_err'] for r in range(world)]}; first: {wrong[:6]}")
    assert checked > 100
    assert torch.equal(out[0]["flat"], out[1]["flat"])                      # identical update on both ranks
    assert torch.equal(out[0]["tg"], out[1]["tg"])


def test_sharded_exchange_equals_the_allreduce_exchange_bitwise():
    """dist.ShardedGradSync: every tensor has one owner rank; reduce-scatter of the per-task gradients, PCGrad on the owned
    tensors only, all-gather of the combined gradient, the foreign pass, then norm / clip / AdamW everywhere.  Two ranks on different shards,
    two steps: combined gradient, clip norm, flags, step counts, parameters and both Adam moments BITWISE equal to the all-reduce exchange's
    (with two ranks a sum of two addends is the same in either collective), replicas identical, every tensor owned exactly once, fewer floats sent."""
    world = 2
    res = {}
    for mode in ("allreduce", "sharded"):
        out = mp.Manager().dict()
        mp.spawn(_worker, args=(world, _free_port(), out, "1", mode), nprocs=world, join=True)
        res[mode] = {r: out[r] for r in range(world)}
    a, s = res["allreduce"], res["sharded"]
    assert s[0]["kind"] == "ShardedGradSync" and a[0]["kind"] == "OverlappedGradSync"
    for r in range(world):
        for key in ("final", "normsq", "flags", "flat", "flat2", "m2", "v2", "steps"):
            assert torch.equal(s[r][key], a[r][key]), (r, key)
    assert torch.equal(s[0]["flat2"], s[1]["flat2"])
    # ownership: the ranks' runs of tensor indices are disjoint and together cover every tensor once
    owned = sorted(k for r in range(world) for msg in s[r]["own"] for (k0, k1) in msg for k in range(k0, k1))
    assert owned == list(range(len(owned))) and len(owned) == int(a[0]["flags"].numel())
    # bytes: a ring all-reduce moves 2 (W-1)/W x message, reduce-scatter + all-gather (W-1)/W x (message + combined gradient)
    allreduce_floats = 2 * sum(n for part in a[0]["parts"] for (_, n) in part)
    assert s[0]["rs_floats"] // world + s[0]["ag_floats"] // world < 0.8 * allreduce_floats
