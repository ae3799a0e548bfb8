"""World-size-2 gloo test of the data-parallel gradient exchange (CPU, no kernels):
the averaged per-task gradients equal the arithmetic mean of the two ranks' gradients, and the
PCGrad result is then identical on both ranks (SURVEY.md section 8e oracle for DP)."""
import os

import pytest
import random
import socket

import torch
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


class Toy(nn.Module):
    def __init__(self):
        super().__init__()
        self.shared = nn.Linear(3, 4)
        self.head_a, self.head_b = nn.Linear(4, 2), nn.Linear(4, 1)


def _losses(m, x):
    h = torch.tanh(m.shared(x))
    return {"a": (m.head_a(h) ** 2).sum(), "b": -(m.head_b(h)).sum() - (m.head_a(h) ** 2).sum() * 0.5}


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from gnn_pretraining_amd import dist as D
    from gnn_pretraining_amd.pretrain.control import GradientSurgery
    D.init_from_env("gloo")
    torch.manual_seed(0)
    m = Toy()                                           # identical replicas
    x = torch.randn(5, 3, generator=torch.Generator().manual_seed(100 + rank))   # different shard per rank
    gs = GradientSurgery(torch.device("cpu"), grad_sync=D.FlatGradSync(), shuffle_rng=random.Random(7))
    gs.apply_gradient_surgery(m, _losses(m, x), ["a", "b"])
    out[rank] = {n: (None if p.grad is None else p.grad.clone()) for n, p in m.named_parameters()}
    dist.destroy_process_group()


def test_dp_pcgrad_equals_mean_of_single_rank_gradients():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    # single-process oracle: mean of the two ranks' per-task gradients, then the same PCGrad
    from gnn_pretraining_amd.pretrain.control import GradientSurgery
    torch.manual_seed(0)
    m = Toy()
    per_rank = []
    for r in range(world):
        x = torch.randn(5, 3, generator=torch.Generator().manual_seed(100 + r))
        g = {}
        for t, l in _losses(m, x).items():
            m.zero_grad(set_to_none=True); l.backward(retain_graph=True)
            g[t] = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
        per_rank.append(g)
    mean = {t: {n: (per_rank[0][t][n] + per_rank[1][t][n]) / 2 for n in per_rank[0][t]} for t in per_rank[0]}
    order = ["a", "b"]; random.Random(7).shuffle(order)
    final, _ = GradientSurgery(torch.device("cpu"))._pcgrad(mean, order)
    for r in range(world):
        for n, g in final.items():
            torch.testing.assert_close(out[r][n], g, rtol=1e-5, atol=1e-6)
    for n in out[0]:                                    # replicas agree exactly
        if out[0][n] is not None:
            assert torch.equal(out[0][n], out[1][n])


def test_overlapped_exchange_groups_parts_into_messages(monkeypatch):
    """dist.OverlappedGradSync: which parts travel together (GMP_DP_GROUPS) and which part each message waits for."""
    import torch
    from gnn_pretraining_amd.dist import OverlappedGradSync
    base = torch.zeros(7 * 64)
    parts = [[(64 * i, 32), (64 * i + 32, 16)] for i in range(7)]          # two slices per part
    monkeypatch.delenv("GMP_DP_GROUPS", raising=False)
    s = OverlappedGradSync(base, parts, comm=None)
    assert s.wait_part == [0, 2, 4, 5, 6] and [p.n for p in s.parts] == [2, 4, 4, 2, 2]
    assert s.total == 7 * 48 and abs(sum(p.share for p in s.parts) - 1.0) < 1e-12
    # a message's table: source offsets of its slices, then the exclusive prefix of their lengths
    assert s.parts[1].table.tolist() == [64, 96, 128, 160, 0, 32, 48, 80, 96]
    monkeypatch.setenv("GMP_DP_GROUPS", "0|1|2|3|4|5|6")
    assert OverlappedGradSync(base, parts, comm=None).wait_part == list(range(7))
    for bad in ("0|1|2|3|4|5", "0|2,1|3|4|5|6", "0|1|1|2|3|4|5|6"):
        monkeypatch.setenv("GMP_DP_GROUPS", bad)
        with pytest.raises(ValueError):
            OverlappedGradSync(base, parts, comm=None)


def test_owner_shares_are_contiguous_cover_everything_and_balance():
    """dist.split_by_weight (the tensor -> owner-rank map of the sharded exchange): contiguous shares, every item once, empty shares allowed,
    and no share heavier than the fair share plus one item."""
    import random
    from gnn_pretraining_amd.dist import split_by_weight
    rng = random.Random(3)
    for world in (1, 2, 3, 8):
        for n in (0, 1, 5, 62):
            w = [rng.choice([4, 256, 512, 65536, 131072]) for _ in range(n)]
            c = split_by_weight(w, world)
            assert len(c) == world + 1 and c[0] == 0 and c[-1] == n and all(a <= b for a, b in zip(c[:-1], c[1:]))
            if n:
                fair, big = sum(w) / world, max(w)
                assert all(sum(w[a:b]) <= fair + big for a, b in zip(c[:-1], c[1:]))
    assert split_by_weight([10, 10, 10], 3) == [0, 1, 2, 3]
