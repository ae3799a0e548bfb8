"""Device-side augmentation and masking (csrc/augment.hip: gmp_aug_two_views, gmp_aug_node_masks) against the reference's rules
(src/pretrain/augmentations.py:17-111, src/models/pretrain_model.py:67-88).  The device draws with Philox keys, so its random
choices are its own; what is checked is (a) every structural rule of the reference -- counts, sortedness, relabelling, edge
order, common-node sets, mask shapes -- and (b) the oracle ITSELF: the device's decisions (which nodes / edges / columns) are
turned into the permutations torch.randperm would have had to return, injected into oracle.augment.create_two_views, and the
two views the oracle then builds must equal the device's arrays exactly."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from gnn_pretraining_amd import ops, synthetic as S                       # noqa: E402
from gnn_pretraining_amd.engine import ViewArrays, device_views_to_host   # noqa: E402
from gnn_pretraining_amd.graph import Batch, Data                         # noqa: E402
from oracle import augment as OA, graph_ops as OG                         # noqa: E402
from oracle.harness import to_oracle, view_to_oracle                      # noqa: E402

DEV = torch.device("cuda:0")


def _batches():
    gen = torch.Generator().manual_seed(11)
    out = [S.domain_batch(gen, d, 8) for d in (7, 4, 37, 21)]
    out.append(S.domain_batch(gen, 2, 3))                                    # fewer than 3 features: no attribute mask possible
    out.append(S.domain_batch(gen, 21, 32))                                  # a validation-sized batch
    out.append(Batch.from_data_list([Data(torch.zeros(1, 4), torch.zeros(2, 0, dtype=torch.long), torch.zeros(1, dtype=torch.long), torch.zeros(12)),
                                     Data(torch.zeros(2, 4), torch.tensor([[0, 1], [1, 0]]), torch.zeros(1, dtype=torch.long), torch.zeros(12)),
                                     S.random_graph(gen, 4, 126.0, 200.0), S.random_graph(gen, 4, 3.0, 2.0)]))   # 1-, 2-node, large, tiny graphs
    return out


def _draw(b: Batch, seed: int, stream: int = 40):
    ptr = torch.tensor(b.ptr_host, dtype=torch.long, device=DEV)
    eptr = torch.tensor(b.edge_ptr_host, dtype=torch.long, device=DEV)
    dv = ops.aug_two_views(ptr, eptr, b.edge_index.to(DEV).contiguous(), b.ptr_host, b.edge_ptr_host, int(b.x.size(1)), seed, stream)
    return device_views_to_host(dv)


def test_views_obey_every_structural_rule_of_the_reference():
    saw_edge_drop = saw_attr_mask = saw_none = 0
    for bi, b in enumerate(_batches()):
        n = np.diff(np.asarray(b.ptr_host))
        F = int(b.x.size(1))
        for seed in range(12):
            views = _draw(b, 1000 * bi + seed)
            for v in views:
                kept = np.diff(v.ptr)
                assert np.array_equal(kept, np.where(n >= 3, n - np.maximum(1, (n * .2).astype(int)), n))          # augmentations.py:49-50
                assert (np.diff(v.rows) > 0).all()                                                                  # sorted kept nodes (:53)
                g_of = np.searchsorted(np.asarray(b.ptr_host), v.rows, side="right") - 1
                assert np.array_equal(np.bincount(g_of, minlength=len(n)), kept)
                full = OG.subgraph(torch.from_numpy(v.rows), b.edge_index, b.num_nodes).numpy()                     # PyG subgraph(relabel)
                if v.edges.shape[1] == full.shape[1]:
                    assert np.array_equal(v.edges, full)
                    saw_none += 1
                else:                                                                                               # some graph dropped edges
                    saw_edge_drop += 1
                    eg = np.searchsorted(v.ptr, full[0], side="right") - 1
                    pos = 0
                    for g in range(len(n)):
                        fg = full[:, eg == g]
                        k = int((np.searchsorted(v.ptr, v.edges[0], side="right") - 1 == g).sum())
                        mine = v.edges[:, pos:pos + k]
                        pos += k
                        ep = fg.shape[1]
                        assert k in (ep, ep - max(1, int(ep * .2)) if ep >= 3 else ep)                              # :36-37
                        # a subsequence of the graph's surviving edges, in their order
                        it = iter(map(tuple, fg.T))
                        assert all(any(e == x for x in it) for e in map(tuple, mine.T))
                if v.rowmask is not None:
                    saw_attr_mask += 1
                    assert F >= 3
                    bits_per_graph = [set(np.unique(v.rowmask[v.ptr[g]:v.ptr[g + 1]])) for g in range(len(n)) if kept[g]]
                    assert all(len(s) == 1 for s in bits_per_graph)                                                 # one column set per graph
                    for s in bits_per_graph:
                        m = int(next(iter(s)))
                        assert m == 0 or (bin(m).count("1") == max(1, int(F * .2)) and m < (1 << F))                # :23
            # common nodes (:77-85): kept in both views, as ids local to each view
            a, c = views
            want = np.intersect1d(a.rows, c.rows)
            assert np.array_equal(a.rows[a.common], want) and np.array_equal(c.rows[c.common], want)
    assert saw_edge_drop and saw_attr_mask and saw_none


class _Injected:
    """torch.randperm / torch.rand stand-ins that hand out prepared values in call order."""

    def __init__(self, values):
        self.values, self.pos = values, 0

    def take(self, kind, n=None):
        k, v = self.values[self.pos]
        assert k == kind, (self.pos, k, kind)
        self.pos += 1
        if kind == "perm":
            assert len(v) == n, (self.pos, len(v), n)
        return v


def _decisions(b: Batch, views):
    """The draws create_two_views (augmentations.py:61-74 order: randperm(n), rand, [randperm(E')], rand, [randperm(F)], view 1 then
    view 2, graph by graph) would have had to make for the device's choices."""
    F = int(b.x.size(1))
    seq = []
    ei = b.edge_index.numpy()
    for g in range(b.num_graphs):
        s, e = b.ptr_host[g], b.ptr_host[g + 1]
        n = e - s
        loc = ei[:, b.edge_ptr_host[g]:b.edge_ptr_host[g + 1]] - s
        for v in views:
            kept = v.rows[v.ptr[g]:v.ptr[g + 1]] - s
            if n >= 3:
                rest = np.setdiff1d(np.arange(n), kept)
                seq.append(("perm", torch.from_numpy(np.concatenate([kept, rest]))))
            relabel = np.full(n, -1)
            relabel[kept] = np.arange(len(kept))
            alive = relabel[loc[:, (relabel[loc[0]] >= 0) & (relabel[loc[1]] >= 0)]]
            mine = v.edges[:, (np.searchsorted(v.ptr, v.edges[0], side="right") - 1) == g] - v.ptr[g]
            dropped = mine.shape[1] < alive.shape[1]
            seq.append(("rand", torch.tensor([0.1 if dropped else 0.9])))
            if dropped:
                # positions of the kept edges among the survivors, in order (duplicates of an edge cannot occur: COO pairs are unique)
                keys = {tuple(x): i for i, x in enumerate(alive.T)}
                kp = np.asarray([keys[tuple(x)] for x in mine.T])
                seq.append(("perm", torch.from_numpy(np.concatenate([kp, np.setdiff1d(np.arange(alive.shape[1]), kp)]))))
            bits = int(v.rowmask[v.ptr[g]]) if (v.rowmask is not None and v.ptr[g + 1] > v.ptr[g]) else 0
            seq.append(("rand", torch.tensor([0.1 if bits else 0.9])))
            if bits:
                cols = np.asarray([c for c in range(F) if (bits >> c) & 1])
                seq.append(("perm", torch.from_numpy(np.concatenate([cols, np.setdiff1d(np.arange(F), cols)]))))
    return seq


@pytest.mark.parametrize("seed", [3, 14, 159])
def test_oracle_rebuilds_the_device_views_from_injected_permutations(monkeypatch, seed):
    for bi, b in enumerate(_batches()):
        views = _draw(b, 77 * seed + bi)
        inj = _Injected(_decisions(b, views))
        monkeypatch.setattr(OA.torch, "randperm", lambda n, generator=None, **kw: inj.take("perm", n))
        monkeypatch.setattr(OA.torch, "rand", lambda *a, generator=None, **kw: inj.take("rand"))
        o1, o2, m1, m2 = OA.create_two_views(to_oracle(b), None)
        monkeypatch.undo()
        assert inj.pos == len(inj.values)
        for ov, om, dv in ((o1, m1, views[0]), (o2, m2, views[1])):
            mine = view_to_oracle(b.host(), dv)
            assert torch.equal(ov.x, mine.x), "features (gathered rows, attribute-masked columns)"
            assert torch.equal(ov.edge_index, mine.edge_index) and torch.equal(ov.ptr, mine.ptr)
            common = torch.cat([torch.nonzero(m).squeeze(1) + int(ov.ptr[g]) for g, m in enumerate(om)]) if om else torch.zeros(0, dtype=torch.long)
            assert torch.equal(common, torch.from_numpy(np.asarray(dv.common, dtype=np.int64)))


def test_node_feature_masks_on_the_device():
    """pretrain_model.py:71-80: per graph with n >= 3, max(1, int(.15 n)) distinct nodes of that graph; here ascending per graph."""
    for bi, b in enumerate(_batches()):
        ptr = torch.tensor(b.ptr_host, dtype=torch.long, device=DEV)
        n = np.diff(np.asarray(b.ptr_host))
        k = np.where(n >= 3, np.maximum(1, (n * .15).astype(int)), 0)
        seen = set()
        for seed in range(6):
            idx = ops.aug_node_masks(ptr, b.ptr_host, 50 * bi + seed, 7).cpu().numpy()
            assert len(np.unique(idx)) == len(idx) == int(k.sum())
            g_of = np.searchsorted(np.asarray(b.ptr_host), idx, side="right") - 1
            assert np.array_equal(np.bincount(g_of, minlength=len(n)), k)
            assert (np.diff(idx) > 0).all()
            seen.add(tuple(idx))
        assert len(seen) > 1 or k.sum() == 0                       # the seed matters


def test_device_draws_have_the_reference_distributions():
    """Uniformity of the k-subsets and the two coins (p = .2), over many seeds on one graph: every node is kept with probability
    keep_n / n, every surviving edge with (E' - drop) / E' when the coin falls, the coins fall one time in five."""
    gen = torch.Generator().manual_seed(2)
    b = Batch.from_data_list([S.random_graph(gen, 21, 20.0, 40.0)])
    n, F, T = b.num_nodes, 21, 600
    kept = np.zeros(n)
    coin_e = coin_a = 0
    full_e = None
    for seed in range(T):
        v, _ = _draw(b, seed)
        kept[v.rows] += 1
        alive = OG.subgraph(torch.from_numpy(v.rows), b.edge_index, n).shape[1]
        coin_e += v.edges.shape[1] < alive
        coin_a += v.rowmask is not None
    keep_n = n - max(1, int(n * .2))
    assert np.abs(kept / T - keep_n / n).max() < 0.08                # sigma ~ 0.016 per node
    assert abs(coin_e / T - 0.2) < 0.06 and abs(coin_a / T - 0.2) < 0.06


def test_batched_launches_equal_the_per_job_calls():
    """gmp_aug_node_masks_batch / gmp_aug_two_views_batch (all the (task, domain) jobs of a step in one / two launches): bit-identical to
    the per-job calls, with more jobs than one launch carries (8), a job without graphs, and graphs of 1 and 2 nodes."""
    bs = _batches() + _batches()[:4]                                    # 11 jobs
    empty = Batch.empty(4)
    bs.insert(3, empty)
    dev = lambda b: (torch.tensor(b.ptr_host, dtype=torch.long, device=DEV), torch.tensor(b.edge_ptr_host, dtype=torch.long, device=DEV),
                     b.edge_index.to(DEV).contiguous())
    on_dev = [dev(b) for b in bs]
    masks = ops.aug_node_masks_batch([(d[0], b.ptr_host, 3 + 2 * i) for i, (b, d) in enumerate(zip(bs, on_dev))], 991)
    views = ops.aug_two_views_batch([(d[0], d[1], d[2], b.ptr_host, b.edge_ptr_host, int(b.x.size(1)), 40 + 2 * i)
                                     for i, (b, d) in enumerate(zip(bs, on_dev))], 991)
    for i, (b, d) in enumerate(zip(bs, on_dev)):
        assert torch.equal(masks[i], ops.aug_node_masks(d[0], b.ptr_host, 991, 3 + 2 * i))
        if b.num_graphs == 0:
            continue
        one = ops.aug_two_views(d[0], d[1], d[2], b.ptr_host, b.edge_ptr_host, int(b.x.size(1)), 991, 40 + 2 * i)
        got = views[i]
        tot = one.totals.cpu().tolist()
        assert got.totals.cpu().tolist() == tot and torch.equal(got.counts, one.counts)
        for v in range(2):
            assert torch.equal(got.rows[v], one.rows[v]) and torch.equal(got.rowmask[v], one.rowmask[v])
            assert torch.equal(got.edges[v][:, :tot[v]], one.edges[v][:, :tot[v]])
            assert torch.equal(got.common[v][:tot[2]], one.common[v][:tot[2]])
