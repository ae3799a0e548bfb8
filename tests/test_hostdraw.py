"""The native host module (csrc_host/hostdraw.cpp) against the Python implementation of the reference's draw order:
for an equal generator state the index arrays are bit-identical AND the generator is left in the same state (so every
later draw of the run -- sampler, evaluation -- is unchanged).  CPU only."""
import numpy as np
import pytest
import torch

from gnn_pretraining_amd import synthetic as S
from gnn_pretraining_amd.engine import StepEngine, hostdraw
from gnn_pretraining_amd.graph import Batch, Data
from gnn_pretraining_amd.models.pretrain_model import draw_mask_indices
from gnn_pretraining_amd.pretrain.tasks import sample_negative_edges

pytestmark = pytest.mark.skipif(hostdraw() is None, reason="gnn_pretraining_amd/_hostdraw.so not built (python -m gnn_pretraining_amd.csrc_host.build)")


def _gens(seed):
    a, b = torch.Generator(), torch.Generator()
    a.manual_seed(seed); b.manual_seed(seed)
    return a, b


def _batches():
    gen = torch.Generator().manual_seed(5)
    out = [S.domain_batch(gen, d, 8) for d in (7, 4, 37, 21)]
    out.append(S.domain_batch(gen, 2, 3))                                            # fewer than 3 features: no attribute mask possible
    out.append(Batch.from_data_list([Data(torch.zeros(1, 4), torch.zeros(2, 0, dtype=torch.long), torch.zeros(1, dtype=torch.long), torch.zeros(12)),
                                     Data(torch.zeros(2, 4), torch.tensor([[0, 1], [1, 0]]), torch.zeros(1, dtype=torch.long), torch.zeros(12)),
                                     S.random_graph(gen, 4, 126.0, 200.0)]))                 # 1-node, 2-node and a large graph
    return out


def _args(b):
    return torch.tensor(b.ptr_host), torch.tensor(b.edge_ptr_host), b.edge_index.contiguous()


@pytest.mark.parametrize("seed", [0, 1, 12345])
def test_native_draws_equal_the_python_reference_order(seed):
    H = hostdraw()
    for b in _batches():
        ptr, eptr, ei = _args(b)
        g1, g2 = _gens(seed)
        assert torch.equal(H.mask_indices(ptr, g1), draw_mask_indices(b.ptr_host, g2))
        assert torch.equal(g1.get_state(), g2.get_state())
        for _ in range(3):                                                       # several rounds: coins fall differently
            got = H.draw_views(ptr, eptr, ei, b.x.size(1), g1)
            want = StepEngine._draw_views_python(b, g2)
            for vi in range(2):
                rows, edges, vptr, rowmask, common = got[5 * vi:5 * vi + 5]
                w = want[vi]
                assert np.array_equal(rows.numpy(), w.rows) and np.array_equal(edges.numpy(), w.edges)
                assert np.array_equal(vptr.numpy(), w.ptr) and np.array_equal(common.numpy(), w.common)
                if w.rowmask is None:
                    assert rowmask.numel() == 0
                else:
                    assert np.array_equal(rowmask.numpy().view(np.uint64), w.rowmask)
            assert torch.equal(g1.get_state(), g2.get_state())


def test_engine_draw_uses_the_native_module_and_matches_python(monkeypatch):
    import sys
    sys.path.insert(0, __file__.rsplit("/", 1)[0])
    from test_host_logic import _Inp, _planner
    e = _planner("reference", "s4")
    gen = torch.Generator().manual_seed(3)
    inp = _Inp(S.pretrain_step_batches(gen, e.domains))
    g1, g2 = _gens(9)
    import random
    e.neg_rng = random.Random(4)
    art_native = e.draw(inp, g1)
    native_state = e.sync_neg_rng().getstate()
    monkeypatch.setattr("gnn_pretraining_amd.engine._HOSTDRAW", None)
    monkeypatch.setattr("gnn_pretraining_amd.engine._HOSTDRAW_TRIED", True)
    e.neg_rng, e._neg_native = random.Random(4), None
    art_python = e.draw(inp, g2)
    assert torch.equal(g1.get_state(), g2.get_state())
    assert e.neg_rng.getstate() == native_state          # the Python-random stream of the negatives too
    for t in art_python:
        for d in art_python[t]:
            a, b = art_native[t][d], art_python[t][d]
            if isinstance(b, tuple):
                for va, vb in zip(a, b):
                    assert all(np.array_equal(getattr(va, f), getattr(vb, f)) for f in ("rows", "edges", "ptr", "common"))
                    assert (va.rowmask is None) == (vb.rowmask is None) and (va.rowmask is None or np.array_equal(va.rowmask, vb.rowmask))
            else:
                assert np.array_equal(np.asarray(a), np.asarray(b)), (t, d)


def _sparse_big_batch(gen):
    """large graphs with few edges: the over-sample is far below the population, random.sample takes its rejection-set branch"""
    return Batch.from_data_list([S.random_graph(gen, 4, 100.0, 20.0), S.random_graph(gen, 4, 126.0, 30.0), S.random_graph(gen, 4, 60.0, 10.0)])


@pytest.mark.parametrize("seed", [0, 7, 2024])
def test_native_negative_sampler_replays_pythons_random(seed):
    """PyG's batched_negative_sampling draws from Python's random.sample: the native CPython-compatible MT19937 must return the
    negatives of pretrain/tasks.py sample_negative_edges for an equal random.Random state and leave the equal state behind --
    over small graphs (every non-edge, no draw), mid-sized ones (pool branch of random.sample) and sparse large ones (set branch)."""
    import random
    H = hostdraw()
    gen = torch.Generator().manual_seed(seed)
    r_py, nat = random.Random(seed), H.PyRandom()
    nat.setstate(torch.tensor(r_py.getstate()[1], dtype=torch.long))
    drew = False
    for b in _batches() + [_sparse_big_batch(gen), S.domain_batch(gen, 21, 32)]:
        before = r_py.getstate()
        want = sample_negative_edges(b, r_py)
        drew |= r_py.getstate() != before
        got = nat.negative_edges(*_args(b))
        assert torch.equal(got, want)
        assert tuple(nat.getstate().tolist()) == r_py.getstate()[1]
    assert drew                                           # at least one batch needed random draws


def test_merged_pairs_hold_the_ordered_list_as_a_multiset(monkeypatch):
    """engine.merge_mirrored_pairs: every ordered pair of the reference's list (positives, then negatives) stands in the merged list
    as its (min, max) pair with the right sign, multiplicities add up to the ordered counts, every ordered position is named by exactly one
    merged row (a row stands for one or two ordered rows: each keeps a dropout mask of its own, heads.py:44-52), and the native and numpy
    versions give the same arrays (a directed edge, a self loop, a duplicate and a triplicate included)."""
    import random
    from gnn_pretraining_amd.engine import merge_mirrored_pairs
    cases = [(b, sample_negative_edges(b, random.Random(2)).numpy()) for b in _batches() if b.num_graphs]
    odd = Batch.from_data_list([Data(torch.zeros(4, 4), torch.tensor([[0, 1, 1, 2, 2, 3], [1, 0, 2, 2, 3, 3]]), torch.zeros(1, dtype=torch.long), torch.zeros(12)),
                                Data(torch.zeros(3, 4), torch.tensor([[0, 0, 1, 2, 1], [1, 1, 0, 1, 2]]), torch.zeros(1, dtype=torch.long), torch.zeros(12))])
    cases.append((odd, np.array([[0, 3, 0, 4, 6], [3, 0, 2, 6, 4]])))              # a directed edge, a self loop, a duplicate, a triplicate
    for b, neg in cases:
        pos = b.edge_index
        pairs, w, od = merge_mirrored_pairs(b, neg, 100, 7)
        with monkeypatch.context() as m:
            m.setattr("gnn_pretraining_amd.engine._HOSTDRAW", None)
            m.setattr("gnn_pretraining_amd.engine._HOSTDRAW_TRIED", True)
            pairs_np, w_np, od_np = merge_mirrored_pairs(b, neg, 100, 7)
        assert np.array_equal(pairs, pairs_np) and np.array_equal(w, w_np) and np.array_equal(od, od_np)
        assert (pairs[0] <= pairs[1]).all() and pairs.dtype == np.int64 and w.dtype == np.float32 and od.dtype == np.int32 and od.shape == pairs.shape
        npos = int((w > 0).sum())
        assert (w[:npos] > 0).all() and (w[npos:] < 0).all()                         # positives first
        assert np.array_equal(np.abs(w), 1.0 + (od[1] >= 0))                         # |w| = how many ordered rows the merged row stands for
        ordered = np.concatenate([pos.numpy(), neg], axis=1)
        named = np.concatenate([od[0], od[1][od[1] >= 0]]) - 7
        assert sorted(named.tolist()) == list(range(ordered.shape[1]))             # every ordered row exactly once
        for r in range(pairs.shape[1]):                                             # ... and it is that row's pair
            for o in od[:, r]:
                if o >= 0:
                    i, j = ordered[:, o - 7]
                    assert (min(i, j) + 100, max(i, j) + 100) == (pairs[0, r], pairs[1, r])
        assert (np.diff(od[0]) > 0).all() and (od[1][od[1] >= 0] > od[0][od[1] >= 0]).all()      # first-occurrence order
        for sign, e, seg, ws in ((1, pos.numpy(), pairs[:, :npos], w[:npos]), (-1, neg, pairs[:, npos:], w[npos:])):
            keys = [(int(min(a, b)) + 100, int(max(a, b)) + 100) for a, b in e.T]
            want = {}
            for k in keys:
                want[k] = want.get(k, 0) + 1
            got = {}
            for a, b, x in zip(seg[0], seg[1], ws):
                got[(int(a), int(b))] = got.get((int(a), int(b)), 0) + int(sign * x)
            assert got == want
            assert list(got) == list(dict.fromkeys(keys))                              # first-occurrence order


def _plan_fields(p):
    names = ("N", "S", "E", "max_seg", "max_seg_edges", "num_tiles", "fwd_cuts", "seg_ptr", "seg_dom", "seg_task", "task_row", "sizes", "skipped",
             "nfm_rows", "nc_rows", "nc_n", "gc_rows", "gc_n", "gc_B", "gc_r0", "gc_M", "gp_rows", "gp_B", "gp_r0", "gp_M", "da_B", "da_r0", "da_M",
             "lp_K", "lp_S", "lp_rows_end", "lp_max_rows", "lp_max_edges")
    out = {}
    for n in names:
        if hasattr(p, n):
            v = getattr(p, n)
            out[n] = [int(x) for x in v] if isinstance(v, (list, np.ndarray)) and n not in ("skipped", "fwd_cuts") else ([tuple(int(y) for y in x) for x in v] if n == "fwd_cuts" else v)
    return out


@pytest.mark.parametrize("scheme", ["s4", "s5", "s1", "s2", "b3", "b2"])
@pytest.mark.parametrize("merge,split", [(True, 3), (False, 1), (True, 2)])
def test_native_layout_equals_the_python_layout(scheme, merge, split):
    """hostdraw.plan_step (one call, GIL released) against engine.StepEngine.plan's numpy code on the same draws: every scalar, every
    list, both upload images element for element and the offsets into them -- over the scheme families, with a domain that has no graph,
    one with a single graph (no graph-contrast pair: tasks.py:241) and one with 1- and 2-node graphs."""
    import sys
    sys.path.insert(0, __file__.rsplit("/", 1)[0])
    from test_host_logic import _Inp, _planner
    from gnn_pretraining_amd.engine import Artefacts
    e = _planner("reference", scheme)
    e.lp_merge, e.fwd_ranges = merge, split
    gen = torch.Generator().manual_seed(13)
    hosts = [S.pretrain_step_batches(gen, e.domains) for _ in range(3)]
    odd = dict(hosts[0])
    names = list(odd)
    odd[names[0]] = Batch.empty(int(odd[names[0]].x.size(1)))
    odd[names[1]] = Batch.from_data_list(hosts[1][names[1]].to_data_list()[:1])
    if len(names) > 2:
        F = int(odd[names[2]].x.size(1))
        mk = lambda n, ei: Data(torch.zeros(n, F), ei, torch.zeros(1, dtype=torch.long), torch.zeros(12))
        odd[names[2]] = Batch.from_data_list([mk(1, torch.zeros(2, 0, dtype=torch.long)), mk(2, torch.tensor([[0, 1], [1, 0]])),
                                              S.random_graph(gen, F, 20.0, 30.0)])
    hosts.append(odd)
    for k, host in enumerate(hosts):
        inp = _Inp(host)
        art = e.draw(inp, torch.Generator().manual_seed(100 + k))
        assert isinstance(art, Artefacts) and art.raw is not None
        e.native_plan = True
        pn = e.plan(inp, art)
        e.native_plan = False
        pp = e.plan(inp, art)
        assert _plan_fields(pn) == _plan_fields(pp)
        assert pn.lay32 == pp.lay32 and pn.lay64 == pp.lay64
        assert pn.cat32.dtype == pp.cat32.dtype and np.array_equal(pn.cat32, pp.cat32)
        assert pn.cat64.dtype == pp.cat64.dtype and np.array_equal(pn.cat64, pp.cat64)
        if "link_pred" in e.tasks:
            assert np.array_equal(pn.lp_labels, pp.lp_labels) and pn.lp_labels.dtype == np.float32
        for name, arr in pp.a32.items():
            assert np.array_equal(pn.a32[name], np.asarray(arr)) and pn.a32[name].shape == np.asarray(arr).shape, name
        for name, arr in pp.a64.items():
            assert np.array_equal(pn.a64[name], np.asarray(arr)) and pn.a64[name].shape == np.asarray(arr).shape, name
