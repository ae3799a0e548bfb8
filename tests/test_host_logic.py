"""CPU tests of the host-side logic of the product path (no kernels): augmentation / masking index
parity with the oracle for an equal generator state (bit-exact), Batch collation."""
import pytest
import torch

from gnn_pretraining_amd import synthetic as S
from gnn_pretraining_amd.graph import Batch
from gnn_pretraining_amd.pretrain.augmentations import GraphAugmentor
from gnn_pretraining_amd.models.pretrain_model import draw_mask_indices
from oracle import augment as OA
from oracle import graph_ops as OG
from oracle.models import PretrainableGNN as OraclePG


def to_oracle(b: Batch) -> OG.Batch:
    return OG.Batch(b.x, b.edge_index, b.batch, b.ptr, torch.tensor(b.edge_ptr_host), b.y, b.graph_properties)


def test_batch_collation_matches_oracle():
    gen = torch.Generator().manual_seed(1)
    graphs = [S.random_graph(gen, 21) for _ in range(6)]
    b = Batch.from_data_list(graphs)
    o = OG.Batch.from_data_list([OG.Data(g.x, g.edge_index, g.y, g.graph_properties) for g in graphs])
    assert torch.equal(b.x, o.x) and torch.equal(b.edge_index, o.edge_index) and torch.equal(b.batch, o.batch)
    assert torch.equal(b.ptr, o.ptr) and torch.equal(b.graph_properties, o.graph_properties)
    back = b.to_data_list()
    for g, h in zip(graphs, back):
        assert torch.equal(g.x, h.x) and torch.equal(g.edge_index, h.edge_index)


def test_two_views_bit_exact_vs_oracle():
    for seed in range(6):
        gen = torch.Generator().manual_seed(100 + seed)
        b = S.domain_batch(gen, 21 if seed % 2 else 4, 8, mean_nodes=33 if seed < 4 else 4, mean_edges=62 if seed < 4 else 3)
        g1, g2 = torch.Generator().manual_seed(seed), torch.Generator().manual_seed(seed)
        v1, v2, m1, m2 = GraphAugmentor.create_two_views(b, g1)
        o1, o2, n1, n2 = OA.create_two_views(to_oracle(b), g2)
        for a, o in ((v1, o1), (v2, o2)):
            assert torch.equal(a.x, o.x) and torch.equal(a.edge_index, o.edge_index)
            assert torch.equal(a.batch, o.batch) and torch.equal(a.ptr, o.ptr)
        assert all(torch.equal(x, y) for x, y in zip(m1, n1)) and all(torch.equal(x, y) for x, y in zip(m2, n2))
        assert torch.equal(torch.rand(3, generator=g1), torch.rand(3, generator=g2))      # same generator state after


def test_mask_indices_bit_exact_vs_oracle():
    gen = torch.Generator().manual_seed(5)
    b = S.domain_batch(gen, 21, 8)
    g1, g2 = torch.Generator().manual_seed(9), torch.Generator().manual_seed(9)
    assert torch.equal(draw_mask_indices(b.ptr_host, g1), OraclePG.draw_mask_indices(b.ptr, g2))


# ---------------------------------------------------------------- engine host planning (no GPU involved)
import numpy as np                                                        # noqa: E402
from gnn_pretraining_amd.engine import StepEngine                          # noqa: E402
from gnn_pretraining_amd.pretrain import pretrain as PT                    # noqa: E402


class _Inp:
    def __init__(self, host):
        self.host, self.domains, self.row_off, r = host, list(host), {}, 0
        for d in host:
            self.row_off[d] = r
            r += host[d].num_nodes


def _planner(mode, scheme="s4"):
    e = StepEngine.__new__(StepEngine)                # host-side methods only: no device, no library
    e.tasks, e.domains = PT.ACTIVE_TASKS[scheme], PT.PRETRAIN_DOMAINS[scheme]
    e.max_rows, e.max_edges, e.S_MAX, e.KMAX, e.rng_mode, e._nprng = 16384, 131072, 64, 131072, mode, None
    import random
    e.neg_rng, e._neg_native, e.lp_merge, e.fwd_ranges, e.native_plan = random.Random(99), None, True, 2, True
    return e


def test_reference_mode_views_equal_oracle_views():
    """Engine index artefacts == what the oracle's create_two_views builds, for an equal generator state."""
    gen = torch.Generator().manual_seed(3)
    host = S.pretrain_step_batches(gen, PT.PRETRAIN_DOMAINS["s4"])
    for d, b in host.items():
        g1, g2 = torch.Generator().manual_seed(17), torch.Generator().manual_seed(17)
        v1, v2 = StepEngine._draw_views(b, g1)
        o1, o2, m1, m2 = OA.create_two_views(to_oracle(b), g2)
        for v, o, m in ((v1, o1, m1), (v2, o2, m2)):
            x = b.x[torch.from_numpy(v.rows)].clone()
            if v.rowmask is not None:
                F = x.size(1)
                bits = ((v.rowmask[:, None] >> np.arange(F, dtype=np.uint64)[None, :]) & np.uint64(1)).astype(bool)
                x[torch.from_numpy(bits)] = 0.0
            assert torch.equal(x, o.x)
            assert torch.equal(torch.from_numpy(v.edges), o.edge_index)
            assert torch.equal(torch.from_numpy(v.ptr), o.ptr)
            assert torch.equal(torch.from_numpy(v.common), torch.cat(m).nonzero().squeeze(1))


def test_vectorized_draws_are_valid_artefacts():
    """The vectorized sampler draws from the same distributions: every structural rule of the reference holds."""
    gen = torch.Generator().manual_seed(5)
    host = S.pretrain_step_batches(gen, PT.PRETRAIN_DOMAINS["s4"])
    e, inp = _planner("vectorized"), _Inp(host)
    saw_edge_drop = saw_attr_mask = False
    for _ in range(12):
        art = e.draw(inp, gen)
        for d, b in host.items():
            n = np.diff(np.asarray(b.ptr_host)); ei = b.edge_index.numpy()
            # NFM: max(1, int(.15 n)) distinct nodes per graph with n >= 3
            idx = art["node_feat_mask"][d]
            per = np.bincount(np.searchsorted(np.asarray(b.ptr_host), idx, side="right") - 1, minlength=len(n))
            assert len(np.unique(idx)) == len(idx)
            assert np.array_equal(per, np.where(n >= 3, np.maximum(1, (n * .15).astype(int)), 0))
            # LP (PyG batched_negative_sampling, num_neg_samples = E of the batch applied per graph): min(E, own non-edges) negatives
            # per graph, all non-adjacent ordered pairs (i != j) inside the graph
            neg = art["link_pred"][d]
            g_of = lambda v: np.searchsorted(np.asarray(b.ptr_host), v, side="right") - 1
            assert np.array_equal(g_of(neg[0]), g_of(neg[1])) and (neg[0] != neg[1]).all()
            eg = np.diff(np.asarray(b.edge_ptr_host))
            assert np.array_equal(np.bincount(g_of(neg[0]), minlength=len(n)), np.minimum(ei.shape[1], n * (n - 1) - eg))
            und = set(map(tuple, ei.T)) | set(map(tuple, ei[::-1].T))
            assert not (set(map(tuple, neg.T)) & und) and len(set(map(tuple, neg.T))) == neg.shape[1]
            for t in ("node_contrast", "graph_contrast"):
                v1, v2 = art[t][d]
                for v in (v1, v2):
                    kept = np.diff(v.ptr)
                    assert np.array_equal(kept, np.where(n >= 3, n - np.maximum(1, (n * .2).astype(int)), n))
                    assert (np.diff(v.rows) > 0).all()                       # kept nodes stay sorted
                    want = OG.subgraph(torch.from_numpy(v.rows), b.edge_index, b.num_nodes).numpy()
                    if v.edges.shape[1] == want.shape[1]:
                        assert np.array_equal(v.edges, want)                  # == PyG subgraph(relabel_nodes=True)
                    else:                                                     # some graph drew an edge drop
                        saw_edge_drop = True
                        assert v.edges.shape[1] < want.shape[1]
                        assert set(map(tuple, v.edges.T)) <= set(map(tuple, want.T))
                    if v.rowmask is not None:
                        saw_attr_mask = True
                        m = max(1, int(b.x.size(1) * .2))
                        pop = np.array([bin(int(w)).count("1") for w in v.rowmask])
                        assert set(pop.tolist()) <= {0, m}
                both = np.intersect1d(v1.rows, v2.rows)
                assert np.array_equal(v1.rows[v1.common], both) and np.array_equal(v2.rows[v2.common], both)
    assert saw_edge_drop and saw_attr_mask


def test_plan_layout_is_task_major_and_consistent():
    gen = torch.Generator().manual_seed(6)
    host = S.pretrain_step_batches(gen, PT.PRETRAIN_DOMAINS["s4"])
    for mode in ("reference", "vectorized"):
        e, inp = _planner(mode), _Inp(host)
        p = e.plan(inp, e.draw(inp, gen))
        assert p.S == 28 and p.seg_task == sorted(p.seg_task) and p.task_row[-1] == p.N
        assert p.a64["edge_index"].min() >= 0 and p.a64["edge_index"].max() < p.N
        seg_of_row = np.searchsorted(np.asarray(p.seg_ptr), np.arange(p.N), side="right") - 1
        ei = p.a64["edge_index"]
        assert np.array_equal(seg_of_row[ei[0]], seg_of_row[ei[1]])           # block diagonal: no edge crosses a segment
        assert p.a32["tiles"].shape == (p.num_tiles, 2)
        assert (p.a64["nfm_idx"] < p.task_row[1]).all() and (p.a64["nc_idx"] >= p.task_row[2]).all()


@pytest.mark.parametrize("mode", ["reference", "vectorized"])
@pytest.mark.parametrize("scheme", ["s4", "s5"])
def test_plan_accepts_steps_with_absent_domains(mode, scheme):
    """Validation passes hand the engine one domain at a time: the others are zero-graph batches.  Their segments exist
    (the layout stays task-major x domain) but are empty, and every row / edge of the plan belongs to a present domain."""
    from gnn_pretraining_amd.constants import DOMAIN_DIMENSIONS
    from gnn_pretraining_amd.graph import Batch
    e = _planner(mode, scheme)
    for present in (["ENZYMES"], ["MUTAG", "NCI1"], []):
        gen = torch.Generator().manual_seed(3)
        real = S.pretrain_step_batches(gen, present, graphs_per_domain=5) if present else {}
        host = {d: (real[d] if d in real else Batch.empty(DOMAIN_DIMENSIONS[d])) for d in e.domains}
        inp = _Inp(host)
        p = e.plan(inp, e.draw(inp, gen))
        seg_ptr, seg_dom = p.a32["seg_ptr"], p.a32["seg_dom"]
        lens = np.diff(seg_ptr)
        absent = [i for i, d in enumerate(e.domains) if d not in present]
        assert p.S == len(seg_dom) and int(seg_ptr[-1]) == p.N
        assert all(lens[seg_dom == i].sum() == 0 for i in absent)
        if not present:
            assert p.N == 0 and p.E == 0
        else:
            assert p.N > 0 and lens[np.isin(seg_dom, [e.domains.index(d) for d in present])].sum() == p.N
