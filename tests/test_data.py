"""Datasets and loaders: the flat store against Batch.from_data_list, the graph-property
calculator against the networkx restatement in oracle/graph_properties.py and closed forms, the processing logic and
the loaders' draw order against src/data/pretrain_data_loaders.py:28-46 semantics.  CPU only."""
import numpy as np
import pytest
import torch

from gnn_pretraining_amd import synthetic
from gnn_pretraining_amd import synthetic as S
from gnn_pretraining_amd.data import data_setup as DS
from gnn_pretraining_amd.data.finetune_data_loaders import create_finetune_data_loader
from gnn_pretraining_amd.data.graph_properties import GraphPropertyCalculator, standardize
from gnn_pretraining_amd.data.pretrain_data_loaders import (BalancedMultiDomainSampler, GraphDataset,
                                                            create_train_data_loader, create_val_data_loader)
from gnn_pretraining_amd.data.store import GraphStore, load_splits
from gnn_pretraining_amd.graph import Batch, Data
from oracle import graph_properties as OGP


def _gen(seed=0):
    g = torch.Generator()
    g.manual_seed(seed)
    return g


def _same_batch(a: Batch, b: Batch):
    for f in ("x", "edge_index", "batch", "ptr", "y", "graph_properties"):
        u, v = getattr(a, f), getattr(b, f)
        assert (u is None) == (v is None), f
        if u is not None:
            assert torch.equal(u, v), f
    assert a.ptr_host == b.ptr_host and a.edge_ptr_host == b.edge_ptr_host


@pytest.fixture(scope="module")
def processed(tmp_path_factory):
    root = tmp_path_factory.mktemp("processed")
    DS.process_synthetic(root, scale=0.05)
    return root


# ---- store ------------------------------------------------------------------------------------------------------------
def test_collate_equals_from_data_list_with_repeats(tmp_path):
    graphs = [synthetic.random_graph(_gen(1), 21) for _ in range(30)]
    store = GraphStore.from_data_list(graphs)
    idx = [3, 3, 29, 0, 17, 3, 11, 0]
    _same_batch(store.collate(idx), Batch.from_data_list([graphs[i] for i in idx]))
    store.save(tmp_path)
    back = GraphStore.load(tmp_path)
    _same_batch(back.collate(idx), store.collate(idx))
    assert len(back) == 30 and back.num_node_features == 21


def test_store_rejects_inconsistent_pointers():
    with pytest.raises(ValueError):
        GraphStore(torch.zeros(5, 3), torch.zeros(2, 4, dtype=torch.long), torch.tensor([0, 4]), torch.tensor([0, 4]))


def test_store_empty_edge_graphs_and_single_graph_node_labels():
    g0 = Data(torch.randn(3, 4), torch.zeros(2, 0, dtype=torch.long), torch.tensor([1]), torch.zeros(12))
    g1 = Data(torch.randn(2, 4), torch.tensor([[0, 1], [1, 0]]), torch.tensor([0]), torch.ones(12))
    store = GraphStore.from_data_list([g0, g1])
    _same_batch(store.collate([1, 0, 0]), Batch.from_data_list([g1, g0, g0]))
    planet = GraphStore.from_data_list([synthetic.cora_like(_gen(2), 60, 100, 16, num_classes=3)])
    assert planet.node_level_labels and planet.collate([0]).y.numel() == 60


# ---- graph properties -------------------------------------------------------------------------------------------------
def _ei(pairs):
    return torch.tensor(pairs, dtype=torch.long).t().contiguous()


@pytest.mark.parametrize("name,n,pairs,want", [
    ("path4", 4, [(0, 1), (1, 2), (2, 3)], [4, 3, 0.5, 1.5, 0.25, 2, 0, 0, 1, 3, -0.5, 1 / 3]),
    ("triangle", 3, [(0, 1), (1, 2), (0, 2)], [3, 3, 1, 2, 0, 2, 1, 1, 1, 1, 0, 0]),
    ("star5", 5, [(0, 1), (0, 2), (0, 3), (0, 4)], [5, 4, 0.4, 1.6, 1.44, 4, 0, 0, 1, 2, -1, 1]),
    ("two_triangles_isolated", 7, [(0, 1), (1, 2), (0, 2), (3, 4), (4, 5), (3, 5)],
     [7, 6, 6 / 21, 12 / 7, 24 / 49, 2, 6 / 7, 1, 3, 1, 0, 2 / 30]),
])
def test_graph_properties_closed_forms(name, n, pairs, want):
    g = Data(torch.zeros(n, 1), _ei(pairs + [(b, a) for a, b in pairs]))
    got = GraphPropertyCalculator()(g).numpy()
    np.testing.assert_allclose(got, np.array(want, dtype=np.float32), rtol=1e-6, atol=1e-6, err_msg=name)
    np.testing.assert_allclose(OGP.graph_properties(g.edge_index.numpy(), n), got, rtol=1e-6, atol=1e-6)


def test_graph_properties_match_networkx_restatement_on_messy_graphs():
    rng = np.random.default_rng(3)
    calc = GraphPropertyCalculator()
    for trial in range(60):
        n = int(rng.integers(1, 40))
        m = int(rng.integers(0, 3 * n))
        ei = rng.integers(0, n, size=(2, m))                       # self loops, duplicates, one direction only, isolated nodes
        got = calc(Data(torch.zeros(n, 1), torch.from_numpy(ei))).numpy()
        np.testing.assert_allclose(got, OGP.graph_properties(ei, n), rtol=1e-5, atol=1e-6, err_msg=f"trial {trial}")


def test_graph_properties_largest_component_tie_takes_first():
    # two components of 4 nodes: a path (diameter 3) holding node 0, and a star (diameter 2)
    pairs = [(0, 1), (1, 2), (2, 3), (4, 5), (4, 6), (4, 7)]
    ei = _ei(pairs)
    assert GraphPropertyCalculator()(Data(torch.zeros(8, 1), ei))[9] == 3.0 == OGP.graph_properties(ei.numpy(), 8)[9]
    swapped = _ei([(0, 1), (0, 2), (0, 3), (4, 5), (5, 6), (6, 7)])
    assert GraphPropertyCalculator()(Data(torch.zeros(8, 1), swapped))[9] == 2.0 == OGP.graph_properties(swapped.numpy(), 8)[9]


def test_standardize_matches_sklearn_scaler():
    rng = np.random.default_rng(0)
    v = rng.normal(size=(50, 12)).astype(np.float32) * 7 + 3
    v[:, 4] = 2.0                                                   # constant column: scale 0 -> 1
    train = rng.permutation(50)[:35]
    np.testing.assert_allclose(standardize(v.astype(np.float64), train), OGP.standardize(v, train), rtol=1e-5, atol=1e-5)


# ---- processing -------------------------------------------------------------------------------------------------------
def test_processed_layout_splits_and_scaling(processed):
    for name in DS.TUDATASETS:
        store, splits = GraphStore.load(processed / name), load_splits(processed / name)
        allidx = torch.cat(list(splits.values()))
        assert allidx.unique().numel() == allidx.numel() == len(store), name            # a partition
        assert set(splits) == ({"train", "val", "test"} if name in DS.DOWNSTREAM_TUDATASETS else {"train", "val"})
        assert (store.graph_properties is not None) == (name in DS.PRETRAIN_TUDATASETS)
        if store.graph_properties is not None:
            tr = store.graph_properties[splits["train"]]
            assert tr.mean(0).abs().max() < 1e-4
    enz, spl = GraphStore.load(processed / "ENZYMES"), load_splits(processed / "ENZYMES")
    assert float(enz.x.min()) >= -3.0 and float(enz.x.max()) <= 3.0
    rows = torch.cat([torch.arange(enz.node_ptr[i], enz.node_ptr[i + 1]) for i in spl["train"]])
    # clip happens after scaling, so the train mean is only approximately 0
    assert enz.x[rows].mean(0).abs().max() < 0.05
    # stratified: every class appears in val and test
    assert enz.y[spl["val"]].unique().numel() == enz.y[spl["test"]].unique().numel() == 6
    assert len(spl["train"]) == round(0.8 * len(enz))


def test_pretrain_only_split_is_sklearn_shufflesplit_seed_42():
    from sklearn.model_selection import ShuffleSplit
    tr, va = next(ShuffleSplit(n_splits=1, test_size=0.1, random_state=42).split(np.arange(188)))
    s = DS.pretrain_only_splits(188)
    assert s["train"].tolist() == tr.tolist() and s["val"].tolist() == va.tolist()


def test_link_prediction_splits(processed):
    store, s = GraphStore.load(processed / "Cora_LP"), load_splits(processed / "Cora_LP")
    g = store.graph(0)
    n, E = g.num_nodes, g.num_edges
    nvt = int(E * 0.2)
    nv = int(nvt * 0.5)
    assert s["train_pos"].size(1) == E - nvt and s["val_pos"].size(1) == nv and s["test_pos"].size(1) == nvt - nv
    assert s["val_neg"].size(1) == nv and s["test_neg"].size(1) == nvt - nv
    key = lambda e: (e[0] * n + e[1])
    pos = torch.cat([key(s["train_pos"]), key(s["val_pos"]), key(s["test_pos"])])
    assert torch.equal(pos.sort().values, key(g.edge_index).sort().values)             # a partition of the edges
    und = set(key(DS.to_undirected_host(s["train_pos"], n)).tolist())
    neg = torch.cat([s["val_neg"], s["test_neg"]], dim=1)
    assert not (set(key(neg).tolist()) & und) and bool((neg[0] != neg[1]).all())
    assert key(neg).unique().numel() == neg.size(1)
    again = DS.create_link_prediction_splits(g)
    assert all(torch.equal(again[k], s[k]) for k in s)                                 # seeded (42): reproducible


# ---- loaders ----------------------------------------------------------------------------------------------------------
def test_balanced_sampler_replays_the_reference_draw_order(processed):
    domains = DS.PRETRAIN_TUDATASETS
    loader = create_train_data_loader(domains, _gen(7), processed)
    sizes = {d: len(loader.domain_datasets[d]) for d in domains}
    assert loader.samples_per_domain == 8 and len(loader) == max(sizes.values()) // 8
    ref_gen = _gen(7)
    for step, batches in enumerate(loader):
        assert list(batches) == domains
        for d in domains:                                   # pretrain_data_loaders.py:38-41, restated
            ds = loader.domain_datasets[d]
            idx = torch.randint(0, len(ds), (8,), generator=ref_gen)
            _same_batch(batches[d], Batch.from_data_list([ds[i.item()] for i in idx]))
            assert batches[d].graph_properties.numel() == 8 * 12
        if step == 3:
            break


def test_single_domain_sampler_takes_32_graphs(processed):
    loader = create_train_data_loader(["ENZYMES"], _gen(0), processed)
    assert loader.samples_per_domain == 32 and next(iter(loader))["ENZYMES"].num_graphs == 32


def test_val_loader_is_sequential_batches_of_32(processed):
    loader = create_val_data_loader("NCI1", _gen(0), processed)
    val = load_splits(processed / "NCI1")["val"]
    batches = list(loader)
    assert len(batches) == len(loader) == -(-len(val) // 32)
    assert sum(b.num_graphs for b in batches) == len(val)
    store = GraphStore.load(processed / "NCI1")
    _same_batch(batches[0], store.collate(val[:32]))


def test_val_loader_draws_the_dataloader_base_seed_like_torch(processed):
    """The reference's validation loaders are torch DataLoaders on the SHARED generator (pretrain_data_loaders.py:65): every
    iteration start draws a base seed from it.  Generator state after two passes over our loader == after two passes over a
    real torch DataLoader of the same length on an equally seeded generator (and batches() makes no draw)."""
    from torch.utils.data import DataLoader
    ga, gb = _gen(9), _gen(9)
    ours = create_val_data_loader("NCI1", ga, processed)
    theirs = DataLoader(list(range(len(ours.dataset))), batch_size=32, generator=gb)
    for _ in range(2):
        assert len(list(ours)) == len(list(theirs))
    assert torch.equal(ga.get_state(), gb.get_state())
    assert not torch.equal(ga.get_state(), _gen(9).get_state())
    before = ga.get_state()
    assert len(ours.batches()) == len(ours)
    assert torch.equal(ga.get_state(), before)
    ours.draw_base_seed()
    next(iter(theirs))
    assert torch.equal(ga.get_state(), gb.get_state())


def test_finetune_loaders(processed):
    g = _gen(0)
    gl = create_finetune_data_loader("ENZYMES", "train", 32, g, processed)
    b = next(iter(gl))
    assert b.num_graphs == 32 and b.y.numel() == 32 and b.graph_properties is None
    nl = create_finetune_data_loader("Cora_NC", "train", -1, g, processed)
    data, idx, y = next(iter(nl))
    assert len(nl) == 1 and idx.numel() == len(nl.dataset) and torch.equal(y, data.y[idx])
    tl = create_finetune_data_loader("Cora_LP", "train", 256, g, processed)
    data, edges, labels = next(iter(tl))
    assert edges.shape == (2, min(256, len(tl.dataset))) and bool((labels == 1).all())
    assert torch.equal(tl.dataset.train_edges, load_splits(processed / "Cora_LP")["train_pos"])
    vl = create_finetune_data_loader("Cora_LP", "val", 256, g, processed)
    lab = torch.cat([l for _, _, l in vl])
    assert lab.numel() == len(vl.dataset) and lab.sum() * 2 == lab.numel()               # positives then negatives
    assert lab[: lab.numel() // 2].min() == 1 and lab[lab.numel() // 2:].max() == 0


# ---- the bridge from the reference's on-disk contract (data_setup.py:66-72) ------------------------------------------------
class _PygLikeData:
    """What the reference pickles: an object with x, edge_index (graph-local ids), y and num_nodes -- the attribute names of
    torch_geometric.data.Data, none of this repository's classes."""

    def __init__(self, x, edge_index, y):
        self.x, self.edge_index, self.y = x, edge_index, y

    @property
    def num_nodes(self):
        return self.x.size(0)


def test_export_bridge_tudataset_shaped(tmp_path):
    from gnn_pretraining_amd.data.export import export_dataset
    gen = _gen(21)
    graphs = [S.random_graph(gen, 7, 18.0, 20.0, num_classes=2) for _ in range(48)]
    pyg_like = [_PygLikeData(g.x.double(), g.edge_index.clone(), g.y) for g in graphs]          # (float64 x: cast on export)
    splits = {"train": torch.arange(0, 38), "val": torch.arange(38, 48)}
    props = torch.randn(48, 12, generator=gen)
    export_dataset(pyg_like, splits, props, tmp_path / "MUTAG")
    store = GraphStore.load(tmp_path / "MUTAG")
    assert len(store) == 48 and store.x.dtype == torch.float32 and torch.equal(store.graph_properties, props)
    pick = [3, 17, 3, 39]
    ref = [Data(graphs[i].x, graphs[i].edge_index, graphs[i].y, props[i]) for i in pick]
    _same_batch(store.collate(pick), Batch.from_data_list(ref))
    assert torch.equal(load_splits(tmp_path / "MUTAG")["val"], splits["val"])
    # ... and the loaders run on the exported tree, drawing as the reference's sampler does (pretrain_data_loaders.py:35-43)
    loader = create_train_data_loader(["MUTAG"], _gen(0), tmp_path)
    want = torch.randint(0, 38, (32,), generator=_gen(0))
    _same_batch(next(iter(loader))["MUTAG"], store.collate(splits["train"][want]))
    assert sum(b.num_graphs for b in create_val_data_loader("MUTAG", _gen(0), tmp_path)) == 10


def test_export_bridge_planetoid_shaped(tmp_path):
    """Cora_NC / Cora_LP: ONE Data with node labels; splits are node-id vectors, or [2, k] edge lists (data_setup.py:116-160)."""
    from gnn_pretraining_amd.data.export import export_dataset
    gen = _gen(22)
    n, f = 120, 33
    ei = torch.randint(0, n, (2, 400), generator=gen)
    x = torch.rand(n, f, generator=gen)
    g = _PygLikeData(x, ei, torch.randint(0, 7, (n,), generator=gen))
    nc = {"train": torch.arange(0, 20), "val": torch.arange(20, 50), "test": torch.arange(50, 120)}
    export_dataset([g], nc, None, tmp_path / "Cora_NC")
    st = GraphStore.load(tmp_path / "Cora_NC")
    assert len(st) == 1 and st.node_level_labels and torch.equal(st.graph(0).y, g.y) and torch.equal(st.graph(0).edge_index, ei)
    lp = {"train_pos": ei[:, :300], "val_pos": ei[:, 300:350], "val_neg": ei[:, 350:400].flip(0), "test_pos": ei[:, 350:], "test_neg": ei[:, :50].flip(0)}
    export_dataset([g], lp, None, tmp_path / "Cora_LP")
    got = load_splits(tmp_path / "Cora_LP")
    assert set(got) == set(lp) and all(torch.equal(got[k], lp[k]) for k in lp)


def test_export_bridge_refuses_global_edge_ids(tmp_path):
    from gnn_pretraining_amd.data.export import export_dataset
    g = _PygLikeData(torch.zeros(3, 2), torch.tensor([[0, 5], [1, 2]]), torch.zeros(1, dtype=torch.long))
    with pytest.raises(ValueError, match="local"):
        export_dataset([g], {"train": torch.arange(1)}, None, tmp_path / "X")


def test_export_tree_walks_the_reference_layout(tmp_path):
    """data/processed/{D}/{data,splits,graph_properties}.pt -> {D}/*.safetensors, with the .pt files written by torch.save the way
    data_setup.save_processed_data does (here: our own picklable stand-ins; the object pickle needs the explicit opt-in, the two tensor files go through the weights-only loader)."""
    from gnn_pretraining_amd.data.export import export_processed_tree
    gen = _gen(23)
    src = tmp_path / "processed"
    for name, dim in (("ENZYMES", 21), ("NCI1", 37)):
        graphs = [S.random_graph(gen, dim) for _ in range(12)]
        (src / name).mkdir(parents=True)
        torch.save(graphs, src / name / "data.pt")
        torch.save({"train": torch.arange(0, 9), "val": torch.arange(9, 12)}, src / name / "splits.pt")
        torch.save(torch.randn(12, 12, generator=gen), src / name / "graph_properties.pt")
    with pytest.raises(ValueError, match="allow_pickle"):            # unpickling objects is never the default
        export_processed_tree(src, tmp_path / "export")
    out = export_processed_tree(src, tmp_path / "export", allow_pickle=True)
    assert sorted(out) == ["ENZYMES", "NCI1"]
    for name in out:
        st = GraphStore.load(tmp_path / "export" / name)
        assert len(st) == 12 and st.graph_properties.shape == (12, 12)
