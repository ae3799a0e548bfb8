"""Module- and step-level parity of the HIP path against the CPU oracle (iii-iv):
same state_dict, same inputs, same RNG artefacts; train-mode BatchNorm, dropout off.

fp32 tolerance 1e-4 relative for outputs/losses (north_star); gradient tolerance: parity_util.assert_grad_close."""
import copy

import numpy as np

import pytest
import torch

pytestmark = pytest.mark.gpu

from gnn_pretraining_amd import synthetic as S                                    # noqa: E402
from gnn_pretraining_amd.models import FinetuneGNN, GINBackbone, GINLayer, InputEncoder, PretrainableGNN   # noqa: E402
from gnn_pretraining_amd.pretrain import pretrain as PT                           # noqa: E402
from gnn_pretraining_amd.pretrain.tasks import TwoViews                           # noqa: E402
from oracle import models as OM, tasks as OTk, train as OTr                       # noqa: E402
from parity_util import assert_close, assert_grad_close, assert_grad_tight, copy_state, set_dropout, to_oracle          # noqa: E402

DEV = torch.device("cuda:0")
OUT_RTOL = 1e-4


def perturb_bn(model, gen):
    """non-trivial affine/running statistics so BN parity is not tested at its identity initialisation"""
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.weight.data = torch.rand(m.weight.shape, generator=gen) + 0.5
            m.bias.data = torch.randn(m.bias.shape, generator=gen) * 0.1
            m.running_mean.data = torch.randn(m.running_mean.shape, generator=gen) * 0.1
            m.running_var.data = torch.rand(m.running_var.shape, generator=gen) + 0.5


@pytest.mark.parametrize("training,seed", [(True, 3), (False, 4), (True, 5), (False, 6)])
def test_backbone_forward_backward(training, seed):
    gen = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    ob = OM.GINBackbone()
    perturb_bn(ob, gen)
    for l in ob.layers:
        l.gin_conv.eps.data.fill_(0.1)
    hb = GINBackbone()
    copy_state(hb, ob)
    hb.to(DEV)
    set_dropout(ob, 0.0); set_dropout(hb, 0.0)
    ob.train(training); hb.train(training)
    b = S.domain_batch(gen, 21, 8)
    h0 = torch.randn(b.num_nodes, 256, generator=gen)
    g = torch.randn(b.num_nodes, 256, generator=gen)
    x1 = h0.clone().requires_grad_()
    y1 = ob(x1, b.edge_index); y1.backward(g)
    x2 = h0.to(DEV).requires_grad_()
    y2 = hb(x2, b.edge_index.to(DEV)); y2.backward(g.to(DEV))
    assert_close(y2, y1, OUT_RTOL, "backbone output")
    assert_grad_close(x2.grad, x1.grad, x1.grad.abs().max().item(), "grad h0")
    op, hp = dict(ob.named_parameters()), dict(hb.named_parameters())
    gmax = max(p.grad.abs().max().item() for p in op.values())
    for n in op:
        assert_grad_close(hp[n].grad, op[n].grad, gmax, f"grad {n}")
    if training:
        for (n, a), (_, o) in zip(hb.named_buffers(), ob.named_buffers()):
            assert_close(a.float(), o.float(), OUT_RTOL, f"buffer {n}")


def test_input_encoder_and_single_layer():
    gen = torch.Generator().manual_seed(4)
    torch.manual_seed(4)
    for dim in (7, 4, 37, 21):
        oe = OM.InputEncoder(dim); he = InputEncoder(dim)
        copy_state(he, oe); he.to(DEV)
        set_dropout(oe, 0.0); set_dropout(he, 0.0)
        x = torch.randn(300, dim, generator=gen).clamp_(-3, 3)
        assert_close(he(x.to(DEV)), oe(x), OUT_RTOL, f"encoder dim {dim}")
    ol = OM.GINLayer(); hl = GINLayer()
    copy_state(hl, ol); hl.to(DEV)
    set_dropout(ol, 0.0); set_dropout(hl, 0.0)
    b = S.domain_batch(gen, 21, 8)
    h = torch.randn(b.num_nodes, 256, generator=gen)
    assert_close(hl(h.to(DEV), b.edge_index.to(DEV)), ol(h, b.edge_index), OUT_RTOL, "GINLayer")


def test_stacked_segments_equal_separate_calls():
    """seg_ptr semantics: one stacked launch == independent forward() calls (per-segment BN statistics)."""
    gen = torch.Generator().manual_seed(8)
    torch.manual_seed(8)
    hb = GINBackbone().to(DEV)
    set_dropout(hb, 0.0)
    hb.train()
    parts = [S.domain_batch(gen, 21, 8) for _ in range(3)]
    hs = [torch.randn(p.num_nodes, 256, generator=gen).to(DEV) for p in parts]
    state0 = copy.deepcopy(hb.state_dict())
    sep = torch.cat([hb(h, p.edge_index.to(DEV)) for h, p in zip(hs, parts)])
    stats_sep = copy.deepcopy(hb.state_dict())
    hb.load_state_dict(state0)
    offs = [0]
    for p in parts:
        offs.append(offs[-1] + p.num_nodes)
    ei = torch.cat([p.edge_index + o for p, o in zip(parts, offs)], dim=1).to(DEV)
    seg = torch.tensor(offs, dtype=torch.int32, device=DEV)
    stacked = hb(torch.cat(hs), ei, seg_ptr=seg, max_seg_rows=max(p.num_nodes for p in parts))
    assert_close(stacked, sep, 1e-6, "stacked vs separate")
    for k, v in hb.state_dict().items():
        assert_close(v.float(), stats_sep[k].float(), 1e-6, f"running stat {k}")


def _models(tasks, domains, seed):
    torch.manual_seed(seed)
    gen = torch.Generator().manual_seed(seed)
    om = OM.PretrainableGNN(torch.device("cpu"), domains, tasks)
    perturb_bn(om, gen)
    hm = PretrainableGNN(torch.device("cpu"), domains, tasks)
    copy_state(hm, om)
    hm.device = DEV
    hm.to(DEV)
    set_dropout(om, 0.0); set_dropout(hm, 0.0)
    om.train(); hm.train()
    return om, hm, gen


def _views_to_oracle(v):
    def mask(n, idx):
        m = torch.zeros(n, dtype=torch.bool); m[idx] = True
        return m
    return OTk.TwoViews(to_oracle(v.v1), to_oracle(v.v2), mask(v.v1.num_nodes, v.common1), mask(v.v2.num_nodes, v.common2))


def _artefacts(state, batches, gen):
    """Draw with the PRODUCT's host code, convert for the oracle: both sides see identical artefacts."""
    art_h, art_o = {}, {}
    for name, task in state.tasks.items():
        a = task.draw(batches, gen)
        art_h[name] = a
        if name in ("node_contrast", "graph_contrast"):
            art_o[name] = {d: (None if v is None else _views_to_oracle(v)) for d, v in a.items()}
        else:
            art_o[name] = a
    return art_h, art_o


@pytest.mark.parametrize("scheme,seed", [("s4", 12), ("b2", 13), ("s5", 14), ("b4", 15)])
def test_task_losses_and_gradients(scheme, seed):
    tasks, domains = PT.ACTIVE_TASKS[scheme], PT.PRETRAIN_DOMAINS[scheme]
    om, hm, gen = _models(tasks, domains, seed)
    cfg = PT.PretrainConfig(scheme, 0)
    state = PT.StepState(hm, cfg, steps_per_epoch=10, epochs=2)
    state.grl.current_step = 15          # non-zero GRL lambda for s5
    host = S.pretrain_step_batches(gen, domains)
    dev_batches = {d: b.to(DEV) for d, b in host.items()}
    o_batches = {d: to_oracle(b) for d, b in host.items()}
    art_h, art_o = _artefacts(state, dev_batches, gen)
    otemp, ogrl = OTr.TemperatureScheduler(20), OTr.GRLScheduler(2, 10)
    ogrl.current_step = 15
    otasks = OTk.instantiate_tasks(om, tasks, ogrl, otemp)
    for name in tasks:
        om.zero_grad(set_to_none=True); hm.zero_grad(set_to_none=True)
        lo, po = otasks[name].loss(o_batches, art_o[name])
        lh, ph = state.tasks[name].loss(dev_batches, art_h[name])
        assert_close(lh, lo, OUT_RTOL, f"{name} loss")
        for d in po:
            assert_close(ph[d], po[d], OUT_RTOL, f"{name}/{d} loss")
        lo.backward(); lh.backward()
        og, hg = dict(om.named_parameters()), dict(hm.named_parameters())
        gmax = max(p.grad.abs().max().item() for p in og.values() if p.grad is not None)
        for n, p in og.items():
            if p.grad is None:
                assert hg[n].grad is None or float(hg[n].grad.abs().max()) == 0.0, f"{name}: {n} should have no grad"
                continue
            assert hg[n].grad is not None, f"{name}: {n} missing grad"
            assert_grad_close(hg[n].grad, p.grad, gmax, f"{name}: grad {n}")


def test_full_s4_train_step():
    """One optimisation step (5 task losses, PCGrad with a fixed task order, clip, AdamW) -- parameters
    after the step and the set of parameters that were updated at all (row a17's quirk)."""
    scheme = "s4"
    tasks, domains = PT.ACTIVE_TASKS[scheme], PT.PRETRAIN_DOMAINS[scheme]
    om, hm, gen = _models(tasks, domains, 21)
    before = {k: v.clone() for k, v in om.state_dict().items()}
    state = PT.StepState(hm, PT.PretrainConfig(scheme, 0), steps_per_epoch=462)
    host = S.pretrain_step_batches(gen, domains)
    dev_batches = {d: b.to(DEV) for d, b in host.items()}
    o_batches = {d: to_oracle(b) for d, b in host.items()}
    art_h, art_o = _artefacts(state, dev_batches, gen)
    order = ["graph_contrast", "node_feat_mask", "graph_prop", "link_pred", "node_contrast"]
    otemp, ogrl = OTr.TemperatureScheduler(462 * 50), OTr.GRLScheduler(50, 462)
    otasks = OTk.instantiate_tasks(om, tasks, ogrl, otemp)
    oopt, obal = OTr.make_optimizer(om, tasks), OTr.AdaptiveLossBalancer()
    # lr 1e-5 moves weights by ~1e-5 per step: compare the UPDATE, not the weights, and use a larger lr
    for g in oopt.param_groups:
        g["lr"] = g["lr"] * 1000
    for g in state.optimizer.optimizer.param_groups:
        g["lr"] = g["lr"] * 1000
    lo, _, to, mo = OTr.train_step(om, otasks, oopt, obal, ogrl, otemp, o_batches, gen, artifacts=art_o, order=order)
    lh, _, th, mh = PT.train_step(state, dev_batches, gen, artefacts=art_h, order=order)
    for n in tasks:
        assert_close(lh[n], lo[n], OUT_RTOL, f"loss {n}")
    assert_close(th, to, OUT_RTOL, "balanced total")
    # 14 shared tensors (biases feeding a train-mode BN) have analytically zero gradients: whether their
    # rounding noise is exactly 0.0 (-> pair skipped) differs between implementations; everything else must agree
    assert abs(mh["gradient_surgery/total_projections"] - mo["gradient_surgery/total_projections"]) <= 14 * 10
    after_o, after_h = om.state_dict(), hm.state_dict()
    moved_o = {k for k in before if before[k].dtype.is_floating_point and not torch.equal(before[k], after_o[k])}
    moved_h = {k for k in before if before[k].dtype.is_floating_point and not torch.equal(before[k], after_h[k].cpu())}
    assert moved_h == moved_o, (sorted(moved_h ^ moved_o))[:10]
    assert "heads.link_pred.predictor.mlp.0.weight" not in moved_o        # a17: neither first-shuffled nor last task
    # Adam's first step is lr*sign(g) for |g| >> eps: an element whose gradient is ~0 can take the opposite
    # sign on the two sides, so bound the mass of the difference, not its max
    num = sum(((after_h[k].cpu() - after_o[k]).double() ** 2).sum().item() for k in moved_o)
    den = sum(((after_o[k] - before[k]).double() ** 2).sum().item() for k in moved_o)
    assert (num / den) ** 0.5 <= 2e-2, f"relative update error {(num / den) ** 0.5:.3e}"


def test_finetune_cora_shape_forward_backward():
    gen = torch.Generator().manual_seed(31)
    torch.manual_seed(31)
    om = OM.FinetuneGNN(torch.device("cpu"), "Cora_NC", "full_finetune")
    hm = FinetuneGNN(torch.device("cpu"), "Cora_NC", "full_finetune")
    copy_state(hm, om); hm.device = DEV; hm.to(DEV)
    set_dropout(om, 0.0); set_dropout(hm, 0.0)
    om.train(); hm.train()
    c = S.cora_like(gen)
    from gnn_pretraining_amd.graph import Batch
    b = Batch.from_data_list([c])
    idx = torch.randperm(c.num_nodes, generator=gen)[:140]
    lo = torch.nn.functional.cross_entropy(om(to_oracle(b))[idx], c.y[idx])
    lo.backward()
    from gnn_pretraining_amd import operators as O
    bd = b.to(DEV)
    logits = hm(bd)
    lh = O.cross_entropy_sum(O.take_rows(logits, idx.to(DEV)), c.y[idx].to(DEV)) / 140
    lh.backward()
    assert_close(lh, lo, OUT_RTOL, "Cora NC loss")
    og, hg = dict(om.named_parameters()), dict(hm.named_parameters())
    gmax = max(p.grad.abs().max().item() for p in og.values())
    for n, p in og.items():
        assert_grad_close(hg[n].grad, p.grad, gmax, f"grad {n}")


def test_finetune_link_prediction_train_step_with_mined_negatives():
    """finetune.py:181-204: embeddings under no_grad -> mined negatives -> scorer on [pos | neg] -> BCE (mean).
    The mined pairs are held against the oracle miner run on the GPU path's own embeddings (index work exact), the
    loss and gradients against the oracle model on the same edges."""
    from gnn_pretraining_amd.data import data_setup as DS
    from gnn_pretraining_amd.data.finetune_data_loaders import LinkLoader, LinkPredictionDataset
    from gnn_pretraining_amd.finetune import finetune as FT
    from gnn_pretraining_amd.graph import Batch
    from oracle import miner as OMi
    gen = torch.Generator().manual_seed(77)
    torch.manual_seed(77)
    c = S.cora_like(gen, num_nodes=600, undirected_edges=1500)
    splits = DS.create_link_prediction_splits(c)
    data = Batch.from_data_list([c])
    loader = LinkLoader(LinkPredictionDataset(data, splits, "train"), 256)
    om = OM.FinetuneGNN(torch.device("cpu"), "Cora_LP", "full_finetune")
    hm = FinetuneGNN(torch.device("cpu"), "Cora_LP", "full_finetune")
    copy_state(hm, om); hm.device = DEV; hm.to(DEV)
    set_dropout(om, 0.0); set_dropout(hm, 0.0)
    om.train(); hm.train()
    batch = next(iter(loader))
    train_edges = splits["train_pos"].to(DEV).contiguous()
    miner = FT.LinkPredictionHardNegativeMiner()
    # what the miner will see: the no_grad embedding pass (it also moves the BN running statistics, as in the reference)
    with torch.no_grad():
        probe = FinetuneGNN(torch.device("cpu"), "Cora_LP", "full_finetune")
        copy_state(probe, om); probe.to(DEV); set_dropout(probe, 0.0); probe.train()
        emb = probe.gnn_backbone(probe.input_encoder(data.x.to(DEV)), train_edges)
    want_neg = OMi.mine_hard_negatives_for_edges(emb.cpu(), batch[1], 256, splits["train_pos"],
                                                 similarity=ops_matrix(emb, train_edges))
    assert torch.equal(miner.mine_hard_negatives_for_edges(emb, batch[1].to(DEV), 256, train_edges).cpu(), want_neg)
    loss, targets, pred, prob = FT.process_batch(hm, batch, DEV, "link_prediction", "Cora_LP", miner, train_edges)
    loss.backward()
    assert targets.numel() == 512 and int(targets.sum()) == 256 and prob.shape == (512, 2)
    # oracle: same two forward passes (the first only for its BN side effect), same edges
    with torch.no_grad():
        om.gnn_backbone(om.input_encoder(c.x), splits["train_pos"])
    all_edges = torch.cat([batch[1], want_neg], dim=1)
    labels = torch.cat([torch.ones(256), torch.zeros(256)])
    lo = torch.nn.functional.binary_cross_entropy(om(to_oracle(data), edge_index=all_edges, message_passing_edges=splits["train_pos"]), labels)
    lo.backward()
    assert_close(loss, lo, OUT_RTOL, "Cora LP loss")
    og, hg = dict(om.named_parameters()), dict(hm.named_parameters())
    gmax = max(p.grad.abs().max().item() for p in og.values())
    for n, p in og.items():
        assert_grad_close(hg[n].grad, p.grad, gmax, f"grad {n}")
    for (n, a), (_, o) in zip(hm.named_buffers(), om.named_buffers()):
        assert_close(a, o, 1e-4, f"buffer {n}")


def ops_matrix(emb, edges):
    from gnn_pretraining_amd import ops
    return ops.hard_negative_topk(emb.contiguous(), edges, 1, return_matrix=True)[1].cpu()


def test_finetune_node_classification_engine_matches_the_oracle_step():
    """BASELINE.json configs[4]: one Cora_NC-shaped full-graph fine-tune step (2,708 x 1,433, 5 GIN layers, CE on 140 nodes,
    AdamW with the reference's parameter groups -- finetune.py:162-179, finetune_model.py:38-64) on the explicit-kernel engine
    (finetune/engine.py) against the oracle model + torch.optim.AdamW: loss, every gradient, every parameter after the update,
    running statistics; the engine's parameters stay views of the module (state_dict keys and values)."""
    from gnn_pretraining_amd.finetune.engine import NodeClassificationEngine
    from gnn_pretraining_amd.graph import Batch
    gen = torch.Generator().manual_seed(33)
    torch.manual_seed(33)
    om = OM.FinetuneGNN(torch.device("cpu"), "Cora_NC", "full_finetune")
    hm = FinetuneGNN(torch.device("cpu"), "Cora_NC", "full_finetune")
    copy_state(hm, om); hm.device = DEV; hm.to(DEV)
    set_dropout(om, 0.0)
    om.train(); hm.train()
    c = S.cora_like(gen)
    idx = torch.randperm(c.num_nodes, generator=gen)[:140]
    eng = NodeClassificationEngine(hm, c.x, c.edge_index, DEV, seed=3)
    eng.dropout_p = 0.0
    keys_before = list(hm.state_dict().keys())
    # oracle step
    oopt = torch.optim.AdamW(om.param_groups)
    for g in oopt.param_groups:
        g["lr"] *= 100                                   # visible update (lr 1e-4 moves weights by 1e-4 x Adam's unit step)
    eng.lr.mul_(100)
    before = {k: v.clone() for k, v in om.state_dict().items()}
    # The HIP step first; the oracle then runs with the ReLU gates the engine used (oracle/gates.py), read from the activations the step leaves
    # behind in call order -- encoder, then the inner and the outer ReLU of each layer -- so that no pre-activation within rounding of zero gates
    # the two sides differently.  (Without: one flipped unit of one node moves a whole row of a weight gradient; with 140 labelled nodes that
    # was 26 % of the tensor's largest entry on one box, where the pre-training step's flips stay near 1e-3.)
    eng.step(idx.to(DEV), c.y[idx].to(DEV))
    torch.cuda.synchronize()
    from oracle import gates as OGt
    masks = [(eng.h[0] > 0).cpu()]
    for l in range(5):
        masks += [(eng.r1[l] > 0).cpu(), (eng.h[l + 1] > 0).cpu()]
    tape = OGt.GateTape(masks)
    with OGt.use_tape(tape):
        lo = torch.nn.functional.cross_entropy(om(to_oracle(Batch.from_data_list([c])))[idx], c.y[idx])
    assert tape.done()
    oopt.zero_grad(); lo.backward(); oopt.step()
    assert abs(eng.loss() - lo.item()) <= 1e-4 * abs(lo.item())
    og = dict(om.named_parameters())
    gmax = max(p.grad.abs().max().item() for p in og.values())
    worst = 0.0
    for n, p in og.items():
        if n.endswith("linear.bias") or n.endswith("gin_conv.nn.0.bias") or n.endswith("gin_conv.nn.3.bias"):
            # analytically ZERO (a bias in front of a train-mode BatchNorm): both sides hold rounding noise of a 2,708-row sum (the fp32
            # ORACLE's was 1.3e-5 of 0.124 in one run, the HIP path's 7e-7)
            assert eng.gradient(n).abs().max().item() <= 1e-4 * gmax and p.grad.abs().max().item() <= 1e-3 * gmax, n
        else:
            worst = max(worst, assert_grad_tight(eng.gradient(n), p.grad, gmax, f"grad {n}"))
    print(f"Cora_NC engine step, shared gates: worst gradient error {worst:.2e} ({tape.flips} gates the oracle would have set differently)")
    eng.flush_counters()
    after_o, after_h = om.state_dict(), hm.state_dict()
    assert list(after_h.keys()) == keys_before
    num = den = 0.0
    for k, v in after_o.items():
        if "running_" in k:
            assert_close(after_h[k], v, 1e-4, f"buffer {k}")
        elif k.endswith("num_batches_tracked"):
            assert int(after_h[k]) == int(v)
        elif k.endswith("linear.bias") or k.endswith("gin_conv.nn.0.bias") or k.endswith("gin_conv.nn.3.bias"):
            # a bias in front of a train-mode BatchNorm has an analytically ZERO gradient: both sides hold rounding noise, and Adam's
            # first step turns noise into +-lr -- nothing to compare beyond "it moved by at most one step"
            assert (after_h[k].cpu() - before[k]).abs().max().item() <= 1.01 * 100 * 1e-3
        else:
            num += ((after_h[k].cpu() - v).double() ** 2).sum().item()
            den += ((v - before[k]).double() ** 2).sum().item()
    assert (num / den) ** 0.5 <= 2e-2, f"relative update error {(num / den) ** 0.5:.3e}"
    # a second step runs from the updated state (moments, step counters) and keeps the loss finite
    eng.step(idx.to(DEV), c.y[idx].to(DEV))
    assert np.isfinite(eng.loss())
    # evaluation goes through the MODULE (finetune.evaluate): its encoder weight is now a row-strided [256, 1433] view of the engine's
    # K-padded slot -- the module forward must take it (round 3: the CLI crashed at its first validation pass) and agree with the engine's
    hm.eval()
    with torch.no_grad():
        logits_mod = hm(Batch.from_data_list([c]).to(DEV))
    assert not hm.input_encoder.linear.weight.is_contiguous()
    logits_eng = eng.forward()
    torch.cuda.synchronize()
    assert_close(logits_mod, logits_eng.cpu(), 1e-4, "module forward on the engine's strided encoder weight")
    hm.train()


def test_finetune_engine_graph_replay_equals_the_eager_steps():
    """The Cora_NC step three ways (finetune/engine.py): replayed from its captured hipGraph (one replay instead of ~120 launches, dropout seed read
    from a device word), launched eagerly with the weight-gradient GEMMs on the side stream (the default), and eagerly on one stream: dropout ON,
    five steps, parameters / moments / running statistics / loss BITWISE equal -- replay k draws the masks of eager step k, and a weight
    gradient is the same GEMM on either stream."""
    from gnn_pretraining_amd.finetune.engine import NodeClassificationEngine
    gen = torch.Generator().manual_seed(35)
    c = S.cora_like(gen)
    idx = torch.randperm(c.num_nodes, generator=gen)[:140].to(DEV)
    y = c.y[idx.cpu()].to(DEV)
    outs = []
    for use_graph, fork in ((True, False), (False, True), (False, False)):
        torch.manual_seed(35)
        hm = FinetuneGNN(torch.device("cpu"), "Cora_NC", "full_finetune")
        hm.device = DEV; hm.to(DEV); hm.train()
        eng = NodeClassificationEngine(hm, c.x, c.edge_index, DEV, seed=9)
        eng.use_graph, eng.fork_wgrads = use_graph, fork
        eng.lr.mul_(30)
        losses = []
        for k in range(5):
            eng.step(idx, y)
            losses.append(eng.loss())
        assert (eng._graph is not None) == use_graph
        eng.flush_counters()
        torch.cuda.synchronize()
        outs.append((eng.flat.clone(), eng.exp_avg.clone(), eng.exp_avg_sq.clone(), {k: v.clone() for k, v in hm.state_dict().items()}, losses, eng.step_count))
    a = outs[0]
    assert len(set(a[4])) == 5                                           # the loss moves: the steps are real
    for b in outs[1:]:
        assert a[5] == b[5] == 5 and a[4] == b[4], (a[4], b[4])
        for i in range(3):
            assert torch.equal(a[i], b[i])
        for k in a[3]:
            assert torch.equal(a[3][k], b[3][k]), k
