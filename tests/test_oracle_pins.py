"""Pin the oracle against material the reference itself provides:
goldens from its four importable modules + parameter counts in its results CSV."""
import json
import os

import pytest
import torch

from oracle import models as OM
from oracle import train as OT

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_callers.json")))


def test_temperature_scheduler():
    for c in GOLD["schedulers"]["temperature"]:
        s = OT.TemperatureScheduler(c["total"])
        s.current_step = c["step"]
        assert s() == pytest.approx(c["value"], rel=0, abs=1e-15)


def test_grl_scheduler():
    for c in GOLD["schedulers"]["grl"]:
        s = OT.GRLScheduler(c["epochs"], c["steps_per_epoch"])
        s.current_step = c["step"]
        assert s() == pytest.approx(c["value"], rel=0, abs=1e-15)
    # SURVEY section 4 known answers
    s = OT.GRLScheduler(10, 10); s.current_step = 70
    assert s() == 0.009866142981514305
    t = OT.TemperatureScheduler(100); t.current_step = 50
    assert t() == pytest.approx(0.31622776601683794, abs=1e-15)


@pytest.mark.parametrize("case", GOLD["balancer"], ids=lambda c: c["name"])
def test_loss_balancer(case):
    b = OT.AdaptiveLossBalancer()
    totals = []
    for i in range(case["calls"]):
        t = b.balance_losses({k: torch.tensor(v) for k, v in case["losses"].items()}, case["lambda"])
        totals.append(float(t))
        if i == 0:
            assert b.get_current_weights() == pytest.approx(case["first_weights"])
    assert totals == pytest.approx(case["totals"], rel=1e-6, abs=1e-12)
    assert b.get_current_weights() == pytest.approx(case["last_weights"], rel=1e-12)


def _toy(case):
    import torch.nn as nn

    class Toy(nn.Module):
        def __init__(self):
            super().__init__()
            self.shared = nn.Linear(3, 4)
            self.head_a = nn.Linear(4, 2)
            self.head_b = nn.Linear(4, 2)
            self.head_c = nn.Linear(4, 1)
            self.unused = nn.Parameter(torch.ones(2))

    m = Toy()
    m.load_state_dict({k: torch.tensor(v, dtype=torch.float32).reshape(m.state_dict()[k].shape)
                       for k, v in case["init"].items()})
    x = torch.tensor(case["x"], dtype=torch.float32).reshape(5, 3)
    h = torch.tanh(m.shared(x))
    losses = {"a": (m.head_a(h) ** 2).sum(),
              "b": -(m.head_b(h)).sum() * 3.0 - (m.head_a(h) ** 2).sum() * 0.5,
              "c": (m.head_c(h) - 1.0).abs().sum()}
    return m, losses


@pytest.mark.parametrize("case", GOLD["pcgrad"]["toy"], ids=lambda c: "-".join(c["order"]) + f"-s{c['seed']}")
def test_pcgrad_toy_model(case):
    """Whole apply_gradient_surgery incl. the a17 quirk (which .grad stay None)."""
    m, losses = _toy(case)
    metrics = OT.apply_gradient_surgery(m, losses, order=list(case["order"]))
    assert metrics == pytest.approx(case["metrics"])
    for n, p in m.named_parameters():
        want = case["final_grads"][n]
        if want is None:
            assert p.grad is None, n
        else:
            assert p.grad is not None, n
            torch.testing.assert_close(p.grad.flatten().double(), torch.tensor(want, dtype=torch.float64),
                                       rtol=2e-5, atol=1e-6)   # fp32 on another CPU rounds tanh / the dot products differently in the last bits


def test_pcgrad_edge_cases():
    e = GOLD["pcgrad"]["edge"]
    tg = {t: {k: torch.tensor(v, dtype=torch.float32) for k, v in d.items()} for t, d in e["task_grads"].items()}
    for c in e["cases"]:
        final, metrics = OT.pcgrad_combine({t: dict(d) for t, d in tg.items()}, c["order"])
        assert metrics == pytest.approx(c["metrics"])
        assert set(final) == set(c["final"])
        for k, v in c["final"].items():
            torch.testing.assert_close(final[k].double(), torch.tensor(v, dtype=torch.float64), rtol=2e-5, atol=1e-6)   # fp32 on another CPU rounds tanh / the dot products differently in the last bits


@pytest.mark.parametrize("scheme", sorted(GOLD["optimizer_groups"]))
def test_optimizer_groups(scheme):
    tasks = OT.SCHEMES[scheme]
    domains = ["ENZYMES"] if scheme == "b4" else OM.PRETRAIN_DOMAINS
    m = OM.PretrainableGNN(torch.device("cpu"), domains, tasks)
    names = {id(p): n for n, p in m.named_parameters()}
    opt = OT.make_optimizer(m, tasks)
    got = [{"name": g["name"], "lr": g["lr"], "weight_decay": g["weight_decay"], "betas": list(g["betas"]),
            "eps": g["eps"], "params": [names[id(p)] for p in g["params"]]} for g in opt.param_groups]
    assert got == GOLD["optimizer_groups"][scheme]


# analysis/results/experiment_results.csv column 12 (trainable_parameters), all 12 distinct rows
CSV_PARAM_COUNTS = {
    ("CiteSeer_LP", "full_finetune"): 2468102, ("CiteSeer_LP", "linear_probe"): 1145857,
    ("CiteSeer_NC", "full_finetune"): 2272523, ("CiteSeer_NC", "linear_probe"): 950278,
    ("Cora_LP", "full_finetune"): 1886982, ("Cora_LP", "linear_probe"): 564737,
    ("Cora_NC", "full_finetune"): 1691660, ("Cora_NC", "linear_probe"): 369415,
    ("ENZYMES", "full_finetune"): 1355915, ("ENZYMES", "linear_probe"): 33670,
    ("PTC_MR", "full_finetune"): 1360775, ("PTC_MR", "linear_probe"): 38530,
}


@pytest.mark.parametrize("key", sorted(CSV_PARAM_COUNTS))
def test_finetune_param_counts(key):
    m = OM.FinetuneGNN(torch.device("cpu"), key[0], key[1])
    assert sum(p.numel() for p in m.parameters() if p.requires_grad) == CSV_PARAM_COUNTS[key]


def test_pretrain_state_dict_keys():
    """the key layout consumed by load_pretrained_weights."""
    m = OM.PretrainableGNN(torch.device("cpu"), OM.PRETRAIN_DOMAINS, OT.SCHEMES["s5"])
    sd = m.state_dict()
    for k in ["gnn_backbone.layers.0.gin_conv.eps", "gnn_backbone.layers.4.gin_conv.nn.0.weight",
              "gnn_backbone.layers.2.gin_conv.nn.1.running_var", "gnn_backbone.layers.2.gin_conv.nn.3.bias",
              "gnn_backbone.layers.1.batch_norm.num_batches_tracked", "input_encoders.ENZYMES.linear.weight",
              "input_encoders.MUTAG.batch_norm.running_mean", "mask_token",
              "heads.node_feat_mask.NCI1.mlp.0.weight", "heads.node_feat_mask.NCI1.mlp.3.bias",
              "heads.link_pred.predictor.mlp.0.weight", "heads.link_pred.predictor.mlp.3.weight",
              "heads.domain_adv.classifier.mlp.3.bias", "heads.graph_prop.PROTEINS.mlp.0.weight"]:
        assert k in sd, k
    assert sd["gnn_backbone.layers.0.gin_conv.eps"].shape == (1,)
    assert sd["mask_token"].shape == (256,)
    one_layer = sum(p.numel() for p in m.gnn_backbone.layers[0].parameters())
    assert one_layer == 264449 and sum(p.numel() for p in m.gnn_backbone.parameters()) == 1322245
    s4 = OM.PretrainableGNN(torch.device("cpu"), OM.PRETRAIN_DOMAINS, OT.SCHEMES["s4"])
    assert sum(p.numel() for p in s4.parameters()) == 3669302


def test_oracle_step_reproduces_its_committed_fixture():
    """tests/golden/oracle_step.json freezes the oracle's numbers for one seeded s4 step (losses, PCGrad counts, parameters
    after the step): an edit to oracle/ that moves them is caught here, on the CPU."""
    import importlib.util
    import json
    import os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_oracle_goldens", os.path.join(here, "make_oracle_goldens.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    got, want = mod.run(), json.load(open(os.path.join(here, "oracle_step.json")))
    assert got["nodes"] == want["nodes"] and got["edges"] == want["edges"]
    for k, v in want["losses"].items():
        assert abs(got["losses"][k] - v) <= 1e-5 * abs(v), k
    # a conflict is "dot product < 0": pairs within rounding of orthogonal flip with the BLAS thread count, so the counts
    # (and, through the few extra projections, the updates) are only reproducible to a few percent / a few 1e-5
    for k in ("gradient_surgery/total_conflicts", "gradient_surgery/total_projections"):
        assert abs(got["pcgrad"][k] - want["pcgrad"][k]) <= 0.05 * want["pcgrad"][k], k
    for k, v in want["param_sq_sum_after"].items():
        # loose on purpose: Adam turns the rounding noise of an (analytically) zero gradient into +-lr steps, and PCGrad's
        # borderline conflicts flip with the BLAS thread count; the test is there to catch real drift
        assert abs(got["param_sq_sum_after"][k] - v) <= 5e-3 * abs(v) + 1e-4, k
    for k, v in want["running_mean_sum_after"].items():
        assert abs(got["running_mean_sum_after"][k] - v) <= 1e-5 * max(abs(v), 1e-3), k
