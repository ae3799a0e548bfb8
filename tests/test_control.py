"""The product's gradient-shaping callers vs the goldens generated from the reference's own modules
(CPU tensors: these classes are device-agnostic torch code, no kernels involved)."""
import json
import os

import pytest
import torch
import torch.nn as nn

from gnn_pretraining_amd.pretrain import control as Cn

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_callers.json")))


def test_schedulers():
    for c in GOLD["schedulers"]["temperature"]:
        s = Cn.TemperatureScheduler(c["total"]); s.current_step = c["step"]
        assert s() == pytest.approx(c["value"], abs=1e-15)
    for c in GOLD["schedulers"]["grl"]:
        s = Cn.GRLScheduler(c["epochs"], c["steps_per_epoch"]); s.current_step = c["step"]
        assert s() == pytest.approx(c["value"], abs=1e-15)


@pytest.mark.parametrize("case", GOLD["balancer"], ids=lambda c: c["name"])
def test_balancer(case):
    b = Cn.AdaptiveLossBalancer()
    totals = [float(b.balance_losses({k: torch.tensor(v) for k, v in case["losses"].items()}, case["lambda"]))
              for _ in range(case["calls"])]
    assert totals == pytest.approx(case["totals"], rel=1e-6, abs=1e-12)
    assert b.get_current_weights() == pytest.approx(case["last_weights"], rel=1e-9)


class Toy(nn.Module):
    def __init__(self):
        super().__init__()
        self.shared = nn.Linear(3, 4)
        self.head_a, self.head_b, self.head_c = nn.Linear(4, 2), nn.Linear(4, 2), nn.Linear(4, 1)
        self.unused = nn.Parameter(torch.ones(2))


@pytest.mark.parametrize("case", GOLD["pcgrad"]["toy"], ids=lambda c: "-".join(c["order"]) + f"-s{c['seed']}")
def test_gradient_surgery_matches_reference(case):
    m = Toy()
    m.load_state_dict({k: torch.tensor(v, dtype=torch.float32).reshape(m.state_dict()[k].shape) for k, v in case["init"].items()})
    x = torch.tensor(case["x"], dtype=torch.float32).reshape(5, 3)
    h = torch.tanh(m.shared(x))
    losses = {"a": (m.head_a(h) ** 2).sum(), "b": -(m.head_b(h)).sum() * 3.0 - (m.head_a(h) ** 2).sum() * 0.5,
              "c": (m.head_c(h) - 1.0).abs().sum()}
    metrics = Cn.GradientSurgery(torch.device("cpu")).apply_gradient_surgery(m, losses, list(losses), order=list(case["order"]))
    assert metrics == pytest.approx(case["metrics"])
    for n, p in m.named_parameters():
        want = case["final_grads"][n]
        if want is None:
            assert p.grad is None, n
        else:
            torch.testing.assert_close(p.grad.flatten().double(), torch.tensor(want, dtype=torch.float64), rtol=1e-5, atol=1e-6)


def test_gradient_surgery_edge_cases():
    e = GOLD["pcgrad"]["edge"]
    tg = {t: {k: torch.tensor(v, dtype=torch.float32) for k, v in d.items()} for t, d in e["task_grads"].items()}
    gs = Cn.GradientSurgery(torch.device("cpu"))
    for c in e["cases"]:
        final, metrics = gs._pcgrad({t: dict(d) for t, d in tg.items()}, c["order"])
        assert metrics == pytest.approx(c["metrics"])
        assert set(final) == set(c["final"])
        for k, v in c["final"].items():
            torch.testing.assert_close(final[k].double(), torch.tensor(v, dtype=torch.float64), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("scheme", sorted(GOLD["optimizer_groups"]))
def test_optimizer_groups(scheme):
    from gnn_pretraining_amd.models import PretrainableGNN
    from gnn_pretraining_amd.pretrain.pretrain import ACTIVE_TASKS, PRETRAIN_DOMAINS
    m = PretrainableGNN(torch.device("cpu"), PRETRAIN_DOMAINS[scheme], ACTIVE_TASKS[scheme])
    names = {id(p): n for n, p in m.named_parameters()}
    opt = Cn.TaskSpecificOptimizer(m, ACTIVE_TASKS[scheme])
    got = [{"name": g["name"], "lr": g["lr"], "weight_decay": g["weight_decay"], "betas": list(g["betas"]), "eps": g["eps"],
            "params": [names[id(p)] for p in g["params"]]} for g in opt.optimizer.param_groups]
    assert got == GOLD["optimizer_groups"][scheme]
