"""Generate golden vectors from the reference's own importable modules.

Run ONLY in the build container (the reference does not travel to the GPU box):

    PYTHONPATH=/root/reference python tests/golden/make_reference_goldens.py

Imports exactly the four reference modules that need nothing beyond torch /
stdlib: src.pretrain.schedulers, adaptive_loss_balancer,
gradient_surgery, optimizers.  Everything else on the hot path needs
torch_geometric, which is not installed.  Output: tests/golden/reference_callers.json
(inputs and expected outputs only -- no reference source text).
"""
import json
import os
import random
import sys

import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))      # repo root, for oracle.models (names only)

from src.pretrain.schedulers import GRLScheduler, TemperatureScheduler            # noqa: E402
from src.pretrain.adaptive_loss_balancer import AdaptiveLossBalancer              # noqa: E402
from src.pretrain.gradient_surgery import GradientSurgery                         # noqa: E402
from src.pretrain.optimizers import TaskSpecificOptimizer                         # noqa: E402
import src.pretrain.gradient_surgery as gs_mod                                    # noqa: E402


def tolist(t):
    return t.detach().double().flatten().tolist()


def schedulers():
    out = {"temperature": [], "grl": []}
    for total in (100, 23100, 7):
        s = TemperatureScheduler(total_steps=total)
        for step in (0, 1, total // 3, total // 2, total - 1, total, total + 5):
            s.current_step = step
            out["temperature"].append({"total": total, "step": step, "value": s()})
    for ep, spe in ((10, 10), (50, 462), (3, 5)):
        g = GRLScheduler(total_epochs=ep, steps_per_epoch=spe)
        n = ep * spe
        for step in (0, int(0.4 * n) - 1, int(0.4 * n), int(0.4 * n) + 1, int(0.7 * n), n - 1, n):
            g.current_step = step
            out["grl"].append({"epochs": ep, "steps_per_epoch": spe, "step": step, "value": g()})
    return out


def balancer():
    cases = []
    # (name, list of calls; each call = dict of task->loss, lambda)
    seqs = {
        "warmup_two": [({"a": 2.0, "b": 4.0}, 0.0)] * 3,
        "single": [({"a": 2.5}, 0.0)] * 2,
        "post_warmup": [({"a": 2.0, "b": 4.0}, 0.0)] * 102,
        "five_tasks": [({"node_feat_mask": 0.9, "link_pred": 0.69, "node_contrast": 5.8,
                         "graph_contrast": 2.7, "graph_prop": 1.1}, 0.0)] * 103,
        "with_domain_adv": [({"a": 0.3, "b": 0.2, "domain_adv": 1.4}, 0.01)] * 102,
        "tiny_total": [({"a": 1e-9, "b": 2e-9}, 0.0)] * 101,
        "zeros": [({"a": 0.0, "b": 0.0}, 0.0)] * 102,
    }
    for name, seq in seqs.items():
        b = AdaptiveLossBalancer()
        totals, weights = [], []
        for losses, lam in seq:
            t = b.balance_losses({k: torch.tensor(v) for k, v in losses.items()}, lam)
            totals.append(float(t))
            weights.append(dict(b.get_current_weights()))
        cases.append({"name": name, "losses": seq[0][0], "lambda": seq[0][1], "calls": len(seq),
                      "totals": totals, "last_weights": weights[-1], "first_weights": weights[0]})
    return cases


class Toy(nn.Module):
    """Two shared tensors, three task heads, one tensor nobody touches."""

    def __init__(self):
        super().__init__()
        self.shared = nn.Linear(3, 4)
        self.head_a = nn.Linear(4, 2)
        self.head_b = nn.Linear(4, 2)
        self.head_c = nn.Linear(4, 1)
        self.unused = nn.Parameter(torch.ones(2))


def toy_losses(model, x):
    h = torch.tanh(model.shared(x))
    return {
        "a": (model.head_a(h) ** 2).sum(),
        "b": -(model.head_b(h)).sum() * 3.0 - (model.head_a(h) ** 2).sum() * 0.5,   # conflicts with a on shared
        "c": (model.head_c(h) - 1.0).abs().sum(),
    }


def pcgrad():
    cases = []
    for seed, order in ((0, ["a", "b", "c"]), (0, ["c", "a", "b"]), (1, ["b", "c", "a"]), (2, ["b", "a", "c"])):
        torch.manual_seed(seed)
        model = Toy()
        x = torch.randn(5, 3)
        init = {k: tolist(v) for k, v in model.state_dict().items()}
        losses = toy_losses(model, x)
        # inject the shuffle result (the reference uses unseeded random.shuffle)
        orig = gs_mod.random.shuffle

        def fixed_shuffle(lst, _o=order):
            lst[:] = _o
        gs_mod.random.shuffle = fixed_shuffle
        try:
            g = GradientSurgery(device=torch.device("cpu"))
            # also capture raw per-task gradients the same way the class does
            per_task = {}
            for name, loss in losses.items():
                model.zero_grad(set_to_none=True)
                loss.backward(retain_graph=True)
                per_task[name] = {n: tolist(p.grad) for n, p in model.named_parameters() if p.grad is not None}
            model.zero_grad(set_to_none=True)
            metrics = g.apply_gradient_surgery(model, losses, list(losses.keys()))
        finally:
            gs_mod.random.shuffle = orig
        final = {n: (None if p.grad is None else tolist(p.grad)) for n, p in model.named_parameters()}
        cases.append({"seed": seed, "order": order, "x": tolist(x), "init": init,
                      "shapes": {n: list(p.shape) for n, p in model.named_parameters()},
                      "task_grads": per_task, "final_grads": final, "metrics": metrics})
    # zero-norm and missing-tensor edge cases through _apply_pcgrad directly
    g = GradientSurgery(device=torch.device("cpu"))
    tg = {"t0": {"w": torch.tensor([1.0, 0.0]), "z": torch.zeros(2), "only0": torch.tensor([2.0])},
          "t1": {"w": torch.tensor([-1.0, 1.0]), "z": torch.tensor([1.0, 1.0])},
          "t2": {"w": torch.tensor([-1.0, -3.0]), "z": torch.tensor([-1.0, 0.0]), "only2": torch.tensor([5.0])}}
    edge = []
    for order in (["t0", "t1", "t2"], ["t2", "t1", "t0"], ["t1", "t2", "t0"]):
        orig = gs_mod.random.shuffle

        def fixed_shuffle(lst, _o=order):
            lst[:] = _o
        gs_mod.random.shuffle = fixed_shuffle
        try:
            final, metrics = g._apply_pcgrad({t: dict(d) for t, d in tg.items()}, list(tg.keys()))
        finally:
            gs_mod.random.shuffle = orig
        edge.append({"order": order, "final": {k: tolist(v) for k, v in final.items()}, "metrics": metrics})
    return {"toy": cases,
            "edge": {"task_grads": {t: {k: tolist(v) for k, v in d.items()} for t, d in tg.items()}, "cases": edge}}


def optimizer_groups():
    from oracle.models import PretrainableGNN       # a model with the reference's parameter NAMES
    from oracle.train import SCHEMES
    out = {}
    for scheme, tasks in SCHEMES.items():
        domains = ["ENZYMES"] if scheme == "b4" else ["MUTAG", "PROTEINS", "NCI1", "ENZYMES"]
        m = PretrainableGNN(torch.device("cpu"), domains, tasks)
        opt = TaskSpecificOptimizer(model=m, active_tasks=tasks)
        ids = {id(p): n for n, p in m.named_parameters()}
        out[scheme] = [{"name": g["name"], "lr": g["lr"], "weight_decay": g["weight_decay"],
                        "betas": list(g["betas"]), "eps": g["eps"],
                        "params": [ids[id(p)] for p in g["params"]]}
                       for g in opt.optimizer.param_groups]
    return out


def main():
    random.seed(0)
    blob = {"generated_from": "alonbebchuk/GNN-Pretraining src/pretrain/{schedulers,adaptive_loss_balancer,"
                              "gradient_surgery,optimizers}.py imported in the build container",
            "schedulers": schedulers(), "balancer": balancer(), "pcgrad": pcgrad(),
            "optimizer_groups": optimizer_groups()}
    path = os.path.join(HERE, "reference_callers.json")
    with open(path, "w") as f:
        json.dump(blob, f)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
