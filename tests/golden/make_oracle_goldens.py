"""Frozen outputs of the CPU oracle on a small seeded pre-training step, so that (a) a change to oracle/ that moves its
numbers is caught on the CPU (tests/test_oracle_pins.py) and (b) the HIP engine is held to committed numbers as well as to
the live oracle (tests/test_gpu_engine.py).  The reference itself cannot produce these (it needs torch_geometric, absent
here): this pins the oracle against drift, it does not pin it to the reference.

    python tests/golden/make_oracle_goldens.py          # rewrites tests/golden/oracle_step.json

Inputs: scheme s4, torch seed 123, batches from gnn_pretraining_amd.synthetic.pretrain_step_batches(Generator(123)),
dropout off, the oracle's own draws from the same generator (link-prediction negatives: PyG's sampler draws from Python's
`random`, here random.Random(123)), PCGrad order fixed."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

ORDER = ["graph_contrast", "node_feat_mask", "graph_prop", "link_pred", "node_contrast"]
SEED, SCHEME = 123, "s4"


def run():
    from gnn_pretraining_amd import synthetic as S
    from oracle import models as OM, tasks as OTk, train as OTr
    from parity_util import set_dropout, to_oracle
    tasks = ["node_feat_mask", "link_pred", "node_contrast", "graph_contrast", "graph_prop"]
    domains = ["MUTAG", "PROTEINS", "NCI1", "ENZYMES"]
    torch.manual_seed(SEED)
    gen = torch.Generator().manual_seed(SEED)
    model = OM.PretrainableGNN(torch.device("cpu"), domains, tasks)
    set_dropout(model, 0.0)
    model.train()
    host = S.pretrain_step_batches(gen, domains)
    batches = {d: to_oracle(b) for d, b in host.items()}
    temp, grl = OTr.TemperatureScheduler(462 * 50), OTr.GRLScheduler(50, 462)
    otasks = OTk.instantiate_tasks(model, tasks, grl, temp)
    import random
    otasks["link_pred"].py_rng = random.Random(SEED)
    opt, bal = OTr.make_optimizer(model, tasks), OTr.AdaptiveLossBalancer()
    for g in opt.param_groups:
        g["lr"] *= 1000
    losses, _, total, metrics = OTr.train_step(model, otasks, opt, bal, grl, temp, batches, gen, order=list(ORDER))
    sd = model.state_dict()
    return {
        "scheme": SCHEME, "seed": SEED, "order": ORDER, "lr_scale": 1000,
        "nodes": {d: int(b.num_nodes) for d, b in host.items()}, "edges": {d: int(b.num_edges) for d, b in host.items()},
        "losses": {k: float(v.detach()) for k, v in losses.items()}, "total": float(total.detach()),
        "pcgrad": {k: float(v) for k, v in metrics.items()},
        "param_sq_sum_after": {k: float((v.double() ** 2).sum()) for k, v in sd.items() if v.dtype.is_floating_point and "running_" not in k},
        "running_mean_sum_after": {k: float(v.double().sum()) for k, v in sd.items() if k.endswith("running_mean")},
    }


if __name__ == "__main__":
    out = run()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_step.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", path, {k: round(v, 6) for k, v in out["losses"].items()})
