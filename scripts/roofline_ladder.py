"""Aggregation-kernel size ladder of SURVEY.md section 8d: G-ENZ batches of 8 / 32 / 1,024 / 65,536 graphs and the
Cora shape, F = 256 fp32, int32 CSR.  Prints algorithmic GB/s per rung (HIP events on the launch stream) and one
Cora_NC full-graph training step (module path) for the fine-tuning config."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnn_pretraining_amd import ops, synthetic as S
from gnn_pretraining_amd.graph import Batch

dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(7)


def time_agg(ei, N, iters=50):
    E = ei.size(1)
    csr = ops.csr_build(ei, N)
    x, eps, out = torch.randn(N, 256, device=dev), torch.zeros(1, device=dev), torch.empty(N, 256, device=dev)
    l = ops.L.lib()
    args = (ops._ptr(x), ops._ptr(csr.rowptr), ops._ptr(csr.col), ops._ptr(eps), ops._ptr(out), N, 256)
    for _ in range(5):
        l.gmp_gin_aggregate_fwd(*args, ops._stream(x))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        l.gmp_gin_aggregate_fwd(*args, ops._stream(x))
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    b = 2 * 4 * 256 * N + 4 * (N + 1) + 4 * E
    return {"rows": N, "edges": E, "us": round(us, 2), "alg_GBps": round(b / us / 1e3, 1), "frac_of_8TBps": round(b / us / 1e3 / 8000, 4)}


out = {}
base = Batch.from_data_list([S.random_graph(gen, 4) for _ in range(1024)])
for g in (8, 32, 1024):
    sub = Batch.from_data_list(base.to_data_list()[:g])
    out[f"{g} graphs"] = time_agg(sub.edge_index.to(dev), sub.num_nodes)
n0, e0 = base.num_nodes, base.num_edges
ei = base.edge_index.to(dev)
big = (ei.view(2, 1, e0) + (torch.arange(64, device=dev) * n0).view(1, 64, 1)).reshape(2, -1).contiguous()
out["65536 graphs"] = time_agg(big, n0 * 64, iters=20)
c = S.cora_like(gen, dim=8)
out["Cora shape"] = time_agg(c.edge_index.to(dev), c.num_nodes)

# Cora_NC full fine-tune: one training step = full-graph fwd + bwd + AdamW (reference: ~0.130 s per epoch on an L4,
# derived from analysis/results/experiment_results.csv:164-166, which also includes eval, sklearn metrics and logging)
from gnn_pretraining_amd.models import FinetuneGNN
from gnn_pretraining_amd import operators as O
torch.manual_seed(0)
m = FinetuneGNN(dev, "Cora_NC", "full_finetune"); m.train()
g = S.cora_like(gen)
data = Batch.from_data_list([g]).to(dev)
idx = torch.randperm(2708, generator=gen)[:140].to(dev); y = g.y.to(dev)[idx]
opt = torch.optim.AdamW(m.param_groups)
def step():
    loss = O.cross_entropy_sum(O.take_rows(m(data), idx), y) / 140
    opt.zero_grad(); loss.backward(); opt.step()
for _ in range(5): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): step()
torch.cuda.synchronize()
out["Cora_NC full fine-tune step (module path)"] = {"ms": round((time.perf_counter() - t0) / 50 * 1e3, 3)}
print(json.dumps(out, indent=1))
