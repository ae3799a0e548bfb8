"""Host cost of one step's index work, single-threaded, no GPU: engine.draw (reference-order draws) and engine.plan (segment
layout) on the s4 bench workload.  python scripts/diag_host_costs.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_host_logic import _Inp, _planner
from gnn_pretraining_amd import synthetic as S

torch.set_num_threads(1)
e = _planner("reference", "s4")
gen = torch.Generator().manual_seed(3)
inps = [_Inp(S.pretrain_step_batches(gen, e.domains)) for _ in range(8)]
g = torch.Generator().manual_seed(9)
for i in range(20):
    e.plan(inps[i % 8], e.draw(inps[i % 8], g))
n, td, tp = 300, 0.0, 0.0
for i in range(n):
    a = time.perf_counter(); art = e.draw(inps[i % 8], g); b = time.perf_counter(); e.plan(inps[i % 8], art); c = time.perf_counter()
    td += b - a; tp += c - b
print("single thread: draw %.3f ms  plan %.3f ms per step; cpus %d" % (td / n * 1e3, tp / n * 1e3, os.cpu_count()))
