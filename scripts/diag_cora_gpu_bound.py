"""GPU time of the Cora_NC fine-tune step with the host parked (main spins while `reps` eager steps are enqueued behind it): what would a launcher that
keeps up buy, with the weight-gradient GEMMs beside the chain (fork) or in it?  python scripts/diag_cora_gpu_bound.py [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnn_pretraining_amd import _lib as L, synthetic as S
from gnn_pretraining_amd.finetune.engine import NodeClassificationEngine
from gnn_pretraining_amd.models import FinetuneGNN

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(0)
torch.manual_seed(0)
g = S.cora_like(gen)
for fork in (False, True, False, True):
    model = FinetuneGNN(dev, "Cora_NC", "full_finetune")
    model.train()
    eng = NodeClassificationEngine(model, g.x, g.edge_index, dev, seed=0)
    eng.use_graph, eng.fork_wgrads = False, fork
    idx = torch.randperm(g.num_nodes, generator=gen)[:140].to(dev)
    y = g.y[idx.cpu()].to(dev)
    for _ in range(10):
        eng._enqueue(idx, y, True, forked=fork)
        eng.step_count += 1
    torch.cuda.synchronize()
    main = torch.cuda.current_stream(dev)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    L.check(L.lib().gmp_spin_us(60000, main.cuda_stream), "spin")
    a.record()
    t0 = time.perf_counter()
    for _ in range(reps):
        eng._enqueue(idx, y, True, forked=fork)          # (step() forks only inside a capture)
        eng.step_count += 1
    t1 = time.perf_counter()
    b.record()
    torch.cuda.synchronize()
    print(f"fork={fork}: GPU {a.elapsed_time(b) / reps:.3f} ms/step ({reps} steps enqueued in {(t1 - t0) * 1e3:.1f} ms)")
