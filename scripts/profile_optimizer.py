"""The optimizer tail of the s4 step alone (PCGrad Gram / solve / combine, clip norm, AdamW over the step's real gradient buffers), re-enqueued
back to back: for rocprofv3 --kernel-trace --stats.  python scripts/profile_optimizer.py [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from gnn_pretraining_amd.engine import StepEngine
from gnn_pretraining_amd.models.pretrain_model import PretrainableGNN
from gnn_pretraining_amd.pretrain import pretrain as PT

dev = torch.device("cuda:0")
torch.set_num_threads(1)
model = PretrainableGNN(device=dev, domain_names=PT.PRETRAIN_DOMAINS["s4"], task_names=PT.ACTIVE_TASKS["s4"])
model.train()
eng = StepEngine(model, PT.ACTIVE_TASKS["s4"], PT.PRETRAIN_DOMAINS["s4"], dev, seed=0, rng_mode="vectorized")
pool = B.make_pool(0, dev, eng.dpad)
gen = torch.Generator().manual_seed(0)
for k in range(10):
    eng.step(pool[k % len(pool)], gen)
p = eng.last_plan
torch.cuda.synchronize()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(reps):
    eng._optimizer(p, None, True)
b.record()
torch.cuda.synchronize()
print(f"optimizer tail: {a.elapsed_time(b) / reps * 1e3:.1f} us per call ({reps} calls back to back)")
