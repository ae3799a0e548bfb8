"""How far ahead of the GPU does the launcher thread run in the benchmark loop?  After N pipelined steps: the time from the host's last enqueue to the
GPU's completion (drain) in steps, plus the host launch time per step.  A lead near zero means the step is (partly) host-paced."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from gnn_pretraining_amd.engine import StepEngine, StepPrefetcher
from gnn_pretraining_amd.models.pretrain_model import PretrainableGNN
from gnn_pretraining_amd.pretrain import pretrain as PT
from gnn_pretraining_amd.pretrain.control import TemperatureScheduler
from gnn_pretraining_amd._host import limit_host_threads

limit_host_threads(1)
dev = torch.device("cuda:0")
if os.environ.get("GMP_MAIN_PRIORITY", "-1") != "0":
    torch.cuda.set_stream(torch.cuda.Stream(device=dev, priority=-1))
model = PretrainableGNN(device=dev, domain_names=PT.PRETRAIN_DOMAINS["s4"], task_names=PT.ACTIVE_TASKS["s4"])
model.train()
eng = StepEngine(model, PT.ACTIVE_TASKS["s4"], PT.PRETRAIN_DOMAINS["s4"], dev, seed=0, rng_mode="reference")
pool = B.make_pool(0, dev, eng.dpad)
gen = torch.Generator().manual_seed(0)
temp = TemperatureScheduler(462 * 50)
total = 80 + 400
pf = StepPrefetcher(eng, (pool[i % len(pool)] for i in range(total)), gen)
it = iter(pf)
B.advance(eng, temp, gen, it, 80)
torch.cuda.synchronize()
t0 = time.perf_counter()
stamps = []
for k in range(400):
    inp, prepared = next(it)
    eng.temperature = temp()
    eng.step(inp, gen, prepared=prepared)
    temp.step()
    stamps.append(time.perf_counter())
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
step_ms = (t2 - t0) / 400 * 1e3
print(f"step {step_ms:.3f} ms; host loop {(t1 - t0) / 400 * 1e3:.3f} ms/step; drain after the last enqueue {(t2 - t1) * 1e3:.2f} ms = {(t2 - t1) * 1e3 / step_ms:.2f} steps of lead; "
      f"host launch {eng.host_ms['launch'] / eng.host_ms['steps']:.3f} upload {eng.host_ms['upload'] / eng.host_ms['steps']:.3f} ms/step")
