"""Where the PCGrad Gram / solve / combine launches spend their time: phase 1 of gmp_mt_pcgrad_clip_adamw_ex over tensor ranges [k0, k1) of the
s4 model (events around 200 back-to-back calls).  python scripts/diag_gram_ranges.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
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
for k in range(6):
    eng.step(pool[k % len(pool)], gen)
torch.cuda.synchronize()
st = torch.cuda.current_stream().cuda_stream
order = (C.c_int32 * eng.T)(*range(eng.T))
lens = eng.t_len.cpu().tolist()
has = eng.has.cpu().view(eng.K, -1)[:, :eng.T].sum(1).tolist()
print(f"K = {eng.K} tensors, {sum(1 for h in has if h >= 2)} shared by >= 2 tasks; lengths of the shared ones: {sorted(set(l for l, h in zip(lens, has) if h >= 2))}")

def run(k0, k1, reps=200):
    def call():
        eng._chk(eng.lib.gmp_mt_pcgrad_clip_adamw_ex(eng.task_grads.data_ptr(), eng.P, eng.T, eng.K, eng.t_off.data_ptr(), eng.t_len.data_ptr(), eng.has.data_ptr(),
                                                      order, eng.T, eng.T - 1, -1, eng.flat.data_ptr(), eng.exp_avg.data_ptr(), eng.exp_avg_sq.data_ptr(), None,
                                                      eng.lr.data_ptr(), eng.wd.data_ptr(), 0.9, 0.999, 1e-8, 0.5, eng.final_grad.data_ptr(), eng.normsq.data_ptr(),
                                                      eng.metrics.data_ptr(), eng.flags.data_ptr(), eng.mt_ws.data_ptr(), eng.mt_ws.numel(), 0, k0, k1, 1, None, st), "x")
    for _ in range(10): call()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): call()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3

print(f"all tensors [0, {eng.K}): {run(0, eng.K):.1f} us for gram + finish + solve + combine")
for (k0, k1) in ((0, 1), (0, 8), (0, 32), (32, 64), (64, eng.K)):
    k1 = min(k1, eng.K)
    big = sum(1 for k in range(k0, k1) if lens[k] >= 65536 and has[k] >= 2)
    print(f"tensors [{k0}, {k1}) ({big} large shared ones): {run(k0, k1):.1f} us")
