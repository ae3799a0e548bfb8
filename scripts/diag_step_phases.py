"""Phase durations of the s4 step on the GPU clock (no profiler): forward / heads / backward (+ optimizer by difference).
GMP_STEP_TIMING=1 python scripts/diag_step_phases.py"""
import ctypes as C, os, sys, time
os.environ["GMP_STEP_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from gnn_pretraining_amd import _lib as L
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
acc, n = [0.0, 0.0, 0.0], 0
out = (C.c_float * 3)()
t_all = 0.0
for k in range(80):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.step(pool[k % len(pool)], gen)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    L.check(L.lib().gmp_step_phase_ms(out), "phase")
    if k >= 20:
        for i in range(3):
            acc[i] += out[i]
        t_all += (t1 - t0) * 1e3
        n += 1
det = (C.c_float * 13)()
L.check(L.lib().gmp_step_phase_detail_ms(det), "detail")
print("last step detail (us): enc %.0f | fwd layers %s | heads %.0f | bwd layers 4..0 %s | tail %.0f" % (
    det[0] * 1e3, [round(det[1 + i] * 1e3) for i in range(5)], det[6] * 1e3, [round(det[7 + i] * 1e3) for i in range(5)], det[12] * 1e3))
print("forward %.3f heads %.3f backward %.3f ms | whole synchronous step %.3f ms" % (acc[0] / n, acc[1] / n, acc[2] / n, t_all / n))

# the same detail for the last step of a PIPELINED run (host several steps ahead, as in bench.py): no idle GPU in front of the step
from gnn_pretraining_amd.pretrain.control import TemperatureScheduler
temp = TemperatureScheduler(462 * 50)
for rep in range(3):
    B.run_steps(eng, temp, pool, gen, 100, start=80 + 100 * rep)
    torch.cuda.synchronize()
    L.check(L.lib().gmp_step_phase_detail_ms(det), "detail")
    print("pipelined, last step (us): enc %.0f | fwd layers %s | heads %.0f | bwd layers 4..0 %s | tail %.0f | sum %.0f" % (
        det[0] * 1e3, [round(det[1 + i] * 1e3) for i in range(5)], det[6] * 1e3, [round(det[7 + i] * 1e3) for i in range(5)], det[12] * 1e3,
        sum(det[i] for i in range(13)) * 1e3))
    hd = (C.c_float * 24)()
    L.check(L.lib().gmp_step_head_ms(hd, 8), "heads")
    print("   heads, us after the forward [start, input half done, weight half done]: " +
          "  ".join(f"{t}: {hd[3 * i] * 1e3:.0f} / {hd[3 * i + 1] * 1e3:.0f} / {hd[3 * i + 2] * 1e3:.0f}" for i, t in enumerate(eng.tasks)))
