"""Step throughput of every pre-training scheme of the reference (b2 ... s5) on the stacked engine: same synthetic ENZYMES-shaped
batches as bench.py, reference-order draws, 300 timed steps after 50.  One process per scheme: every engine creates its own HIP
streams, and a process that has created many maps them onto the four hardware queues less favourably (the 5th engine of one process
ran s4 at 2.9 ms instead of 2.1 ms)."""
import json, os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from gnn_pretraining_amd import synthetic as S
from gnn_pretraining_amd._host import limit_host_threads
from gnn_pretraining_amd.engine import StepEngine, StepInputs
from gnn_pretraining_amd.models import PretrainableGNN
from gnn_pretraining_amd.pretrain import pretrain as PT
from gnn_pretraining_amd.pretrain.control import TemperatureScheduler

if len(sys.argv) != 2:
    import subprocess
    res = {}
    for sc in (sys.argv[1:] or ["b2", "b3", "b4", "s1", "s2", "s3", "s4", "s5"]):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), sc], capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        res.update(json.loads(line[-1]) if line else {sc: r.stderr[-300:]})
        print(sc, res[sc], flush=True)
    print(json.dumps(res))
    sys.exit(0)

limit_host_threads(1)
dev = torch.device("cuda:0")
out = {}
for scheme in sys.argv[1:]:
    torch.manual_seed(0)
    doms, tasks = PT.PRETRAIN_DOMAINS[scheme], PT.ACTIVE_TASKS[scheme]
    model = PretrainableGNN(dev, doms, tasks); model.train()
    eng = StepEngine(model, tasks, doms, dev, seed=0, shuffle_rng=random.Random(0), rng_mode="reference")
    gen = torch.Generator().manual_seed(1)
    gpd = 32 // len(doms)
    pool = [StepInputs(S.pretrain_step_batches(gen, doms, graphs_per_domain=gpd), dev, eng.dpad) for _ in range(8)]
    temp = TemperatureScheduler(462 * 50)
    B.run_steps(eng, temp, pool, gen, 50)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    B.run_steps(eng, temp, pool, gen, 300, start=50)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    out[scheme] = {"tasks": len(tasks), "domains": len(doms), "ms_per_step": round(dt / 300 * 1e3, 3), "graphs_per_s": round(32 * 300 / dt)}
    del eng, model, pool
    torch.cuda.empty_cache()
print(json.dumps(out))
