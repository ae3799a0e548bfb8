"""What the data-parallel exchange costs a step on ONE GPU (no peer: RCCL short-circuits a one-rank all-reduce, so this is the
machinery only -- events, pack / unpack kernels, the extra stream joins): s4 step time with no exchange, with the one-message
exchange after the backward (GMP_DP_OVERLAP=0) and with the exchange in six parts beside the backward.  One process per mode."""
import json, os, random, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) == 1:
    res = {}
    # the last two: a 400 us spin kernel stands in for the collective (split over the parts by size): how much of it the backward hides
    for mode, fake in (("none", "0"), ("flat", "0"), ("overlap", "0"), ("flat", "400"), ("overlap", "400")):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), mode], capture_output=True, text=True, env=dict(os.environ, GMP_DP_FAKE_US=fake))
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        res[mode + "+" + fake] = json.loads(line[-1]) if line else r.stderr[-400:]
        print(mode, "fake collective", fake, "us:", res[mode + "+" + fake], flush=True)
    sys.exit(0)

mode = sys.argv[1]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29581", RANK="0", WORLD_SIZE="1", GMP_DP_FORCE="1",
                  GMP_DP_OVERLAP="1" if mode == "overlap" else "0")
import torch
import torch.distributed as dist
import bench as B
from gnn_pretraining_amd import dist as D, streams as ST, synthetic as S
from gnn_pretraining_amd._host import limit_host_threads
from gnn_pretraining_amd.engine import StepEngine, StepInputs
from gnn_pretraining_amd.models import PretrainableGNN
from gnn_pretraining_amd.pretrain import pretrain as PT
from gnn_pretraining_amd.pretrain.control import TemperatureScheduler

limit_host_threads(1)
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
if mode != "none":
    dist.init_process_group("nccl", device_id=dev)
torch.manual_seed(0)
doms, tasks = PT.PRETRAIN_DOMAINS["s4"], PT.ACTIVE_TASKS["s4"]
model = PretrainableGNN(dev, doms, tasks); model.train()
eng = StepEngine(model, tasks, doms, dev, seed=0, shuffle_rng=random.Random(0), rng_mode="reference",
                 grad_sync=None if mode == "none" else D.FlatGradSync())
gen = torch.Generator().manual_seed(1)
pool = [StepInputs(S.pretrain_step_batches(gen, doms), dev, eng.dpad) for _ in range(8)]
temp = TemperatureScheduler(462 * 50)
NS = int(os.environ.get("GMP_DIAG_STEPS", "400"))
B.run_steps(eng, temp, pool, gen, 80 if NS >= 400 else 20)
torch.cuda.synchronize(); t0 = time.perf_counter()
B.run_steps(eng, temp, pool, gen, NS, start=80)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
hm = eng.host_ms
print(json.dumps({"ms_per_step": round(dt / NS * 1e3, 3), "host_launch_ms": round(hm["launch"] / hm["steps"], 3), "host_upload_ms": round(hm["upload"] / hm["steps"], 3),
                  "host_exchange_ms": round(getattr(eng._packed_sync, "host_s", 0.0) / max(getattr(eng._packed_sync, "calls", 0), 1) * 1e3, 3), "sync": type(eng._packed_sync).__name__,
                  "message_MB": round(getattr(eng._packed_sync, "total", 0) * 4 / 1e6, 1), "streams": dict(ST.last_report)}))
if mode != "none":
    dist.destroy_process_group()
