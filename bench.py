#!/usr/bin/env python
"""Headline benchmark: s4 multi-task pre-training steps on synthetic ENZYMES-shaped batches.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

One "step" = one full optimisation step of scheme s4 (5 task losses over 4 domains x 8 graphs,
5 per-task backwards, PCGrad, clip 0.5, AdamW) == run_training's loop body
(reference src/pretrain/pretrain.py:113-155).  Prints ONE JSON line (rank 0) with
  value      graphs/s over all ranks (weak scaling: every rank runs its own 32-graph step),
  roofline   the GIN aggregation kernel on the 65,536-graph rung (x = 2.2 GB, streams from HBM),
             algorithmic bytes / live HIP-event time, against the 8 TB/s HBM3E peak,
  cpu_baseline  the CPU oracle's s4 step on this box's host cores (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # before anything initialises HIP: RCCL peer mappings need dmabuf IPC here


def log(msg: str) -> None:
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)



def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--roofline-only", action="store_true", help="only the aggregation-kernel leg (used for the PMC passes)")
    ap.add_argument("--no-cora", action="store_true", help="skip the Cora_NC fine-tune leg (BASELINE.json configs[4]) of the result line")
    ap.add_argument("--rng", choices=["reference", "vectorized", "device"], default=None,
                    help="how the step's augmentation/mask/negative indices are drawn: 'reference' = the exact "
                         "per-graph draw sequence of the reference from the CPU torch.Generator; 'vectorized' = same "
                         "distributions, all graphs of a domain at once (numpy); 'device' = masks and views built on the GPU "
                         "(csrc/augment.hip, Philox).  Default: reference when the native host module is built, else vectorized")
    return ap.parse_args(argv)


if __name__ == "__main__" and "WORLD_SIZE" not in os.environ:
    _a = parse_args()
    if _a.gpus > 1:
        # `python bench.py --gpus N`: start the N ranks ourselves, as child processes, before this process has touched the GPU
        # (gnn_pretraining_amd/launch.py); rank 0's result line is relayed, any failing rank fails the run, and a line that does
        # not say n_gpus == N is an error.  Under torch.distributed.run (WORLD_SIZE set) this block is skipped.
        from gnn_pretraining_amd.launch import run_and_relay
        sys.exit(run_and_relay(_a.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:], log))

import torch  # noqa: E402

from gnn_pretraining_amd import dist as D, ops, synthetic as S  # noqa: E402
from gnn_pretraining_amd.engine import StepEngine, StepInputs, StepPrefetcher  # noqa: E402
from gnn_pretraining_amd.graph import Batch  # noqa: E402
from gnn_pretraining_amd.models import PretrainableGNN  # noqa: E402
from gnn_pretraining_amd.pretrain import pretrain as PT  # noqa: E402
from gnn_pretraining_amd.pretrain.control import TemperatureScheduler  # noqa: E402

SCHEME = "s4"
GRAPHS_PER_STEP = 32
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy ceiling
# HBM bytes per launch of the aggregation kernel at exactly this rung, from separate rocprofv3 --pmc passes (FETCH_SIZE x 2: gfx950
# reports half of wide coalesced reads, MI355X_MICROARCH.md section HBM; + WRITE_SIZE; x 1024).  PMC passes cannot run inside this
# process, so the figure is a constant -- tied to the source it was measured on: `roofline.traffic` is reported only while the
# lines of csrc/aggregate.hip between its [pmc-stamp-begin] / [pmc-stamp-end] markers (the kernel the passes measured, and its
# launcher) still hash to PMC_SOURCE_SHA16 and the rung is PMC_SHAPE; null otherwise (re-measure: profiles/README.md).
PMC_SHAPE = (2146816, 8068480)
PMC_TRAFFIC_BYTES = int((1230567.3 * 2 + 2146853.9) * 1024)     # profiles/r03_pmc_aggregate_fwd_bwd.csv (forward: 1.063 x algorithmic)
PMC_TRAFFIC_BYTES_BWD = int((2744854.8 * 2 + 2169640.2) * 1024) # same file, backward with the eps row products (1.18 x algorithmic)
PMC_SOURCE_SHA16 = "e27842a84e2faf0b"                           # sha256 of the marked region of the measured source, first 16 hex digits


def pmc_source_sha16():
    import hashlib
    try:
        txt = open(os.path.join(ROOT, "gnn_pretraining_amd", "csrc", "aggregate.hip"), encoding="utf-8").read()
        region = txt[txt.index("// [pmc-stamp-begin]"):txt.index("// [pmc-stamp-end]")]
    except (OSError, ValueError):
        return None
    return hashlib.sha256(region.encode()).hexdigest()[:16]


def pmc_traffic(shape, backward: bool = False):
    return (PMC_TRAFFIC_BYTES_BWD if backward else PMC_TRAFFIC_BYTES) if (shape == PMC_SHAPE and pmc_source_sha16() == PMC_SOURCE_SHA16) else None


ROOFLINE_BWD = None
PRIME_STEPS = 30                # untimed set-up steps before the caller's warm-up (kernel code objects, workspaces, clocks)
POOL = 8                       # distinct pre-generated step inputs, resident in HBM before the timed region


def host_cores() -> int:
    """Cores this process may actually use (the GPU box gives one GPU's job a 16-core share of a big host)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:                                        # cgroup v2 CPU quota, if any
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("GMP_BENCH_MAX_CORES", "16"))))


def make_pool(seed: int, device, dpad: int):
    gen = torch.Generator().manual_seed(seed)
    domains = PT.PRETRAIN_DOMAINS[SCHEME]
    return [StepInputs(S.pretrain_step_batches(gen, domains, enzymes_shaped=True), device, dpad) for _ in range(POOL)]


def run_steps(engine, temperature, pool, gen, n, start=0):
    """The body of run_training (reference pretrain.py:113-155): draw artefacts, 5 task losses, per-task
    gradients, PCGrad, clip, AdamW, scheduler step -- one engine.step per optimisation step."""
    pf = StepPrefetcher(engine, (pool[(start + i) % len(pool)] for i in range(n)), gen)
    advance(engine, temperature, gen, iter(pf), n)
    return pf


def advance(engine, temperature, gen, it, n):
    """n steps from a running StepPrefetcher iterator (set-up, warm-up and timed steps share ONE prefetcher, so the timed
    window starts with the host pipeline full instead of waiting for a new thread's first prepare())."""
    for _ in range(n):
        inp, prepared = next(it)
        engine.temperature = temperature()
        engine.step(inp, gen, prepared=prepared)
        temperature.step()


def aggregation_roofline(device, graphs: int = 65536, distinct: int = 1024, iters: int = 20):
    """Time gmp_gin_aggregate_fwd alone on the 65,536-graph rung."""
    gen = torch.Generator().manual_seed(7)
    base = Batch.from_data_list([S.random_graph(gen, 4) for _ in range(distinct)])
    reps = graphs // distinct
    n0, e0 = base.num_nodes, base.num_edges
    N, E = n0 * reps, e0 * reps
    ei = base.edge_index.to(device)
    offs = (torch.arange(reps, device=device) * n0).view(1, reps, 1)
    ei_big = (ei.view(2, 1, e0) + offs).reshape(2, E).contiguous()       # the 1,024 graphs tiled 64x
    csr = ops.csr_build(ei_big, N)
    del ei_big
    x = torch.randn(N, 256, device=device)
    eps = torch.zeros(1, device=device)
    out = torch.empty_like(x)
    l = ops.L.lib()
    args = (ops._ptr(x), ops._ptr(csr.rowptr), ops._ptr(csr.col), ops._ptr(eps), ops._ptr(out), N, 256)
    for _ in range(10):                                    # warm-up: clocks ramp from the small-kernel regime of the step
        ops.L.check(l.gmp_gin_aggregate_fwd(*args, ops._stream(x)), "aggregate")
    torch.cuda.synchronize(device)
    # HIP events on the stream the kernel is launched on (torch's current stream)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    ev[0].record()
    for i in range(iters):
        ops.L.check(l.gmp_gin_aggregate_fwd(*args, ops._stream(x)), "aggregate")
        ev[i + 1].record()
    torch.cuda.synchronize(device)
    ms = sum(ev[i].elapsed_time(ev[i + 1]) for i in range(iters)) / iters
    alg_bytes = 2 * 4 * 256 * N + 4 * (N + 1) + 4 * E
    achieved = alg_bytes / (ms * 1e-3) / 1e9
    # the backward of the same rung (gmp_gin_aggregate_bwd: g_x = (1 + eps) g + sum over the TRANSPOSED CSR of g, and the eps
    # gradient sum_r <g[r], x[r]>): the same LDS-resident-tile kernel on the by-source CSR, with the layer input x streamed beside
    # it for the row products.  Algorithmic bytes: read g, read x, write g_x (3 x 1 KiB per row) + the transposed index arrays +
    # the row products written and read back once by the two-stage sum
    g = torch.randn(N, 256, device=device)
    gx, geps = torch.empty_like(x), torch.empty(1, device=device)
    ws = torch.empty(l.gmp_gin_aggregate_bwd_workspace_bytes(N, 256), dtype=torch.uint8, device=device)
    bargs = (ops._ptr(g), ops._ptr(csr.rowptr_t), ops._ptr(csr.col_t), ops._ptr(eps), ops._ptr(x), ops._ptr(gx), ops._ptr(geps), N, 256,
             ops._ptr(ws), ws.numel())
    for _ in range(5):
        ops.L.check(l.gmp_gin_aggregate_bwd(*bargs, ops._stream(x)), "aggregate bwd")
    torch.cuda.synchronize(device)
    evb = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    evb[0].record()
    for i in range(iters):
        ops.L.check(l.gmp_gin_aggregate_bwd(*bargs, ops._stream(x)), "aggregate bwd")
        evb[i + 1].record()
    torch.cuda.synchronize(device)
    ms_b = sum(evb[i].elapsed_time(evb[i + 1]) for i in range(iters)) / iters
    bytes_b = 3 * 4 * 256 * N + 4 * (N + 1) + 4 * E + 2 * 4 * N
    ach_b = bytes_b / (ms_b * 1e-3) / 1e9
    global ROOFLINE_BWD
    ROOFLINE_BWD = {"bound": "hbm", "achieved": round(ach_b, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach_b / HBM_PEAK_GBS, 4),
                    "traffic": pmc_traffic((N, E), backward=True), "kernel": "gin_aggregate_ldstile_kernel<DOT> + 2 small sum kernels (gmp_gin_aggregate_bwd with the eps "
                                               "gradient, N >= 65536)", "rows": N, "edges": E, "bytes_per_launch": bytes_b,
                    "avg_launch_ms": round(ms_b, 4), "launches": iters, "frac_of_copy_ceiling": round(ach_b / 6290.0, 4)}
    return {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": pmc_traffic((N, E)), "traffic_source": f"rocprofv3 --pmc passes on the marked kernel region of aggregate.hip, sha256 {PMC_SOURCE_SHA16}", "kernel": "gin_aggregate_ldstile_kernel (gmp_gin_aggregate_fwd, N >= 65536)",
            "rows": N, "edges": E, "bytes_per_launch": alg_bytes, "avg_launch_ms": round(ms, 4), "launches": iters,
            # for orientation only (frac above is against the 8 TB/s spec): the part's measured float4 copy rate, MI355X_MICROARCH.md
            "measured_copy_ceiling": 6290.0, "frac_of_copy_ceiling": round(achieved / 6290.0, 4)}


def gemm_roofline(device, rows: int = 7392):
    """The dense feature-transform GEMM of a backbone layer at the step's size (rows of the stacked s4 batch x 256 -> 512, fp32
    MFMA, bias epilogue) against the 157 TFLOP/s fp32 MFMA peak: 50 launches back to back between two HIP events on the launch
    stream (so the ~2 us gap between dependent launches is inside the figure; an event pair around every single launch reads
    2 us HIGHER than that -- the records cost more than the gap -- and rocprofv3's per-dispatch duration ~2 us lower)."""
    A, B = torch.randn(rows, 256, device=device), torch.randn(512, 256, device=device)
    bias, out = torch.randn(512, device=device), torch.empty(rows, 512, device=device)
    for _ in range(10):
        ops.gemm(ops.NT, A, B, bias, out)
    iters = 50
    torch.cuda.synchronize(device)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(iters):
        ops.gemm(ops.NT, A, B, bias, out)
    ev[1].record()
    torch.cuda.synchronize(device)
    ms = ev[0].elapsed_time(ev[1]) / iters
    flops = 2.0 * rows * 512 * 256
    ach = flops / (ms * 1e-3) / 1e12
    return {"bound": "mfma", "achieved": round(ach, 1), "peak": 157.0, "unit": "TFLOP/s", "frac": round(ach / 157.0, 4), "traffic": None,
            "kernel": "gemm_pipe_kernel<1,1,NT> (gmp_gemm_f32: GIN layer 256 -> 512 + bias; csrc/gemm_pipe.h)", "M": rows, "N": 512, "K": 256,
            "flops_per_launch": flops, "avg_launch_ms": round(ms, 4), "launches": iters,
            "note": "back-to-back dependent launches (launch gap included).  The main loop is MFMA-bound at the clock the chip holds "
                    "under this load (~2.05 GHz = 134 TFLOP/s: 2.0 us per 32-deep K-step of a 128x128 tile); the rest of a launch is "
                    "its prologue (first DMA, ~3 us with the launch gap) and its 15 MB epilogue + end-of-kernel L2 write-back (~5 us)"}


def cora_finetune(device, seed: int, steps: int = 200, cpu_budget_s: float = 6.0):
    """BASELINE.json configs[4]: Cora_NC full-graph fine-tune (2,708 nodes / 5,429 undirected edges / 1,433 features, 256 hidden,
    5 GIN layers, CE on the 140 training nodes, AdamW) -- one optimisation step = one epoch of the reference
    (src/finetune/finetune.py:162-179).  Reported beside the headline line: ms per step on the explicit-kernel engine
    (finetune/engine.py), the encoder GEMM 2,708 x 1,433(->1,440) x 256 against the fp32 MFMA peak, and the CPU oracle's step."""
    from gnn_pretraining_amd.finetune.engine import NodeClassificationEngine
    from gnn_pretraining_amd.models import FinetuneGNN
    gen = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    g = S.cora_like(gen)
    model = FinetuneGNN(device, "Cora_NC", "full_finetune")
    model.train()
    eng = NodeClassificationEngine(model, g.x, g.edge_index, device, seed=seed)
    idx = torch.randperm(g.num_nodes, generator=gen)[:140]
    y = g.y[idx].to(device)
    idx_d = idx.to(device)
    for _ in range(20):
        eng.step(idx_d, y)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.step(idx_d, y)
    torch.cuda.synchronize(device)
    ms = (time.perf_counter() - t0) / steps * 1e3
    loss = eng.loss()
    # the encoder GEMM alone (the one large dense product of the step), back-to-back launches between two events
    A, W, b, out = eng.x, torch.randn(256, eng.dpad, device=device), torch.randn(256, device=device), torch.empty(eng.N, 256, device=device)
    for _ in range(10):
        ops.gemm(ops.NT, A, W, b, out)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize(device)
    ev[0].record()
    for _ in range(50):
        ops.gemm(ops.NT, A, W, b, out)
    ev[1].record()
    torch.cuda.synchronize(device)
    gms = ev[0].elapsed_time(ev[1]) / 50
    flops = 2.0 * eng.N * 256 * eng.d_in                       # algorithmic: the padded columns are zeros
    out_d = {"workload": "Cora_NC-shaped full-graph fine-tune step (N=2708, E=10858 directed, F=1433, 7 classes, 140 training nodes)",
             "ms_per_step": round(ms, 3), "steps_per_s": round(1e3 / ms, 1), "loss_after": round(loss, 4), "steps": steps,
             "roofline_gemm": {"bound": "mfma", "achieved": round(flops / (gms * 1e-3) / 1e12, 1), "peak": 157.0, "unit": "TFLOP/s",
                               "frac": round(flops / (gms * 1e-3) / 1e12 / 157.0, 4), "kernel": "gemm_pipe_kernel NT (input encoder 2708 x 1433 -> 256)",
                               "avg_launch_ms": round(gms, 4), "launches": 50}}
    # CPU oracle: the same step (oracle model + torch AdamW with the reference's groups) on the host cores
    from oracle import models as OM
    from oracle.harness import to_oracle
    from gnn_pretraining_amd.graph import Batch
    cores = host_cores()
    torch.set_num_threads(cores)
    om = OM.FinetuneGNN(torch.device("cpu"), "Cora_NC", "full_finetune")
    om.train()
    opt = torch.optim.AdamW(om.param_groups)
    ob = to_oracle(Batch.from_data_list([g]))
    times, t_end = [], time.time() + cpu_budget_s
    while (time.time() < t_end and len(times) < 40) or len(times) < 3:
        t1 = time.time()
        lo = torch.nn.functional.cross_entropy(om(ob)[idx], g.y[idx])
        opt.zero_grad(); lo.backward(); opt.step()
        times.append(time.time() - t1)
    torch.set_num_threads(1)
    times.sort()
    med = times[len(times) // 2]
    out_d["cpu_baseline"] = {"value": round(1.0 / med, 2), "unit": "steps/s", "cores": cores, "kind": "port",
                             "sample": f"{len(times)} steps of the same workload on the torch-only oracle (median {med * 1e3:.1f} ms/step)"}
    return out_d


def cpu_baseline(seed: int, budget_s: float = 20.0):
    """The CPU oracle (a port: the reference itself needs torch_geometric) on this box's host cores."""
    from oracle import models as OM, tasks as OTk, train as OTr
    from oracle.harness import to_oracle
    cores = host_cores()
    domains, tasks = PT.PRETRAIN_DOMAINS[SCHEME], PT.ACTIVE_TASKS[SCHEME]

    def run(threads: int, budget: float, cap: int):
        torch.set_num_threads(threads)
        torch.manual_seed(seed)
        random.seed(seed)             # PyG's negative sampler (oracle/augment.py) draws from Python's global `random`
        gen = torch.Generator().manual_seed(seed)
        model = OM.PretrainableGNN(torch.device("cpu"), domains, tasks)
        model.train()
        temp, grl = OTr.TemperatureScheduler(462 * 50), OTr.GRLScheduler(50, 462)
        otasks = OTk.instantiate_tasks(model, tasks, grl, temp)
        opt, bal = OTr.make_optimizer(model, tasks), OTr.AdaptiveLossBalancer()
        pool = [{d: to_oracle(b) for d, b in S.pretrain_step_batches(gen, domains, enzymes_shaped=True).items()} for _ in range(4)]
        for i in range(2):
            OTr.train_step(model, otasks, opt, bal, grl, temp, pool[i % 4], gen)
        times, t_end = [], time.time() + budget
        while time.time() < t_end and len(times) < cap:
            t0 = time.time()
            OTr.train_step(model, otasks, opt, bal, grl, temp, pool[len(times) % 4], gen)
            times.append(time.time() - t0)
        times.sort()
        return times[len(times) // 2], len(times)

    med, n = run(cores, budget_s, 50)
    med1, n1 = run(1, budget_s / 2, 12)                  # "also a 1-thread figure"
    torch.set_num_threads(1)
    return {"value": round(GRAPHS_PER_STEP / med, 2), "unit": "graphs/s", "cores": cores, "kind": "port",
            "sample": f"{n} s4 steps of 32 synthetic ENZYMES-shaped graphs (median {med * 1e3:.1f} ms/step), "
                      f"torch-only oracle, torch.set_num_threads({cores})",
            "one_thread": {"value": round(GRAPHS_PER_STEP / med1, 2), "cores": 1,
                           "sample": f"{n1} steps of the same workload (median {med1 * 1e3:.1f} ms/step), torch.set_num_threads(1)"}}


def main() -> None:
    a = parse_args()
    # Only the result line may reach stdout: RCCL prints a banner (HIP / ROCm version, hostname, library path) on stdout when a
    # communicator is created, on every rank.  Everything printed before the result goes to stderr instead.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(line: dict) -> None:
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(line), flush=True)
        os.dup2(2, 1)

    from gnn_pretraining_amd.engine import hostdraw
    if a.rng is None:       # the reference's exact draw order when its native implementation is built (GPU-bound either way)
        a.rng = "reference" if hostdraw() is not None else "vectorized"

    if a.roofline_only:
        torch.cuda.set_device(0)
        emit({"roofline": aggregation_roofline(torch.device("cuda:0")), "roofline_bwd": ROOFLINE_BWD})
        return
    from gnn_pretraining_amd._host import limit_host_threads
    limit_host_threads(1)      # the host side of a step is tiny index work: torch's default pool (every core of the node, per rank) only hurts
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = max(torch.cuda.device_count(), 1)
    device = torch.device(f"cuda:{local % ndev}")       # one rank per GPU; (rehearsals on a 1-GPU box share it over gloo)
    torch.cuda.set_device(device)
    world = D.init_from_env(os.environ.get("GMP_DIST_BACKEND"))
    if world != a.gpus:        # never print an n_gpus:1 line for --gpus 8
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    rank = D.rank()
    backend = torch.distributed.get_backend() if world > 1 else None
    if world > 1:
        # one collective BEFORE the engine measures which of its streams own a hardware queue: the communicator (and whatever
        # streams / queues RCCL creates with it) exists from here on, so the calibration sees the process as it will run
        t = torch.ones(1, device=device)
        torch.distributed.all_reduce(t)
        if int(t.item()) != world:
            raise SystemExit(f"rank {rank}: the first all-reduce over {backend} returned {t.item()} for {world} ranks")

    # The step's main stream carries the critical chain (forward, input-gradient chain of the backward, optimizer); its small BatchNorm /
    # aggregation kernels compete for CU slots with the weight-gradient GEMMs on the other streams.  On a high-priority queue their
    # workgroups are dispatched first: 1.452 against 1.463 ms per step (three interleaved pairs).  GMP_MAIN_PRIORITY=0: default queue.
    if os.environ.get("GMP_MAIN_PRIORITY", "-1") != "0":
        torch.cuda.set_stream(torch.cuda.Stream(device=device, priority=int(os.environ.get("GMP_MAIN_PRIORITY", "-1"))))
    seed = 42
    torch.manual_seed(seed)                        # identical replicas on every rank
    model = PretrainableGNN(device, PT.PRETRAIN_DOMAINS[SCHEME], PT.ACTIVE_TASKS[SCHEME])
    model.train()
    sync = D.FlatGradSync() if world > 1 else None
    engine = StepEngine(model, PT.ACTIVE_TASKS[SCHEME], PT.PRETRAIN_DOMAINS[SCHEME], device, seed=seed + rank,
                        shuffle_rng=random.Random(seed), grad_sync=sync, rng_mode=a.rng)     # same PCGrad task order on every rank
    temperature = TemperatureScheduler(total_steps=462 * PT.EPOCHS)
    pool = make_pool(seed + 1000 * rank, device, engine.dpad)   # every rank draws its own batches (weak scaling)
    gen = torch.Generator().manual_seed(seed + rank)

    # Set-up, not measurement: 30 untimed steps load every kernel's code object, grow the lazily sized workspaces and bring the
    # clocks out of idle, so that a short run (--steps 5 --warmup 2) times the same steady state a long one does.  The W warm-up
    # steps and the K timed steps the caller asked for follow, unchanged.
    gates_checked = None
    if world > 1:
        # Gates are only sound between streams on hardware queues of their own, and that was measured, not promised: in a
        # data-parallel process prove it once with the communicator live (probe steps with gates vs events, bitwise, on every
        # rank; StepEngine.verify_gates) and fall back to events otherwise.  GMP_DP_GATES=0 / 1 skips the check.
        force = os.environ.get("GMP_DP_GATES")
        if force == "0":
            engine.use_gates = False
        elif force != "1" and engine.use_gates:
            gates_checked = engine.verify_gates(pool[0])
            log(f"rank {rank}: gates vs events under the live communicator: {engine.gates_verified}")
        torch.distributed.barrier()                 # ranks start stepping together (imports / pool building differ by seconds)
    total = PRIME_STEPS + a.warmup + a.steps
    pf = StepPrefetcher(engine, (pool[i % len(pool)] for i in range(total)), gen)
    it = iter(pf)
    advance(engine, temperature, gen, it, PRIME_STEPS)
    from gnn_pretraining_amd import streams as ST
    log(f"streams: {ST.last_report}, cross-stream sync: {'gates' if engine.use_gates else 'events'}")
    log(f"rank {rank}/{world}: model + {POOL} step inputs resident, {PRIME_STEPS} set-up steps done, warming up {a.warmup} steps")
    advance(engine, temperature, gen, it, a.warmup)
    log("timing")
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize(device)
    busy0, wait0, draw0 = pf.busy_s, pf.wait_s, pf.draw_s
    t0 = time.perf_counter()
    advance(engine, temperature, gen, it, a.steps)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - t0
    pf.busy_s, pf.wait_s, pf.draw_s = pf.busy_s - busy0, pf.wait_s - wait0, pf.draw_s - draw0
    for _ in it:                                    # (the prefetcher's end marker)
        pass
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())

    hm = engine.host_ms
    log(f"{a.steps} steps in {elapsed:.3f} s; last-step losses {engine.losses()}")
    log("host ms/step: " + ", ".join(f"{k} {hm[k] / max(hm['steps'], 1):.2f}" for k in ("draw", "plan", "upload", "launch")) +
        f"; prefetch threads busy: draws {pf.draw_s / max(a.steps, 1) * 1e3:.2f}, layout {pf.busy_s / max(a.steps, 1) * 1e3:.2f}, launcher waited for them {pf.wait_s / max(a.steps, 1) * 1e3:.2f}")
    roof = roof_gemm = cpu = None
    if rank == 0 and not a.no_roofline:
        roof = aggregation_roofline(device)
        log(f"roofline {roof['achieved']} GB/s")
        roof_gemm = gemm_roofline(device)
        log(f"gemm {roof_gemm['achieved']} TFLOP/s")
    cora = None
    if rank == 0 and world == 1 and not a.no_cora:
        cora = cora_finetune(device, seed)
        log(f"cora fine-tune step {cora['ms_per_step']} ms (CPU oracle {cora['cpu_baseline']['value']} steps/s)")
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        log("cpu baseline ...")
        cpu = cpu_baseline(seed)
    if rank == 0:
        ms = elapsed / a.steps * 1e3
        line = {
            "metric": "pretrain graphs/sec (s4 multi-task)", "value": round(GRAPHS_PER_STEP * world * a.steps / elapsed, 2),
            "unit": "graphs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "fp32", "data": "synthetic",
            "config": {"workload": "s4 (NFM+LP+NC+GC+GP) pre-training step, 4 domains x 8 synthetic ENZYMES-shaped graphs "
                                   "per rank (input dims 7/4/37/21, hidden 256, 5 GIN layers), PCGrad + clip + AdamW",
                       "global_batch_graphs": GRAPHS_PER_STEP * world, "parallelism": f"dp{world}", "setup_steps": PRIME_STEPS, "index_rng": a.rng,
                       "index_draws": ("native (csrc_host/hostdraw.cpp)" if (a.rng == "reference" and hostdraw() is not None) else
                                       "device (csrc/augment.hip; negatives on the host)" if a.rng == "device" else "python/numpy"),
                       "cross_stream_sync": "gates" if engine.use_gates else "events",
                       "lp_rows": ("one per unordered pair through the 768->256 layer; dropout mask, score and BCE term per ORDERED row (gmp_lp_pair_*)"
                                   if engine.lp_merge else "the reference's ordered list (GMP_LP_MERGE=0)"), "gates_verified_under_communicator": gates_checked,
                       "ranks": world, "backend": ("rccl (torch 'nccl')" if backend == "nccl" else backend),
                       "devices_visible": torch.cuda.device_count(),
                       "gradient_exchange": (type(engine._shard_sync_obj).__name__ if engine._shard_sync_obj is not None else
                                             type(engine._packed_sync).__name__ if engine._packed_sync is not None else None)},
            "roofline": roof, "cpu_baseline": cpu, "roofline_gemm": roof_gemm, "roofline_bwd": ROOFLINE_BWD, "cora_finetune": cora,
        }
        emit(line)
    if world > 1:
        torch.distributed.barrier()                 # rank 0 may still be in its roofline leg: leave together
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
