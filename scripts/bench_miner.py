"""Timing of the hard-negative miner (csrc/hardneg.hip) at Cora / CiteSeer size, beside (a) the reference's own
sequence of torch ops run on the same GPU (finetune.py:45-75 as written: dense mask, torch.where, torch.topk) and
(b) the CPU oracle.  Usage: python scripts/bench_miner.py [--n 2708] [--k 256]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_pretraining_amd import ops                                               # noqa: E402


def torch_sequence(emb, edges, k):
    import torch.nn.functional as F
    n = emb.size(0)
    zn = F.normalize(emb, dim=1)
    sim = torch.mm(zn, zn.t())
    mask = torch.zeros(n, n, device=emb.device, dtype=torch.bool)
    mask[edges[0], edges[1]] = True
    mask[edges[1], edges[0]] = True
    mask.fill_diagonal_(True)
    pot = ~mask
    scores = sim[pot]
    idx = torch.where(pot)
    _, top = torch.topk(scores, k, largest=True)
    return torch.stack([idx[0][top], idx[1][top]])


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3          # us


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--n", type=int, default=2708)
    p.add_argument("--d", type=int, default=256)
    p.add_argument("--edges", type=int, default=8446)        # 80 % of Cora's 10,556 directed entries
    p.add_argument("--k", type=int, default=256)
    p.add_argument("--iters", type=int, default=50)
    a = p.parse_args()
    g = torch.Generator().manual_seed(0)
    emb = torch.randn(a.n, a.d, generator=g).cuda()
    edges = torch.randint(0, a.n, (2, a.edges), generator=g).cuda()
    t_hip = timed(lambda: ops.hard_negative_topk(emb, edges, a.k), a.iters)
    t_torch = timed(lambda: torch_sequence(emb, edges, a.k), max(a.iters // 5, 3))
    from oracle import miner as OM
    t0 = time.time()
    OM.mine_hard_negatives_for_edges(emb.cpu(), edges[:, :a.k].cpu(), a.k, edges.cpu())
    t_cpu = (time.time() - t0) * 1e6
    # algorithmic traffic: write S once (n^2*4), read it 7 times (6 select passes + gather), read emb twice
    alg = a.n * a.n * 4 * 8 + 2 * a.n * a.d * 4
    print(json.dumps({"n": a.n, "d": a.d, "k": a.k, "hip_us": round(t_hip, 1), "torch_gpu_us": round(t_torch, 1),
                      "cpu_oracle_us": round(t_cpu, 1), "speedup_vs_torch_gpu": round(t_torch / t_hip, 2),
                      "algorithmic_GBps": round(alg / t_hip / 1e3, 1), "gemm_TFLOPs_if_alone": round(2 * a.n * a.n * a.d / t_hip / 1e6, 2)}))


if __name__ == "__main__":
    main()
