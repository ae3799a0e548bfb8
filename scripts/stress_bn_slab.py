"""Stress of the BatchNorm slab form's hand-off (batchnorm.hip, gmp_bn_config.sync): the Cora-sized segment, forward + backward at 256 and 512
channels, thousands of launches back to back with a GEMM stream running beside them (uneven load), every output compared BITWISE with
the first round's and the time-out word read at the end.  python scripts/stress_bn_slab.py [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnn_pretraining_amd import ops

dev = torch.device("cuda:0")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 500
gen = torch.Generator().manual_seed(5)
rows = 2708
segd = torch.tensor([0, rows], dtype=torch.int32, device=dev)
sync = torch.zeros(ops.bn_sync_words(512, 1), dtype=torch.int32, device=dev)
cases = []
for C in (256, 512, 256):
    x = (torch.randn(rows, C, generator=gen) * 2 + 0.3).to(dev)
    gy = torch.randn(rows, C, generator=gen).to(dev)
    gamma, beta = (torch.rand(C, generator=gen) + 0.5).to(dev), torch.randn(C, generator=gen).to(dev)
    cases.append((C, x, gy, gamma, beta))
side = torch.cuda.Stream(dev)
A, Bm = torch.randn(4096, 512, device=dev), torch.randn(512, 512, device=dev)
ref, bad = None, 0
for r in range(rounds):
    if r % 3 == 0:
        with torch.cuda.stream(side):
            for _ in range(6):
                ops.gemm(ops.NT, A, Bm)
    outs = []
    for k, (C, x, gy, gamma, beta) in enumerate(cases):
        cfg = ops.make_bn_config(True, True, dropout_p=0.2, seed=7, stream_id=k, sync=sync)
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        y, sm, sr = ops.bn_fwd(x, None, segd, rows, gamma, beta, rm, rv, cfg)
        gu, gg, gb = ops.bn_bwd(gy, x, None, segd, rows, gamma, beta, rm, rv, sm, sr, cfg)
        outs += [y, sm, sr, rm, rv, gu, gg, gb]
    if ref is None:
        ref = outs
    elif r % 10 == 0 or r == rounds - 1:
        torch.cuda.synchronize()
        for i, (a, b) in enumerate(zip(outs, ref)):
            if not torch.equal(a, b):
                bad += 1
                print(f"round {r}: output {i} differs from round 0 (max |d| {(a - b).abs().max().item():.3e})", flush=True)
torch.cuda.synchronize()
print(f"{rounds} rounds x 6 launches: {bad} outputs differed; sync words [error, generation, departures] = {sync[:3].tolist()}")
