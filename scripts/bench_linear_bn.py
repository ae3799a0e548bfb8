"""Fused Linear + BatchNorm launches against the separate ones at the stacked step's shapes (28 segments, ~7.4 k rows).
us per call incl. launch gap, back-to-back on one stream.  python scripts/bench_linear_bn.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnn_pretraining_amd import ops

dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(0)
n = [264] * 12 + [211] * 16
ptr = [0]
for v in n:
    ptr.append(ptr[-1] + v)
rows, mx, S = ptr[-1], max(n), len(n)
seg = torch.tensor(ptr, dtype=torch.int32, device=dev)


def timeit(f, reps=200):
    for _ in range(20):
        f()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e6


for K, N, res, pdrop in ((256, 512, False, 0.0), (512, 256, True, 0.2)):
    x = torch.randn(rows, K, generator=gen).to(dev)
    w = (torch.randn(N, K, generator=gen) / K ** 0.5).to(dev)
    b = torch.randn(N, generator=gen).to(dev)
    r = torch.randn(rows, N, generator=gen).to(dev) if res else None
    gam, bet = torch.ones(N, device=dev), torch.zeros(N, device=dev)
    cfg = ops.make_bn_config(True, True, pdrop, seed=1, stream_id=3)
    z = torch.empty(rows, N, device=dev)
    t_f = timeit(lambda: ops.linear_bn_fwd(x, w, b, r, seg, mx, gam, bet, cfg))
    t_g = timeit(lambda: ops.gemm(ops.NT, x, w, b, out=z))
    t_s = timeit(lambda: (ops.gemm(ops.NT, x, w, b, out=z), ops.bn_fwd(z, r, seg, mx, gam, bet, None, None, cfg)))
    print(f"fwd {K}->{N} rows {rows}: fused {t_f:.1f} us | gemm {t_g:.1f} + bn = {t_s:.1f} us")
K, N = 256, 512
g_out = torch.randn(rows, K, generator=gen).to(dev)
w = (torch.randn(K, N, generator=gen) / K ** 0.5).to(dev)
x = torch.randn(rows, N, generator=gen).to(dev)
gam, bet = torch.ones(N, device=dev), torch.zeros(N, device=dev)
cfg = ops.make_bn_config(True, True)
_, sm, sr = ops.bn_fwd(x, None, seg, mx, gam, bet, None, None, cfg)
gy = torch.empty(rows, N, device=dev)
t_f = timeit(lambda: ops.linear_bn_bwd_input(g_out, w, x, seg, mx, gam, bet, sm, sr, cfg))
t_g = timeit(lambda: ops.gemm(ops.NN, g_out, w, out=gy))
t_s = timeit(lambda: (ops.gemm(ops.NN, g_out, w, out=gy), ops.bn_bwd(gy, x, None, seg, mx, gam, bet, None, None, sm, sr, cfg)))
print(f"bwd {K}->{N} rows {rows}: fused {t_f:.1f} us | gemm {t_g:.1f} + bn_bwd = {t_s:.1f} us")
