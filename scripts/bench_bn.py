"""BatchNorm kernels at the s4 step's shape: 28 segments of ~265 rows, C = 256 (residual + ReLU + dropout) and C = 512 (ReLU).
Prints us per call (HIP events, allocation included in the wrapper, so a small constant overhead) and effective GB/s."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_pretraining_amd import ops

dev = "cuda"
gen = torch.Generator().manual_seed(0)
S = 28
sizes = torch.randint(230, 300, (S,), generator=gen)
ptr = torch.zeros(S + 1, dtype=torch.int32); ptr[1:] = sizes.cumsum(0)
N, mx = int(ptr[-1]), int(sizes.max())
ptr = ptr.to(dev)


def timed(fn, iters=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


for C_, res, p in ((256, True, 0.2), (512, False, 0.0)):
    x = torch.randn(N, C_, device=dev); r = torch.randn(N, C_, device=dev) if res else None
    g = torch.randn(N, C_, device=dev)
    gam, bet = torch.rand(C_, device=dev) + 0.5, torch.randn(C_, device=dev) * 0.1
    cfg = ops.make_bn_config(True, True, dropout_p=p, seed=1, stream_id=3)
    y, sm, sr = ops.bn_fwd(x, r, ptr, mx, gam, bet, None, None, cfg)
    tf = timed(lambda: ops.bn_fwd(x, r, ptr, mx, gam, bet, None, None, cfg))
    tb = timed(lambda: ops.bn_bwd(g, x, r, ptr, mx, gam, bet, None, None, sm, sr, cfg, [0, 7, 14, 21, 28]))
    te = timed(lambda: (torch.empty_like(x), torch.empty(S, C_, device=dev), torch.empty(S, C_, device=dev), torch.empty(4096, dtype=torch.uint8, device=dev)))
    mb = N * C_ * 4 / 1e6
    print(f"C={C_} rows={N}: fwd {tf:.1f} us ({(2 + res) * mb / tf * 1e-3:.2f} TB/s)  bwd {tb:.1f} us ({(3 + res) * mb / tb * 1e-3:.2f} TB/s)  wrapper allocs ~{te:.1f} us")
