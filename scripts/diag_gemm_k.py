"""Fixed vs per-K-step cost of the fp32 MFMA GEMM at the backbone shape (M = 6,507 rows): NT, N = 512, K swept."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnn_pretraining_amd import ops

dev = "cuda"
M = 6507
big = torch.zeros(256 * 1024 * 1024, device=dev)


def timed(fn, n=300):
    torch.cuda.synchronize()
    for _ in range(100): big.add_(1.0)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for N in (512, 256):
    for K in (32, 64, 128, 256, 512, 1024):
        A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); bias = torch.randn(N, device=dev)
        out = torch.empty(M, N, device=dev)
        us = timed(lambda: ops.gemm(ops.NT, A, B, bias, out))
        print(f"M={M} N={N} K={K:5d}: {us:7.2f} us  {2 * M * N * K / us / 1e6:6.1f} TF/s")
