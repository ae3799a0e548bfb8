"""Micro-benchmark of the f32 MFMA GEMM at the shapes of the stacked step (tuning aid, not a test)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnn_pretraining_amd import ops
dev = "cuda:0"
def timeit(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3   # us
M = int(os.environ.get("M", 6400))
for (mode, name, m, n, k) in ((ops.NT, "NT fwd  x[M,256] W1[512,256]", M, 512, 256), (ops.NT, "NT fwd  r1[M,512] W2[256,512]", M, 256, 512),
                              (ops.NN, "NN dgrad g[M,256] W2[256,512]", M, 512, 256), (ops.NN, "NN dgrad g[M,512] W1[512,256]", M, 256, 512),
                              (ops.NT, "NT lp   feat[7700,768] W[256,768]", 7700, 256, 768)):
    if mode == ops.NT:
        A, B = torch.randn(m, k, device=dev), torch.randn(n, k, device=dev)
    else:
        A, B = torch.randn(m, k, device=dev), torch.randn(k, n, device=dev)
    out = torch.empty(m, n, device=dev)
    us = timeit(lambda: ops.gemm(mode, A, B, out=out))
    print(f"{name:36s} {us:8.1f} us  {2*m*n*k/us/1e6:7.1f} TF/s")
# torch reference (rocBLAS) for orientation only
A, B = torch.randn(M, 256, device=dev), torch.randn(512, 256, device=dev)
us = timeit(lambda: torch.mm(A, B.t()))
print(f"{'torch.mm (rocBLAS) M x512x256':36s} {us:8.1f} us  {2*M*512*256/us/1e6:7.1f} TF/s")
