"""What the vendor libraries reach on the step's fp32 GEMM shapes (torch.addmm -> rocBLAS / hipBLASLt), for scale: NOT on the product path.
python scripts/probe_blas.py"""
import os, sys, torch, time
dev = torch.device("cuda:0")
def bench(f, n=200):
    for _ in range(20): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for (M, K, N) in ((7392, 256, 512), (7392, 512, 256), (3700, 256, 512), (2708, 512, 256), (15000, 768, 256)):
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); bias = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev)
    for pref in ("cublas", "cublaslt"):
        try:
            torch.backends.cuda.preferred_blas_library(pref)
        except Exception as e:
            print("pref", pref, e); continue
        t = bench(lambda: torch.addmm(bias, A, W.t(), out=out))
        print(f"{M}x{K}->{N} {pref}: {t:.1f} us = {2*M*K*N/t/1e6:.1f} TF/s")
